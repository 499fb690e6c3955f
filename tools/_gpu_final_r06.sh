# final collection of round 6 after the 128 x 128 tile form: suite, trace of the default command, counters, bench lines
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/final2_r06
mkdir -p $O; cd $R
python -m pytest tests -q -m gpu --durations=15 > $O/gpu_suite.txt 2>&1; tail -2 $O/gpu_suite.txt
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_graph
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_graph -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-variants --no-roofline-pass --no-rocprof --no-pmc --no-ceiling > $O/graph_bench.log 2>&1
cp $(find /tmp/p_graph -name "*kernel_stats.csv" | head -1) $O/r06_graph_run_kernel_stats.csv
python3 $R/tools/trace_stats.py /tmp/p_graph $O/r06_graph_timed_region.csv --last-solves 10 --total-solves 16 --time-steps 100 --label "bench.py --steps 10 --warmup 2 (default options: -pn_graph_capture auto; tapes retained; grouped pn_linear_wgrad launches on the sweep's stream, 128 x 128 tiles): the 10 timed replays only" > /dev/null
unset DEBUG_CLR_GRAPH_PACKET_CAPTURE
cd $R
bash tools/_gpu_pmc_wgrad.sh > /dev/null 2>&1; cp gpurun_out/pmc_wgrad/r06_pmc_wgrad.txt $O/
timeout 1200 python bench.py > $O/r06_bench.json 2> $O/r06_bench.err; echo "rc $?" >> $O/r06_bench.err
for i in 1 2 3; do timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-variants >> $O/r06_bench_repeat.jsonl 2>> $O/r06_bench_repeat.err; done
timeout 900 python tools/fuzz_modes.py 60 6 > $O/fuzz_modes.txt 2>&1; tail -1 $O/fuzz_modes.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
