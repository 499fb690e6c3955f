R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/final_r05
mkdir -p $O
cd $R
python -m pytest tests -q -m gpu --durations=25 > $O/gpu_suite.txt 2>&1; tail -3 $O/gpu_suite.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
cd $R
timeout 900 python bench.py > $O/r05_bench.json 2> $O/r05_bench.err; echo "rc $?" >> $O/r05_bench.err
for i in ${REPEATS:-1 2 3}; do timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-variants >> $O/r05_bench_repeat.jsonl 2>> $O/r05_bench_repeat.err; done
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_graph
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_graph -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-variants --no-roofline-pass --no-rocprof --no-pmc --no-ceiling > $O/graph_bench.log 2>&1
cp $(find /tmp/p_graph -name "*kernel_stats.csv" | head -1) $O/r05_graph_run_kernel_stats.csv
python3 $R/tools/trace_stats.py /tmp/p_graph $O/r05_graph_timed_region.csv --last-solves 10 --total-solves 16 --time-steps 100 --label "bench.py --steps 10 --warmup 2 (no launch option: -pn_graph_capture auto; tapes retained): the 10 timed replays only" > /dev/null
cd $R
{ echo "== LD_LIBRARY_PATH=pnode_amd/lib tools/mb_wgrad_abi (twice)"; LD_LIBRARY_PATH=pnode_amd/lib timeout 120 ./tools/mb_wgrad_abi; LD_LIBRARY_PATH=pnode_amd/lib timeout 120 ./tools/mb_wgrad_abi; } > $O/mb_abi.txt 2>&1
tail -2 $O/r05_graph_timed_region.csv; head -c 300 $O/r05_bench.json; echo; cat $O/mb_abi.txt | tail -4
