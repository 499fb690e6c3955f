# Round-3 profile collection (one MI355X).  Raw traces stay in /tmp; summaries go to gpurun_out/prof_r03/.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_r03
mkdir -p $O
cd $R
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
# (0) the whole -m gpu suite
timeout 2400 python -m pytest tests -x -q -m gpu > $O/gpu_suite.log 2>&1; echo "suite rc $?" >> $O/gpu_suite.log
# (1) C5 shard, theta methods: every Krylov configuration, reference-literal Conv1d func and the stencil form of the same operator
timeout 1500 python tools/bench_c5_theta.py > $O/r03_c5_theta.txt 2>&1
timeout 900 python tools/bench_c5_theta.py --only-default --tunableop > $O/r03_c5_theta_tunableop.txt 2>&1
# (2) kernel trace of the default Krylov configuration (stencil func)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_kr
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_kr -- python3 $R/tools/prof_krylov.py default stencil --trace-only > $O/kr_trace.log 2>&1
cp $(find /tmp/p_kr -name "*kernel_stats.csv" | head -1) $O/r03_krylov_stencil_kernel_stats.csv
python3 $R/tools/krylov_trace_summary.py /tmp/p_kr > $O/r03_krylov_stencil_trace_summary.txt 2>&1
# (3) C3b: the fused solution-update + error-norm kernel in place
rm -rf /tmp/p_c3b
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_c3b -- python3 $R/tools/prof_c3b.py --solves 3 > $O/c3b.log 2>&1
cp $(find /tmp/p_c3b -name "*kernel_stats.csv" | head -1) $O/r03_c3b_kernel_stats.csv
python3 $R/tools/trace_stats.py /tmp/p_c3b $O/r03_c3b_trace_stats.csv --label "tools/prof_c3b.py --solves 3 (C3b: dopri5 adaptive, 4096 x 512 fp32, max_cps 50), whole run" > /dev/null
for v in "wfin=1" "wvpt=1" "wvpt=4"; do
  rm -rf /tmp/p_c3b_x
  PN_TUNE="$v" timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_c3b_x -- python3 $R/tools/prof_c3b.py --solves 3 > $O/c3b_$v.log 2>&1
  python3 $R/tools/trace_stats.py /tmp/p_c3b_x "$O/r03_c3b_${v}_trace_stats.csv" --label "PN_TUNE=$v tools/prof_c3b.py --solves 3" > /dev/null
done
# (4) headline: graph-replayed timed region under the profiler, then the default bench lines
rm -rf /tmp/p_graph
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_graph -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-variants --no-roofline-pass > $O/graph_bench.log 2>&1
cp $(find /tmp/p_graph -name "*kernel_stats.csv" | head -1) $O/r03_graph_run_kernel_stats.csv
python3 $R/tools/trace_stats.py /tmp/p_graph $O/r03_graph_timed_region.csv --last-solves 10 --total-solves 15 --time-steps 100 --label "bench.py --steps 10 --warmup 2 (default: hipGraph replay, tapes retained): the 10 timed replays only" > /dev/null
cd $R
timeout 900 python bench.py --steps 20 --warmup 5 > $O/r03_bench.json 2> $O/r03_bench.err; echo "rc $?" >> $O/r03_bench.err
for c in c2 c3b c4 c5; do timeout 900 python bench.py --config $c --steps 5 --warmup 2 > $O/r03_bench_$c.json 2> $O/r03_bench_$c.err; echo "rc $?" >> $O/r03_bench_$c.err; done
# (5) host overhead of eager launches
timeout 300 python tools/host_overhead.py > $O/r03_host_overhead.txt 2>&1
timeout 300 python tools/profile_host_noop.py > $O/r03_profile_host_noop.txt 2>&1
tail -4 $O/gpu_suite.log
grep "C5 shard" $O/r03_c5_theta.txt $O/r03_c5_theta_tunableop.txt
cat $O/r03_krylov_stencil_trace_summary.txt | head -16
grep -h combine $O/*c3b*trace_stats.csv
tail -3 $O/r03_graph_timed_region.csv
for f in $O/r03_bench*.json; do echo $f; head -c 400 $f; echo; done
cat $O/r03_host_overhead.txt
