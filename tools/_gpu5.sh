R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r02
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
rm -rf /tmp/p_graph
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_graph -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-variants --no-roofline-pass > $O/graph_bench.log 2>&1
cp $(find /tmp/p_graph -name "*kernel_stats.csv" | head -1) $O/r02_graph_run_kernel_stats.csv
python3 $R/tools/trace_stats.py /tmp/p_graph $O/r02_graph_timed_region.csv --last-solves 10 --total-solves 15 --time-steps 100 --label "bench.py --steps 10 --warmup 2 (default: hipGraph replay, tapes retained): the 10 timed replays only" | tail -3
# C2 (small N) graph-mode and eager traces
rm -rf /tmp/c2g /tmp/c2e
rocprofv3 --kernel-trace --output-format csv -d /tmp/c2g -- python3 $R/tools/c2_trace.py > $O/c2_graph.log 2>&1
python3 $R/tools/trace_stats.py /tmp/c2g $O/r02_c2_graph_timed_region.csv --last-solves 10 --total-solves 14 --time-steps 100 --label "C2 4096x2 rk4 x100, hipGraph replay: the 10 timed replays" | tail -3
rocprofv3 --kernel-trace --output-format csv -d /tmp/c2e -- python3 $R/tools/c2_trace.py eager > $O/c2_eager.log 2>&1
python3 $R/tools/trace_stats.py /tmp/c2e $O/r02_c2_eager_timed_region.csv --last-solves 10 --total-solves 14 --time-steps 100 --label "C2 4096x2 rk4 x100, eager launches: the 10 timed solves" | tail -3
cd $R
python3 tools/c2_trace.py; python3 tools/c2_trace.py eager
timeout 900 python3 bench.py --steps 10 --warmup 3 > gpurun_out/bench_v3.json 2> gpurun_out/bench_v3.err; tail -3 gpurun_out/bench_v3.err
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > gpurun_out/t5.log; tail -6 gpurun_out/t5.log
