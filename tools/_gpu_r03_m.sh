R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_m
mkdir -p $O
cd $R
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
timeout 2400 python -m pytest tests -x -q -m gpu > $O/gpu_suite.log 2>&1; echo "suite rc $?" >> $O/gpu_suite.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?" >> $O/smoke.log
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?" >> $O/bench_default.err
tail -n 4 $O/gpu_suite.log; tail -n 2 $O/smoke.log; tail -n 2 $O/bench_default.err; head -c 600 $O/bench_default.json
