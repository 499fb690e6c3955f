# kernel trace of the timed region of the default bench (graph replay); $1 = tag, further args go to bench.py
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-a}; shift
O=$R/gpurun_out/trace_r06_$TAG
mkdir -p $O
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_graph
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_graph -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-variants --no-roofline-pass --no-rocprof --no-pmc --no-ceiling "$@" > $O/graph_bench.log 2>&1
cp $(find /tmp/p_graph -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
python3 $R/tools/trace_stats.py /tmp/p_graph $O/timed_region.csv --last-solves 10 --total-solves 16 --time-steps 100 --label "bench.py --steps 10 --warmup 2 $*: the 10 timed replays only" > /dev/null
tail -1 $O/graph_bench.log | head -c 400; echo
column -s, -t $O/timed_region.csv | cut -c1-200 | head -40
