# Round-4 profile collection (one MI355X).  Raw traces stay in /tmp; summaries go to gpurun_out/prof_r04/.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_r04
mkdir -p $O
cd $R
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
# (1) the adaptive workload that adapts (bench.py --config c3b --stiff): rocprofv3 kernel trace + stats of the whole command, 3 solves
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_stiff
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_stiff -- python3 $R/bench.py --config c3b --stiff --steps 3 --warmup 1 --no-cpu-baseline --no-variants --no-roofline-pass --no-rocprof --no-pmc > $O/stiff_trace.log 2>&1
cp $(find /tmp/p_stiff -name "*kernel_stats.csv" | head -1) $O/r04_c3b_stiff_kernel_stats.csv
python3 $R/tools/trace_stats.py /tmp/p_stiff $O/r04_c3b_stiff_trace_stats.csv --label "rocprofv3 --kernel-trace --stats -- python3 bench.py --config c3b --stiff --steps 3 --warmup 1 (4 solves of 161 accepted + 207 rejected dopri5 attempts, 4096 x 512 fp32, max_cps 50), whole run" > /dev/null
# (2) headline: the default-constructed solver (-pn_graph_capture auto) under the profiler: the 10 timed replays only
rm -rf /tmp/p_graph
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_graph -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-variants --no-roofline-pass --no-rocprof --no-pmc > $O/graph_bench.log 2>&1
cp $(find /tmp/p_graph -name "*kernel_stats.csv" | head -1) $O/r04_graph_run_kernel_stats.csv
python3 $R/tools/trace_stats.py /tmp/p_graph $O/r04_graph_timed_region.csv --last-solves 10 --total-solves 16 --time-steps 100 --label "bench.py --steps 10 --warmup 2 (no launch option: -pn_graph_capture auto; tapes retained): the 10 timed replays only" > /dev/null
cd $R
# (3) bench lines of the final tree
timeout 900 python bench.py --steps 20 --warmup 5 > $O/r04_bench.json 2> $O/r04_bench.err; echo "rc $?" >> $O/r04_bench.err
timeout 900 python bench.py --config c3b --stiff --steps 5 --warmup 2 > $O/r04_bench_c3b_stiff.json 2> $O/r04_bench_c3b_stiff.err; echo "rc $?" >> $O/r04_bench_c3b_stiff.err
for c in c2 c3b c4 c5; do timeout 900 python bench.py --config $c --steps 5 --warmup 2 > $O/r04_bench_$c.json 2> $O/r04_bench_$c.err; echo "rc $?" >> $O/r04_bench_$c.err; done
# (4) host cost of eager launches: C++ step loops against the Python stage loop
timeout 300 python tools/host_overhead.py > $O/r04_host_overhead.txt 2>&1
# (5) C5 shard, theta methods: the captured product in the eager path's arithmetic form (default) and with the double-VJP form forced
timeout 1200 python tools/bench_c5_theta.py --only-default > $O/r04_c5_theta_default.txt 2>&1
timeout 1200 python tools/bench_c5_theta.py --only-default --form dvjp > $O/r04_c5_theta_dvjp.txt 2>&1
timeout 1500 python tools/bench_c5_theta.py --only-default --conv1d > $O/r04_c5_theta_conv1d.txt 2>&1
# (6) recompute with tapes (difference 20) and the robustness runs of the final tree
timeout 900 python tools/bench_recompute_tapes.py > $O/r04_recompute_tapes.txt 2>&1
timeout 900 python tools/prof_stiff_phases.py > $O/r04_stiff_phases.txt 2>&1
timeout 900 python tools/fuzz_modes.py 300 11 > $O/r04_fuzz_modes.txt 2>&1
timeout 900 python tools/fuzz_imex.py > $O/r04_fuzz_imex.txt 2>&1
ITERS=100 timeout 600 python tools/soak_graph.py > $O/r04_soak_graph.txt 2>&1
timeout 900 python tools/leak_check.py > $O/r04_leak_check.txt 2>&1
grep -h combine $O/r04_c3b_stiff_trace_stats.csv
tail -3 $O/r04_graph_timed_region.csv
for f in $O/r04_bench*.json; do echo $f; head -c 300 $f; echo; done
cat $O/r04_host_overhead.txt
grep -h "C5 shard" $O/r04_c5_theta_*.txt
