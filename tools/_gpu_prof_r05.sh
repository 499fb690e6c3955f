# Round-5 profile collection (one MI355X).  Raw traces stay in /tmp; summaries go to gpurun_out/prof_r05/.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_r05
mkdir -p $O
cd $R
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
cd /tmp && export TMPDIR=/tmp
# (1) headline: the default-constructed solver (-pn_graph_capture auto) under the profiler: the 10 timed replays only
rm -rf /tmp/p_graph
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_graph -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-variants --no-roofline-pass --no-rocprof --no-pmc --no-ceiling > $O/graph_bench.log 2>&1
cp $(find /tmp/p_graph -name "*kernel_stats.csv" | head -1) $O/r05_graph_run_kernel_stats.csv
python3 $R/tools/trace_stats.py /tmp/p_graph $O/r05_graph_timed_region.csv --last-solves 10 --total-solves 16 --time-steps 100 --label "bench.py --steps 10 --warmup 2 (no launch option: -pn_graph_capture auto; tapes retained): the 10 timed replays only" > /dev/null
# (2) the streaming microbenchmark behind roofline.copy_ceiling, the profiler's own statistics
rm -rf /tmp/p_ceil
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_ceil -- python3 $R/bench.py --ceiling-only > $O/ceiling.log 2>&1
cp $(find /tmp/p_ceil -name "*kernel_stats.csv" | head -1) $O/r05_ceiling_kernel_stats.csv
# (3) the adaptive workload under the profiler (GPU-busy share of the wall time)
rm -rf /tmp/p_stiff
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_stiff -- python3 $R/bench.py --config c3b --stiff --steps 3 --warmup 1 --no-cpu-baseline --no-variants --no-roofline-pass --no-rocprof --no-pmc --no-ceiling > $O/stiff_trace.log 2>&1
python3 $R/tools/trace_stats.py /tmp/p_stiff $O/r05_c3b_stiff_trace_stats.csv --label "rocprofv3 --kernel-trace --stats -- python3 bench.py --config c3b --stiff --steps 3 --warmup 1, whole run" > /dev/null
cd $R
# (4) bench lines of the final tree
timeout 900 python bench.py > $O/r05_bench.json 2> $O/r05_bench.err; echo "rc $?" >> $O/r05_bench.err
timeout 900 python bench.py --config c3b --stiff --steps 5 --warmup 2 > $O/r05_bench_c3b_stiff.json 2> $O/r05_bench_c3b_stiff.err; echo "rc $?" >> $O/r05_bench_c3b_stiff.err
for c in c2 c3b c4 c5; do timeout 900 python bench.py --config $c --steps 5 --warmup 2 > $O/r05_bench_$c.json 2> $O/r05_bench_$c.err; echo "rc $?" >> $O/r05_bench_$c.err; done
for i in 1 2 3; do timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-variants >> $O/r05_bench_repeat.jsonl 2>> $O/r05_bench_repeat.err; done
# (5) what the periodic re-validation costs: 230 timed replayed calls contain two re-validating calls (every 100th)
timeout 900 python bench.py --steps 230 --warmup 5 --no-cpu-baseline --no-variants --no-rocprof --no-pmc --no-ceiling --no-roofline-pass > $O/r05_bench_230_steps.json 2> $O/r05_bench_230_steps.err
# (6) robustness runs of the final tree
timeout 900 python tools/fuzz_guard.py 150 5 24 > $O/r05_fuzz_guard.txt 2>&1
timeout 1200 python tools/fuzz_modes.py 300 12 > $O/r05_fuzz_modes.txt 2>&1
timeout 900 python tools/fuzz_imex.py > $O/r05_fuzz_imex.txt 2>&1
ITERS=100 timeout 600 python tools/soak_graph.py > $O/r05_soak_graph.txt 2>&1
timeout 900 python tools/leak_check.py > $O/r05_leak_check.txt 2>&1
timeout 600 python tools/prof_stiff_phases.py > $O/r05_stiff_phases.txt 2>&1
# (7) the fused dW + db MFMA kernel beside the library's GEMM (twice: the first lines of a process run on ramping clocks)
{ echo "== LD_LIBRARY_PATH=pnode_amd/lib tools/mb_wgrad_abi (twice)"; LD_LIBRARY_PATH=pnode_amd/lib timeout 120 ./tools/mb_wgrad_abi; LD_LIBRARY_PATH=pnode_amd/lib timeout 120 ./tools/mb_wgrad_abi;
  echo "== python tools/mb_wgrad_lib.py"; timeout 300 python tools/mb_wgrad_lib.py 2>/dev/null;
  echo "== tools/mb_wgrad (the bare product, tile / split / slab variants)"; timeout 120 ./tools/mb_wgrad; } > $O/r05_microbench_wgrad.txt 2>&1
tail -3 $O/r05_graph_timed_region.csv
for f in $O/r05_bench*.json; do echo $f; head -c 250 $f; echo; done
tail -2 $O/r05_fuzz_guard.txt $O/r05_fuzz_modes.txt $O/r05_fuzz_imex.txt $O/r05_soak_graph.txt $O/r05_leak_check.txt
