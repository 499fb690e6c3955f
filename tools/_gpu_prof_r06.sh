# Round-6 profile collection (one MI355X).  Raw traces stay in /tmp; summaries go to gpurun_out/prof_r06/.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_r06
mkdir -p $O
cd $R
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
cd /tmp && export TMPDIR=/tmp
# (1) headline: the default-constructed solver under the profiler: the 10 timed replays only
rm -rf /tmp/p_graph
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_graph -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-variants --no-roofline-pass --no-rocprof --no-pmc --no-ceiling > $O/graph_bench.log 2>&1
cp $(find /tmp/p_graph -name "*kernel_stats.csv" | head -1) $O/r06_graph_run_kernel_stats.csv
python3 $R/tools/trace_stats.py /tmp/p_graph $O/r06_graph_timed_region.csv --last-solves 10 --total-solves 16 --time-steps 100 --label "bench.py --steps 10 --warmup 2 (default options: -pn_graph_capture auto; tapes retained; grouped pn_linear_wgrad launches on the sweep's stream): the 10 timed replays only" > /dev/null
# (1b) the same in fp64
rm -rf /tmp/p_graph64
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_graph64 -- python3 $R/bench.py --dtype f64 --steps 10 --warmup 2 --no-cpu-baseline --no-variants --no-roofline-pass --no-rocprof --no-pmc --no-ceiling > $O/graph_bench_f64.log 2>&1
python3 $R/tools/trace_stats.py /tmp/p_graph64 $O/r06_graph_timed_region_f64.csv --last-solves 10 --total-solves 16 --time-steps 100 --label "bench.py --dtype f64 --steps 10 --warmup 2: the 10 timed replays only" > /dev/null
# (2) the adaptive workload under the profiler (GPU-busy share of the wall time)
rm -rf /tmp/p_stiff
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_stiff -- python3 $R/bench.py --config c3b --stiff --steps 3 --warmup 1 --no-cpu-baseline --no-variants --no-roofline-pass --no-rocprof --no-pmc --no-ceiling > $O/stiff_trace.log 2>&1
python3 $R/tools/trace_stats.py /tmp/p_stiff $O/r06_c3b_stiff_trace_stats.csv --label "rocprofv3 --kernel-trace --stats -- python3 bench.py --config c3b --stiff --steps 3 --warmup 1, whole run" > /dev/null
cd $R
# (3) bench lines of the final tree
timeout 1200 python bench.py > $O/r06_bench.json 2> $O/r06_bench.err; echo "rc $?" >> $O/r06_bench.err
timeout 900 python bench.py --dtype f64 --no-variants --no-cpu-baseline > $O/r06_bench_f64.json 2> $O/r06_bench_f64.err; echo "rc $?" >> $O/r06_bench_f64.err
timeout 900 python bench.py --config c3b --stiff --steps 5 --warmup 2 > $O/r06_bench_c3b_stiff.json 2> $O/r06_bench_c3b_stiff.err; echo "rc $?" >> $O/r06_bench_c3b_stiff.err
for c in c2 c3b c4 c5; do timeout 900 python bench.py --config $c --steps 5 --warmup 2 > $O/r06_bench_$c.json 2> $O/r06_bench_$c.err; echo "rc $?" >> $O/r06_bench_$c.err; done
for i in 1 2 3; do timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-variants >> $O/r06_bench_repeat.jsonl 2>> $O/r06_bench_repeat.err; done
# (4) robustness runs of the final tree
timeout 900 python tools/fuzz_guard.py 100 5 24 > $O/r06_fuzz_guard.txt 2>&1
timeout 1500 python tools/fuzz_modes.py 300 12 > $O/r06_fuzz_modes.txt 2>&1
timeout 900 python tools/fuzz_imex.py > $O/r06_fuzz_imex.txt 2>&1
ITERS=100 timeout 600 python tools/soak_graph.py > $O/r06_soak_graph.txt 2>&1
timeout 900 python tools/leak_check.py > $O/r06_leak_check.txt 2>&1
timeout 600 python tools/profile_stiff_reverse.py > $O/r06_stiff_reverse_host.txt 2>&1
timeout 600 python tools/accuracy_c3a.py > $O/r06_accuracy.txt 2>&1
# (5) the fused dW + db MFMA kernel: the ABI loop (single and grouped launches), the structure microbenchmarks, the ladder
{ echo "== LD_LIBRARY_PATH=pnode_amd/lib tools/mb_wgrad_abi (twice: the first lines of a process run on ramping clocks)"; LD_LIBRARY_PATH=pnode_amd/lib timeout 120 ./tools/mb_wgrad_abi; LD_LIBRARY_PATH=pnode_amd/lib timeout 120 ./tools/mb_wgrad_abi;
  echo "== MB_EXACT=1 tools/mb_wgrad_abi (the grouped launches on the fp32 matrix instruction)"; MB_EXACT=1 LD_LIBRARY_PATH=pnode_amd/lib timeout 120 ./tools/mb_wgrad_abi | tail -2;
  echo "== MB_F64=1 tools/mb_wgrad_abi"; MB_F64=1 LD_LIBRARY_PATH=pnode_amd/lib timeout 120 ./tools/mb_wgrad_abi | tail -2;
  echo "== tools/mb_wgrad_bf16x3 (the fp32 product on the bf16 matrix cores: terms, split, loop order; error against float64)"; timeout 200 ./tools/mb_wgrad_bf16x3 | sed -n '/pass 1/,$p';
  echo "== tools/mb_mfma_ladder (what each ingredient of an LDS-staged fp32 MFMA loop costs)"; timeout 200 ./tools/mb_mfma_ladder | sed -n '/pass 1/,$p';
  echo "== tools/mb_wgrad6 (loop structures of the product, partial-tile read-modify-write included, no bias)"; timeout 200 ./tools/mb_wgrad6 | sed -n '/pass 1/,$p'; } > $O/r06_microbench.txt 2>&1
bash tools/_gpu_pmc_wgrad.sh > /dev/null 2>&1; cp gpurun_out/pmc_wgrad/r06_pmc_wgrad.txt $O/
tail -3 $O/r06_graph_timed_region.csv
for f in $O/r06_bench*.json; do echo $f; head -c 250 $f; echo; done
tail -2 $O/r06_fuzz_guard.txt $O/r06_fuzz_modes.txt $O/r06_fuzz_imex.txt $O/r06_soak_graph.txt $O/r06_leak_check.txt
