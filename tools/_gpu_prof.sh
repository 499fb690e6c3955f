# Round-2 profile collection (one MI355X).  Raw traces stay in /tmp; summaries go to gpurun_out/prof_r02/.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_r02
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
set -x
# (1) eager launches only
rm -rf /tmp/p_eager
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_eager -- python3 $R/bench.py --mode eager --steps 3 --warmup 1 --no-cpu-baseline --no-variants --no-roofline-pass > $O/eager_bench.log 2>&1
cp $(find /tmp/p_eager -name "*kernel_stats.csv" | head -1) $O/r02_eager_kernel_stats.csv
python3 $R/tools/trace_stats.py /tmp/p_eager $O/r02_eager_timed_region.csv --last-solves 3 --total-solves 4 --time-steps 100 --label "bench.py --mode eager --steps 3 --warmup 1: the 3 timed solves" > /dev/null
# (2) the default (graph-replayed) timed region only: 2 eager warm-up + the capturing call (replays once) + 2 warm-up replays + 10 timed replays
rm -rf /tmp/p_graph
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_graph -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-variants --no-roofline-pass > $O/graph_bench.log 2>&1
cp $(find /tmp/p_graph -name "*kernel_stats.csv" | head -1) $O/r02_graph_run_kernel_stats.csv
python3 $R/tools/trace_stats.py /tmp/p_graph $O/r02_graph_timed_region.csv --last-solves 10 --total-solves 15 --time-steps 100 --label "bench.py --steps 10 --warmup 2 (default: hipGraph replay, tapes retained): the 10 timed replays only" > /dev/null
# (3) the default command as the driver runs it (graph region + eager event pass + variants)
rm -rf /tmp/p_default
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_default -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/default_bench_under_rocprof.log 2>&1
cp $(find /tmp/p_default -name "*kernel_stats.csv" | head -1) $O/r02_default_kernel_stats.csv
# (4) PMC: HBM bytes (separate passes), then L2 hit/miss
rm -rf /tmp/pmc_f /tmp/pmc_w /tmp/pmc_l2
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_f -- python3 $R/bench.py --mode eager --steps 1 --warmup 0 --nt 16 --no-cpu-baseline --no-variants --no-roofline-pass > /tmp/pmc_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_w -- python3 $R/bench.py --mode eager --steps 1 --warmup 0 --nt 16 --no-cpu-baseline --no-variants --no-roofline-pass > /tmp/pmc_w.log 2>&1
python3 $R/tools/pmc_traffic.py /tmp/pmc_f /tmp/pmc_w $O/r02_pmc_traffic.json "round 2: batched parameter accumulation, tapes retained" > $O/pmc_traffic.txt 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d /tmp/pmc_l2 -- python3 $R/bench.py --mode eager --steps 1 --warmup 0 --nt 16 --no-cpu-baseline --no-variants --no-roofline-pass > /tmp/pmc_l2.log 2>&1
python3 - <<PY > $O/r02_pmc_l2.txt 2>&1
import csv, glob
per = {}
for f in glob.glob("/tmp/pmc_l2/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "pn_" not in n: continue
        n = n[n.index("pn_"):].split("(")[0]
        per.setdefault(n, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
print("L2 (TCC) hit rate of the solver kernels in place, rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum, bench.py --mode eager --nt 16")
for n, c in sorted(per.items()):
    h, m = sum(c.get("TCC_HIT_sum", [0])), sum(c.get("TCC_MISS_sum", [0]))
    print("%-60s launches %4d  hit %.3e  miss %.3e  hit rate %.3f" % (n[:60], len(c.get("TCC_HIT_sum", [])), h, m, h / max(h + m, 1)))
PY
# (5) un-profiled bench lines for reference
python3 $R/bench.py --steps 10 --warmup 3 > $O/r02_bench.json 2> $O/r02_bench.err
python3 $R/bench.py --config c4 --steps 10 --warmup 3 > $O/r02_bench_c4.json 2> $O/r02_bench_c4.err
tail -3 $O/*.log | tail -40
ls -la $O
