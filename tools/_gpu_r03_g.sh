R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_g
mkdir -p $O
cd $R
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
timeout 900 python -m pytest tests/test_gpu_krylov.py -x -q -m gpu > $O/krylov_tests.log 2>&1; echo "krylov rc $?" >> $O/krylov_tests.log
timeout 300 python tools/prof_krylov.py default stencil > $O/prof_krylov_stencil_default.txt 2>&1
timeout 600 python tools/bench_c5_theta.py --only-default > $O/c5_default.txt 2>&1
timeout 900 python tools/bench_c5_theta.py --only-default --tunableop > $O/c5_default_tunableop.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_kr
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_kr -- python3 $R/tools/prof_krylov.py default stencil --trace-only > $O/kr_trace.log 2>&1
python3 - <<PY > $O/r03_krylov_stencil_trace_summary.txt 2>&1
import csv, glob, collections
rows=[]
for f in glob.glob("/tmp/p_kr/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
idx=[i for i,r in enumerate(rows) if "kr_dots_kernel" in r[2]]
print("tools/prof_krylov.py default stencil --trace-only under rocprofv3 --kernel-trace: C5 shard (64 x 1024 fp64), cn, Newton-GMRES,")
print("device-resident GMRES + replayed linearisations; 6 solves in the trace, the last 2 summarised")
print("kernels", len(rows), "kr_dots launches", len(idx))
n=len(idx)//6 if len(idx)>=6 else len(idx)
cut=idx[len(idx)-2*n] if n else 0
sub=rows[cut:]
span=(sub[-1][1]-sub[0][0])/1e3
busy=sum(e-s for s,e,_ in sub)/1e3
print("last 2 solves: kernels %d, sum of durations %.1f us, first start -> last end %.1f us, GPU busy %.1f %%" % (len(sub), busy, span, 100*busy/span))
per=collections.defaultdict(list)
for s,e,nm in sub:
    k=nm
    if "pn_" in k or "kr_" in k:
        k=k[k.index("kr_") if "kr_" in k else k.index("pn_"):].split("(")[0]
    else:
        k=k.split("(")[0][:60]
    per[k].append((e-s)/1e3)
for k,v in sorted(per.items(), key=lambda kv:-sum(kv[1]))[:25]:
    print("%-62s calls %5d total %9.1f us avg %7.2f" % (k[:62], len(v), sum(v), sum(v)/len(v)))
PY
tail -4 $O/krylov_tests.log
head -6 $O/prof_krylov_stencil_default.txt
grep "C5 shard" $O/c5_default.txt $O/c5_default_tunableop.txt
cat $O/r03_krylov_stencil_trace_summary.txt
