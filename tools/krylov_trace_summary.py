#!/usr/bin/env python3
"""Summary of a rocprofv3 --kernel-trace of tools/prof_krylov.py ... --trace-only (6 solves in the trace): the last two solves --
kernels, GPU busy fraction, per-kernel totals.   krylov_trace_summary.py <trace dir>"""
import collections
import csv
import glob
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
idx = [i for i, r in enumerate(rows) if "kr_dots_kernel" in r[2]]
print("tools/prof_krylov.py default stencil --trace-only under rocprofv3 --kernel-trace: C5 shard (64 x 1024 fp64), cn, Newton-GMRES,")
print("device-resident GMRES + replayed linearisations (530 Krylov iterations per solve); 6 solves in the trace, the last 2 summarised")
print("kernels in the trace", len(rows), " kr_dots launches", len(idx))
n = len(idx) // 6 if len(idx) >= 6 else len(idx)
sub = rows[idx[len(idx) - 2 * n]:] if n else rows
span = (sub[-1][1] - sub[0][0]) / 1e3
busy = sum(e - s for s, e, _ in sub) / 1e3
print("last 2 solves: kernels %d, sum of durations %.1f us, first start -> last end %.1f us, GPU busy %.1f %%" % (len(sub), busy, span, 100 * busy / span))
per = collections.defaultdict(list)
for s, e, nm in sub:
    k = nm
    if "pn_" in k or "kr_" in k:
        k = k[k.index("kr_") if "kr_" in k else k.index("pn_"):].split("(")[0]
    else:
        k = k.split("(")[0][:60]
    per[k].append((e - s) / 1e3)
for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1]))[:25]:
    print("%-62s calls %5d total %9.1f us avg %7.2f  share %.3f" % (k[:62], len(v), sum(v), sum(v) / len(v), sum(v) / busy))
