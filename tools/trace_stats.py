#!/usr/bin/env python3
"""Per-kernel statistics from a rocprofv3 --kernel-trace CSV, optionally restricted to the LAST part of
the run (the timed region of bench.py is the last thing the process does when --no-roofline-pass
--no-variants --no-cpu-baseline are given), so that a graph-replayed region can be summarised without the
eager warm-up launches that precede it.

  trace_stats.py <dir with *_kernel_trace.csv> <out.csv> [--last-solves K --total-solves M] [--label text]

Output columns: kernel (pn_* kernels by full template name, everything else grouped by leading name),
calls, total_us, avg_us, min_us, max_us, share of GPU kernel time.  Also prints the pn_* sum per time step
when --time-steps is given."""
import argparse
import csv
import glob
import re
import statistics
import sys


def short(name):
    if "pn_" in name:
        n = name[name.index("pn_"):]
        return n.split("(")[0]
    n = re.sub(r"^void ", "", name)
    n = re.split(r"[<(]", n)[0]
    return n[:70]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir"); ap.add_argument("out")
    ap.add_argument("--last-solves", type=int, default=0); ap.add_argument("--total-solves", type=int, default=0)
    ap.add_argument("--time-steps", type=int, default=0, help="time steps per solve (for the per-time-step sums)")
    ap.add_argument("--label", default="")
    a = ap.parse_args()
    rows = []
    for f in glob.glob(a.dir + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    if not rows:
        sys.exit("no kernel trace found under " + a.dir)
    if a.last_solves and a.total_solves:
        # cut at the pn_* launch count: every solve launches the same number of pn_lincomb kernels
        idx = [i for i, r in enumerate(rows) if "pn_lincomb_kernel" in r[2]]
        if len(idx) % a.total_solves:
            sys.exit("%d pn_lincomb launches do not divide into %d solves: wrong --total-solves (graph mode: 2 eager calls + the "
                     "capturing call, which replays once + warm-up + timed)" % (len(idx), a.total_solves))
        per = len(idx) // a.total_solves
        first = idx[len(idx) - per * a.last_solves]
        # the solve starts a little before its first pn_ launch (pn_copy of u0 is the first one)
        rows = rows[first:]
    per = {}
    for s, e, n in rows:
        per.setdefault(short(n), []).append((e - s) / 1e3)
    tot = sum(sum(v) for v in per.values())
    span = (rows[-1][1] - rows[0][0]) / 1e3
    with open(a.out, "w") as fh:
        fh.write("# %s\n" % a.label)
        fh.write("# kernels %d, sum of kernel durations %.1f us, first start -> last end %.1f us\n" % (len(rows), tot, span))
        w = csv.writer(fh)
        w.writerow(["kernel", "calls", "total_us", "avg_us", "median_us", "min_us", "max_us", "share"])
        for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
            w.writerow([k, len(v), "%.1f" % sum(v), "%.3f" % (sum(v) / len(v)), "%.3f" % statistics.median(v),
                        "%.3f" % min(v), "%.3f" % max(v), "%.4f" % (sum(v) / tot)])
        if a.time_steps and a.last_solves:
            nts = a.time_steps * a.last_solves
            vec = sum(sum(v) for k, v in per.items() if k.startswith("pn_lincomb_kernel") and not k.startswith("pn_lincomb_kernel<float, 1,")
                      and not k.startswith("pn_lincomb_kernel<double, 1,"))
            par = sum(sum(v) for k, v in per.items() if k.startswith(("pn_param_accum", "pn_colsum")))
            wgrad = sum(sum(v) for k, v in per.items() if "pn_linear_" in k)        # the fused dW + db MFMA kernel and its finish passes
            allpn = sum(sum(v) for k, v in per.items() if k.startswith("pn_") or "::pn_" in k)
            fh.write("# per time step (%d time steps): state-vector kernels %.2f us, parameter accumulation passes (HBM-bound) %.2f us, "
                     "pn_linear_wgrad (MFMA-bound) %.2f us, all pn_* %.2f us, all kernels %.1f us, wall %.1f us\n"
                     % (nts, vec / nts, par / nts, wgrad / nts, allpn / nts, tot / nts, span / nts))
    print(open(a.out).read()[:3000])


if __name__ == "__main__":
    main()
