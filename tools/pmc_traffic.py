#!/usr/bin/env python3
"""HBM bytes per launch of the pn_* kernels from two rocprofv3 counter passes (FETCH_SIZE and WRITE_SIZE do
not fit into one pass: MI355X_MICROARCH.md, "rocprofv3 PMC slots").  Units and the gfx950 correction follow
that guide: both counters are KiB; FETCH_SIZE reports half of the bytes of a wide coalesced streaming read
and is doubled, WRITE_SIZE is exact.

    tools/pmc_traffic.sh            # on the GPU box: the two passes + this script
    python tools/pmc_traffic.py <dir with FETCH pass> <dir with WRITE pass> <out.json> [label]
"""
import csv
import glob
import json
import sys


def load(d, counter):
    per = {}
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            name = r["Kernel_Name"]
            if "pn_" not in name:
                continue
            name = name[name.index("pn_"):].split("(")[0]
            per.setdefault(name, []).append(float(r["Counter_Value"]))
    return per


def main():
    fdir, wdir, out = sys.argv[1:4]
    label = sys.argv[4] if len(sys.argv) > 4 else ""
    fetch, write = load(fdir, "FETCH_SIZE"), load(wdir, "WRITE_SIZE")
    kernels, total, launches, ptotal, plaunches = {}, 0.0, 0, 0.0, 0
    for name in sorted(set(fetch) | set(write)):
        fv, wv = fetch.get(name, []), write.get(name, [])
        n = max(len(fv), len(wv))
        fr = sum(fv) / max(len(fv), 1)
        wr = sum(wv) / max(len(wv), 1)
        rb, wb = 2.0 * fr * 1024.0, wr * 1024.0
        kernels[name] = {"launches": n, "FETCH_SIZE_KiB_raw": fr, "read_bytes_corrected": rb, "WRITE_SIZE_KiB": wr,
                         "write_bytes": wb, "hbm_bytes_per_launch": rb + wb}
        if "pn_lincomb_kernel" in name and not name.startswith("pn_lincomb_kernel<float, 1,"):
            total += n * (rb + wb)
            launches += n
        if "pn_param_accum" in name:
            ptotal += n * (rb + wb)
            plaunches += n
    res = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py "
                     "--mode eager --steps 1 --warmup 0 --nt 10 --no-cpu-baseline --no-variants" + (" ; " + label if label else ""),
           "units": "counter values are KiB; FETCH_SIZE is doubled per MI355X_MICROARCH.md (gfx950 reports 1/2 of a wide "
                    "coalesced read stream); WRITE_SIZE is exact",
           "kernels": kernels}
    if launches:
        n, w = 4096 * 512, 4
        per_step = total / launches * 8.0
        nt = launches / 8.0                      # time steps in the trace (8 state-vector launches each)
        npar = 4 * (512 * 512 + 512)
        ppar = ptotal / nt
        res["rk4_time_step"] = {"launches": 8 + plaunches / nt, "hbm_bytes": per_step + ppar,
                                "hbm_bytes_per_launch_avg": (per_step + ppar) / (8 + plaunches / nt),
                                "algorithmic_bytes_survey_8d": 32 * n * w + 12 * npar * w,
                                "moved_over_algorithmic": (per_step + ppar) / (32 * n * w + 12 * npar * w),
                                "state_vector_kernels": {"launches": 8, "hbm_bytes": per_step, "algorithmic_bytes": 32 * n * w,
                                                         "moved_over_algorithmic": per_step / (32 * n * w)},
                                "parameter_accumulation": {"launches": plaunches / nt, "hbm_bytes": ppar,
                                                           "algorithmic_bytes": 12 * npar * w,
                                                           "moved_over_algorithmic": ppar / (12 * npar * w)}}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res.get("rk4_time_step"), indent=1))
    for k, v in kernels.items():
        print("%-62s launches %4d  read %.3f MB  write %.3f MB" % (k[:62], v["launches"], v["read_bytes_corrected"] / 1e6, v["write_bytes"] / 1e6))


if __name__ == "__main__":
    main()
