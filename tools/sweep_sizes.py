#!/usr/bin/env python3
"""Duration and moved GB/s of the streaming kernel (3-vector stage launch and 6-vector combine
launch, fp32) as a function of the vector size, operands rotating through a > 2 GiB slab (cold),
HIP start/stop events per dispatch.  Writes gpurun_out/size_sweep.json."""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pnode_amd import _lib
from pnode_amd.petsc_adjoint import HipVecOps
dev = torch.device("cuda:0")
lib = _lib.load()
out = []
if len(sys.argv) > 1:
    lib.pn_tune_set(sys.argv[1].encode())
for logn in (range(14, 28) if len(sys.argv) <= 2 else range(int(sys.argv[2]), 28)):
    n = 1 << logn
    slots = max(16, min(4096, (3 << 30) // (4 * n)))
    slab = torch.zeros(slots, n, device=dev)
    ops = HipVecOps(dev, torch.float32, n)
    row = {"n": n, "MiB_per_vector": 4 * n / 2 ** 20}
    for nin, name in ((2, "3-vector"), (5, "6-vector")):
        reps = 200 if logn < 22 else 60
        k = 0
        def launch():
            global k
            xs = [slab[(k + j) % slots] for j in range(nin)]
            ops.rk_stage(slab[(k + nin) % slots], xs[0], xs[1:], [0.5] * (nin - 1))
            k += nin + 1
        for _ in range(10):
            launch()
        torch.cuda.synchronize()
        lib.pn_prof_enable(1)
        for _ in range(reps):
            launch()
        L = (ctypes.c_int64 * len(_lib.KERNEL_IDS))(); us = (ctypes.c_double * len(_lib.KERNEL_IDS))(); by = (ctypes.c_double * len(_lib.KERNEL_IDS))()
        lib.pn_prof_collect(len(L), L, us, by)
        lib.pn_prof_enable(0)
        row[name] = {"avg_us": us[0] / L[0], "GBps": by[0] / us[0] / 1e3, "frac_of_8TBps": by[0] / us[0] / 1e3 / 8000}
    out.append(row)
    print("%9d elems %8.2f MiB | 3-vec %7.2f us %7.1f GB/s | 6-vec %7.2f us %7.1f GB/s" % (
        n, row["MiB_per_vector"], row["3-vector"]["avg_us"], row["3-vector"]["GBps"], row["6-vector"]["avg_us"], row["6-vector"]["GBps"]), flush=True)
    del slab
    torch.cuda.empty_cache()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "size_sweep.json"), "w"), indent=1)
