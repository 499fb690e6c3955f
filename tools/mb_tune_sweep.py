#!/usr/bin/env python3
"""Interleaved sweep of the launch policy knobs (pn_tune_set) on the two launch shapes of the headline config
(3 vectors = stage/theta, 6 vectors = combine/accum; 8 MiB fp32 each), each timed behind a GEMM and behind a
streaming kernel.  6 rounds x 200 launches per cell, medians."""
import ctypes, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pnode_amd import _lib
from pnode_amd.petsc_adjoint import HipVecOps
lib = _lib.load(); dev = torch.device("cuda:0")
n = 4096 * 512
ops = HipVecOps(dev, torch.float32, n)
u, y, z = (torch.randn(n, device=dev) for _ in range(3))
ks = [torch.randn(n, device=dev) for _ in range(4)]
X = torch.randn(4096, 512, device=dev); W = torch.randn(512, 512, device=dev) * 0.02
out = torch.empty(4096, 512, device=dev)
K = len(_lib.KERNEL_IDS); I = _lib.KERNEL_IDS.index("pn_rk_stage")
cfgs = sys.argv[1:] or ["", "vpt=1", "vpt=4", "cap=2048", "cap=1024 vpt=4", "st=0", "ld=1"]
shapes = {"3vec": lambda: ops.rk_stage(y, u, [ks[0]], [0.5]), "6vec": lambda: ops.rk_stage(y, u, ks, [0.1, 0.2, 0.3, 0.4])}
befores = {"gemm": lambda: torch.mm(X, W, out=out), "stream": lambda: ops.copy(z, ks[0])}
res = {}
for r in range(6):
    for c in cfgs:
        lib.pn_tune_set(c.encode() if c else None)
        for sn, sf in shapes.items():
            for bn, bf in befores.items():
                for _ in range(10): bf(); sf()
                torch.cuda.synchronize(); lib.pn_prof_enable(1)
                for _ in range(200): bf(); sf()
                torch.cuda.synchronize()
                L = (ctypes.c_int64 * K)(); us = (ctypes.c_double * K)(); by = (ctypes.c_double * K)()
                lib.pn_prof_collect(len(L), L, us, by); lib.pn_prof_enable(0)
                res.setdefault((c, sn, bn), []).append(us[I] / L[I])
lib.pn_tune_set(None)
print("%-18s %12s %12s %12s %12s   per time step (3x stage + combine behind GEMM, 3x theta + accum behind stream)" % ("policy", "3vec|gemm", "3vec|stream", "6vec|gemm", "6vec|stream"))
for c in cfgs:
    m = {k: statistics.median(res[(c,) + k]) for k in (("3vec", "gemm"), ("3vec", "stream"), ("6vec", "gemm"), ("6vec", "stream"))}
    step = 3 * m[("3vec", "gemm")] + m[("6vec", "gemm")] + 3 * m[("3vec", "stream")] + m[("6vec", "stream")]
    print("%-18s %12.2f %12.2f %12.2f %12.2f   %.2f us -> frac %.3f" % (c or "(default)", m[("3vec", "gemm")], m[("3vec", "stream")], m[("6vec", "gemm")], m[("6vec", "stream")], step, 32 * n * 4 / step / 1e3 / 8000))
