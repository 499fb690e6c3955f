#!/usr/bin/env python3
"""Floor of the solver kernels of one rk4 time step at C3a (8 MiB fp32 vectors), measured in isolation.

One time step launches 6 three-vector and 2 six-vector pn_lincomb kernels, every one of them directly behind a
kernel of func (forward sweep: the last Linear's GEMM; reverse sweep: the last kernel of the stage VJP, a
weight-gradient GEMM or a bias-gradient reduction).  For each (predecessor, launch shape) pair this tool times
the launch with HIP start/stop events bound to the dispatch (the events bench.py uses) while the queue is kept
saturated, and it times a 64 KiB launch of the same kernel in the same position: that one moves no data worth
mentioning, so its duration is the dispatch floor of that position.  Output: a table and the floor of the sum
  T_step >= 3*t(gemm_fwd, 3v) + t(gemm_fwd, 6v) + 3*t(vjp_tail, 3v) + t(vjp_tail, 6v)."""
import ctypes, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pnode_amd import _lib
from pnode_amd.petsc_adjoint import HipVecOps
lib = _lib.load(); dev = torch.device("cuda:0")
B, D = 4096, 512
n = B * D
ops = HipVecOps(dev, torch.float32, n)
small = HipVecOps(dev, torch.float32, 16384)
vecs = [torch.randn(n, device=dev) for _ in range(8)]
X = torch.randn(B, D, device=dev); W = torch.randn(D, D, device=dev) * 0.02; bias = torch.zeros(D, device=dev)
G = torch.randn(B, D, device=dev)
out = torch.empty(B, D, device=dev); dW = torch.empty(D, D, device=dev); db = torch.empty(D, device=dev)
K = len(_lib.KERNEL_IDS)
preds = {
    "streaming kernel (pn_copy, nt stores)": lambda: ops.copy(vecs[7], vecs[6]),
    "forward GEMM + bias (addmm 4096x512x512)": lambda: torch.addmm(bias, X, W, out=out),
    "weight-gradient GEMM (X^T G, 512x4096x512)": lambda: torch.mm(X.t(), G, out=dW),
    "bias-gradient reduction (sum over batch)": lambda: torch.sum(G, 0, out=db),
}
shapes = {
    "tiny (64 KiB)": lambda: small.rk_stage(vecs[5], vecs[0], [vecs[1]], [0.5]),
    "3-vector (24 MiB)": lambda: ops.rk_stage(vecs[5], vecs[0], [vecs[1]], [0.5]),
    "6-vector (48 MiB)": lambda: ops.rk_stage(vecs[5], vecs[0], vecs[1:5], [0.1, 0.2, 0.2, 0.1]),
}


def measure(before, launch, reps=400):
    for _ in range(30):
        before(); launch()
    torch.cuda.synchronize(); lib.pn_prof_enable(1)
    for _ in range(reps):
        before(); launch()
    torch.cuda.synchronize()
    L = (ctypes.c_int64 * K)(); us = (ctypes.c_double * K)(); by = (ctypes.c_double * K)()
    lib.pn_prof_collect(len(L), L, us, by); lib.pn_prof_enable(0)
    i = _lib.KERNEL_IDS.index("pn_rk_stage")
    return us[i] / L[i]


res = {}
for rep in range(3):
    for pn, pf in preds.items():
        for sn, sf in shapes.items():
            res.setdefault((pn, sn), []).append(measure(pf, sf))
print("%-46s" % "predecessor \\ launch" + "".join("%20s" % s for s in shapes))
T = {}
for pn in preds:
    row = []
    for sn in shapes:
        T[(pn, sn)] = statistics.median(res[(pn, sn)])
        row.append("%17.2f us" % T[(pn, sn)])
    print("%-46s" % pn + "".join(row), flush=True)
g, v, r = list(preds)[1], list(preds)[2], list(preds)[3]
for tail_name, tail in (("weight-gradient GEMM", v), ("bias-gradient reduction", r)):
    full = 3 * T[(g, "3-vector (24 MiB)")] + T[(g, "6-vector (48 MiB)")] + 3 * T[(tail, "3-vector (24 MiB)")] + T[(tail, "6-vector (48 MiB)")]
    floor = 4 * T[(g, "tiny (64 KiB)")] + 4 * T[(tail, "tiny (64 KiB)")]
    print("VJP tail = %-24s: sum of the 8 launches in place %.1f us (%.3f of 8 TB/s on 32*N*w); "
          "the same 8 positions with 64 KiB launches (pure dispatch) %.1f us" % (tail_name, full, 32.0 * n * 4 / full / 1e3 / 8000, floor))
s = list(preds)[0]
ideal = 6 * T[(s, "3-vector (24 MiB)")] + 2 * T[(s, "6-vector (48 MiB)")]
print("behind a streaming kernel everywhere (no func in between): %.1f us (%.3f)" % (ideal, 32.0 * n * 4 / ideal / 1e3 / 8000))
