#!/usr/bin/env python3
"""What does the kernel in front of a streaming solver kernel cost it?  The same 3-vector pn_rk_stage launch
(8 MiB fp32 vectors, HIP start/stop events bound to the dispatch) is timed behind (a) another streaming
kernel with non-temporal stores, (b) a Linear(512,512) GEMM at batch 4096 whose 8 MiB output it then reads,
(c) the same GEMM but reading other data, (d) a tiny kernel."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pnode_amd import _lib
from pnode_amd.petsc_adjoint import HipVecOps
lib = _lib.load(); dev = torch.device("cuda:0")
n = 4096 * 512
ops = HipVecOps(dev, torch.float32, n)
u, k, y, z = (torch.randn(n, device=dev) for _ in range(4))
X = torch.randn(4096, 512, device=dev); W = torch.randn(512, 512, device=dev) * 0.02
out = torch.empty(4096, 512, device=dev)
tiny = torch.zeros(64, device=dev)
K = len(_lib.KERNEL_IDS)


def measure(name, before, src):
    for _ in range(20):
        before(); ops.rk_stage(y, u, [src], [0.5])
    torch.cuda.synchronize(); lib.pn_prof_enable(1)
    for _ in range(300):
        before(); ops.rk_stage(y, u, [src], [0.5])
    torch.cuda.synchronize()
    L = (ctypes.c_int64 * K)(); us = (ctypes.c_double * K)(); by = (ctypes.c_double * K)()
    lib.pn_prof_collect(len(L), L, us, by); lib.pn_prof_enable(0)
    i = _lib.KERNEL_IDS.index("pn_rk_stage")
    print("%-64s pn_rk_stage avg %.2f us  (%.2f TB/s)" % (name, us[i] / L[i], 3 * n * 4 / (us[i] / L[i]) / 1e6), flush=True)


measure("(a) behind a streaming kernel (copy, nt stores)", lambda: ops.copy(z, k), k)
measure("(b) behind the GEMM whose output it reads", lambda: torch.mm(X, W, out=out), out.view(-1))
measure("(c) behind the same GEMM, reading other data", lambda: torch.mm(X, W, out=out), k)
measure("(d) behind a tiny elementwise kernel", lambda: tiny.add_(1.0), k)
