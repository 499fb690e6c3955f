"""Microbenchmark (round 5): ways to form the bias sensitivity mu_b += alpha * colsum(G) for G = 4096 x 512 fp32 (8 MiB):
PyTorch's reduction (what autograd runs), rocBLAS / hipBLASLt through torch, pn_colsum_accum one source at a time, and
pn_colsum_accum_multi over the 16 cotangents of an rk4 time step (4 layers x 4 stages) and over 32.  GPU time per SOURCE."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pnode_amd.petsc_adjoint import HipVecOps
dev = torch.device("cuda:0")
for rows, cols, dtype in ((4096, 512, torch.float32), (4096, 512, torch.float64), (64, 1152, torch.float64)):
    ops = HipVecOps(dev, dtype, 64)
    Gs = [torch.randn(rows, cols, device=dev, dtype=dtype) for _ in range(32)]
    mu = torch.zeros(cols, device=dev, dtype=dtype)
    mus = [torch.zeros(cols, device=dev, dtype=dtype) for _ in range(4)]
    ones = torch.ones(rows, device=dev, dtype=dtype)
    def t(fn, reps=100, per=1):
        for k in range(5): fn(k)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for k in range(reps): fn(k)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3 / per
    print("%d x %d %s, us per source:" % (rows, cols, dtype))
    print("   G.sum(0) + add_                 %6.2f" % t(lambda k: mu.add_(Gs[k % 32].sum(0), alpha=0.5)))
    print("   addmv_(G^T, ones)               %6.2f" % t(lambda k: torch.addmv(mu, Gs[k % 32].t(), ones, beta=1.0, alpha=0.5, out=mu)))
    print("   pn_colsum_accum (1 source)      %6.2f" % t(lambda k: ops.colsum_accum(Gs[k % 32], mu, 0.5)))
    for n in (4, 16, 32):
        items = [(Gs[j], mus[j % 4], 0.5) for j in range(n)]
        us = t(lambda k: ops.colsum_accum_multi(items), reps=40, per=n)
        print("   pn_colsum_accum_multi (%2d)      %6.2f   = %.0f GB/s" % (n, us, rows * cols * Gs[0].element_size() / (us * 1e-6) / 1e9))
