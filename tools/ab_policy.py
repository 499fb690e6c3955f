#!/usr/bin/env python3
"""Interleaved A/B timing of kernel policies on the target workload, in ONE process
(cdna_hip_programming.md rule 24).  Usage: python tools/ab_policy.py [--rounds R] cfg1 cfg2 ...
Each cfg is a PN_TUNE spec ("" = defaults).  Prints, per cfg, the median over rounds of the
solver-kernel time per rk4 time step and the per-entry-point average launch durations."""
import argparse
import ctypes
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from pnode_amd import _lib, options, petsc_adjoint  # noqa: E402
from problems import MLPFunc  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=6)
ap.add_argument("--nt", type=int, default=100)
ap.add_argument("--batch", type=int, default=4096)
ap.add_argument("--dim", type=int, default=512)
ap.add_argument("cfgs", nargs="+")
args = ap.parse_args()

lib = _lib.load()
dev = torch.device("cuda:0")
options.set_option("ts_adapt_type", "none")
options.set_option("ts_trajectory_solution_only", "0")
torch.manual_seed(0)
func = MLPFunc(args.dim, torch.float32).to(dev)
y0 = torch.randn(args.batch, args.dim, device=dev)
t = torch.tensor([0.01 * args.nt])
ode = petsc_adjoint.ODEPetsc()
ode.setupTS(y0, func, step_size=0.01, method="rk4")


def solve():
    for p in func.parameters():
        p.grad = None
    y = y0.detach().requires_grad_(True)
    ode.odeint_adjoint(y, t).abs().mean().backward()


solve()
res = {c: [] for c in args.cfgs}
for r in range(args.rounds):
    for c in args.cfgs:
        lib.pn_tune_set(c.encode() if c else None)
        torch.cuda.synchronize()
        lib.pn_prof_enable(1)
        solve()
        torch.cuda.synchronize()
        L = (ctypes.c_int64 * len(_lib.KERNEL_IDS))()
        us = (ctypes.c_double * len(_lib.KERNEL_IDS))()
        by = (ctypes.c_double * len(_lib.KERNEL_IDS))()
        lib.pn_prof_collect(len(L), L, us, by)
        lib.pn_prof_enable(0)
        res[c].append([us[i] / max(L[i], 1) for i in range(len(_lib.KERNEL_IDS))] + [(us[0] + us[2] + us[3]) / args.nt])
n = args.batch * args.dim
for c in args.cfgs:
    med = [statistics.median(x[i] for x in res[c]) for i in range(len(_lib.KERNEL_IDS) + 1)]
    mn = min(x[-1] for x in res[c])
    print("%-30s us/step median %6.2f min %6.2f -> %6.1f GB/s frac %.3f | stage %.2f theta %.2f accum %.2f param %.2f"
          % (c or "(defaults)", med[-1], mn, 32 * n * 4 / med[-1] / 1e3, 32 * n * 4 / med[-1] / 1e3 / 8000, med[0], med[2], med[3], med[4]))
