#!/usr/bin/env python3
"""In-situ A/B of kernel policies on the larger-vector configs (C4 conv shard: 32 MiB vectors;
C3a fp64: 16 MiB vectors; C3a fp32: 8 MiB), interleaved in one process.  usage: ab_configs.py cfg1 cfg2 ..."""
import ctypes, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from pnode_amd import _lib, options, petsc_adjoint
from problems import MLPFunc
lib = _lib.load(); dev = torch.device("cuda:0")
cfgs = sys.argv[1:]
import importlib.util
spec = importlib.util.spec_from_file_location("bc", os.path.join(ROOT, "tools", "bench_configs.py"))
src = open(os.path.join(ROOT, "tools", "bench_configs.py")).read().split("out = []")[0]
ns = {"__file__": os.path.join(ROOT, "tools", "bench_configs.py")}; exec(compile(src, "bench_configs_head", "exec"), ns)
ConvBlock = ns["ConvBlock"]
options.set_option("ts_adapt_type", "none"); options.set_option("ts_trajectory_solution_only", "0")
torch.manual_seed(0)
cases = [("C4 32MiB f32", ConvBlock(64).to(dev), torch.randn(128, 64, 32, 32, device=dev), torch.tensor([1.0]), 0.25),
         ("C3a 16MiB f64", MLPFunc(512, torch.float64).to(dev), torch.randn(4096, 512, device=dev, dtype=torch.float64), torch.tensor([0.2]), 0.01),
         ("C3a 8MiB f32", MLPFunc(512, torch.float32).to(dev), torch.randn(4096, 512, device=dev), torch.tensor([0.2]), 0.01)]
for name, f, y0, t, h in cases:
    ode = petsc_adjoint.ODEPetsc(); ode.setupTS(y0, f, step_size=h, method="rk4")
    def solve():
        for p in f.parameters(): p.grad = None
        y = y0.detach().requires_grad_(True); ode.odeint_adjoint(y, t).abs().mean().backward()
    solve(); solve()
    res = {c: [] for c in cfgs}
    for r in range(6):
        for c in cfgs:
            lib.pn_tune_set(c.encode() if c else None)
            torch.cuda.synchronize(); lib.pn_prof_enable(1); solve(); torch.cuda.synchronize()
            K = len(_lib.KERNEL_IDS); L = (ctypes.c_int64 * K)(); us = (ctypes.c_double * K)(); by = (ctypes.c_double * K)()
            lib.pn_prof_collect(len(L), L, us, by); lib.pn_prof_enable(0)
            res[c].append((us[0] + us[2] + us[3]) / ode.num_steps)
    n = y0.numel(); w = y0.element_size()
    for c in cfgs:
        med = statistics.median(res[c])
        print("%-14s %-16s solver us/step median %7.2f min %7.2f  frac(32Nw) %.3f" % (name, c or "(default)", med, min(res[c]), 32 * n * w / med / 1e3 / 8000), flush=True)
    lib.pn_tune_set(None)
