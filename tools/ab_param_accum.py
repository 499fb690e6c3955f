#!/usr/bin/env python3
"""In-situ A/B of the parameter-sensitivity accumulation on the headline config (C3a fp32, rk4,
100 steps): one pn_param_accum launch per stage ("stage", 4 per time step) against one
pn_param_accum_multi launch per time step ("step").  Both variants interleaved in one process.
Reports whole-solve wall time (eager and hipGraph replay) and per-entry-point kernel time
from HIP start/stop events (eager)."""
import ctypes, os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from pnode_amd import _lib, options, petsc_adjoint
from problems import MLPFunc
lib = _lib.load(); dev = torch.device("cuda:0")
torch.manual_seed(0)
f = MLPFunc(512, torch.float32).to(dev)
y0 = torch.randn(4096, 512, device=dev)
t = torch.tensor([1.0])
NT = 100


def make(mode, graph):
    options.clear()
    options.set_option("ts_adapt_type", "none"); options.set_option("ts_trajectory_solution_only", "0")
    options.set_option("pn_param_accum", mode)
    if graph:
        options.set_option("pn_graph_capture", "1")
    o = petsc_adjoint.ODEPetsc(); o.setupTS(y0, f, step_size=0.01, method="rk4")
    options.clear()
    return o


def solve(o):
    for p in f.parameters(): p.grad = None
    y = y0.detach().requires_grad_(True); o.odeint_adjoint(y, t).abs().mean().backward()
    return torch.cat([p.grad.reshape(-1) for p in f.parameters()])


odes = {(m, g): make(m, g) for m in ("stage", "step") for g in (False, True)}
grads = {}
for k, o in odes.items():
    for _ in range(3):
        grads[k] = solve(o).clone()
ref = grads[("stage", False)]
for k, g in grads.items():
    print(k, "bit-identical to per-stage eager:", bool(torch.equal(g, ref)),
          "rel diff %.3e" % ((g - ref).norm() / ref.norm()).item(), "nonfinite", int((~torch.isfinite(g)).sum()), flush=True)
for k, o in odes.items():
    g2 = solve(o)
    print(k, "4th solve identical to own 3rd:", bool(torch.equal(g2, grads[k])), "to ref:", bool(torch.equal(g2, ref)), flush=True)

wall = {k: [] for k in odes}
kern = {m: [] for m in ("stage", "step")}
K = len(_lib.KERNEL_IDS)
for r in range(5):
    for k, o in odes.items():
        torch.cuda.synchronize(); t0 = time.perf_counter(); solve(o); torch.cuda.synchronize()
        wall[k].append(time.perf_counter() - t0)
    for m in ("stage", "step"):
        o = odes[(m, False)]
        torch.cuda.synchronize(); lib.pn_prof_enable(1); solve(o); torch.cuda.synchronize()
        L = (ctypes.c_int64 * K)(); us = (ctypes.c_double * K)(); by = (ctypes.c_double * K)()
        lib.pn_prof_collect(len(L), L, us, by); lib.pn_prof_enable(0)
        kern[m].append([(L[i], us[i]) for i in range(K)])
for k in odes:
    med = statistics.median(wall[k])
    print("%-6s %-6s wall ms median %7.2f min %7.2f  -> %6.1f time-steps/s" % (k[0], "graph" if k[1] else "eager", 1e3 * med, 1e3 * min(wall[k]), NT / med), flush=True)
for m in ("stage", "step"):
    print(m)
    tot = 0.0
    for i, name in enumerate(_lib.KERNEL_IDS):
        L = kern[m][0][i][0]
        if not L:
            continue
        us = statistics.median(r[i][1] for r in kern[m])
        tot += us
        print("   %-22s launches/step %5.2f  avg us %6.2f  us/step %7.2f" % (name, L / NT, us / L, us / NT))
    print("   all pn_* kernels us/step %.2f" % (tot / NT), flush=True)
