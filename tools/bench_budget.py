#!/usr/bin/env python3
"""C3a (4096x512 fp32, rk4, 100 steps) under a checkpoint budget: state-only checkpoints
(-ts_trajectory_solution_only 1) against checkpoints that carry their step's stage values (0)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from pnode_amd import options, petsc_adjoint
from problems import MLPFunc
dev = torch.device("cuda:0")
f = MLPFunc(512, torch.float32).to(dev); y0 = torch.randn(4096, 512, device=dev); t = torch.tensor([1.0])
def run(extra, reps=3):
    options.clear(); options.set_option("ts_adapt_type", "none")
    options.set_option("pn_graph_capture", 0)        # eager launches, as in round 1's table (the default since round 4 is `auto`)
    for k, v in extra.items(): options.set_option(k, v)
    ode = petsc_adjoint.ODEPetsc(); ode.setupTS(y0, f, step_size=0.01, method="rk4"); options.clear()
    def solve():
        for p in f.parameters(): p.grad = None
        y = y0.detach().requires_grad_(True); ode.odeint_adjoint(y, t).abs().mean().backward()
        return torch.cat([p.grad.reshape(-1) for p in f.parameters()])
    g = solve(); solve(); torch.cuda.synchronize(); n0 = ode.nfe_forward; t0 = time.perf_counter()
    for _ in range(reps): solve()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    return g, 100 / dt, (ode.nfe_forward - n0) // reps
ref, r0, n0 = run({"ts_trajectory_solution_only": 0})
print("store-all                      : %6.1f time-steps/s  NFE-F %4d" % (r0, n0))
g, r, n = run({"ts_trajectory_solution_only": 1}); print("solution-only (every state)    : %6.1f time-steps/s  NFE-F %4d  bitwise %s" % (r, n, bool(torch.equal(g, ref))))
for c in (100, 50, 20, 10, 5):
    for so in (1, 0):
        g, r, n = run({"ts_trajectory_solution_only": so, "ts_trajectory_max_cps_ram": c})
        print("max_cps %3d %-18s : %6.1f time-steps/s  NFE-F %4d  bitwise %s" % (c, "state-only" if so else "state+stages", r, n, bool(torch.equal(g, ref))), flush=True)
