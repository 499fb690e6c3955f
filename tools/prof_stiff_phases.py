#!/usr/bin/env python3
"""Where the time of the adaptive workload (bench.py --config c3b --stiff) goes: forward sweep and reverse sweep timed apart (wall,
to completion on the device), with and without the recompute tapes, host time of each sweep (wall until the call returns)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from pnode_amd import options, petsc_adjoint
from problems import SwitchedMLPFunc
dev = torch.device("cuda:0")
torch.manual_seed(0)
y0 = torch.randn(4096, 512, device=dev)
f = SwitchedMLPFunc(512, torch.float32).to(dev)
t = torch.tensor([SwitchedMLPFunc.T_END])
for extra in ({}, {"pn_trajectory_retain_graph": 0}, {"ts_trajectory_solution_only": 0}, {"pn_step_loop": "python"}):
    options.clear(); options.set_option("ts_trajectory_max_cps_ram", 50)
    for k, v in extra.items(): options.set_option(k, v)
    ode = petsc_adjoint.ODEPetsc(); ode.setupTS(y0, f, step_size=0.01, method="dopri5"); options.clear()
    rows = []
    for it in range(4):
        for p in f.parameters(): p.grad = None
        y = y0.detach().requires_grad_(True)
        torch.cuda.synchronize(); a = time.perf_counter()
        out = ode.odeint_adjoint(y, t)
        b = time.perf_counter(); torch.cuda.synchronize(); c = time.perf_counter()
        loss = out.abs().mean()
        torch.cuda.synchronize(); d = time.perf_counter()
        loss.backward()
        e = time.perf_counter(); torch.cuda.synchronize(); g = time.perf_counter()
        rows.append((b - a, c - a, e - d, g - d))
    r = rows[-1]
    print("%-40s steps %d rej %d | forward host %6.1f ms, done %6.1f ms | reverse host %6.1f ms, done %6.1f ms | NFE-F %d NFE-B %d"
          % (extra, ode._nsteps, ode.num_rejections, 1e3 * r[0], 1e3 * r[1], 1e3 * r[2], 1e3 * r[3], ode.nfe_forward // 4, ode.nfe_backward // 4), flush=True)
