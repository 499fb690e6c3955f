#!/usr/bin/env python3
"""Gradient accuracy of the fp32 engine at the full headline config (C3a: 4096 x 512, rk4, 100
steps) against the fp64 engine (which the -m gpu tests hold within 1e-11 of the fp64 oracle)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from pnode_amd import options, petsc_adjoint
from problems import MLPFunc, flat_grads, rel_err
dev = torch.device("cuda:0")
options.set_option("ts_adapt_type", "none"); options.set_option("ts_trajectory_solution_only", "0")
torch.manual_seed(0)
y0 = torch.randn(4096, 512); target = torch.randn(1, 4096, 512); t = torch.tensor([1.0], dtype=torch.float64)
res = {}
for dt in (torch.float64, torch.float32):
    f = MLPFunc(512, dt).to(dev)
    ode = petsc_adjoint.ODEPetsc(); ode.setupTS(y0.to(dev, dt), f, step_size=0.01, method="rk4")
    y = y0.to(dev, dt).requires_grad_(True)
    out = ode.odeint_adjoint(y, t.to(dev))
    torch.mean(torch.abs(out - target.to(dev, dt))).backward()
    res[dt] = (out.detach().double().cpu(), y.grad.double().cpu(), flat_grads(f).double().cpu(), ode._nsteps)
a, b = res[torch.float64], res[torch.float32]
print("C3a full size, %d steps: fp32 vs fp64  forward %.2e  dL/dy0 %.2e  dL/dtheta %.2e"
      % (b[3], rel_err(b[0], a[0]), rel_err(b[1], a[1]), rel_err(b[2], a[2])))
