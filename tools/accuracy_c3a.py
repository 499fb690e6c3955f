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
for dt in (torch.float64, torch.float32, "fp32-exact", "fp32-autograd"):
    # fp32: the default (Linear sensitivities by the engine, operands split into three bf16 terms); fp32-exact: the same kernel on
    # the fp32 matrix instruction (-pn_linear_wgrad_exact 1); fp32-autograd: -pn_linear_param_grads 0 (what the reference computes)
    tag = dt
    if dt == "fp32-exact":
        options.set_option("pn_linear_wgrad_exact", "1")
    if dt == "fp32-autograd":
        options.set_option("pn_linear_wgrad_exact", "0"); options.set_option("pn_linear_param_grads", "0")
    dt = torch.float32 if isinstance(dt, str) else dt
    f = MLPFunc(512, dt).to(dev)
    ode = petsc_adjoint.ODEPetsc(); ode.setupTS(y0.to(dev, dt), f, step_size=0.01, method="rk4")
    y = y0.to(dev, dt).requires_grad_(True)
    out = ode.odeint_adjoint(y, t.to(dev))
    torch.mean(torch.abs(out - target.to(dev, dt))).backward()
    res[tag] = (out.detach().double().cpu(), y.grad.double().cpu(), flat_grads(f).double().cpu(), ode._nsteps, ode.linear_param_grads)
a = res[torch.float64]
for tag in (torch.float32, "fp32-exact", "fp32-autograd"):
  b = res[tag]
  print(str(tag).ljust(16), b[4][:60].ljust(62), end=" ")
  print("C3a full size, %d steps: fp32 vs fp64  forward %.2e  dL/dy0 %.2e  dL/dtheta %.2e"
      % (b[3], rel_err(b[0], a[0]), rel_err(b[1], a[1]), rel_err(b[2], a[2])))
