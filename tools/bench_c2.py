#!/usr/bin/env python3
"""BASELINE config C2 (batched spiral: batch 4096 x state_dim 2, rk4, 100 steps h=0.025, fp32,
adjoint on): launch-latency-bound.  Prints time-steps/s and the per-time-step latency."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from pnode_amd import options, petsc_adjoint  # noqa: E402
from problems import SpiralFunc  # noqa: E402

dev = torch.device("cuda:0")
options.set_option("ts_adapt_type", "none")
for so, graph in (("1", "0"), ("0", "0"), ("0", "1"), ("1", "1")):
    options.set_option("ts_trajectory_solution_only", so)
    options.set_option("pn_graph_capture", graph)
    torch.manual_seed(0)
    func = SpiralFunc(torch.float32).to(dev)
    y0 = torch.randn(4096, 2, device=dev)
    t = torch.tensor([2.5])
    ode = petsc_adjoint.ODEPetsc()
    ode.setupTS(y0, func, step_size=0.025, method="rk4")

    def solve():
        for p in func.parameters():
            p.grad = None
        y = y0.detach().requires_grad_(True)
        ode.odeint_adjoint(y, t).abs().mean().backward()

    for _ in range(3):
        solve()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 10
    for _ in range(K):
        solve()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    print("C2 solution_only=%s hipgraph=%s: %.1f time-steps/s, %.1f us per time step (fwd+adjoint), nsteps %d"
          % (so, graph, ode._nsteps / dt, 1e6 * dt / ode._nsteps, ode._nsteps))
