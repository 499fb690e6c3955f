#!/usr/bin/env python3
"""cProfile of the eager host path on a launch-bound problem (spiral model, 20x1x2, 9 rk4 steps, 10 outputs)."""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from pnode_amd import options, petsc_adjoint
from problems import SpiralFunc
dev = torch.device("cuda:0")
options.set_option("ts_adapt_type", "none")
f = SpiralFunc(torch.float32).to(dev)
y0 = torch.randn(20, 1, 2, device=dev)
t = torch.linspace(0.0, 0.225, 10)
ode = petsc_adjoint.ODEPetsc(); ode.setupTS(y0, f, step_size=0.025, method="rk4")
def solve():
    for p in f.parameters(): p.grad = None
    y = y0.clone().requires_grad_(True)
    ode.odeint_adjoint(y, t).abs().mean().backward()
for _ in range(20): solve()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(100): solve()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 100
print("eager: %.2f ms per solve (9 steps fwd+adjoint) = %.0f us per time step" % (1e3 * dt, 1e6 * dt / 9))
pr = cProfile.Profile(); pr.enable()
for _ in range(100): solve()
torch.cuda.synchronize(); pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)
