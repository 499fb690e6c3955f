#!/usr/bin/env python3
"""cProfile of the engine's own host work in an eager C3a-shaped solve (4096 x 512 fp32, rk4, 40 steps, tapes retained) with func
replaced by a one-kernel no-op (y * a): what remains is Python orchestration + ctypes launches + autograd's call overhead."""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from pnode_amd import options, petsc_adjoint
dev = torch.device("cuda:0"); NT = 40
options.set_option("ts_adapt_type", "none"); options.set_option("ts_trajectory_solution_only", 0)
for a in sys.argv[1:]:
    if "=" in a:
        k, v = a.lstrip("-").split("=", 1); options.set_option(k, v)
y0 = torch.randn(4096, 512, device=dev); t = torch.tensor([0.01 * NT])


class Cheap(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.a = torch.nn.Parameter(torch.ones(1, device=dev))
    def forward(self, t, y):
        return y * self.a


func = Cheap()
ode = petsc_adjoint.ODEPetsc(); ode.setupTS(y0, func, step_size=0.01, method="rk4")
def solve():
    for p in func.parameters(): p.grad = None
    y = y0.detach().requires_grad_(True)
    ode.odeint_adjoint(y, t).abs().mean().backward()
for _ in range(5): solve()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): solve()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
print("no-op func, eager: %.2f ms per solve = %.1f us per time step (fwd + reverse)" % (1e3 * dt, 1e6 * dt / NT))
# the same call pattern without the engine: 4 forwards (grad on) + 4 autograd.grad per time step
ys = [torch.randn(4096, 512, device=dev) for _ in range(4)]
w = torch.randn(4096, 512, device=dev)
def alone():
    outs = []
    for _ in range(NT):
        for y in ys:
            with torch.enable_grad():
                yy = y.detach().requires_grad_(True)
                outs.append((yy, func(0.0, yy)))
    for yy, o in reversed(outs):
        torch.autograd.grad(o, (yy,) + tuple(func.parameters()), w)
for _ in range(3): alone()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): alone()
torch.cuda.synchronize(); da = (time.perf_counter() - t0) / 10
print("func + autograd.grad alone, same call pattern: %.1f us per time step  ->  engine's own host work %.1f us per time step"
      % (1e6 * da / NT, 1e6 * (dt - da) / NT))
pr = cProfile.Profile(); pr.enable()
for _ in range(10): solve()
torch.cuda.synchronize(); pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(45)
