#!/usr/bin/env python3
"""cProfile of one C5-shard cn solve (64 x 1024 fp64, Newton-GMRES) per Krylov configuration: where the host time goes."""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, torch.nn as nn
from pnode_amd import options, petsc_adjoint
from problems import BurgersEX, BurgersIM
dev = torch.device("cuda:0"); n5, NT = 1024, 10
torch.manual_seed(0)
y0 = torch.rand(64, n5, dtype=torch.float64, device=dev)
class StencilIM(nn.Module):
    def __init__(s, n, alpha=8e-4):
        super().__init__(); s.k = alpha * float(n) ** 2
    def forward(s, t, y): return s.k * (torch.roll(y, 1, -1) - 2.0 * y + torch.roll(y, -1, -1))
STENCIL = "stencil" in sys.argv
class Full(nn.Module):
    def __init__(s):
        super().__init__(); s.fI, s.fE = (StencilIM(n5) if STENCIL else BurgersIM(n5)).to(dev), BurgersEX(n5).to(dev)
    def forward(s, t, y): return s.fI(t, y) + s.fE(t, y)
f = Full(); t = torch.tensor([0.01 * NT], dtype=torch.float64)
params = [p for p in f.parameters() if p.requires_grad]
which = sys.argv[1] if len(sys.argv) > 1 else "default"
extra = {"default": {}, "nograph": {"pn_krylov_graph": 0}, "host": {"pn_krylov": "host", "pn_krylov_graph": 0}}[which]
options.clear(); options.set_option("ts_adapt_type", "none"); options.set_option("pn_krylov_log", 1)
for k, v in extra.items(): options.set_option(k, v)
ode = petsc_adjoint.ODEPetsc()
ode.setupTS(y0, f, step_size=0.01, method="cn", implicit_form=True, batch_size=64, linear_solver="petsc")
options.clear()
def solve():
    for p in params: p.grad = None
    y = y0.detach().requires_grad_(True); ode.odeint_adjoint(y, t).abs().mean().backward()
for _ in range(3): solve()
torch.cuda.synchronize(); t0 = time.perf_counter(); solve(); torch.cuda.synchronize()
print("%s: %.1f ms per solve, gmres its %d, syncs %d" % (which, 1e3 * (time.perf_counter() - t0), ode._theta.linear_its, ode._theta.host_syncs))
print("iterations per linear solve (this solve):", getattr(ode._theta, "_its_log", None))
if "--trace-only" in sys.argv:
    for _ in range(2): solve()
    torch.cuda.synchronize(); sys.exit(0)
# a graph replay on its own
th = ode._theta
if th._op_graphs:
    e = next(iter(th._op_graphs.values()))
    g = next(iter(e.B.values()))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): g.replay()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("product graph: %.1f us host per replay, %.1f us until the GPU is done" % (1e6 * (t1 - t0) / 50, 1e6 * (t2 - t0) / 50))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): e.gA.replay()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("linearisation graph: %.1f us host per replay, %.1f us until the GPU is done" % (1e6 * (t1 - t0) / 50, 1e6 * (t2 - t0) / 50))
pr = cProfile.Profile(); pr.enable(); solve(); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
