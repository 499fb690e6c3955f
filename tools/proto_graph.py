#!/usr/bin/env python3
"""Prototype: capture a whole fixed-step forward sweep and a whole reverse sweep as hipGraphs."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from pnode_amd import options, petsc_adjoint
from problems import SpiralFunc, MLPFunc, flat_grads

dev = torch.device("cuda:0")
cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
options.set_option("ts_adapt_type", "none")
options.set_option("ts_trajectory_solution_only", "0")
torch.manual_seed(0)
if cfg == "c2":
    func = SpiralFunc(torch.float32).to(dev); y0 = torch.randn(4096, 2, device=dev); t = torch.tensor([2.5]); h = 0.025
else:
    func = MLPFunc(512, torch.float32).to(dev); y0 = torch.randn(4096, 512, device=dev); t = torch.tensor([1.0]); h = 0.01
params = [p for p in func.parameters()]
ode = petsc_adjoint.ODEPetsc()
ode.setupTS(y0, func, step_size=h, method="rk4")

def eager():
    for p in params: p.grad = None
    y = y0.detach().requires_grad_(True)
    out = ode.odeint_adjoint(y, t); out.abs().mean().backward()
    return out.detach().clone(), y.grad.clone(), flat_grads(func).clone()

s = torch.cuda.current_stream()

with torch.cuda.stream(s):
    for _ in range(3): ref = eager()

torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): ref = eager()
torch.cuda.synchronize(); te = (time.perf_counter() - t0) / 5
print("eager  %.1f ms/solve  %.1f time-steps/s" % (te * 1e3, ode._nsteps / te))

static_y0 = y0.clone()
gout = torch.zeros((1,) + tuple(y0.shape), device=dev)
g_f = torch.cuda.CUDAGraph(); g_b = torch.cuda.CUDAGraph()
pool = torch.cuda.graph_pool_handle()
tc = time.perf_counter()
with torch.cuda.graph(g_f, pool=pool):
    with torch.no_grad():
        sol = ode._odeint(static_y0, t, True)
with torch.cuda.graph(g_b, pool=pool):
    with torch.no_grad():
        g = gout.view(1, -1)
        ode._begin_adjoint(g[0])
        ode._adjoint_steps(ode._nsteps, None)
torch.cuda.synchronize()
print("capture %.1f ms" % ((time.perf_counter() - tc) * 1e3))

def graphed():
    static_y0.copy_(y0)
    g_f.replay()
    out = sol.clone()
    gout.copy_(torch.sign(out) / out.numel())
    g_b.replay()
    return out, ode._shaped(ode.adj_u_flat).clone(), ode.adj_p_tensor.clone()

r = graphed(); torch.cuda.synchronize()
print("match:", torch.equal(r[0], ref[0]), torch.equal(r[1], ref[1]), torch.equal(r[2], ref[2]),
      (r[2] - ref[2]).abs().max().item())
t0 = time.perf_counter()
for _ in range(10): r = graphed()
torch.cuda.synchronize(); tg = (time.perf_counter() - t0) / 10
print("graph  %.1f ms/solve  %.1f time-steps/s  (x%.2f)" % (tg * 1e3, ode._nsteps / tg, te / tg))
