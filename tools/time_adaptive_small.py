"""Development timing: the reference's spiral dynamics (Linear(2, 50)-Tanh-Linear(50, 2) on y^3, 4096 x 2) under dopri5 ADAPTIVE,
eager launches against per-evaluation hipGraphs (the default)."""
import os
import sys
import time

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
import pnode_amd
from pnode_amd import options, petsc_adjoint
from problems import SpiralFunc

dev = torch.device("cuda:0")


def make(opts):
    options.clear()
    for k, v in opts.items():
        options.set_option(k, v)
    torch.manual_seed(0)
    f = SpiralFunc(torch.float32).to(dev)
    y0 = torch.randn(4096, 2, device=dev)
    o = petsc_adjoint.ODEPetsc()
    o.setupTS(y0, f, step_size=0.025, method="dopri5")
    options.clear()
    return o, f, y0


def solve(o, f, y0):
    for p in f.parameters():
        p.grad = None
    y = y0.detach().requires_grad_(True)
    o.odeint_adjoint(y, torch.tensor([2.5])).abs().mean().backward()


for name, opts in (("eager", {"pn_graph_capture": "0"}), ("default", {})):
    o, f, y0 = make(opts)
    for _ in range(6):
        solve(o, f, y0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        solve(o, f, y0)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print("%-8s %7.2f ms per solve  %4d steps + %d rejected  -> %7.1f time-steps/s   %s" % (name, 1e3 * dt, o.num_steps, o.num_rejections, o.num_steps / dt, o.graph_status))
