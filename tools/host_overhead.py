#!/usr/bin/env python3
"""Where the host time of an eager C3a solve goes (4096 x 512 fp32, rk4, 40 steps, tapes retained): wall time of the
forward / reverse sweeps with the GPU left to run behind, against the same sweeps with func replaced by a no-op of the
same output (the engine's own host cost), and against func + autograd alone."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from pnode_amd import options, petsc_adjoint
from problems import MLPFunc
dev = torch.device("cuda:0"); NT = 40
options.set_option("ts_adapt_type", "none"); options.set_option("ts_trajectory_solution_only", 0)
options.set_option("pn_graph_capture", 0)            # the eager sweeps are what is measured here (the default is `auto` since round 4)
f = MLPFunc(512, torch.float32).to(dev)
y0 = torch.randn(4096, 512, device=dev); t = torch.tensor([0.01 * NT])


class Cheap(torch.nn.Module):
    """One tiny kernel per call, one parameter: the engine's own host work is what remains."""
    def __init__(self):
        super().__init__()
        self.a = torch.nn.Parameter(torch.ones(1, device=dev))
    def forward(self, t, y):
        return y * self.a


def run(func, label, loop="native"):
    options.set_option("pn_step_loop", loop)         # native: pn_rk_attempt / pn_rk_adjoint_step (round 4); python: rounds 1-3
    ode = petsc_adjoint.ODEPetsc(); ode.setupTS(y0, func, step_size=0.01, method="rk4")
    def solve():
        for p in func.parameters(): p.grad = None
        y = y0.detach().requires_grad_(True)
        t0 = time.perf_counter()
        out = ode.odeint_adjoint(y, t)
        t1 = time.perf_counter()
        out.abs().mean().backward()
        t2 = time.perf_counter()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        return t1 - t0, t2 - t1, t3 - t0
    for _ in range(3): solve()
    r = [solve() for _ in range(5)]
    fw = min(x[0] for x in r); bw = min(x[1] for x in r); tot = min(x[2] for x in r)
    print("%-34s forward host %7.1f us/step  reverse host %7.1f us/step  until GPU done %7.1f us/step"
          % (label, 1e6 * fw / NT, 1e6 * bw / NT, 1e6 * tot / NT), flush=True)

run(f, "C3a func, C++ step loops")
run(f, "C3a func, Python stage loop", "python")
run(Cheap(), "no-op func, C++ step loops")
run(Cheap(), "no-op func, Python stage loop", "python")
# the no-op func's own cost with the same call pattern (4 forwards with grad + 4 autograd.grad per time step): what is left of the
# no-op row after subtracting this is the engine's own host work (Python orchestration + ctypes launches)
cheap = Cheap()
ysn = [torch.randn(4096, 512, device=dev) for _ in range(4)]
wn = torch.randn(4096, 512, device=dev)
def alone_noop():
    outs = []
    t0 = time.perf_counter()
    for _ in range(NT):
        for y in ysn:
            with torch.enable_grad():
                yy = y.detach().requires_grad_(True)
                outs.append((yy, cheap(0.0, yy)))
    t1 = time.perf_counter()
    for yy, o in reversed(outs):
        torch.autograd.grad(o, (yy,) + tuple(cheap.parameters()), wn)
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    return t1 - t0, t2 - t1
for _ in range(2): alone_noop()
an = [alone_noop() for _ in range(3)]
print("%-34s forward host %7.1f us/step  reverse host %7.1f us/step   (subtract from the no-op row: the engine's own host work)"
      % ("no-op func + autograd.grad alone", 1e6 * min(x[0] for x in an) / NT, 1e6 * min(x[1] for x in an) / NT))
# func + autograd alone, same call pattern: 4 forwards, 4 x (backward through a retained tape) per step
ys = [torch.randn(4096, 512, device=dev, requires_grad=True) for _ in range(4)]
w = torch.randn(4096, 512, device=dev)
def alone():
    t0 = time.perf_counter()
    outs = []
    for _ in range(NT):
        for y in ys:
            with torch.enable_grad(): outs.append((y, f(0.0, y)))
    t1 = time.perf_counter()
    for y, o in reversed(outs):
        torch.autograd.grad(o, (y,) + tuple(f.parameters()), w)
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    return t1 - t0, t2 - t1
for _ in range(2): alone()
a = [alone() for _ in range(3)]
print("%-34s forward host %7.1f us/step  reverse host %7.1f us/step" % ("func + autograd.grad alone", 1e6 * min(x[0] for x in a) / NT, 1e6 * min(x[1] for x in a) / NT))
