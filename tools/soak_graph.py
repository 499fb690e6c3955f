#!/usr/bin/env python3
"""Soak: a training loop driven by hipGraph replays next to the same loop with eager launches, two
models updated with their own gradients (SGD, in place).  Bit-identical gradients keep the two models
identical forever; any divergence is reported with the iteration it first appears at."""
import copy, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from pnode_amd import options, petsc_adjoint
from problems import MLPFunc
dev = torch.device("cuda:0")
ITERS, NT = int(os.environ.get("ITERS", 150)), int(os.environ.get("NT", 20))
torch.manual_seed(0)
fa = MLPFunc(512, torch.float32).to(dev); fb = copy.deepcopy(fa)
t = torch.tensor([0.0, 0.01 * (NT // 2), 0.01 * NT])


def make(f, graph):
    options.clear()
    options.set_option("ts_adapt_type", "none"); options.set_option("ts_trajectory_solution_only", 0)
    if not graph: options.set_option("pn_graph_capture", 0)      # (graph: no option at all -- `auto`, the default since round 4)
    o = petsc_adjoint.ODEPetsc(); o.setupTS(torch.empty(4096, 512, device=dev), f, step_size=0.01, method="rk4")
    options.clear(); return o


oa, ob = make(fa, True), make(fb, False)
gen = torch.Generator(device=dev).manual_seed(1)
bad = None
t0 = time.time()
for it in range(ITERS):
    y0 = torch.randn(4096, 512, device=dev, generator=gen)
    tgt = torch.randn(4096, 512, device=dev, generator=gen)
    res = []
    for f, o in ((fa, oa), (fb, ob)):
        for p in f.parameters(): p.grad = None
        y = y0.clone().requires_grad_(True)
        sol = o.odeint_adjoint(y, t)
        loss = (sol[2] - tgt).pow(2).mean() + sol[1].abs().mean()
        loss.backward()
        res.append((loss.detach().clone(), y.grad.clone(), [p.grad.clone() for p in f.parameters()]))
        with torch.no_grad():
            for p in f.parameters(): p.add_(p.grad, alpha=-0.05)
    if it % 7 == 3: torch.cuda.synchronize()
    if it % 11 == 5: torch.cuda.current_stream().synchronize()
    same = torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1]) and all(torch.equal(a, b) for a, b in zip(res[0][2], res[1][2]))
    if not same and bad is None:
        bad = it
        print("DIVERGED at iteration", it, flush=True)
    if it % 25 == 0:
        print("iter %4d loss %.6f same=%s captured=%s" % (it, float(res[0][0]), same, bool(oa.graphs_captured)), flush=True)
print("soak: %d iterations x %d time steps, %.1f s, first divergence: %s, parameters equal at the end: %s"
      % (ITERS, NT, time.time() - t0, bad, all(torch.equal(a, b) for a, b in zip(fa.parameters(), fb.parameters()))))
