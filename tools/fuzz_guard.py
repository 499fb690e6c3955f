#!/usr/bin/env python3
"""Randomised check of the capture guard (pnode_amd/_funcguard.py, _sweepgraphs.py) on the GPU: two solvers, one with
-pn_graph_capture 0 and one with the default (auto), each with its own copy of a func that reads scalar attributes, a flag,
a dictionary of hyper-parameters, a re-assignable tensor attribute, a buffer, a host tensor and a list of tensors.  Between
calls a random mutation is applied to BOTH copies -- the things the reference's callers do between solves
(examples-sinode/grand/src/base_classes.py:58-60, block_pnode.py:61-63; annealing; train/eval) -- and EVERY call's states,
dL/dy0 and dL/dtheta must be equal bit for bit.  usage: fuzz_guard.py [cases] [seed] [calls per case]"""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import torch.nn as nn
from pnode_amd import options, petsc_adjoint
from problems import flat_grads

dev = torch.device("cuda:0")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
calls = int(sys.argv[3]) if len(sys.argv) > 3 else 24
D = 16


class Func(nn.Module):
    def __init__(self, seed):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.l1, self.l2 = nn.Linear(D, D), nn.Linear(D, D)
        with torch.no_grad():
            for p in self.parameters():
                p.copy_(torch.randn(p.shape, generator=g) * 0.3)
        self.bn = nn.BatchNorm1d(D)
        self.register_buffer("mask", torch.ones(D))
        self.alpha, self.use_skip, self.nfe = 0.7, True, 0
        self.opt = {"beta": 0.2, "gains": [1.0, 0.5]}
        self.x0 = torch.zeros(1, D)
        self.edges = [torch.arange(D), torch.ones(D)]
        self.scale = torch.tensor(1.0)                      # host tensor: a kernel argument

    def forward(self, t, y):
        self.nfe += 1
        idx, w = self.edges
        h = torch.tanh(self.bn(self.l1(y))) * self.mask
        out = self.alpha * self.l2(h)[:, idx] * w * self.scale + self.opt["beta"] * self.x0 * torch.cos(y) * self.opt["gains"][1]
        return out - y if self.use_skip else out


MUTATIONS = ["none", "none", "alpha", "flag", "dict", "x0_new", "x0_new_keep", "x0_inplace", "mask_replace", "mask_inplace", "edges",
             "host_fill", "params_step", "eval", "train", "counter_reset", "time", "y0"]


def mutate(kind, f, st, r):
    """Apply mutation `kind` with the pre-drawn random numbers r (the same for both copies)."""
    if kind == "alpha":
        f.alpha = 0.3 + 0.6 * r[0]
    elif kind == "flag":
        f.use_skip = not f.use_skip
    elif kind == "dict":
        f.opt["beta"] = 0.1 + r[0]
        f.opt["gains"][1] = 0.25 + r[1]
    elif kind in ("x0_new", "x0_new_keep"):
        if kind == "x0_new_keep":
            st["keep"].append(f.x0)
        f.x0 = (st["y0"] * (0.5 + r[0])).clone().detach()
    elif kind == "x0_inplace":
        f.x0.mul_(0.5 + r[0])
    elif kind == "mask_replace":
        st["keep"].append(f.mask)
        f.mask = torch.full((D,), 0.5 + 0.5 * r[0], device=dev)
    elif kind == "mask_inplace":
        f.mask.fill_(0.5 + 0.5 * r[0])
    elif kind == "edges":
        g = torch.Generator().manual_seed(int(r[0] * 1e6))
        f.edges = [torch.randperm(D, generator=g).to(dev), torch.rand(D, generator=g).to(dev)]
    elif kind == "host_fill":
        f.scale.fill_(0.5 + r[0])
    elif kind == "params_step":
        with torch.no_grad():
            for p in f.parameters():
                p.mul_(1.0 - 0.01 * r[0])
    elif kind == "eval":
        f.eval()
    elif kind == "train":
        f.train()
    elif kind == "counter_reset":
        f.nfe = 0
    elif kind == "time":
        st["t"] = [0.2, 0.3, 0.35][int(r[0] * 3) % 3]
    elif kind == "y0":
        st["y0"] = st["y0"] * (0.9 + 0.2 * r[0])


bad = 0
t0 = time.time()
captured_cases = 0
for case in range(cases):
    seed = rng.randrange(1 << 30)
    reval = rng.choice([1, 3, 100])
    # a few mutation kinds per case, so that configurations repeat often enough to be captured
    kinds = ["none", "none", "none"] + rng.sample(MUTATIONS[2:], rng.choice([1, 2, 3]))
    plan = [(rng.choice(kinds), [rng.random(), rng.random()]) for _ in range(calls)]
    sides = {}
    for tag, opts in (("eager", {"pn_graph_capture": 0}), ("auto", {"pn_graph_revalidate": reval})):
        options.clear()
        for k, v in dict({"ts_adapt_type": "none"}, **opts).items():
            options.set_option(k, v)
        f = Func(seed).to(dev)
        torch.manual_seed(seed)
        st = {"y0": torch.randn(32, D, device=dev) * 0.5, "t": 0.3, "keep": []}
        f.x0 = st["y0"].clone()                              # (plain tensor attributes do not follow .to(dev))
        f.edges = [torch.arange(D, device=dev), torch.ones(D, device=dev)]
        ode = petsc_adjoint.ODEPetsc()
        ode.setupTS(st["y0"], f, step_size=0.05, method="rk4")
        options.clear()
        res = []
        for it, (kind, r) in enumerate(plan):
            mutate(kind, f, st, r)
            for p in f.parameters():
                p.grad = None
            y = st["y0"].clone().requires_grad_(True)
            out = ode.odeint_adjoint(y, torch.tensor([st["t"]]))
            (out * (1.0 + 0.1 * it)).sum().backward()
            res.append((out.detach().clone(), y.grad.clone(), flat_grads(f).clone(), f.nfe, f.bn.running_mean.clone()))
        sides[tag] = (res, ode)
    (re, _), (ra, ode_a) = sides["eager"], sides["auto"]
    captured_cases += bool(ode_a.graphs_captured)
    for it, (a, b) in enumerate(zip(re, ra)):
        same = all(torch.equal(x, y) for x, y in zip(a[:3], b[:3])) and a[3] == b[3] and torch.equal(a[4], b[4])
        if not same:
            bad += 1
            print("MISMATCH case", case, "call", it, "mutation", plan[it][0], "revalidate", reval, "kinds", kinds, "status", ode_a.graph_status,
                  "nfe", a[3], b[3], flush=True)
            break
    if case % 5 == 4:
        print("case %d/%d done, %d mismatches, %d cases with captured graphs, %.0f s" % (case + 1, cases, bad, captured_cases, time.time() - t0), flush=True)
print("fuzz_guard: %d cases x %d calls, mismatches: %d, cases that captured: %d" % (cases, calls, bad, captured_cases))
sys.exit(1 if bad else 0)
