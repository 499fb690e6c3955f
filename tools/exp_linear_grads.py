"""Experiment (round 5): what would it buy at C3a if the engine accumulated the parameter sensitivities of func's nn.Linear
layers itself (dW by an accumulating GEMM into mu, db by a column sum) instead of taking dW / db from autograd (whose bias
gradient is a 14.7 us reduction, 16 per time step)?  Times one stage VJP of the C3a MLP, GPU time by events, both ways."""
import os, sys, functools
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, torch.nn as nn
from problems import MLPFunc
dev = torch.device("cuda:0")
torch.manual_seed(0)
f = MLPFunc(512, torch.float32).to(dev)
params = [p for p in f.parameters()]
y0 = torch.randn(4096, 512, device=dev)
w = torch.randn(4096, 512, device=dev)
npar = sum(p.numel() for p in params)
mu = torch.zeros(npar, device=dev)
offs, o = {}, 0
for p in params:
    offs[id(p)] = o; o += p.numel()
ones = torch.ones(4096, device=dev)
state = {"on": False}

def fwd_hook(m, inp, out):
    if state["on"] and out.requires_grad:
        x = inp[0].detach()
        out.register_hook(functools.partial(grad_hook, m, x))

def grad_hook(m, x, g):
    ow = offs[id(m.weight)]
    mw = mu[ow: ow + m.weight.numel()].view_as(m.weight)
    torch.addmm(mw, g.t(), x, beta=1.0, alpha=0.5, out=mw)
    if m.bias is not None:
        ob = offs[id(m.bias)]
        mb = mu[ob: ob + m.bias.numel()]
        if state["bias"] == "addmv":
            torch.addmv(mb, g.t(), ones, beta=1.0, alpha=0.5, out=mb)
        elif state["bias"] == "sum":
            mb.add_(g.sum(0), alpha=0.5)

for m in f.modules():
    if type(m) is nn.Linear:
        m.register_forward_hook(fwd_hook)

def tape():
    y = y0.detach().requires_grad_(True)
    with torch.enable_grad():
        out = f(0.0, y)
    return y, out

def timeit(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    tot = 0.0
    for _ in range(reps):
        t = tape_fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record(); fn(t); e1.record(); torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / reps * 1e3

def run_autograd(t=None):
    y, out = t if t else tape()
    g = torch.autograd.grad(out, (y,) + tuple(params), w)
    for p, gp in zip(params, g[1:]):
        mu[offs[id(p)]: offs[id(p)] + p.numel()].add_(gp.reshape(-1), alpha=0.5)

def run_hooks(t=None):
    y, out = t if t else tape()
    torch.autograd.grad(out, (y,), w)

# GPU time via graph capture of ONE VJP each way (removes host dispatch from the measurement)
def graph_time(mode, bias):
    state["on"] = mode == "hooks"; state["bias"] = bias
    y, out = tape()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    # warm
    if mode == "hooks":
        torch.autograd.grad(out, (y,), w, retain_graph=True)
    else:
        torch.autograd.grad(out, (y,) + tuple(params), w, retain_graph=True)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        if mode == "hooks":
            torch.autograd.grad(out, (y,), w, retain_graph=True)
        else:
            gs = torch.autograd.grad(out, (y,) + tuple(params), w, retain_graph=True)
            for p, gp in zip(params, gs[1:]):
                mu[offs[id(p)]: offs[id(p)] + p.numel()].add_(gp.reshape(-1), alpha=0.5)
    for _ in range(5): g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 50 * 1e3

import pnode_amd  # sets DEBUG_CLR_GRAPH_PACKET_CAPTURE
print("one stage VJP at C3a (4096 x 512 fp32, 4 Linear layers), GPU us per replayed graph:")
print("  autograd (y + params) + mu.add_      : %.1f" % graph_time("autograd", None))
print("  hooks: addmm_ into mu, bias by addmv  : %.1f" % graph_time("hooks", "addmv"))
print("  hooks: addmm_ into mu, bias by sum(0) : %.1f" % graph_time("hooks", "sum"))
print("  hooks: addmm_ into mu, no bias        : %.1f" % graph_time("hooks", "none"))
# correctness of the hook path
mu.zero_(); state["on"] = False
y, out = tape(); gs = torch.autograd.grad(out, (y,) + tuple(params), w)
ref = torch.cat([g.reshape(-1) for g in gs[1:]]) * 0.5
mu.zero_(); state["on"] = True; state["bias"] = "addmv"
y, out = tape(); torch.autograd.grad(out, (y,), w)
print("rel err hooks vs autograd:", float((mu - ref).norm() / ref.norm()))
