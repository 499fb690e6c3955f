"""Development probe: funcs of various shapes through an adaptive solve, default launch mode against eager launches (6 calls each)."""
import sys, warnings
import os
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch, torch.nn as nn, torch.nn.functional as F
from pnode_amd import options, petsc_adjoint
from problems import flat_grads

class LN(nn.Module):
    def __init__(s): super().__init__(); s.a, s.n, s.b = nn.Linear(6, 6), nn.LayerNorm(6), nn.Linear(6, 6)
    def forward(s, t, y): return s.b(torch.tanh(s.n(s.a(y))))
class Branch(nn.Module):
    def __init__(s): super().__init__(); s.a, s.b, s.c = nn.Linear(6, 6), nn.Linear(6, 6), nn.Linear(6, 6)
    def forward(s, t, y):
        h = s.a(y); return s.b(torch.tanh(h)) + s.c(torch.sin(h)) + 0.1 * h
class Loop(nn.Module):
    def __init__(s): super().__init__(); s.ls = nn.ModuleList([nn.Linear(6, 6) for _ in range(3)])
    def forward(s, t, y):
        for l in s.ls: y = torch.tanh(l(y))
        return y
class Transposed(nn.Module):
    def __init__(s): super().__init__(); s.a = nn.Linear(5, 5); s.b = nn.Linear(6, 6)
    def forward(s, t, y): return s.b(torch.tanh(s.a(y.t()).t()))          # Linear over the batch dimension (non-contiguous in/out)
class Frozen(nn.Module):
    def __init__(s): super().__init__(); s.a, s.b = nn.Linear(6, 6), nn.Linear(6, 6); s.a.weight.requires_grad_(False)
    def forward(s, t, y): return s.b(torch.tanh(s.a(y)))
class SpectralN(nn.Module):
    def __init__(s): super().__init__(); s.a = nn.utils.parametrizations.spectral_norm(nn.Linear(6, 6)); s.b = nn.Linear(6, 6)
    def forward(s, t, y): return s.b(torch.tanh(s.a(y)))
class InnerNoGrad(nn.Module):
    def __init__(s): super().__init__(); s.a, s.b = nn.Linear(6, 6), nn.Linear(6, 6)
    def forward(s, t, y):
        with torch.no_grad(): g = torch.sigmoid(s.a(y))
        return s.b(torch.tanh(s.a(y))) * g
class Ckpt(nn.Module):
    def __init__(s): super().__init__(); s.a, s.b = nn.Linear(6, 6), nn.Linear(6, 6)
    def forward(s, t, y):
        from torch.utils.checkpoint import checkpoint
        return s.b(checkpoint(lambda z: torch.tanh(s.a(z)), y, use_reentrant=False))
class OneD(nn.Module):
    def __init__(s): super().__init__(); s.a = nn.Linear(30, 30)
    def forward(s, t, y): return torch.tanh(s.a(y.reshape(-1))).reshape(y.shape)
class Detached(nn.Module):
    def __init__(s): super().__init__(); s.a, s.b = nn.Linear(6, 6), nn.Linear(6, 6)
    def forward(s, t, y): return s.b(torch.tanh(s.a(y))) + s.a(y).detach()

dev = torch.device("cuda:0")
def solve(make, opts, calls):
    options.clear()
    for k, v in opts.items(): options.set_option(k, v)
    torch.manual_seed(2); f = make().to(dev); torch.manual_seed(3)
    y0 = torch.randn(5, 6, device=dev); w = torch.randn(3, 5, 6, device=dev)
    ode = petsc_adjoint.ODEPetsc(); ode.setupTS(y0, f, step_size=0.05, method="dopri5"); options.clear()
    outs = []
    with warnings.catch_warnings(record=True) as c:
        warnings.simplefilter("always")
        for it in range(calls):
            for p in f.parameters(): p.grad = None
            y = y0.clone().requires_grad_(True)
            out = ode.odeint_adjoint(y, torch.tensor([0.0, 0.1, 0.3])); (out * w).sum().backward()
            outs.append((out.detach().clone(), y.grad.clone(), torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in f.parameters()])))
    return outs, ode, [str(m.message)[:160] for m in c if "pnode_amd" in str(m.message)]

class BNf(nn.Module):
    def __init__(s): super().__init__(); s.a, s.n, s.b = nn.Linear(6, 6), nn.BatchNorm1d(6), nn.Linear(6, 6)
    def forward(s, t, y): return s.b(torch.tanh(s.n(s.a(y))))
class Diverg(nn.Module):
    """FFJORD-like: differentiates through its own layers inside forward (Hutchinson's estimator with a fixed probe)."""
    def __init__(s): super().__init__(); s.a, s.b = nn.Linear(6, 6), nn.Linear(6, 6); s.register_buffer("e", torch.randn(5, 6))
    def forward(s, t, y):
        with torch.enable_grad():
            z = y if y.requires_grad else y.detach().requires_grad_(True)
            dz = s.b(torch.tanh(s.a(z)))
            (ge,) = torch.autograd.grad(dz, z, s.e, create_graph=True)
        return dz + 0.01 * (ge * s.e).sum(-1, keepdim=True)
class Counter(nn.Module):
    def __init__(s): super().__init__(); s.a, s.b = nn.Linear(6, 6), nn.Linear(6, 6); s.nfe = 0
    def forward(s, t, y): s.nfe += 1; return s.b(torch.tanh(s.a(y))) * torch.cos(t if isinstance(t, torch.Tensor) else torch.tensor(t))
class ConcatT(nn.Module):
    def __init__(s): super().__init__(); s.a, s.b = nn.Linear(7, 16), nn.Linear(16, 6)
    def forward(s, t, y): return s.b(torch.tanh(s.a(torch.cat([y, torch.ones_like(y[:, :1]) * t], 1))))
for cls in (Diverg, Counter, ConcatT, LN, BNf, Branch, Loop, Transposed, Frozen, SpectralN, InnerNoGrad, Ckpt, OneD, Detached):
    base = {"ts_rtol": 1e-6, "ts_atol": 1e-6}
    try:
        ref, _, _ = solve(cls, dict(base, pn_graph_capture=0), 6)
        got, ode, msgs = solve(cls, base, 6)
        same = all(all(torch.equal(x, y) for x, y in zip(a, b)) for a, b in zip(got, ref))
        print("%-12s bitwise equal to eager over 6 calls: %s   %s   %s" % (cls.__name__, same, ode.graph_status[:110], msgs[:1]))
    except Exception as exc:
        print("%-12s EXC %s: %s" % (cls.__name__, type(exc).__name__, str(exc)[:200]))
