import os, sys, gc
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from pnode_amd.petsc_adjoint import HipVecOps
from problems import SpiralFunc
dev = torch.device("cuda:0")
flags = sys.argv[1].split(",")
n = 8192
ops = HipVecOps(dev, torch.float32, n)
u = torch.randn(n, device=dev); k = torch.randn(n, device=dev); y = torch.empty(n, device=dev)
mu = torch.zeros(300, device=dev)
f = SpiralFunc(torch.float32).to(dev); params = tuple(f.parameters())
def vjp(inp, cot):
    with torch.enable_grad():
        yy = inp.view(4096, 2).detach().requires_grad_(True); out = f(0.0, yy)
        return torch.autograd.grad(out, (yy,) + params, cot.view(4096, 2), allow_unused=True)
vjp(u, k); ops.copy(y, u)
gs = [torch.randn(100, device=dev), None, torch.randn(150, device=dev)]
if "h_param" in flags: ops.param_accum(mu, 1.0, gs, [0, 100, 150], [100, 50, 150])
if "h_theta" in flags: ops.adj_theta(y, u, 0.5, [k], [0.25])
if "h_accum" in flags: ops.adj_accum(y, y, [k], [0.5], None)
if "h_stage" in flags: ops.rk_stage(y, u, [k, k, k, k], [0.5, 0.1, 0.2, 0.3])
if "h_bwd" in flags:
    yy = u.view(4096, 2).detach().requires_grad_(True); f(0.0, yy).sum().backward()
if "h_cat" in flags:
    fp = torch.cat([p.view(-1) for p in params])
torch.cuda.synchronize(); gc.collect()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    with torch.no_grad():
        if "copy" in flags: ops.copy(y, u)
        if "tcopy" in flags: y.copy_(u)
        if "zero" in flags: mu.zero_()
        if "vjp" in flags: r = vjp(u, k)
        if "vjpy" in flags: r = vjp(y, k)
        if "after" in flags: ops.copy(y, u)
print("captured", flags, flush=True); g.replay(); torch.cuda.synchronize(); print("replayed", flush=True)
