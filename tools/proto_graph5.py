import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from pnode_amd.petsc_adjoint import HipVecOps
from problems import SpiralFunc
dev = torch.device("cuda:0")
which = sys.argv[1]
n = 8192
ops = HipVecOps(dev, torch.float32, n)
u = torch.randn(n, device=dev); k = torch.randn(n, device=dev); y = torch.empty(n, device=dev)
mu = torch.zeros(300, device=dev); gs = [torch.randn(100, device=dev), None, torch.randn(150, device=dev)]
f = SpiralFunc(torch.float32).to(dev); params = tuple(f.parameters())
def vjp():
    with torch.enable_grad():
        yy = u.view(4096, 2).detach().requires_grad_(True); out = f(0.0, yy)
        return torch.autograd.grad(out, (yy,) + params, k.view(4096, 2), allow_unused=True)
ops.param_accum(mu, 1.0, gs, [0, 100, 150], [100, 50, 150]); vjp(); ops.adj_theta(y, u, 0.5, [k], [0.25]); ops.adj_accum(u, u, [k], [0.5], None)
torch.cuda.synchronize()
import gc; gc.collect()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    if which == "param": ops.param_accum(mu, 1.0, gs, [0, 100, 150], [100, 50, 150])
    if which == "zero": mu.zero_()
    if which == "vjp": r = vjp()
    if which == "theta": ops.adj_theta(y, u, 0.5, [k], [0.25])
    if which == "accum": ops.adj_accum(u, u, [k], [0.5], None)
print("captured", which, flush=True); g.replay(); torch.cuda.synchronize(); print("replayed", which, flush=True)
