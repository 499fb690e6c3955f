import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from pnode_amd.petsc_adjoint import HipVecOps
dev = torch.device("cuda:0")
which = sys.argv[1]
n = 8192
ops = HipVecOps(dev, torch.float32, n)
u = torch.randn(n, device=dev); k = torch.randn(n, device=dev); y = torch.empty(n, device=dev)
ops.rk_stage(y, u, [k], [0.5]); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
if which == "mine":
    with torch.cuda.graph(g):
        ops.rk_stage(y, u, [k], [0.5])
        ops.rk_stage(u, y, [k], [0.25])
elif which == "torch":
    with torch.cuda.graph(g):
        z = torch.tanh(u) * 2
elif which == "mixed":
    with torch.cuda.graph(g):
        z = torch.tanh(u)
        ops.rk_stage(y, u, [z], [0.5])
        z2 = torch.tanh(y)
print("captured", which); g.replay(); torch.cuda.synchronize(); print("replayed", which)
