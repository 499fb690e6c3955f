import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from pnode_amd import _lib, options, petsc_adjoint
from problems import MLPFunc
dev = torch.device("cuda:0")
torch.manual_seed(0)
f = MLPFunc(512, torch.float32).to(dev)
y0 = torch.randn(4096, 512, device=dev)
t = torch.tensor([1.0])
def make(mode, graph):
    options.clear()
    options.set_option("ts_adapt_type", "none"); options.set_option("ts_trajectory_solution_only", "0")
    options.set_option("pn_param_accum", mode)
    if graph: options.set_option("pn_graph_capture", "1")
    o = petsc_adjoint.ODEPetsc(); o.setupTS(y0, f, step_size=0.01, method="rk4"); options.clear(); return o
def solve(o):
    for p in f.parameters(): p.grad = None
    y = y0.detach().requires_grad_(True); out = o.odeint_adjoint(y, t); out.abs().mean().backward()
    return torch.cat([p.grad.reshape(-1) for p in f.parameters()]).clone(), y.grad.clone(), out.detach().clone()
order = sys.argv[1:]
res = {}
for name in order:
    m, g = name.split(":")
    o = make(m, g == "graph")
    rs = [solve(o) for _ in range(5)]
    res[name] = rs
    print(name, "captured", o.graphs_captured, "run-to-run identical (runs 0..4 vs run 4):", [bool(torch.equal(r[0], rs[-1][0])) for r in rs], flush=True)
ref = res[order[0]][-1]
for name in order:
    r = res[name][-1]
    print(name, "vs", order[0], "gp rel %.3e  gy rel %.3e  out rel %.3e" % tuple(((a - b).norm() / b.norm()).item() for a, b in zip(r, ref)))
