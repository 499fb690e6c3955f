"""cProfile of graph-mode adaptive solves (development)."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import pnode_amd
from pnode_amd import petsc_adjoint, options
from problems import SwitchedMLPFunc

dev = torch.device("cuda:0")
torch.manual_seed(0)
f = SwitchedMLPFunc(512, torch.float32).to(dev)
options.clear()
options.set_option("ts_trajectory_type", "memory")
options.set_option("ts_trajectory_max_cps_ram", "50")
if os.environ.get("EAGER"):
    options.set_option("pn_graph_capture", "0")
o = petsc_adjoint.ODEPetsc()
y0 = torch.randn(4096, 512, device=dev) * 0.5
o.setupTS(y0, f, step_size=0.01, method="dopri5", enable_adjoint=True)
options.clear()


def solve():
    for p in f.parameters():
        p.grad = None
    y = y0.detach().requires_grad_(True)
    out = o.odeint_adjoint(y, torch.tensor([4.0]))
    out.abs().mean().backward()


for _ in range(5):
    solve()
torch.cuda.synchronize()
print(o.graph_status, o.num_steps, o.num_rejections)
t0 = time.perf_counter()
for _ in range(3):
    solve()
torch.cuda.synchronize()
print("ms per solve", 1e3 * (time.perf_counter() - t0) / 3)
if os.environ.get("CPROF"):
    pr = cProfile.Profile()
    pr.enable()
    solve()
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(22)
