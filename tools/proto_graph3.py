import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from pnode_amd import options, petsc_adjoint
from problems import SpiralFunc
dev = torch.device("cuda:0")
nt, save, so = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
options.set_option("ts_adapt_type", "none"); options.set_option("ts_trajectory_solution_only", so)
torch.manual_seed(0)
func = SpiralFunc(torch.float32).to(dev); y0 = torch.randn(4096, 2, device=dev); t = torch.tensor([0.025 * nt])
ode = petsc_adjoint.ODEPetsc(); ode.setupTS(y0, func, step_size=0.025, method="rk4")
with torch.no_grad():
    ode._odeint(y0, t, bool(save)); ode._odeint(y0, t, bool(save))
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    with torch.no_grad():
        sol = ode._odeint(y0, t, bool(save))
print("captured", nt, save, so); g.replay(); torch.cuda.synchronize(); print("replayed")
