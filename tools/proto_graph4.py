import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from pnode_amd import options, petsc_adjoint
from problems import SpiralFunc
dev = torch.device("cuda:0")
mode = sys.argv[1]
options.set_option("ts_adapt_type", "none"); options.set_option("ts_trajectory_solution_only", "0")
torch.manual_seed(0)
func = SpiralFunc(torch.float32).to(dev); y0 = torch.randn(4096, 2, device=dev); t = torch.tensor([0.025 * int(os.environ.get("NT", "100"))])
ode = petsc_adjoint.ODEPetsc(); ode.setupTS(y0, func, step_size=0.025, method="rk4")
if "eagerbwd" in mode:
    y = y0.detach().requires_grad_(True); ode.odeint_adjoint(y, t).abs().mean().backward()
else:
    with torch.no_grad(): ode._odeint(y0, t, True)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph(); g2 = torch.cuda.CUDAGraph()
kw = {"pool": torch.cuda.graph_pool_handle()} if "pool" in mode else {}
if "tl" in mode: kw["capture_error_mode"] = "thread_local"
if "gc" in mode:
    import gc; gc.collect(); torch.cuda.synchronize()
if "side" in mode:
    pass
static_y0 = y0.clone()
with torch.cuda.graph(g, **kw):
    with torch.no_grad():
        sol = ode._odeint(static_y0, t, True)
print("captured fwd", mode); g.replay(); torch.cuda.synchronize(); print("replayed fwd")
if "bwd" in mode.split(",")[-1]:
    gout = torch.zeros((1,) + tuple(y0.shape), device=dev)
    with torch.cuda.graph(g2, **kw):
        with torch.no_grad():
            ode._begin_adjoint(gout.view(1, -1)[0]); ode._adjoint_steps(ode._nsteps, None)
    print("captured bwd"); g.replay(); g2.replay(); torch.cuda.synchronize(); print("replayed bwd")
