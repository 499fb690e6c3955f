import os, sys, gc
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from pnode_amd import options, petsc_adjoint
from problems import SpiralFunc
dev = torch.device("cuda:0")
stage = sys.argv[1]
options.set_option("ts_adapt_type", "none"); options.set_option("ts_trajectory_solution_only", "0")
torch.manual_seed(0)
func = SpiralFunc(torch.float32).to(dev); y0 = torch.randn(4096, 2, device=dev); t = torch.tensor([0.025])
ode = petsc_adjoint.ODEPetsc(); ode.setupTS(y0, func, step_size=0.025, method="rk4")
if "noreent" in stage:
    with torch.no_grad():
        ode._odeint(y0, t, True); ode._begin_adjoint(torch.ones(y0.numel(), device=dev)); ode._adjoint_steps(ode._nsteps, None)
else:
    y = y0.detach().requires_grad_(True); ode.odeint_adjoint(y, t).abs().mean().backward()
torch.cuda.synchronize(); gc.collect()
g = torch.cuda.CUDAGraph(); g2 = torch.cuda.CUDAGraph()
static_y0 = y0.clone()
if "eagerfwd" in stage:
    with torch.no_grad(): sol = ode._odeint(static_y0, t, True)
else:
    with torch.cuda.graph(g):
        with torch.no_grad(): sol = ode._odeint(static_y0, t, True)
    g.replay()
torch.cuda.synchronize(); gc.collect()
print("fwd done", flush=True)
gout = torch.zeros((1,) + tuple(y0.shape), device=dev)
PLAIN_IN = torch.randn(8192, device=dev); PLAIN_COT = torch.randn(8192, device=dev)
ops = ode._ops
kw = {"capture_error_mode": "relaxed"} if "relaxed" in stage else {}
with torch.cuda.graph(g2, **kw):
    with torch.no_grad():
        if "nobegin" not in stage:
            ode._begin_adjoint(gout.view(1, -1)[0])
        if "B" in stage:
            Y = ode._stages_of(0)
        if "C" in stage:
            if "inline" in stage:
                with torch.enable_grad():
                    src = PLAIN_IN if "plainin" in stage else Y[3][:ode.n]
                    cot = PLAIN_COT if "plaincot" in stage else ode.adj_u_flat[:ode.n]
                    yy = src.view(4096, 2).detach().requires_grad_(True); out = func(0.0, yy)
                    r = torch.autograd.grad(out, (yy,) + tuple(func.parameters()), cot.view(4096, 2), allow_unused=True)
                gy, gp = r[0].reshape(-1), list(r[1:])
            else:
                gy, gp = ode._vjp(0.0, Y[3], ode.adj_u_flat)
        if "D" in stage:
            ops.param_accum(ode.adj_p_tensor, 0.5, gp, ode._poff, ode._plen)
        if "E" in stage:
            w = ode._buf("w_a"); ops.adj_theta(w, ode.adj_u_flat, 0.1, [gy], [0.2])
        if "F" in stage:
            ops.adj_accum(ode.adj_u_flat, ode.adj_u_flat, [gy], [0.3], None)
print("captured bwd", stage, flush=True); g2.replay(); torch.cuda.synchronize(); print("replayed", flush=True)
