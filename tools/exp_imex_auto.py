"""Experiment (round 5): why do the eager twin and the first replay of the IMEX direct-solve reverse sweep differ by an ulp
when both run in the same call?  Runs the capturable ARKIMEX configuration through the auto machinery (SweepGraphs.AUTO_THETA)
and compares, inside one process: eager reverse sweep vs itself, replay vs itself, eager vs replay."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from pnode_amd import options, petsc_adjoint, _sweepgraphs
from problems import DiffusionIM, ReactionEX

dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "3"
options.clear()
for k, v in {"ts_adapt_type": "none", "ts_arkimex_type": name, "snes_type": "ksponly"}.items():
    options.set_option(k, v)
torch.manual_seed(5)
fI, fE = DiffusionIM(16).to(dev), ReactionEX(16).to(dev)
y0 = torch.randn(8, 16, dtype=torch.float64, device=dev)
t = torch.tensor([0.0, 0.1, 0.25], dtype=torch.float64)
ode = petsc_adjoint.ODEPetsc()
ode.setupTS(y0, fI, step_size=0.05, method="imex", implicit_form=True, imex_form=True, func2=fE, batch_size=8,
            linear_solver="torch", matrixfree_jacobian=False)
options.clear()
params = list(fI.parameters()) + list(fE.parameters())


def eager_pair():
    """forward + reverse, eager, returns (sol, adj_u, adj_p)"""
    with torch.no_grad():
        sol = ode._odeint(y0, t, True)
        g = torch.ones_like(sol).view(sol.shape[0], -1) / sol.numel()
        ode._reverse_sweep(g, sol.shape[0])
        return sol.clone(), ode.adj_u_flat[:ode.n].clone(), ode.adj_p_tensor.clone()


a = eager_pair()
b = eager_pair()
print("eager vs eager      :", [bool(torch.equal(x, y)) for x, y in zip(a, b)])
# reverse sweep alone, twice, on the same trajectory
with torch.no_grad():
    sol = ode._odeint(y0, t, True)
    g = torch.ones_like(sol).view(sol.shape[0], -1) / sol.numel()
    st = ode._host_state()
    ode._reverse_sweep(g, 3); r1 = (ode.adj_u_flat[:ode.n].clone(), ode.adj_p_tensor.clone())
    ode._set_host_state(st)
    try:
        ode._reverse_sweep(g, 3); r2 = (ode.adj_u_flat[:ode.n].clone(), ode.adj_p_tensor.clone())
        print("reverse twice on one trajectory:", [bool(torch.equal(x, y)) for x, y in zip(r1, r2)])
    except Exception as exc:
        print("second reverse sweep on the same trajectory failed:", type(exc).__name__, str(exc)[:200])
# the factors: eager cache vs what graph_prepare writes
th = ode._theta
th.graph_prepare(y0)
for key in th._lu:
    LUe, pive = th._lu[key][:2]
    LUs, pivs = th._static_lu[key][:2]
    print("shift", key, "LU equal", bool(torch.equal(LUe, LUs)), "piv equal", bool(torch.equal(pive, pivs)),
          "strides", LUe.stride(), LUs.stride())
    R = torch.randn(8, LUe.shape[0], dtype=torch.float64, device=dev)
    for adj in (False, True):
        x1 = torch.linalg.lu_solve(LUe, pive, R, left=False, adjoint=adj)
        x2 = torch.linalg.lu_solve(LUs, pivs, R, left=False, adjoint=adj)
        x3 = torch.linalg.lu_solve(LUe, pive, R, left=False, adjoint=adj)
        print("   adjoint", adj, "solve eager-vs-static equal", bool(torch.equal(x1, x2)), "eager-vs-eager", bool(torch.equal(x1, x3)))
# now the real thing: auto over theta
_sweepgraphs.SweepGraphs.AUTO_THETA = True
ode2 = petsc_adjoint.ODEPetsc()
for k, v in {"ts_adapt_type": "none", "ts_arkimex_type": name, "snes_type": "ksponly"}.items():
    options.set_option(k, v)
ode2.setupTS(y0, fI, step_size=0.05, method="imex", implicit_form=True, imex_form=True, func2=fE, batch_size=8,
             linear_solver="torch", matrixfree_jacobian=False)
options.clear()
for it in range(5):
    for p in params:
        p.grad = None
    yin = (y0 * (1.0 + 0.1 * it)).requires_grad_(True)
    sol = ode2.odeint_adjoint(yin, t.to(dev))
    sol.abs().mean().backward()
    e = next(iter(ode2._graphs.values()), None)
    print("call", it, ode2.graph_status, "replay_diff", getattr(e, "replay_diff", None))
