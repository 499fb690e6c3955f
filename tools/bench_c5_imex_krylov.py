#!/usr/bin/env python3
"""C5 shard (Burgers IMEX split, 64 x 1024 fp64, 10 steps), ARKIMEX type 3 + -snes_type ksponly with the reference's DEFAULT linear
solver (linear_solver="petsc": matrix-free GMRES on shift*I - d funcIM/du, pa.py:547, 701-714) instead of the run script's torch LU:
round 2's host-driven loop against the device-resident GMRES + replayed linearisations.  funcIM = the fixed circular Laplacian, as
nn.Conv1d in double (the reference's layer) and written with torch.roll."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, torch.nn as nn
from pnode_amd import options, petsc_adjoint
from problems import BurgersEX, BurgersIM
dev = torch.device("cuda:0"); n5, NT = 1024, 10
torch.manual_seed(0)
y0 = torch.rand(64, n5, dtype=torch.float64, device=dev)
t = torch.tensor([0.01 * NT], dtype=torch.float64)


class StencilIM(nn.Module):
    def __init__(s, n, alpha=8e-4):
        super().__init__(); s.k = alpha * float(n) ** 2
    def forward(s, t, y): return s.k * (torch.roll(y, 1, -1) - 2.0 * y + torch.roll(y, -1, -1))


for fname, fI in (("conv1d", BurgersIM(n5).to(dev)), ("stencil", StencilIM(n5).to(dev))):
    fE = BurgersEX(n5).to(dev)
    params = [p for p in list(fI.parameters()) + list(fE.parameters()) if p.requires_grad]
    for label, extra in (("host loop, eager operator (round 2)", {"pn_krylov": "host", "pn_krylov_graph": 0}),
                         ("device GMRES, eager operator", {"pn_krylov_graph": 0}), ("default", {})):
        options.clear()
        for k, v in dict({"ts_adapt_type": "none", "ts_arkimex_type": "3", "snes_type": "ksponly"}, **extra).items():
            options.set_option(k, v)
        ode = petsc_adjoint.ODEPetsc()
        ode.setupTS(y0, fI, step_size=0.01, method="imex", implicit_form=True, imex_form=True, func2=fE, batch_size=64)
        options.clear()
        def solve():
            for p in params: p.grad = None
            y = y0.detach().requires_grad_(True); ode.odeint_adjoint(y, t).abs().mean().backward()
        for _ in range(3): solve()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3): solve()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
        th = ode._theta
        print("C5 shard IMEX 3 ksponly, matrix-free, funcIM %-7s %-38s %8.2f ms/solve %6.1f time-steps/s  gmres its/solve %d, host syncs/solve %d, second passes %d, "
              "captured linearisations %d%s" % (fname, label, 1e3 * dt, NT / dt, th.linear_its, th.host_syncs, th.second_passes, th._op_stats[1],
                                               "" if not th._graphs_dropped else "  [graphs dropped: %s]" % th._graphs_dropped), flush=True)
