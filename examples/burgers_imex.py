#!/usr/bin/env python3
"""Semi-implicit neural ODE on viscous Burgers, through the drop-in surface -- the workload of the reference's
examples-sinode/Burgers/Burgers.py in its `--imex --linear_solver torch` form: u_t = nu u_xx (stiff, linear,
treated implicitly by a fixed circular finite-difference operator) + N(u) (a small network, explicit), trained to
reproduce trajectories of the true equation u_t = nu u_xx - u u_x.  Written for this package; synthetic data.

    python examples/burgers_imex.py -ts_adapt_type none -ts_arkimex_type 3 -snes_type ksponly
    python examples/burgers_imex.py -ts_adapt_type none -ts_arkimex_type l2 -snes_type ksponly -pn_graph_capture 1
"""
import argparse
import math
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import pnode  # noqa: E402  (before the first CUDA call)

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=128, help="grid points")
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--nu", type=float, default=5e-3)
ap.add_argument("--step_size", type=float, default=0.01)
ap.add_argument("--horizon", type=int, default=10, help="time steps per training window")
ap.add_argument("--niters", type=int, default=200)
ap.add_argument("--lr", type=float, default=2e-3)
args, solver_argv = ap.parse_known_args()
pnode.init([sys.argv[0]] + solver_argv)

import torch  # noqa: E402
import torch.nn as nn  # noqa: E402
from pnode import petsc_adjoint  # noqa: E402

if not torch.cuda.is_available():
    sys.exit("this package runs on MI355X HIP devices only")
dev, dtype = torch.device("cuda:0"), torch.float64
torch.manual_seed(0)
n, dx = args.n, 1.0 / args.n


def laplacian(u):
    return (torch.roll(u, 1, -1) - 2.0 * u + torch.roll(u, -1, -1)) / dx ** 2


class Diffusion(nn.Module):                      # implicit part: fixed, linear, no trainable parameter
    def forward(self, t, u):
        return args.nu * laplacian(u)


class Advection(nn.Module):                      # the truth's explicit part
    def forward(self, t, u):
        return -u * (torch.roll(u, -1, -1) - torch.roll(u, 1, -1)) / (2.0 * dx)


class Learned(nn.Module):                        # the model's explicit part: a stencil network on (u_{i-1}, u_i, u_{i+1})
    def __init__(self):
        super().__init__()
        self.net = nn.Sequential(nn.Linear(3, 32), nn.Tanh(), nn.Linear(32, 1))

    def forward(self, t, u):
        st = torch.stack([torch.roll(u, 1, -1), u, torch.roll(u, -1, -1)], dim=-1)
        return self.net(st).squeeze(-1) / dx * 0.05


x = torch.arange(n, dtype=dtype, device=dev) * dx
phase = torch.rand(args.batch, 1, dtype=dtype, device=dev)
amp = 0.5 + torch.rand(args.batch, 1, dtype=dtype, device=dev)
u0 = amp * torch.sin(2 * math.pi * (x + phase)) + 0.3 * torch.cos(4 * math.pi * (x - phase))
times = torch.linspace(0.0, args.step_size * args.horizon, 3, dtype=dtype)

kw = dict(step_size=args.step_size, method="imex", implicit_form=True, imex_form=True, batch_size=args.batch,
          linear_solver="torch", matrixfree_jacobian=False, fixed_jacobian=True)
truth_solver = petsc_adjoint.ODEPetsc()
truth_solver.setupTS(u0, Diffusion(), func2=Advection(), enable_adjoint=False, **kw)
with torch.no_grad():
    target = truth_solver.odeint_adjoint(u0, times)

model = Learned().to(dev).to(dtype)
solver = petsc_adjoint.ODEPetsc()
solver.setupTS(u0, Diffusion(), func2=model, **kw)
opt = torch.optim.Adam(model.parameters(), lr=args.lr)
t0 = time.time()
for it in range(1, args.niters + 1):
    opt.zero_grad()
    pred = solver.odeint_adjoint(u0, times)
    loss = (pred - target).pow(2).mean()
    loss.backward()
    opt.step()
    if it == 1 or it % 50 == 0:
        print("iter %4d | loss %.3e | %d time steps per solve | graphs %s | %.1f s"
              % (it, loss.item(), solver.num_steps, bool(solver.graphs_captured), time.time() - t0), flush=True)
