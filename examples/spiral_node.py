#!/usr/bin/env python3
"""Neural ODE on the 2-D spiral, through the drop-in surface (`from pnode import petsc_adjoint`).

The workload of the reference's examples-pnode/ode_demo_petsc.py (ground truth y' = y^3 A integrated
with ODEPetsc.odeint; a Linear(2,50)-Tanh-Linear(50,2) model on y^3 trained on random windows of the
trajectory with odeint_adjoint), written for this package.  Solver options are PETSc-style and go
after the script's own, exactly as with the reference:

    python examples/spiral_node.py --niters 200 -ts_adapt_type none -ts_trajectory_type memory
    python examples/spiral_node.py --niters 200 --method dopri5            # adaptive steps
    python examples/spiral_node.py --niters 200 -ts_adapt_type none -pn_graph_capture 0   # plain launches (fixed-step
                                            # sweeps are replayed from hipGraphs by default once validated: ode.graph_status)
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import pnode  # noqa: E402  (import before the first CUDA call)

ap = argparse.ArgumentParser()
ap.add_argument("--method", default="rk4", choices=["euler", "midpoint", "rk2", "bosh3", "rk4", "dopri5"])
ap.add_argument("--step_size", type=float, default=0.025)
ap.add_argument("--data_size", type=int, default=1001)
ap.add_argument("--batch_time", type=int, default=10)
ap.add_argument("--batch_size", type=int, default=20)
ap.add_argument("--niters", type=int, default=400)
ap.add_argument("--test_freq", type=int, default=50)
ap.add_argument("--lr", type=float, default=5e-3)
ap.add_argument("--double_prec", action="store_true")
ap.add_argument("--seed", type=int, default=0)
args, solver_argv = ap.parse_known_args()
pnode.init([sys.argv[0]] + solver_argv)            # where the reference calls petsc4py.init(sys.argv)

import torch  # noqa: E402
import torch.nn as nn  # noqa: E402
from pnode import petsc_adjoint  # noqa: E402

if not torch.cuda.is_available():
    sys.exit("this package runs on MI355X HIP devices only")
dev = torch.device("cuda:0")
dtype = torch.float64 if args.double_prec else torch.float32
torch.manual_seed(args.seed)

A = torch.tensor([[-0.1, 2.0], [-2.0, -0.1]], dtype=dtype, device=dev)
y_start = torch.tensor([[2.0, 0.0]], dtype=dtype, device=dev)
times = torch.linspace(0.0, 25.0, args.data_size, dtype=dtype)


class Truth(nn.Module):
    def forward(self, t, y):
        return torch.mm(y ** 3, A)


class Model(nn.Module):
    def __init__(self):
        super().__init__()
        self.net = nn.Sequential(nn.Linear(2, 50), nn.Tanh(), nn.Linear(50, 2))
        for m in self.net:
            if isinstance(m, nn.Linear):
                nn.init.normal_(m.weight, mean=0.0, std=0.1)
                nn.init.zeros_(m.bias)
        self.nfe = 0

    def forward(self, t, y):
        self.nfe += 1
        return self.net(y ** 3)


truth_solver = petsc_adjoint.ODEPetsc()
truth_solver.setupTS(y_start, Truth(), step_size=args.step_size, method=args.method, enable_adjoint=False)
with torch.no_grad():
    truth = truth_solver.odeint(y_start, times)                       # (data_size, 1, 2)

model = Model().to(dev).to(dtype)
window_t = times[: args.batch_time]
train = petsc_adjoint.ODEPetsc()                                      # separate objects for training and testing,
train.setupTS(torch.empty(args.batch_size, 1, 2, dtype=dtype, device=dev), model, step_size=args.step_size, method=args.method)
test = petsc_adjoint.ODEPetsc()                                       # as the reference's callers do
test.setupTS(y_start, model, step_size=args.step_size, method=args.method, enable_adjoint=False)
opt = torch.optim.RMSprop(model.parameters(), lr=args.lr)

t0 = time.time()
for it in range(1, args.niters + 1):
    starts = torch.randperm(args.data_size - args.batch_time, device=dev)[: args.batch_size]
    y0 = truth[starts]                                                 # (M, 1, 2)
    target = torch.stack([truth[starts + k] for k in range(args.batch_time)], dim=0)
    opt.zero_grad()
    pred = train.odeint_adjoint(y0, window_t)
    loss = (pred - target).abs().mean()
    loss.backward()
    opt.step()
    if it % args.test_freq == 0 or it == 1:
        with torch.no_grad():
            full = test.odeint_adjoint(y_start, times)
            err = (full - truth).abs().mean().item()
        print("iter %4d | train loss %.6f | whole-trajectory error %.6f | steps/solve %d | func evals %d | %.1f s"
              % (it, loss.item(), err, train.num_steps, model.nfe, time.time() - t0), flush=True)
print("done: %d iterations in %.1f s (%s, %s)" % (args.niters, time.time() - t0, args.method, "fp64" if args.double_prec else "fp32"))
