#!/usr/bin/env python3
"""Measures every BASELINE.md configuration that fits one MI355X (beyond bench.py's headline
C3a) and writes gpurun_out/configs.json.  Each entry: time-steps/s (fwd+adjoint), per-time-step
latency, accepted steps, rejections, solver-kernel share where events are available."""
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402
from pnode_amd import _lib, options, petsc_adjoint  # noqa: E402
from problems import MLPFunc, SpiralFunc  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.load()


class ConvBlock(nn.Module):
    """BasicBlock2(64)-shaped func of BASELINE config C4 (reference examples-pnode/models/
    sqnxt_PETSc.py:70-121): five conv layers 64->32->32->32->32->64 (1x1, 1x1, 3x1, 1x3, 1x1) with
    eval-mode BatchNorm + ReLU, written from the layer shapes (not the reference's code)."""

    def __init__(self, dim=64):
        super().__init__()
        h = dim // 2
        self.layers = nn.ModuleList([
            nn.Conv2d(dim, h, 1), nn.Conv2d(h, h, 1), nn.Conv2d(h, h, (3, 1), padding=(1, 0)),
            nn.Conv2d(h, h, (1, 3), padding=(0, 1)), nn.Conv2d(h, dim, 1)])
        self.bns = nn.ModuleList([nn.BatchNorm2d(c) for c in (h, h, h, h, dim)])
        self.eval()

    def forward(self, t, x):
        for conv, bn in zip(self.layers, self.bns):
            x = torch.relu(bn(conv(x)))
        return x


class BurgersIM(nn.Module):
    """Fixed circular 3-point Laplacian alpha/dx^2 [1,-2,1] as a Conv1d, no trainable parameter
    (the shape of examples-sinode/Burgers/Burgers.py:170-195 with fixed_linear=True)."""

    def __init__(self, n, alpha=8e-4, dtype=torch.float64):
        super().__init__()
        self.A = nn.Conv1d(1, 1, 3, padding="same", padding_mode="circular", bias=False)
        dx = 1.0 / n
        self.A.weight = nn.Parameter(torch.tensor([[[alpha / dx ** 2, -2 * alpha / dx ** 2, alpha / dx ** 2]]]),
                                     requires_grad=False)
        self.to(dtype)

    def forward(self, t, y):
        return self.A(y.unsqueeze(1)).squeeze(1)


class BurgersEX(nn.Module):
    """Five Linear layers N -> 9N/8 -> 9N/8 -> 9N/8 -> 9N/8 -> N with ReLU, W ~ N(0, 0.1/sqrt-free as the
    reference: std 0.1), zero bias (Burgers.py:134-160)."""

    def __init__(self, n, dtype=torch.float64):
        super().__init__()
        w = n * 9 // 8
        dims = [n, w, w, w, w, n]
        layers = []
        g = torch.Generator().manual_seed(0)
        for i in range(5):
            lin = nn.Linear(dims[i], dims[i + 1])
            with torch.no_grad():
                lin.weight.copy_(torch.randn(lin.weight.shape, generator=g) * 0.1 / (dims[i] ** 0.5) * 3.0)
                lin.bias.zero_()
            layers.append(lin)
            if i < 4:
                layers.append(nn.ReLU())
        self.net = nn.Sequential(*layers).to(dtype)

    def forward(self, t, y):
        return self.net(y)


def run(name, func, y0, t, step, method, opts, reps=5, warm=3, dtype=torch.float32, **setup_kw):
    options.clear()
    for k, v in opts.items():
        options.set_option(k, v)
    ode = petsc_adjoint.ODEPetsc()
    ode.setupTS(y0, func, step_size=step, method=method, **setup_kw)
    params = [p for p in func.parameters()] + ([p for p in setup_kw["func2"].parameters()] if "func2" in setup_kw else [])
    params = [p for p in params if p.requires_grad]

    def solve():
        for p in params:
            p.grad = None
        y = y0.detach().requires_grad_(True)
        ode.odeint_adjoint(y, t).abs().mean().backward()

    for _ in range(warm):
        solve()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        solve()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    res = {"config": name, "method": method, "options": opts, "state_shape": list(y0.shape), "dtype": str(dtype),
           "accepted_steps": ode._nsteps, "rejections": int(lib.pn_ts_rejections(ode._ts)),
           "ms_per_solve": 1e3 * dt, "time_steps_per_s": ode._nsteps / dt, "us_per_time_step": 1e6 * dt / max(ode._nsteps, 1),
           "nfe_forward_per_solve": ode.nfe_forward // (reps + warm), "nfe_backward_per_solve": ode.nfe_backward // (reps + warm)}
    if not options.truthy(opts.get("pn_graph_capture"), False):
        lib.pn_prof_enable(1)
        solve()
        torch.cuda.synchronize()
        L, us, by = (ctypes.c_int64 * len(_lib.KERNEL_IDS))(), (ctypes.c_double * len(_lib.KERNEL_IDS))(), (ctypes.c_double * len(_lib.KERNEL_IDS))()
        lib.pn_prof_collect(len(L), L, us, by)
        lib.pn_prof_enable(0)
        res["solver_kernels"] = {n: {"launches": int(L[i]), "avg_us": us[i] / L[i], "GBps_moved": by[i] / us[i] / 1e3}
                                 for i, n in enumerate(_lib.KERNEL_IDS) if L[i]}
    print(json.dumps(res), flush=True)
    options.clear()
    return res


out = []
torch.manual_seed(0)
# C1: spiral demo, 20 x 1 x 2, 10 output times, rk4, fp64 (ode_demo_petsc.py training shape)
f = SpiralFunc(torch.float64).to(dev)
out.append(run("C1 spiral 20x1x2, 10 outputs", f, torch.randn(20, 1, 2, dtype=torch.float64, device=dev),
               torch.linspace(0.0, 0.225, 10, dtype=torch.float64), 0.025, "rk4", {"ts_adapt_type": "none"}, dtype=torch.float64))
# C2: batched spiral 4096 x 2, rk4 100 steps
f = SpiralFunc(torch.float32).to(dev)
y0 = torch.randn(4096, 2, device=dev)
for o in ({"ts_adapt_type": "none", "ts_trajectory_solution_only": "0"},
          {"ts_adapt_type": "none", "ts_trajectory_solution_only": "0", "pn_graph_capture": "1"}):
    out.append(run("C2 batched spiral 4096x2", f, y0, torch.tensor([2.5]), 0.025, "rk4", o))
# C3b: MLP 4096 x 512, dopri5 adaptive rtol=atol=1e-4, h0 = 0.01, T = 1, max_cps = 50
f = MLPFunc(512, torch.float32).to(dev)
y0 = torch.randn(4096, 512, device=dev)
out.append(run("C3b MLP 4096x512 dopri5 adaptive max_cps=50", f, y0, torch.tensor([1.0]), 0.01, "dopri5",
               {"ts_trajectory_max_cps_ram": "50"}))
out.append(run("C3b' same, all steps + stages kept", f, y0, torch.tensor([1.0]), 0.01, "dopri5",
               {"ts_trajectory_solution_only": "0"}))
out.append(run("C3a fp64 (same as headline, double)", MLPFunc(512, torch.float64).to(dev), y0.double(), torch.tensor([0.2]), 0.01, "rk4",
               {"ts_adapt_type": "none", "ts_trajectory_solution_only": "0"}, dtype=torch.float64))
# C4 shard: conv block on 128 x 64 x 32 x 32 (one GPU's share of batch 1024), rk4, t = [1.0], Nt in {1, 4}
f = ConvBlock(64).to(dev)
y0 = torch.randn(128, 64, 32, 32, device=dev)
for nt in (1, 4):
    out.append(run("C4 conv block 128x64x32x32, Nt=%d" % nt, f, y0, torch.tensor([1.0]), 1.0 / nt, "rk4",
                   {"ts_adapt_type": "none", "ts_trajectory_solution_only": "0"}))
# C5 shard: SINODE Burgers, IMEX split, 64 x 1024 fp64 (one GPU's share of batch 512), 10 steps;
# the run script's variants that are available: ARKIMEX type 3 + ksponly + torch LU; cn / beuler matrix-free
n5 = 1024
y0 = torch.rand(64, n5, dtype=torch.float64, device=dev)
fI, fE = BurgersIM(n5).to(dev), BurgersEX(n5).to(dev)
out.append(run("C5 shard Burgers IMEX type 3, ksponly, linear_solver=torch", fI, y0, torch.tensor([0.1], dtype=torch.float64), 0.01,
               "imex", {"ts_adapt_type": "none", "ts_arkimex_type": "3", "snes_type": "ksponly"}, reps=3, warm=2,
               dtype=torch.float64, implicit_form=True, imex_form=True, func2=fE, batch_size=64, linear_solver="torch",
               matrixfree_jacobian=False))
out.append(run("C5 shard Burgers IMEX type 3, Newton-GMRES (matrix-free)", fI, y0, torch.tensor([0.1], dtype=torch.float64), 0.01,
               "imex", {"ts_adapt_type": "none", "ts_arkimex_type": "3", "snes_type": "ksponly"}, reps=2, warm=1,
               dtype=torch.float64, implicit_form=True, imex_form=True, func2=fE, batch_size=64))


class BurgersFull(nn.Module):
    def __init__(self, a, b):
        super().__init__()
        self.a, self.b = a, b

    def forward(self, t, y):
        return self.a(t, y) + self.b(t, y)


out.append(run("C5 shard Burgers, cn implicit (matrix-free Newton-GMRES)", BurgersFull(fI, fE), y0,
               torch.tensor([0.1], dtype=torch.float64), 0.01, "cn", {"ts_adapt_type": "none"}, reps=2, warm=1,
               dtype=torch.float64, implicit_form=True))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "configs.json"), "w"), indent=1)
