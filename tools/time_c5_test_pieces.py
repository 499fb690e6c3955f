"""Where the time of tests/test_gpu_configs.py::test_c5_burgers_imex_shard_64x1024 goes on the GPU box (host oracle vs device)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from problems import BurgersIM, BurgersEX
from oracle.arkimex_oracle import odeint_adjoint_arkimex, odeint_adjoint_arkimex_direct
from pnode_amd import options, petsc_adjoint
print("cpus", os.cpu_count(), "torch threads", torch.get_num_threads(), flush=True)
n, B, h = 1024, 64, 1e-3
torch.manual_seed(0)
y0 = torch.rand(B, n, dtype=torch.float64); w = torch.randn(1, B, n, dtype=torch.float64)
dev = torch.device("cuda:0")
for threads in (None, 16):
    if threads:
        torch.set_num_threads(threads)
    for name in ("3",):
        t = torch.tensor([4 * h], dtype=torch.float64)
        fI, fE = BurgersIM(n), BurgersEX(n)
        y2 = y0[:2].clone().requires_grad_(True)
        t0 = time.time(); p = odeint_adjoint_arkimex(fI, fE, y2, t, h, name); (p * w[:, :2]).sum().backward(); t1 = time.time()
        y3 = y0.clone().requires_grad_(True)
        p = odeint_adjoint_arkimex_direct(fI, fE, y3, t, h, name); (p * w).sum().backward(); t2 = time.time()
        print("threads", threads, name, "exact-newton 2 rows %.1fs" % (t1 - t0), "direct 64 rows %.1fs" % (t2 - t1), flush=True)
fI, fE = BurgersIM(n).to(dev), BurgersEX(n).to(dev)
opts = {"ts_adapt_type": "none", "ts_arkimex_type": "3", "snes_type": "ksponly"}
for extra, steps, calls in (({}, 4, 1), ({}, 10, 1), ({"pn_graph_capture": 1}, 10, 4)):
    options.clear()
    for k, v in dict(opts, **extra).items():
        options.set_option(k, v)
    ode = petsc_adjoint.ODEPetsc()
    yd = y0.to(dev)
    ode.setupTS(yd, fI, step_size=h, method="imex", func2=fE, implicit_form=True, imex_form=True, batch_size=B, linear_solver="torch", matrixfree_jacobian=False)
    options.clear()
    for it in range(calls):
        torch.cuda.synchronize(); t0 = time.time()
        y = yd.clone().requires_grad_(True)
        out = ode.odeint_adjoint(y, torch.tensor([steps * h], dtype=torch.float64))
        (out * w.to(dev)).sum().backward()
        torch.cuda.synchronize()
        print("device", extra, steps, "call", it, "%.2fs" % (time.time() - t0), flush=True)
