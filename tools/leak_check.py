#!/usr/bin/env python3
"""Many solves per mode; allocated device memory must be flat after warm-up."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from pnode_amd import options, petsc_adjoint
from problems import MLPFunc, TimeDependent, DiffusionIM, ReactionEX
dev = torch.device("cuda:0")
def run(name, opts, make, iters=60):
    options.clear()
    for k, v in opts.items(): options.set_option(k, v)
    ode, f, y0, t, extra = make()
    params = [p for p in f.parameters()] + ([p for p in extra.get("func2").parameters()] if "func2" in extra else [])
    mem = []
    for it in range(iters):
        for p in params: p.grad = None
        y = y0.detach().requires_grad_(True)
        ode.odeint_adjoint(y, t).abs().mean().backward()
        if it in (10, iters - 1):
            torch.cuda.synchronize(); mem.append(torch.cuda.memory_allocated())
    print("%-28s allocated after 10 its %8.1f MiB, after %d its %8.1f MiB  %s" % (name, mem[0] / 2**20, iters, mem[1] / 2**20, "OK" if abs(mem[1] - mem[0]) < 2**20 else "GROWS"), flush=True)
    del ode; torch.cuda.empty_cache()
def mlp():
    f = MLPFunc(128, torch.float32).to(dev); y0 = torch.randn(512, 128, device=dev)
    ode = petsc_adjoint.ODEPetsc(); ode.setupTS(y0, f, step_size=0.05, method="rk4"); return ode, f, y0, torch.tensor([0.0, 0.5, 1.0]), {}
def mlp_dopri():
    f = MLPFunc(128, torch.float32).to(dev); y0 = torch.randn(512, 128, device=dev)
    ode = petsc_adjoint.ODEPetsc(); ode.setupTS(y0, f, step_size=0.05, method="dopri5"); return ode, f, y0, torch.tensor([1.0]), {}
def theta():
    f = TimeDependent(6, torch.float64).to(dev); y0 = torch.randn(64, 6, dtype=torch.float64, device=dev)
    ode = petsc_adjoint.ODEPetsc(); ode.setupTS(y0, f, step_size=0.1, method="cn", implicit_form=True); return ode, f, y0, torch.tensor([0.5], dtype=torch.float64), {}
def imex():
    fI, fE = DiffusionIM(6).to(dev), ReactionEX(6).to(dev); y0 = torch.randn(8, 6, dtype=torch.float64, device=dev)
    ode = petsc_adjoint.ODEPetsc(); ode.setupTS(y0, fI, step_size=0.05, method="imex", implicit_form=True, imex_form=True, func2=fE, batch_size=8, linear_solver="torch")
    return ode, fI, y0, torch.tensor([0.2], dtype=torch.float64), {"func2": fE}
base = {"ts_adapt_type": "none", "ts_trajectory_solution_only": "0"}
run("eager store-all", dict(base, pn_graph_capture="0"), mlp)
run("eager solution-only", {"ts_adapt_type": "none", "pn_graph_capture": "0"}, mlp)
run("eager budget 3", {"ts_adapt_type": "none", "ts_trajectory_max_cps_ram": "3", "pn_graph_capture": "0"}, mlp)
run("retain-graph", dict(base, pn_trajectory_retain_graph="1", pn_graph_capture="0"), mlp)
run("default launch mode (auto)", {"ts_adapt_type": "none"}, mlp)
run("two-level budget 2 + 3", {"ts_adapt_type": "none", "ts_trajectory_max_cps_ram": "2", "ts_trajectory_max_cps_disk": "3",
                               "ts_trajectory_dirname": "/tmp/pn_leak_ckpt"}, mlp, iters=30)
run("hipGraph", dict(base, pn_graph_capture="1"), mlp)
run("hipGraph + retain", dict(base, pn_graph_capture="1", pn_trajectory_retain_graph="1"), mlp)
run("dopri5 adaptive", {}, mlp_dopri)
run("cn (theta)", {"ts_adapt_type": "none"}, theta, iters=30)
run("cn, replayed linearisations", {"ts_adapt_type": "none", "pn_krylov_graph": "1"}, theta, iters=30)
run("cn, forward-mode graphs", {"ts_adapt_type": "none", "pn_krylov_graph": "1", "pn_krylov_graph_form": "jvp"}, theta, iters=30)
run("cn, host-driven GMRES", {"ts_adapt_type": "none", "pn_krylov": "host", "pn_krylov_graph": "0"}, theta, iters=30)
run("disk tier + budget 3", {"ts_adapt_type": "none", "ts_trajectory_type": "basic", "ts_trajectory_max_cps_ram": "3",
                             "ts_trajectory_dirname": "/tmp/pn_leak_ckpt"}, mlp, iters=30)
run("imex type 3, torch LU", {"ts_adapt_type": "none", "snes_type": "ksponly"}, imex, iters=30)
