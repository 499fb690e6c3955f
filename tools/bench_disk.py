#!/usr/bin/env python3
"""Cost of the trajectory's disk tier at C3a (4096 x 512 fp32, rk4, 100 steps, eager launches): -ts_trajectory_type memory vs
basic, store-all (32 MiB per checkpoint) and solution-only (8 MiB), files under /tmp (tmpfs or local disk of the box)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from pnode_amd import options, petsc_adjoint
from problems import MLPFunc
dev = torch.device("cuda:0"); NT = 100
f = MLPFunc(512, torch.float32).to(dev)
y0 = torch.randn(4096, 512, device=dev); t = torch.tensor([0.01 * NT])
for so in (0, 1):
    for ttype in ("memory", "basic"):
        options.clear()
        for k, v in {"ts_adapt_type": "none", "ts_trajectory_solution_only": so, "ts_trajectory_type": ttype,
                     "ts_trajectory_dirname": os.environ.get("PN_CKPT_DIR", "/tmp/pn_ckpt"), "pn_trajectory_retain_graph": 0}.items():
            options.set_option(k, v)
        ode = petsc_adjoint.ODEPetsc(); ode.setupTS(y0, f, step_size=0.01, method="rk4"); options.clear()
        def solve():
            for p in f.parameters(): p.grad = None
            y = y0.detach().requires_grad_(True)
            t0 = time.perf_counter(); out = ode.odeint_adjoint(y, t); torch.cuda.synchronize(); t1 = time.perf_counter()
            out.abs().mean().backward(); torch.cuda.synchronize(); t2 = time.perf_counter()
            return t1 - t0, t2 - t1
        solve()
        r = [solve() for _ in range(3)]
        fw = min(x[0] for x in r); bw = min(x[1] for x in r)
        st = ode._traj.stats() if ode._traj.on_disk else {}
        print("solution_only=%d %-6s forward %7.1f ms  reverse %7.1f ms  -> %6.1f time-steps/s  %s"
              % (so, ttype, 1e3 * fw, 1e3 * bw, NT / (fw + bw), ("GB written %.2f, waits %d" % (st["bytes_written"] / 1e9, st["waits"])) if st else ""), flush=True)
        ode._traj = None
