#!/usr/bin/env python3
"""Round 4: what recording autograd's tape while a reversed step's stage values are recomputed is worth (DESIGN section 3,
difference 20).  C3a (4096 x 512 fp32, rk4, 100 steps) in the modes that recompute -- PETSc's solution-only default and
checkpoint budgets -- with the tapes (default) and without (-pn_trajectory_retain_graph 0, the reference's way), eager launches
and the default launch mode (hipGraph replay once validated); and the adaptive workload of bench.py --config c3b --stiff."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from pnode_amd import options, petsc_adjoint
from problems import MLPFunc, SwitchedMLPFunc
dev = torch.device("cuda:0")
torch.manual_seed(0)
y0 = torch.randn(4096, 512, device=dev)


def run(f, method, t, extra, reps=3, warm=4):
    options.clear()
    for k, v in extra.items(): options.set_option(k, v)
    ode = petsc_adjoint.ODEPetsc(); ode.setupTS(y0, f, step_size=0.01, method=method); options.clear()
    def solve():
        for p in f.parameters(): p.grad = None
        y = y0.detach().requires_grad_(True); ode.odeint_adjoint(y, t).abs().mean().backward()
        return torch.cat([p.grad.reshape(-1) for p in f.parameters()])
    import gc
    gc.collect(); torch.cuda.empty_cache()
    for _ in range(warm): g = solve()
    torch.cuda.synchronize(); nf, nb = ode.nfe_forward, ode.nfe_backward; t0 = time.perf_counter()
    for _ in range(reps): solve()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    return g, ode._nsteps / dt, (ode.nfe_forward - nf) // reps, (ode.nfe_backward - nb) // reps, ode.graph_status


fs = SwitchedMLPFunc(512, torch.float32).to(dev)
ref = None
for retain in ("auto", 0, "auto", 0):          # (alternating: the first adaptive solves of a process run slower)
    g, rate, nf, nb, st = run(fs, "dopri5", torch.tensor([SwitchedMLPFunc.T_END]), {"ts_trajectory_max_cps_ram": 50, "pn_trajectory_retain_graph": retain},
                              reps=4, warm=2)
    ref = g if ref is None else ref
    print("C3b --stiff dopri5, max_cps 50               %-14s tapes %-4s: %6.1f time-steps/s  NFE-F %5d NFE-B %5d  bitwise %s"
          % (st[:14], "yes" if retain == "auto" else "no", rate, nf, nb, bool(torch.equal(g, ref))), flush=True)

f = MLPFunc(512, torch.float32).to(dev)
t1 = torch.tensor([1.0])
ref = None
for label, base in (("solution-only (every state)", {"ts_trajectory_solution_only": 1}), ("max_cps 50, state-only", {"ts_trajectory_max_cps_ram": 50}),
                    ("max_cps 10, state-only", {"ts_trajectory_max_cps_ram": 10})):
    for launch in ({"pn_graph_capture": 0}, {}):
        for retain in ("auto", 0):
            g, rate, nf, nb, st = run(f, "rk4", t1, dict(base, ts_adapt_type="none", pn_trajectory_retain_graph=retain, **launch))
            ref = g if ref is None else ref
            print("C3a rk4 x 100  %-28s %-14s tapes %-4s: %6.1f time-steps/s  NFE-F %4d NFE-B %4d  bitwise %s"
                  % (label, st[:14], "yes" if retain == "auto" else "no", rate, nf, nb, bool(torch.equal(g, ref))), flush=True)
