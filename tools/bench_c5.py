#!/usr/bin/env python3
"""BASELINE config C5, one GPU's shard (64 x 1024 fp64 of the Burgers batch 512): ARKIMEX + ksponly +
linear_solver="torch" (examples-sinode/Burgers/run_a100_512.sh:20-23).  Wall time per solve, a phase
breakdown (factorisation / forward sweep / reverse sweep, with synchronisations), eager vs hipGraph."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from pnode_amd import options, petsc_adjoint, arkimex
src = open(os.path.join(ROOT, "tools", "bench_configs.py")).read().split("def run(")[0]
ns = {"__file__": os.path.join(ROOT, "tools", "bench_configs.py")}; exec(compile(src, "bench_configs_head", "exec"), ns)
BurgersIM, BurgersEX = ns["BurgersIM"], ns["BurgersEX"]
dev = torch.device("cuda:0")
n5, NT = 1024, int(os.environ.get("NT", 10))
types = sys.argv[1:] or ["3"]
torch.manual_seed(0)
y0 = torch.rand(64, n5, dtype=torch.float64, device=dev)
fI, fE = BurgersIM(n5).to(dev), BurgersEX(n5).to(dev)
t = torch.tensor([0.01 * NT], dtype=torch.float64)
params = [p for p in list(fI.parameters()) + list(fE.parameters()) if p.requires_grad]


def make(kind, graph):
    options.clear()
    for k, v in {"ts_adapt_type": "none", "ts_arkimex_type": kind, "snes_type": "ksponly"}.items():
        options.set_option(k, v)
    if graph:
        options.set_option("pn_graph_capture", 1)
    ode = petsc_adjoint.ODEPetsc()
    ode.setupTS(y0, fI, step_size=0.01, method="imex", implicit_form=True, imex_form=True, func2=fE, batch_size=64,
                linear_solver="torch", matrixfree_jacobian=False, fixed_jacobian=os.environ.get("FIXED", "0") == "1")
    options.clear()
    return ode


def solve(ode):
    for p in params:
        p.grad = None
    y = y0.detach().requires_grad_(True)
    ode.odeint_adjoint(y, t).abs().mean().backward()
    return torch.cat([p.grad.reshape(-1) for p in params]), y.grad


def timeit(ode, reps=5, warm=3):
    for _ in range(warm):
        solve(ode)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        solve(ode)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


for kind in types:
    e = make(kind, False)
    ge, ye = solve(e)
    dt = timeit(e)
    print("C5 shard imex %-6s eager : %7.2f ms/solve  %7.1f time-steps/s  (fixed_jacobian=%s)" % (kind, 1e3 * dt, NT / dt, os.environ.get("FIXED", "0")), flush=True)
    # phase breakdown (synchronised, so the sum exceeds the pipelined wall time)
    acc = {"factor": 0.0, "forward": 0.0, "reverse": 0.0}
    S = arkimex.ArkimexStepper
    orig = {k: getattr(S, k) for k in ("_direct_factor", "odeint", "adjoint_steps")}
    def wrap(name, key):
        def f(self, *a, **kw):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            r = orig[name](self, *a, **kw)
            torch.cuda.synchronize(); acc[key] += time.perf_counter() - t0
            return r
        return f
    S._direct_factor, S.odeint, S.adjoint_steps = wrap("_direct_factor", "factor"), wrap("odeint", "forward"), wrap("adjoint_steps", "reverse")
    for _ in range(3):
        solve(e)
    for k, v in orig.items():
        setattr(S, k, v)
    print("   phases per solve (ms): factorisation calls %.2f | forward sweep incl. them %.2f | reverse sweep incl. them %.2f"
          % (1e3 * acc["factor"] / 3, 1e3 * acc["forward"] / 3, 1e3 * acc["reverse"] / 3), flush=True)
    try:
        g = make(kind, True)
        gg, yg = None, None
        for _ in range(4):
            gg, yg = solve(g)
        if not g.graphs_captured:
            print("   hipGraph: not captured"); continue
        dtg = timeit(g)
        print("C5 shard imex %-6s graph : %7.2f ms/solve  %7.1f time-steps/s   bit-identical to eager: dtheta %s dy0 %s"
              % (kind, 1e3 * dtg, NT / dtg, bool(torch.equal(gg, ge)), bool(torch.equal(yg, ye))), flush=True)
    except Exception as exc:
        print("   hipGraph failed:", repr(exc)[:300], flush=True)
