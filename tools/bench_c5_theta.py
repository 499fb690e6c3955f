#!/usr/bin/env python3
"""C5 shard (Burgers, 64 x 1024 fp64, 10 steps) with the implicit theta methods on the FULL right-hand side
(diffusion + network), as examples-sinode/Burgers/run_a100_512.sh:26-27 runs them: matrix-free Newton-GMRES
(--linear_solver petsc) against the direct solve with the frozen one-sample Jacobian (--linear_solver torch)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, torch.nn as nn
from pnode_amd import options, petsc_adjoint
src = open(os.path.join(ROOT, "tools", "bench_configs.py")).read().split("def run(")[0]
ns = {"__file__": os.path.join(ROOT, "tools", "bench_configs.py")}; exec(compile(src, "bench_configs_head", "exec"), ns)
BurgersIM, BurgersEX = ns["BurgersIM"], ns["BurgersEX"]
ONLY_KRYLOV = int("--only-default" in sys.argv)      # profiling runs: the default Krylov configuration alone
if "--tunableop" in sys.argv:                        # let PyTorch's TunableOp pick func's fp64 GEMM kernels (M = 64 rows: the stock
    torch.cuda.tunable.enable(True)                  # heuristic runs them at ~2.6 TFLOP/s, 65 us each; they are 70 % of the GPU time
    torch.cuda.tunable.set_filename("/tmp/pnode_amd_tunableop_c5.csv")     # of a matrix-free solve, profiles/r03_krylov_*)
dev = torch.device("cuda:0"); n5, NT = 1024, 10
torch.manual_seed(0)
y0 = torch.rand(64, n5, dtype=torch.float64, device=dev)
class StencilIM(nn.Module):
    """The same fixed circular Laplacian alpha/dx^2 [1, -2, 1] written with torch.roll instead of nn.Conv1d.  MIOpen has no
    fp64 convolution: PyTorch's fallback for a double Conv1d loops over the batch (128 tiny launches, 1.6 ms of host time per
    call at 64 x 1024) -- func's cost, not the solver's; this variant shows the solver without it."""
    def __init__(s, n, alpha=8e-4):
        super().__init__(); s.k = alpha * float(n) ** 2
    def forward(s, t, y): return s.k * (torch.roll(y, 1, -1) - 2.0 * y + torch.roll(y, -1, -1))
class Full(nn.Module):
    def __init__(s, stencil):
        super().__init__(); s.fI, s.fE = (StencilIM(n5) if stencil else BurgersIM(n5)).to(dev), BurgersEX(n5).to(dev)
    def forward(s, t, y): return s.fI(t, y) + s.fE(t, y)
t = torch.tensor([0.01 * NT], dtype=torch.float64)
CONFIGS = [
    ("petsc", {"pn_krylov": "host", "pn_krylov_graph": 0}),       # round 2: host-driven GMRES, eager operator
    ("petsc", {"pn_krylov_graph": 0}),                            # device-resident GMRES, eager operator
    ("petsc", {}),                                                # the default: + replayed linearisations
    ("torch", {}), ("torch", {"snes_type": "ksponly"}), ("torch", {"snes_type": "ksponly", "pn_graph_capture": 1}),
]
if ONLY_KRYLOV:
    CONFIGS = [("petsc", {})]
    if "--form" in sys.argv:       # force the arithmetic form of the captured product J v: jvp | dvjp (default: the eager path's form)
        CONFIGS = [("petsc", {"pn_krylov_graph_form": sys.argv[sys.argv.index("--form") + 1]})]
VARIANTS = [("stencil", True)] if ONLY_KRYLOV else [("conv1d", False), ("stencil", True)]
if "--conv1d" in sys.argv:         # the reference's own layer (nn.Conv1d in double), alone
    VARIANTS = [("conv1d", False)]
for (fname, stencil), method in [(v, m) for v in VARIANTS for m in ("cn", "beuler")]:
    f = Full(stencil); params = [p for p in f.parameters() if p.requires_grad]
    for ls, extra in CONFIGS:
        if stencil and ls == "torch" and extra:
            continue
        options.clear(); options.set_option("ts_adapt_type", "none")
        for k, v in extra.items(): options.set_option(k, v)
        ode = petsc_adjoint.ODEPetsc()
        ode.setupTS(y0, f, step_size=0.01, method=method, implicit_form=True, batch_size=64, linear_solver=ls)
        options.clear()
        def solve():
            for p in params: p.grad = None
            y = y0.detach().requires_grad_(True); ode.odeint_adjoint(y, t).abs().mean().backward()
        try:
            for _ in range(4): solve()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(3): solve()
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
            th = ode._theta
            print("C5 shard %-7s %-6s linear_solver=%-5s %-44s %8.2f ms/solve %7.1f time-steps/s  newton its/solve %d, gmres its/solve %d, "
                  "host syncs/solve %d, second passes %d, captured linearisations %d%s"
                  % (fname, method, ls, str(extra), 1e3 * dt, NT / dt, th.newton_its, th.linear_its, th.host_syncs, th.second_passes, th._op_stats[1],
                     "" if not th._graphs_dropped else "  [graphs dropped: %s]" % th._graphs_dropped), flush=True)
        except Exception as exc:
            print("C5 shard %s %s %s %s FAILED: %r" % (fname, method, ls, extra, exc), flush=True)
