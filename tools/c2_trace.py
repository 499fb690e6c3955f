#!/usr/bin/env python3
"""BASELINE config C2 (4096 x 2 fp32, rk4 x 100) replayed from hipGraphs -- the workload of the small-N question
(SURVEY 8a-K last row): run under `rocprofv3 --kernel-trace` and summarise with tools/trace_stats.py to see
how much of a time step belongs to the solver's own launches and how much to func's.
  rocprofv3 --kernel-trace --output-format csv -d /tmp/c2 -- python3 tools/c2_trace.py [eager]"""
import os
import sys
import time

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from pnode_amd import options, petsc_adjoint  # noqa: E402
from problems import SpiralFunc  # noqa: E402

dev = torch.device("cuda:0")
graph = not (len(sys.argv) > 1 and sys.argv[1] == "eager")
options.set_option("ts_adapt_type", "none")
options.set_option("ts_trajectory_solution_only", "0")
if graph:
    options.set_option("pn_graph_capture", "1")
torch.manual_seed(0)
func = SpiralFunc(torch.float32).to(dev)
y0 = torch.randn(4096, 2, device=dev)
t = torch.tensor([2.5])
ode = petsc_adjoint.ODEPetsc()
ode.setupTS(y0, func, step_size=0.025, method="rk4")


def solve():
    for p in func.parameters():
        p.grad = None
    y = y0.detach().requires_grad_(True)
    ode.odeint_adjoint(y, t).abs().mean().backward()


for _ in range(4):            # 2 eager + capture + 1 replay (graph mode)
    solve()
torch.cuda.synchronize()
t0 = time.perf_counter()
K = 10
for _ in range(K):
    solve()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
print("C2 %s: %.1f time-steps/s, %.1f us per time step (fwd+adjoint), graphs %s, tapes %s"
      % ("graph" if graph else "eager", ode._nsteps / dt, 1e6 * dt / ode._nsteps, ode.graphs_captured, ode._tapes is not None))
