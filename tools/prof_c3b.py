#!/usr/bin/env python3
"""BASELINE config 3 as written (C3b): MLP 4096 x 512 fp32, dopri5 adaptive (rtol = atol = 1e-4, h0 = 0.01, T = 1), adjoint on,
-ts_trajectory_max_cps_ram 50.  Meant to run under `rocprofv3 --kernel-trace --stats`: the fused solution-update +
error-norm kernel (pn_combine_wrms_kernel, one launch per step attempt) is the hot kernel of row a-4; its algorithmic bytes
are 7 state vectors per launch for 5dp (u_{n+1} = Y_6 and the six stage derivatives with a non-zero error weight).

  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_c3b -- python3 tools/prof_c3b.py [--solves 3]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import torch  # noqa: E402
from pnode_amd import options, petsc_adjoint  # noqa: E402
from problems import MLPFunc  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--solves", type=int, default=3)
ap.add_argument("--batch", type=int, default=4096)
ap.add_argument("--dtype", default="f32")
a = ap.parse_args()
dev = torch.device("cuda:0")
dt = torch.float32 if a.dtype == "f32" else torch.float64
torch.manual_seed(0)
f = MLPFunc(512, dt).to(dev)
y0 = torch.randn(a.batch, 512, device=dev, dtype=dt)
options.clear()
options.set_option("ts_trajectory_max_cps_ram", "50")
ode = petsc_adjoint.ODEPetsc()
ode.setupTS(y0, f, step_size=0.01, method="dopri5")
options.clear()
t = torch.tensor([1.0])


def solve():
    for p in f.parameters():
        p.grad = None
    y = y0.detach().requires_grad_(True)
    ode.odeint_adjoint(y, t).abs().mean().backward()


solve()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.solves):
    solve()
torch.cuda.synchronize()
el = (time.perf_counter() - t0) / a.solves
n = y0.numel() * y0.element_size()
print("C3b %s: %d accepted steps, %d rejections, %.1f ms per solve, %.1f time-steps/s; combine kernel algorithmic bytes per launch %d"
      % (a.dtype, ode.num_steps, ode.num_rejections, 1e3 * el, ode.num_steps / el, 7 * n))
