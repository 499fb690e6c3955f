#!/usr/bin/env python3
"""Host-side cProfile of the eager forward sweep and reverse sweep at C3a (4096x512 fp32, rk4); the reverse sweep
is called directly (outside loss.backward()) so that cProfile sees inside it."""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from pnode_amd import options, petsc_adjoint
from problems import MLPFunc
dev = torch.device("cuda:0"); NT = 20
options.set_option("ts_adapt_type", "none"); options.set_option("ts_trajectory_solution_only", 0)
f = MLPFunc(512, torch.float32).to(dev)
y0 = torch.randn(4096, 512, device=dev); t = torch.tensor([0.01 * NT])
ode = petsc_adjoint.ODEPetsc(); ode.setupTS(y0, f, step_size=0.01, method="rk4")
g = torch.randn(1, 4096 * 512, device=dev)
def fwd():
    with torch.no_grad(): return ode._odeint(y0, t, True)
def rev():
    with torch.no_grad(): ode._reverse_sweep(g, 1)
for _ in range(3): fwd(); rev()
torch.cuda.synchronize()
for name, fn in (("forward", fwd), ("reverse", rev)):
    if name == "reverse": fwd()
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("%s sweep: host enqueue %.2f ms, until GPU done %.2f ms  (%d time steps)" % (name, 1e3 * (t1 - t0), 1e3 * (t2 - t0), NT))
    if name == "reverse": fwd()
    torch.cuda.synchronize()
    pr = cProfile.Profile(); pr.enable(); fn(); pr.disable(); torch.cuda.synchronize()
    st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(14)
