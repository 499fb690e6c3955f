#!/usr/bin/env python3
"""cProfile of the reverse sweep of the adaptive workload (bench.py --config c3b --stiff, max_cps 50): where the host time of a
reversed step goes -- func / autograd.grad (PyTorch's) against the engine's own bookkeeping."""
import cProfile, os, pstats, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from pnode_amd import options, petsc_adjoint
from problems import SwitchedMLPFunc
dev = torch.device("cuda:0")
torch.manual_seed(0)
y0 = torch.randn(4096, 512, device=dev)
f = SwitchedMLPFunc(512, torch.float32).to(dev)
t = torch.tensor([SwitchedMLPFunc.T_END])
options.clear(); options.set_option("ts_trajectory_max_cps_ram", 50)
ode = petsc_adjoint.ODEPetsc(); ode.setupTS(y0, f, step_size=0.01, method="dopri5"); options.clear()
for it in range(3):
    for p in f.parameters(): p.grad = None
    y = y0.detach().requires_grad_(True)
    out = ode.odeint_adjoint(y, t)
    loss = out.abs().mean()
    torch.cuda.synchronize()
    if it < 2:
        loss.backward()
    else:
        # the autograd engine runs OdeintAdjointMethod.backward on its own thread, where cProfile does not follow: call the
        # body of that backward here, on this thread
        g = torch.autograd.grad(loss, out)[0].contiguous().view(1, -1)
        pr = cProfile.Profile(); pr.enable()
        with torch.no_grad():
            ode._reverse_sweep(g, 1)
        pr.disable()
    torch.cuda.synchronize()
print("accepted steps", ode._nsteps)
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
