#!/usr/bin/env python3
"""Round-2 in-situ A/B at C3a (4096 x 512 fp32, rk4, 20 steps), interleaved in one process:
  * -pn_param_accum batch | step | stage  x  tapes retained | recomputed
  * PN_TUNE launch geometries of the streaming kernel (block=512/1024 are real since round 2)
Per arm: solver-kernel microseconds per time step by HIP start/stop events (vector kernels, parameter
accumulation), and the two roofline fractions bench.py reports.  usage: ab_r02.py [tune1 tune2 ...]"""
import ctypes, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from pnode_amd import _lib, options, petsc_adjoint
from problems import MLPFunc
lib = _lib.load(); dev = torch.device("cuda:0")
tunes = sys.argv[1:] or [""]
torch.manual_seed(0)
f = MLPFunc(512, torch.float32).to(dev)
y0 = torch.randn(4096, 512, device=dev)
t = torch.tensor([0.2])
n, w, npar = y0.numel(), 4, sum(p.numel() for p in f.parameters())
K = len(_lib.KERNEL_IDS)

def make(mode, retain):
    options.clear()
    for k, v in {"ts_adapt_type": "none", "ts_trajectory_solution_only": "0", "pn_param_accum": mode,
                 "pn_trajectory_retain_graph": retain}.items():
        options.set_option(k, v)
    ode = petsc_adjoint.ODEPetsc(); ode.setupTS(y0, f, step_size=0.01, method="rk4"); options.clear()
    return ode

def solve(ode):
    for p in f.parameters(): p.grad = None
    y = y0.detach().requires_grad_(True); ode.odeint_adjoint(y, t).abs().mean().backward()

arms = [(m, r) for r in ("auto", "0") for m in ("batch", "step", "stage")]
odes = {a: make(*a) for a in arms}
for o in odes.values():
    solve(o); solve(o)
res = {(a, c): [] for a in arms for c in tunes}
for rep in range(6):
    for c in tunes:
        lib.pn_tune_set(c.encode() if c else None)
        for a in arms:
            torch.cuda.synchronize(); lib.pn_prof_enable(1); solve(odes[a]); torch.cuda.synchronize()
            L = (ctypes.c_int64 * K)(); us = (ctypes.c_double * K)(); by = (ctypes.c_double * K)()
            lib.pn_prof_collect(len(L), L, us, by); lib.pn_prof_enable(0)
            ns = odes[a].num_steps
            res[(a, c)].append(((us[0] + us[2] + us[3]) / ns, us[4] / ns, us[0] / max(L[0], 1), us[2] / max(L[2], 1), us[3] / max(L[3], 1),
                                us[4] / max(L[4], 1), L[4] / ns))
lib.pn_tune_set(None)
print("arm (param mode, tapes)   tune              vec us/step  par us/step | stage   theta   accum   param(avg us, launches/step) | frac vec  frac incl")
for c in tunes:
    for a in arms:
        r = res[(a, c)]
        med = [statistics.median(x[i] for x in r) for i in range(7)]
        fv = 32.0 * n * w / med[0] / 1e3 / 8000
        fi = (32.0 * n * w + 12.0 * npar * w) / (med[0] + med[1]) / 1e3 / 8000
        print("%-7s tapes=%-5s       %-16s %8.2f     %8.2f   | %6.2f  %6.2f  %6.2f  %6.2f x %.3f | %.3f     %.3f"
              % (a[0], a[1], c or "(default)", med[0], med[1], med[2], med[3], med[4], med[5], med[6], fv, fi), flush=True)
