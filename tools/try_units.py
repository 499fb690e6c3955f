"""Development check of the per-evaluation hipGraphs (pnode_amd/_stagegraphs.py): an adaptive solve both ways."""
import os
import sys
import time
import warnings

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import pnode_amd
from pnode_amd import petsc_adjoint, options
from problems import MLPFunc, SwitchedMLPFunc

warnings.simplefilter("always")
dev = torch.device("cuda:0")
B = int(os.environ.get("B", 4096))
D = int(os.environ.get("D", 512))
T_END = float(os.environ.get("T_END", 1.0))
MAXCPS = os.environ.get("MAXCPS", "50")
dt = torch.float32 if os.environ.get("DT", "f32") == "f32" else torch.float64


def make(opts):
    torch.manual_seed(0)
    f = SwitchedMLPFunc(D, dt).to(dev)
    options.clear()
    for k, v in opts.items():
        options.set_option(k, v)
    if MAXCPS != "0":
        options.set_option("ts_trajectory_type", "memory")
        options.set_option("ts_trajectory_max_cps_ram", MAXCPS)
    o = petsc_adjoint.ODEPetsc()
    y0 = torch.randn(B, D, dtype=dt, device=dev) * 0.5
    o.setupTS(y0, f, step_size=0.01, method="dopri5", enable_adjoint=True)
    options.clear()
    return o, f, y0


def solve(o, f, y0):
    for p in f.parameters():
        p.grad = None
    y = y0.detach().requires_grad_(True)
    out = o.odeint_adjoint(y, torch.tensor([T_END]))
    out.abs().mean().backward()
    return out.detach().clone(), y.grad.clone(), torch.cat([p.grad.reshape(-1) for p in f.parameters()])


oe, fe, y0 = make({"pn_graph_capture": "0"})
ref = solve(oe, fe, y0)
print("eager: steps", oe.num_steps, "rejections", oe.num_rejections, oe.graph_status)
og, fg_, _ = make({})
for k in range(7):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    got = solve(og, fg_, y0)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    d = [float((a.double() - b.double()).abs().max() / b.double().abs().max()) for a, b in zip(got, ref)]
    print("call %d: %.1f ms  steps %d  status %s  rel diff out/dy0/dtheta %.1e %.1e %.1e  nfe %d/%d" % (
        k, 1e3 * el, og.num_steps, og.graph_status, d[0], d[1], d[2], og.nfe_forward, og.nfe_backward))
for k in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    solve(oe, fe, y0)
    torch.cuda.synchronize()
    print("eager call: %.1f ms  nfe %d/%d" % (1e3 * (time.perf_counter() - t0), oe.nfe_forward, oe.nfe_backward))
e = next(iter(og._graphs.values()), None)
if e is not None and e.sg is not None:
    print("units:", sorted((k[0], k[1], u.uses) for k, u in e.sg.units.items()), "captured", e.sg.captured, "replayed", e.sg.replayed)
