#!/usr/bin/env python3
"""Repro harness: several graph-capturing ODEPetsc objects alive in one process."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from pnode_amd import _lib, options, petsc_adjoint
from problems import MLPFunc
dev = torch.device("cuda:0")
torch.manual_seed(0)
B, D, NT = int(os.environ.get("B", 4096)), int(os.environ.get("D", 512)), int(os.environ.get("NT", 100))
f = MLPFunc(D, torch.float32).to(dev)
y0 = torch.randn(B, D, device=dev)
t = torch.tensor([NT * 0.01])
def make(graph, extra=None):
    options.clear()
    options.set_option("ts_adapt_type", "none"); options.set_option("ts_trajectory_solution_only", "0")
    for k, v in (extra or {}).items(): options.set_option(k, v)
    if graph: options.set_option("pn_graph_capture", "1")
    o = petsc_adjoint.ODEPetsc(); o.setupTS(y0, f, step_size=0.01, method="rk4"); options.clear(); return o
def solve(o):
    for p in f.parameters(): p.grad = None
    y = y0.detach().requires_grad_(True); out = o.odeint_adjoint(y, t); out.abs().mean().backward()
    solve.last = (out.detach().clone(), y.grad.clone())
    return torch.cat([p.grad.reshape(-1) for p in f.parameters()]).clone()
def rel(a, b): return ((a - b).norm() / b.norm()).item()
scenario = sys.argv[1]
if scenario == "eager_first_then_graph":
    E = make(False); A = make(True)
    ref = [solve(E) for _ in range(3)][-1]
    for i in range(4): print("A solve", i, "rel", rel(solve(A), ref), "captured", A.graphs_captured)
elif scenario == "graph_alone_then_eager":
    A = make(True)
    ga = [solve(A) for _ in range(4)]
    E = make(False); ref = solve(E)
    for i, g in enumerate(ga): print("A solve", i, "rel", rel(g, ref))
elif scenario == "eager_alive_constructed_later":
    E = make(False); ref = [solve(E) for _ in range(3)][-1]
    A = make(True)
    for i in range(4): print("A solve", i, "rel", rel(solve(A), ref), "captured", A.graphs_captured)
elif scenario == "two_graphs":
    E = make(False); ref = solve(E); del E
    A = make(True); Bo = make(True)
    for i in range(3): print("A solve", i, "rel", rel(solve(A), ref))
    for i in range(3): print("B solve", i, "rel", rel(solve(Bo), ref))
    for i in range(2):
        print("A again", rel(solve(A), ref)); print("B again", rel(solve(Bo), ref))
elif scenario == "graph_alone_detail":
    A = make(True, {"pn_param_accum": os.environ.get("PA", "step")})
    res = []
    for i in range(int(os.environ.get("NS", 6))):
        g = solve(A); res.append((g,) + solve.last)
    E = make(False); ref = (solve(E),) + solve.last
    for i, r in enumerate(res):
        print("A solve", i, "gp rel %.3e  out rel %.3e  gy rel %.3e" % (rel(r[0], ref[0]), rel(r[1], ref[1]), rel(r[2], ref[2])))
elif scenario == "detail2":
    # reference first in a subprocess-free way: eager ode E, deleted before A exists
    mode = os.environ.get("MODE", "plain")
    A = make(True)
    res = []
    for i in range(6):
        g = solve(A)
        if mode == "sync":
            torch.cuda.synchronize()
        res.append((g,) + solve.last)
    if mode == "sync_end":
        torch.cuda.synchronize()
    if mode == "hold":
        held = [p.grad for p in f.parameters()]
    if mode == "check_before_E":
        torch.cuda.synchronize()
        print("before E: last vs prev gp rel %.3e" % rel(res[-1][0], res[-2][0]))
    E = make(False); ref = (solve(E),) + solve.last
    for i, r in enumerate(res):
        print("A solve", i, "gp rel %.3e  out rel %.3e  gy rel %.3e" % (rel(r[0], ref[0]), rel(r[1], ref[1]), rel(r[2], ref[2])))
elif scenario == "detail3":
    import time
    mode = os.environ.get("MODE", "sync")
    if os.environ.get("BLAS"):
        torch.backends.cuda.preferred_blas_library(os.environ["BLAS"])
    E = make(False); refg = solve(E); refp = [p.grad.clone() for p in f.parameters()]; del E
    torch.cuda.synchronize()
    if os.environ.get("CAPMODE"):
        petsc_adjoint.ODEPetsc.GRAPH_CAPTURE_MODE = os.environ["CAPMODE"]
    if os.environ.get("OWNPOOL"):
        orig_gb = petsc_adjoint.ODEPetsc._graph_backward
        def gb(self, e, g, T):
            if e.g_b is None: e.pool = torch.cuda.graph_pool_handle()
            return orig_gb(self, e, g, T)
        petsc_adjoint.ODEPetsc._graph_backward = gb
    A = make(True)
    for i in range(6):
        g = solve(A)
        if mode == "sync": torch.cuda.synchronize()
        elif mode == "sleep": time.sleep(1.0)
        elif mode == "stream_sync": torch.cuda.current_stream().synchronize()
        elif mode == "event_sync":
            ev = torch.cuda.Event(); ev.record(); ev.synchronize()
        elif mode == "item": float(g[0])
        per = [rel(p.grad, r) for p, r in zip(f.parameters(), refp)]
        print("A solve", i, "captured", A.graphs_captured, "per-param rel:", " ".join("%.1e" % x for x in per), flush=True)
elif scenario == "snapshot":
    A = make(True)
    log = []
    orig = petsc_adjoint.ODEPetsc._vjp
    def spy(self, t_, y_flat, w_flat, tape=None, which="EX"):
        gy, gp = orig(self, t_, y_flat, w_flat, tape, which)
        if torch.cuda.is_current_stream_capturing():
            log.append(("y", y_flat.data_ptr())); log.append(("w", w_flat.data_ptr())); log.append(("gy", gy.data_ptr()))
            for k, g in enumerate(gp): log.append(("gp%d" % k, g.data_ptr()))
        return gy, gp
    petsc_adjoint.ODEPetsc._vjp = spy
    for i in range(3): solve(A)
    torch.cuda.synchronize()
    snap = torch.cuda.memory_snapshot()
    segs = sorted((s["address"], s["address"] + s["total_size"], s.get("segment_pool_id"), s["segment_type"]) for s in snap)
    def find(p):
        for a, b, pid, ty in segs:
            if a <= p < b: return pid, ty
        return None
    seen = {}
    for name, p in log[:24]:
        print(name, hex(p), find(p))
    print("adj_p", find(A.adj_p_tensor.data_ptr()), "adj_u", find(A.adj_u_tensor.data_ptr()), "w_a", find(A._work["w_a"].data_ptr()))
    print("pools:", sorted(set((pid, ty) for _, _, pid, ty in segs)))
