#!/usr/bin/env python3
"""Stand-alone escalation (no pnode_amd kernels): N Linear-MLP VJPs captured in one hipGraph, optionally
captured from inside an autograd backward (engine worker thread); replays separated by stream syncs."""
import gc, os, sys, torch
dev = torch.device("cuda:0")
torch.manual_seed(0)
NV = int(os.environ.get("NV", 8)); INBWD = os.environ.get("INBWD", "0") == "1"; SYNC = os.environ.get("SYNC", "stream")
x = torch.randn(4096, 512, device=dev)
net = torch.nn.Sequential(*[m for _ in range(4) for m in (torch.nn.Linear(512, 512), torch.nn.Tanh())][:-1]).to(dev)
names = [n for n, _ in net.named_parameters()]
acc = torch.zeros(sum(p.numel() for p in net.parameters()), device=dev)

def body(inp):
    acc.zero_()
    lam = inp
    for _ in range(NV):
        with torch.enable_grad():
            y = lam.detach().requires_grad_(True)
            alias = [p.detach().requires_grad_(True) for p in net.parameters()]
            out = torch.func.functional_call(net, dict(zip(names, alias)), (y,))
            gr = torch.autograd.grad(out, [y] + alias, lam)
        lam = lam + 0.01 * gr[0]
        o = 0
        for g in gr[1:]:
            acc[o:o + g.numel()] += g.reshape(-1); o += g.numel()
    return lam, acc

def run_eager():
    with torch.no_grad():
        l, a = body(x)
    return l.clone(), a.clone()
ref = run_eager(); run_eager()
state = {}
def capture():
    gc.collect(); torch.cuda.synchronize()
    state["static"] = x.clone()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, pool=torch.cuda.graph_pool_handle(), capture_error_mode="thread_local"):
        with torch.no_grad():
            state["outs"] = body(state["static"])
    state["g"] = g

class InBwd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z): return z.clone()
    @staticmethod
    def backward(ctx, gz):
        if "g" not in state: capture()
        state["static"].copy_(x); state["g"].replay()
        return gz
for i in range(5):
    if INBWD:
        z = torch.ones(1, device=dev, requires_grad=True); InBwd.apply(z).sum().backward()
    else:
        if "g" not in state: capture()
        state["static"].copy_(x); state["g"].replay()
    got = [o.clone() for o in state["outs"]]
    if SYNC == "stream": torch.cuda.current_stream().synchronize()
    elif SYNC == "device": torch.cuda.synchronize()
    if os.environ.get("PRINT") == "equal":
        print("replay", i, [bool(torch.equal(a, b)) for a, b in zip(got, ref)], flush=True)
        continue
    sizes = [p.numel() for p in net.parameters()]
    per = [((a - b).norm() / (b.norm() + 1e-30)).item() for a, b in zip(got[1].split(sizes), ref[1].split(sizes))]
    print("replay", i, "lam %.1e" % ((got[0] - ref[0]).norm() / ref[0].norm()).item(), "params", " ".join("%.0e" % v for v in per), flush=True)
