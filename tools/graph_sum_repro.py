#!/usr/bin/env python3
"""Stand-alone check (no pnode_amd): does a captured column-sum / Linear backward replay correctly after
torch.cuda.synchronize()?"""
import gc, sys, torch
dev = torch.device("cuda:0")
torch.manual_seed(0)
x = torch.randn(4096, 512, device=dev)
lin = torch.nn.Linear(512, 512).to(dev)
mode = sys.argv[1] if len(sys.argv) > 1 else "sum"

def body(inp):
    if mode == "sum":
        return (inp.sum(0),)
    if mode == "linear":
        with torch.enable_grad():
            y = inp.detach().requires_grad_(True)
            alias = [p.detach().requires_grad_(True) for p in lin.parameters()]
            out = torch.tanh(torch.func.functional_call(lin, dict(zip([n for n, _ in lin.named_parameters()], alias)), (y,)))
            return torch.autograd.grad(out, [y] + alias, inp)
ref = [r.clone() for r in body(x)]
for _ in range(2): body(x)
static = x.clone()
gc.collect(); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, pool=torch.cuda.graph_pool_handle(), capture_error_mode="thread_local"):
    outs = body(static)
for i in range(4):
    static.copy_(x)
    g.replay()
    got = [o.clone() for o in outs]
    if len(sys.argv) > 2: torch.cuda.synchronize()
    print(mode, "replay", i, ["%.1e" % ((a - b).norm() / b.norm()).item() for a, b in zip(got, ref)], flush=True)
