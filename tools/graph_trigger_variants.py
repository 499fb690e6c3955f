import gc, os, sys, torch, torch.nn.functional as F
dev = torch.device("cuda:0")
variant = sys.argv[1]
torch.manual_seed(0)
x = torch.randn(4096, 512, device=dev)
if variant in ("module", "module_nofc"):
    net = torch.nn.Sequential(*[m for _ in range(4) for m in (torch.nn.Linear(512, 512), torch.nn.Tanh())][:-1]).to(dev)
    params = list(net.parameters()); names = [n for n, _ in net.named_parameters()]
else:
    params = []
    for _ in range(4):
        params += [torch.randn(512, 512, device=dev) * 0.05, torch.randn(512, device=dev) * 0.05]
    if variant == "req":
        params = [p.requires_grad_(True) for p in params]
sizes = [p.numel() for p in params]
acc = torch.zeros(sum(sizes), device=dev)
def fwd(y, alias):
    if variant == "module":
        return torch.func.functional_call(net, dict(zip(names, alias)), (y,))
    h = y
    for k in range(4):
        h = F.linear(h, alias[2 * k], alias[2 * k + 1])
        if k < 3: h = torch.tanh(h)
    return h
def body(inp):
    acc.zero_(); lam = inp
    for _ in range(2):
        with torch.enable_grad():
            y = lam.detach().requires_grad_(True)
            alias = [p.detach().requires_grad_(True) for p in params]
            gr = torch.autograd.grad(fwd(y, alias), [y] + alias, lam)
        lam = lam + 0.01 * gr[0]
        o = 0
        for g, n in zip(gr[1:], sizes):
            acc[o:o + n] += g.reshape(-1); o += n
    return lam, acc
with torch.no_grad():
    ref = [r.clone() for r in body(x)]; body(x)
static = x.clone(); gc.collect(); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, pool=torch.cuda.graph_pool_handle(), capture_error_mode="thread_local"):
    with torch.no_grad(): outs = body(static)
oks = []
for i in range(4):
    static.copy_(x); g.replay(); got = [o.clone() for o in outs]; torch.cuda.current_stream().synchronize()
    if os.environ.get("EAGER") == "norm":
        _ = [(a - b).norm().item() for a, b in zip(got, ref)]
    elif os.environ.get("EAGER") == "sum":
        _ = x.sum(0); _ = (x * 2).sum(0)
    elif os.environ.get("EAGER") == "body":
        with torch.no_grad(): body(x)
    oks.append(all(torch.equal(a, b) for a, b in zip(got, ref)))
print(variant, oks)
