"""One-time self-test of hipGraph replay in this process.

ROCm 7.2's graph launches with AQL packet capture replay PyTorch's two-pass reductions wrongly after a stream
synchronisation (profiles/r01_graph_packet_capture.txt).  pnode_amd switches packet capture off at import,
but that only works when the HIP runtime is not initialised yet -- which cannot be queried (a profiler's tool
library, for one, initialises it before Python starts).  So before the first capture the failing pattern
itself is run once: VJPs of a 4-layer 512-wide MLP at batch 4096 captured in a graph, replayed across a
stream synchronisation, compared bit for bit with the eager result (~0.1 s, ~60 MB, freed afterwards)."""
import gc

import torch
import torch.nn.functional as F

_verdict = {}


def replay_is_sound(device):
    key = torch.device(device).index or 0
    if key not in _verdict:
        _verdict[key] = _run(torch.device("cuda", key))
    return _verdict[key]


def _run(dev):
    # the pattern of tools/graph_sum_repro2.py, which fails from the second replay on when the defect is active
    cpu_rng = torch.random.get_rng_state()
    gpu_rng = torch.cuda.get_rng_state(dev)
    try:
        torch.manual_seed(0)
        x = torch.randn(4096, 512, device=dev)
        net = torch.nn.Sequential(*[m for _ in range(4) for m in (torch.nn.Linear(512, 512), torch.nn.Tanh())][:-1]).to(dev)
    finally:
        torch.random.set_rng_state(cpu_rng)
        torch.cuda.set_rng_state(gpu_rng, dev)
    names = [n for n, _ in net.named_parameters()]
    acc = torch.zeros(sum(p.numel() for p in net.parameters()), device=dev)

    def body(inp):
        acc.zero_()
        lam = inp
        for _ in range(2):
            with torch.enable_grad():
                y = lam.detach().requires_grad_(True)
                alias = [p.detach().requires_grad_(True) for p in net.parameters()]
                out = torch.func.functional_call(net, dict(zip(names, alias)), (y,))
                gr = torch.autograd.grad(out, [y] + alias, lam)
            lam = lam + 0.01 * gr[0]
            o = 0
            for g in gr[1:]:
                acc[o:o + g.numel()] += g.reshape(-1)
                o += g.numel()
        return lam, acc

    def run_eager():
        with torch.no_grad():
            l, a = body(x)
        return l.clone(), a.clone()

    ref = run_eager()
    run_eager()
    state = {}

    def capture():
        gc.collect()
        torch.cuda.synchronize(dev)
        state["static"] = x.clone()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, pool=torch.cuda.graph_pool_handle(), capture_error_mode="thread_local"):
            with torch.no_grad():
                state["outs"] = body(state["static"])
        state["g"] = g

    ok = True
    try:
        for _ in range(4):
            if "g" not in state:
                capture()
            state["static"].copy_(x)
            state["g"].replay()
            got = [o.clone() for o in state["outs"]]
            torch.cuda.current_stream(dev).synchronize()
            ok = ok and all(bool(torch.equal(a, b)) for a, b in zip(got, ref))
    except Exception:
        ok = False
    state.clear()
    gc.collect()
    return ok
