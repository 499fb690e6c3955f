#!/usr/bin/env python3
"""Minimal: hipMemsetAsync node + dependent kernel in one hipGraph; replays separated by stream syncs."""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pnode_amd import _lib
lib = _lib.load()
dev = torch.device("cuda:0")
N = int(os.environ.get("N", 512)); REPS = int(os.environ.get("REPS", 1)); SYNC = os.environ.get("SYNC", "stream")
bufs = [torch.full((N,), 7.0, device=dev) for _ in range(REPS)]
big = torch.randn(4096, 512, device=dev)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, capture_error_mode="thread_local"):
    for b in bufs:
        tmp = big @ big.t()[:512, :].t() if False else big * 2.0        # some preceding kernel
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        _lib.check(lib.pn_zero(st, 0, N, b.data_ptr()))                 # hipMemsetAsync -> memset node
        b.add_(1.0)                                                    # dependent kernel
for i in range(5):
    g.replay()
    vals = [float(b.sum()) / N for b in bufs]
    if SYNC == "stream": torch.cuda.current_stream().synchronize()
    elif SYNC == "device": torch.cuda.synchronize()
    print("replay", i, "expected 1.0 got", sorted(set(round(v, 3) for v in vals)), flush=True)
