#!/usr/bin/env python3
"""Micro-benchmark of the library's fused dW + db kernel (pn_linear_wgrad) beside torch.addmm on the same operands.

What it separates: the kernel on operands that rotate through 8 pairs (128 MiB: the Infinity Cache holds them) or through 40
(640 MiB: HBM), with and without the bias columns, and -- the situation inside a reverse sweep -- with another GEMM of the
same size between two launches (the dX product of the layer, which evicts what the launch before left in the L2s).

    python tools/mb_wgrad_lib.py [--rows 4096] [--width 512] [--reps 400]
"""
import argparse
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pnode_amd import _lib  # noqa: E402


def timed(fn, reps):
    for _ in range(20):
        fn(0)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(reps):
        fn(i)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=4096)
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--reps", type=int, default=400)
    ap.add_argument("--dist", choices=("normal", "uniform", "zeros"), default="normal",
                    help="operand values: the MFMA rate of this chip depends on them (power management)")
    ap.add_argument("--layers", type=int, default=4, help="partial buffers the launches rotate through (one per layer in a sweep)")
    a = ap.parse_args()
    lib = _lib.load()
    dev = torch.device("cuda:0")
    K, W = a.rows, a.width
    st = torch.cuda.current_stream().cuda_stream
    nb = ctypes.c_int64()
    nw = lib.pn_linear_wgrad_work_bytes(W, W, ctypes.byref(nb))
    flop = 2.0 * K * W * W
    for npairs in (8, 40):
        def fill():
            if a.dist == "normal":
                return torch.randn(K, W, device=dev)
            return torch.rand(K, W, device=dev) - 0.5 if a.dist == "uniform" else torch.zeros(K, W, device=dev)
        G = [fill() for _ in range(npairs)]
        X = [fill() for _ in range(npairs)]
        Wt = torch.randn(W, W, device=dev)
        pws = [torch.zeros(nw // 4, device=dev) for _ in range(4)]
        pbs = [torch.zeros(nb.value // 8, device=dev, dtype=torch.float64) for _ in range(4)]
        mus = [torch.zeros(W, W, device=dev) for _ in range(4)]
        dx = torch.empty(K, W, device=dev)

        def fused(i, bias=True):
            j = i % npairs
            rc = lib.pn_linear_wgrad(st, 0, K, W, W, G[j].data_ptr(), X[j].data_ptr(), 1.0, pws[i % a.layers].data_ptr(),
                                     pbs[i % a.layers].data_ptr() if bias else None)
            assert rc == 0

        def blas(i):
            j = i % npairs
            torch.addmm(mus[i % 4], G[j].t(), X[j], out=mus[i % 4])

        def dxgemm(i):
            torch.mm(G[i % npairs], Wt, out=dx)

        t_dx = timed(dxgemm, a.reps)
        rows = [("pn_linear_wgrad  (dW + db)", timed(fused, a.reps)),
                ("pn_linear_wgrad  (dW only)", timed(lambda i: fused(i, False), a.reps)),
                ("torch.addmm      (dW only)", timed(blas, a.reps)),
                ("dX GEMM + pn_linear_wgrad ", timed(lambda i: (dxgemm(i), fused(i)), a.reps) - t_dx),
                ("dX GEMM + torch.addmm     ", timed(lambda i: (dxgemm(i), blas(i)), a.reps) - t_dx)]
        print("rows %d, %d x %d, %s operands, %d pairs (%d MiB), dX GEMM alone %.2f us" % (K, W, W, a.dist, npairs, npairs * 2 * K * W * 4 >> 20, t_dx))
        for name, us in rows:
            print("   %-28s %7.2f us   %6.1f TFLOP/s" % (name, us, flop / us * 1e-6))


if __name__ == "__main__":
    main()
