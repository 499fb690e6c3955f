"""Microbenchmark (round 5): formulations of the weight-sensitivity GEMM  mu_W += alpha * G^T X  (G, X: 4096 x 512 fp32;
mu_W: 512 x 512) as the engine-side Linear accumulation issues it, GPU time per call by events."""
import torch
dev = torch.device("cuda:0")
for dtype in (torch.float32, torch.float64):
    G = [torch.randn(4096, 512, device=dev, dtype=dtype) for _ in range(8)]
    X = [torch.randn(4096, 512, device=dev, dtype=dtype) for _ in range(8)]
    mw = torch.zeros(512, 512, device=dev, dtype=dtype)
    def t(fn, reps=200):
        for k in range(10): fn(k % 8)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for k in range(reps): fn(k % 8)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3
    print(dtype)
    print("  addmm(mw, G^T, X, out=mw)          %6.2f us" % t(lambda k: torch.addmm(mw, G[k].t(), X[k], beta=1.0, alpha=0.5, out=mw)))
    print("  addmm(mw^T, X^T, G, out=mw^T)      %6.2f us" % t(lambda k: torch.addmm(mw.t(), X[k].t(), G[k], beta=1.0, alpha=0.5, out=mw.t())))
    print("  mm(G^T, X) (fresh output)          %6.2f us" % t(lambda k: torch.mm(G[k].t(), X[k])))
    print("  mm(G^T, X) + mw.add_               %6.2f us" % t(lambda k: mw.add_(torch.mm(G[k].t(), X[k]), alpha=0.5)))
    print("  forward-shaped mm(X, W^T)          %6.2f us" % t(lambda k: torch.mm(X[k], mw.t())))
    print("  dX-shaped mm(G, W)                 %6.2f us" % t(lambda k: torch.mm(G[k], mw)))
    # split-K by hand: the K = 4096 rows in `c` chunks, one batched GEMM, then the chunk sum folded into the accumulation
    for c in (2, 4, 8, 16):
        def split(k, c=c):
            part = torch.bmm(G[k].view(c, 4096 // c, 512).transpose(1, 2), X[k].view(c, 4096 // c, 512))
            mw.add_(part.sum(0), alpha=0.5)
        print("  bmm split-K x%-2d + sum + add_        %6.2f us" % (c, t(split)))
