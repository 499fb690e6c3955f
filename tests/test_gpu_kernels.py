"""-m gpu: every device entry point of the C ABI against a numpy restatement of the same
Vec-op sequence, on ragged / misaligned / empty inputs and at the target size."""
import ctypes

import numpy as np
import pytest
import torch

from conftest import require_gpu
from oracle import ts_oracle

pytestmark = pytest.mark.gpu

SIZES = [1, 3, 4, 5, 63, 64, 255, 256, 257, 1000, 4096 * 2, 65537, 4096 * 512]


def _ops(dtype, n):
    from pnode_amd.petsc_adjoint import HipVecOps
    return HipVecOps(require_gpu(), dtype, n)


def _tol(dtype):
    return 2e-6 if dtype == torch.float32 else 1e-14


def _rand(n, dtype, k, dev, offset=0):
    """k random vectors; offset>0 gives storage-offset (16-byte misaligned) views."""
    g = torch.Generator().manual_seed(n * 7 + k)
    out = []
    for _ in range(k):
        base = torch.randn(n + offset, generator=g, dtype=dtype).to(dev)
        out.append(base[offset:])
    return out


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("n", SIZES)
@pytest.mark.parametrize("nk", [0, 1, 3, 6])
def test_rk_stage(dtype, n, nk):
    ops = _ops(dtype, n)
    u, *K = _rand(n, dtype, nk + 1, ops.device)
    coef = [0.3 * (j + 1) * (-1) ** j for j in range(nk)]
    y = torch.full((n,), float("nan"), dtype=dtype, device=ops.device)
    ops.rk_stage(y, u, K, coef)
    ref = u.double().cpu()
    for c, k in zip(coef, K):
        ref = ref + c * k.double().cpu()
    assert torch.allclose(y.double().cpu(), ref, rtol=_tol(dtype), atol=_tol(dtype))


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("n", [5, 257, 4099])
def test_misaligned_views_take_the_scalar_path(dtype, n):
    ops = _ops(dtype, n)
    u, k1, k2 = _rand(n, dtype, 3, ops.device, offset=1)
    assert u.data_ptr() % 16 != 0
    y = torch.zeros(n + 1, dtype=dtype, device=ops.device)[1:]
    ops.rk_stage(y, u, [k1, k2], [0.5, -0.25])
    ref = u.double() + 0.5 * k1.double() - 0.25 * k2.double()
    assert torch.allclose(y.double(), ref, rtol=_tol(dtype), atol=_tol(dtype))


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("n", SIZES)
@pytest.mark.parametrize("fsal", [False, True])
def test_combine_wrms(dtype, n, fsal):
    ops = _ops(dtype, n)
    nk = 6
    u, *K = _rand(n, dtype, nk + 1, ops.device)
    cb = [0.01 * (j + 1) for j in range(nk)]
    ce = [1e-4 * (-1) ** j * (j + 1) for j in range(nk)]
    atol = rtol = 1e-4
    unew = None if fsal else torch.full((n,), float("nan"), dtype=dtype, device=ops.device)
    ops.combine_wrms(unew, u, K, cb, ce, atol, rtol)
    got = ops.read_enorm()
    # restatement in the storage precision, the way the reference's Vec ops would run
    npd = np.float32 if dtype == torch.float32 else np.float64
    un = u.cpu().numpy().astype(npd)
    if not fsal:
        for c, k in zip(cb, K):
            un = (un + npd(c) * k.cpu().numpy()).astype(npd)
        assert np.allclose(unew.cpu().numpy(), un, rtol=_tol(dtype) * 4, atol=_tol(dtype) * 4)
        un = unew.cpu().numpy()
    err = np.zeros(n, dtype=npd)
    for c, k in zip(ce, K):
        err = (err + npd(c) * k.cpu().numpy()).astype(npd)
    uh = (un + err).astype(npd)
    want = ts_oracle.wrms(un, uh, atol, rtol)
    assert got == pytest.approx(want, rel=5e-3 if dtype == torch.float32 else 1e-9)


def test_wrms_flags_nan_and_inf():
    ops = _ops(torch.float32, 1000)
    u, k = _rand(1000, torch.float32, 2, ops.device)
    k[517] = float("nan")
    ops.combine_wrms(None, u, [k], [0.0], [1e-3], 1e-4, 1e-4)
    assert np.isnan(ops.read_enorm())
    k[517] = float("inf")
    ops.combine_wrms(None, u, [k], [0.0], [1e-3], 1e-4, 1e-4)
    v = ops.read_enorm()
    assert np.isnan(v) or np.isinf(v)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("n", [1, 257, 65537, 4096 * 512])
@pytest.mark.parametrize("with_lam", [True, False])
def test_adj_theta(dtype, n, with_lam):
    ops = _ops(dtype, n)
    lam, *d = _rand(n, dtype, 4, ops.device)
    coef = [0.2, -0.4, 0.7]
    w = torch.full((n,), float("nan"), dtype=dtype, device=ops.device)
    ops.adj_theta(w, lam if with_lam else None, 0.125, d, coef)
    ref = (0.125 * lam.double() if with_lam else 0) + sum(c * x.double() for c, x in zip(coef, d))
    assert torch.allclose(w.double(), ref, rtol=_tol(dtype), atol=_tol(dtype))


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("n", [1, 257, 65537, 4096 * 512])
@pytest.mark.parametrize("forcing", [True, False])
@pytest.mark.parametrize("fuse_next", [True, False])
def test_adj_accum_in_place(dtype, n, forcing, fuse_next):
    ops = _ops(dtype, n)
    lam, g, *d = _rand(n, dtype, 6, ops.device)
    coefs = [1.0, 0.0025, 1.0, -0.5]
    ref = lam.double() + sum(c * x.double() for c, x in zip(coefs, d)) + (g.double() if forcing else 0)
    wn = torch.full((n,), float("nan"), dtype=dtype, device=ops.device) if fuse_next else None
    ops.adj_accum(lam, lam, d, coefs, g if forcing else None, wn, 0.0025)
    assert torch.allclose(lam.double(), ref, rtol=_tol(dtype), atol=_tol(dtype))
    if fuse_next:
        assert torch.allclose(wn.double(), 0.0025 * ref, rtol=_tol(dtype), atol=_tol(dtype))


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_param_accum_ragged_segments(dtype):
    dev = require_gpu()
    lens = [1, 50 * 2, 50, 2 * 50, 2, 7, 512 * 512, 3, 0, 1025]
    ops = _ops(dtype, 8)
    offs, off = [], 0
    for l in lens:
        offs.append(off)
        off += l
    g = torch.Generator().manual_seed(3)
    mu = torch.randn(off, generator=g, dtype=dtype).to(dev)
    grads = [torch.randn(l, generator=g, dtype=dtype).to(dev) if l and i != 5 else None for i, l in enumerate(lens)]
    ref = mu.clone()
    for gr, o, l in zip(grads, offs, lens):
        if gr is not None:
            ref[o:o + l] += gr
    ops.param_accum(mu, 1.0, grads, offs, lens)
    assert torch.equal(mu, ref)
    ops.param_accum(mu, -0.5, grads, offs, lens)
    for gr, o, l in zip(grads, offs, lens):
        if gr is not None:
            ref[o:o + l] += -0.5 * gr
    assert torch.allclose(mu, ref, rtol=_tol(dtype), atol=_tol(dtype))
    # more tensors than one launch takes (48)
    lens = [5] * 130
    offs = [5 * i for i in range(130)]
    mu = torch.zeros(650, dtype=dtype, device=dev)
    grads = [torch.full((5,), float(i), dtype=dtype, device=dev) for i in range(130)]
    ops.param_accum(mu, 1.0, grads, offs, lens)
    assert torch.equal(mu.view(130, 5)[:, 0].cpu(), torch.arange(130, dtype=dtype))


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("nsets", [1, 4, 6, 8, 11])
def test_param_accum_multi_equals_successive_single_launches_bitwise(dtype, nsets):
    """All stages' parameter gradients of one time step in one launch: same rounding as one
    pn_param_accum per stage, in the same order; missing gradients (allow_unused) and misaligned
    tensors (odd offsets) included; more tensors than one launch takes (24)."""
    dev = require_gpu()
    lens = [1, 100, 50, 7, 512 * 512, 3, 0, 1025] + [5] * 30
    ops = _ops(dtype, 8)
    offs, off = [], 0
    for l in lens:
        offs.append(off)
        off += l
    g = torch.Generator().manual_seed(7)
    mu0 = torch.randn(off, generator=g, dtype=dtype).to(dev)
    sets, alphas = [], []
    for j in range(nsets):
        sets.append([torch.randn(l, generator=g, dtype=dtype).to(dev) if l and (i + j) % 5 != 3 else None
                     for i, l in enumerate(lens)])
        alphas.append(0.25 * (j + 1) * (-1) ** j)
    sets[0][4] = None
    one = mu0.clone()
    for a, gs in zip(alphas, sets):
        ops.param_accum(one, a, gs, offs, lens)
    multi = mu0.clone()
    ops.param_accum_multi(multi, alphas, sets, offs, lens)
    assert torch.equal(one, multi)
    ref = mu0.double().cpu()
    for a, gs in zip(alphas, sets):
        for gr, o, l in zip(gs, offs, lens):
            if gr is not None:
                ref[o:o + l] += a * gr.double().cpu()
    assert torch.allclose(multi.double().cpu(), ref, rtol=10 * _tol(dtype), atol=10 * _tol(dtype))


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("n", [1, 5, 257, 65537, 4096 * 512])
@pytest.mark.parametrize("nk", [1, 3, 8, 11])
def test_dots(dtype, n, nk):
    ops = _ops(dtype, n)
    x, *ys = _rand(n, dtype, nk + 1, ops.device)
    got = ops.dots(x, ys)
    want = [float(torch.dot(x.double().cpu(), y.double().cpu())) for y in ys]
    scale = float(x.double().norm() * max(y.double().norm() for y in ys)) + 1e-300
    assert len(got) == nk
    for g, w in zip(got, want):
        assert abs(g - w) <= 1e-12 * scale        # products and sums are carried in double
    assert ops.dots(x, ys) == got                 # bit-reproducible
    assert ops.dots(x, [x])[0] == pytest.approx(float(x.double().pow(2).sum()), rel=1e-13)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("n", [3, 257, 4096 * 512])
def test_lincomb_general_and_in_place(dtype, n):
    ops = _ops(dtype, n)
    xs = _rand(n, dtype, 8, ops.device)
    cs = [0.5, -1.25, 2.0, 0.125, -0.75, 1.5, -2.5, 0.0625]
    ref = sum(c * x.double() for c, x in zip(cs, xs))
    out = torch.empty(n, dtype=dtype, device=ops.device)
    ops.lincomb(out, xs, cs)
    assert torch.allclose(out.double(), ref, rtol=_tol(dtype) * 4, atol=_tol(dtype) * 4)
    ref2 = xs[0].double() - 0.5 * xs[1].double()
    ops.lincomb(xs[0], [xs[0], xs[1]], [1.0, -0.5])            # out aliases an input
    assert torch.allclose(xs[0].double(), ref2, rtol=_tol(dtype), atol=_tol(dtype))


def test_linearity_property_at_target_size():
    """stage(u, K, a) + stage(u, K, b) - u == stage(u, K, a+b) up to rounding, N = 4096*512."""
    n = 4096 * 512
    ops = _ops(torch.float64, n)
    u, k1, k2 = _rand(n, torch.float64, 3, ops.device)
    ya, yb, yc = (torch.empty(n, dtype=torch.float64, device=ops.device) for _ in range(3))
    ops.rk_stage(ya, u, [k1, k2], [0.25, 0.5])
    ops.rk_stage(yb, u, [k1, k2], [0.5, -0.125])
    ops.rk_stage(yc, u, [k1, k2], [0.75, 0.375])
    assert torch.allclose(ya + yb - u, yc, rtol=1e-13, atol=1e-13)


def test_profiler_counts_launches_and_bytes():
    from pnode_amd import _lib
    lib = _lib.load()
    n = 4096 * 512
    ops = _ops(torch.float32, n)
    u, k = _rand(n, torch.float32, 2, ops.device)
    y = torch.empty_like(u)
    lib.pn_prof_enable(1)
    for _ in range(5):
        ops.rk_stage(y, u, [k], [0.5])
    L = (ctypes.c_int64 * len(_lib.KERNEL_IDS))()
    us = (ctypes.c_double * len(_lib.KERNEL_IDS))()
    by = (ctypes.c_double * len(_lib.KERNEL_IDS))()
    assert lib.pn_prof_collect(len(L), L, us, by) == 0
    lib.pn_prof_enable(0)
    assert L[0] == 5 and by[0] == 5 * 3 * n * 4
    assert 1.0 < us[0] / 5 < 1000.0      # a few microseconds per launch


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_combine_wrms_single_launch_is_reproducible_and_independent_of_the_geometry(dtype):
    """Round 3: one launch per attempt.  The norm is finished on the host from the workgroups' partial sums in a pinned
    block (default), or inside the launch (PN_TUNE wfin=1: the block that arrives last adds the block partials in index
    order; cross-XCD hand-off by write-through stores + sharded agent-scope arrival counters).  Checked: many
    back-to-back launches on one work area, of alternating sizes and interleaved with other kernels, return the same bits
    for the same input (a lost or stale partial, or a counter that is not back at zero, would show); one and two vectors
    per thread and both ways of finishing agree to round-off of the double sum; the value equals the fp64 restatement."""
    from pnode_amd import _lib
    lib = _lib.load()
    n = 4096 * 512
    ops = _ops(dtype, n)
    small = _ops(dtype, 70001)
    nk = 6
    u, *K = _rand(n, dtype, nk + 1, ops.device)
    ce = [1e-4 * (-1) ** j * (j + 1) for j in range(nk)]
    su, *sK = [x[:70001].clone() for x in [u] + K]
    a = torch.randn(1024, 1024, device=ops.device)
    vals, svals = [], []
    try:
        for rep in range(40):
            if rep % 3 == 0:
                a = a @ a * 1e-3                      # something else in the queue: blocks arrive unevenly
            ops.combine_wrms(None, u, K, [0.0] * nk, ce, 1e-4, 1e-4)
            vals.append(ops.read_enorm())
            small.combine_wrms(None, su, sK, [0.0] * nk, ce, 1e-4, 1e-4)
            svals.append(small.read_enorm())
        assert len(set(vals)) == 1 and len(set(svals)) == 1
        lib.pn_tune_set(b"wvpt=1")
        ops.combine_wrms(None, u, K, [0.0] * nk, ce, 1e-4, 1e-4)
        v1 = ops.read_enorm()
        lib.pn_tune_set(b"wvpt=2")
        ops.combine_wrms(None, u, K, [0.0] * nk, ce, 1e-4, 1e-4)
        v2 = ops.read_enorm()
        lib.pn_tune_set(b"wvpt=4")
        ops.combine_wrms(None, u, K, [0.0] * nk, ce, 1e-4, 1e-4)
        v4 = ops.read_enorm()
        # the in-launch finish (arrival counters; the default finishes on the host): same value to round-off of the sum,
        # the same bits launch after launch
        lib.pn_tune_set(b"wfin=1")
        fin = []
        for rep in range(20):
            if rep % 3 == 0:
                a = a @ a * 1e-3
            ops.combine_wrms(None, u, K, [0.0] * nk, ce, 1e-4, 1e-4)
            fin.append(ops.read_enorm())
        assert len(set(fin)) == 1 and fin[0] == pytest.approx(v4, rel=1e-13)
    finally:
        lib.pn_tune_set(None)
    assert v1 == pytest.approx(v2, rel=1e-13) and v4 == pytest.approx(v2, rel=1e-13) and vals[0] in (v1, v2, v4)
    npd = np.float32 if dtype == torch.float32 else np.float64
    un = u.cpu().numpy()
    err = np.zeros(n, dtype=npd)
    for c, k in zip(ce, K):
        err = (err + npd(c) * k.cpu().numpy()).astype(npd)
    want = ts_oracle.wrms(un, (un + err).astype(npd), 1e-4, 1e-4)
    assert vals[0] == pytest.approx(want, rel=5e-3 if dtype == torch.float32 else 1e-9)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("n", [1, 255, 65537, 4096 * 512 + 3])
def test_dots_single_launch_matches_fp64_and_is_reproducible(dtype, n):
    ops = _ops(dtype, n)
    x, *ys = _rand(n, dtype, 12, ops.device)
    got = [ops.dots(x, ys + [x]) for _ in range(6)]
    assert all(g == got[0] for g in got)
    want = [float(torch.dot(x.double(), y.double())) for y in ys + [x]]
    scale = float(x.double().norm()) * max(float(y.double().norm()) for y in ys + [x])
    assert np.allclose(got[0], want, rtol=0, atol=1e-14 * scale * n ** 0.5 if dtype == torch.float64 else 1e-6 * scale)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("rows,cols", [(1, 1), (3, 5), (64, 512), (4096, 512), (4096, 2), (1000, 257), (33, 1152), (20000, 64), (7, 4096)])
def test_colsum_accum(dtype, rows, cols):
    """pn_colsum_accum: mu += alpha * column sums of G -- the sensitivity of a bias (round 5; the reduction autograd takes for
    d/d(bias) of a Linear layer fused with the accumulation into mu, pa.py:341-363).  Against a float64 sum; twice into the same
    mu (accumulation), with an unaligned mu slice and a ragged column count; bit-reproducible from launch to launch."""
    ops = _ops(dtype, 64)
    dev = ops.device
    g = torch.Generator().manual_seed(rows * 31 + cols)
    G = torch.randn(rows, cols, generator=g, dtype=dtype).to(dev)
    base = torch.randn(cols + 3, generator=g, dtype=dtype).to(dev)
    mu = base[3:].clone() if cols % 2 else base[:cols].clone()
    mu0 = mu.clone()
    ops.colsum_accum(G, mu, 0.75)
    ops.colsum_accum(G, mu, -0.25)
    ref = mu0.double() + 0.5 * G.double().sum(0)
    scale = float(G.double().abs().sum(0).max()) + 1.0
    tol = (4e-7 if dtype == torch.float32 else 1e-15) * scale
    assert float((mu.double() - ref).abs().max()) <= tol
    again = mu0.clone()
    ops.colsum_accum(G, again, 0.75)
    ops.colsum_accum(G, again, -0.25)
    assert torch.equal(again, mu)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("nsrc", [2, 5, 16, 32, 40])
def test_colsum_accum_multi(dtype, nsrc):
    """pn_colsum_accum_multi: the bias sensitivities of several layers / stages / time steps in ONE pass (what the engine queues
    under -pn_param_accum batch|step).  Sources of different shapes, several of them adding to the same mu slice; equal to the
    float64 sums, and BIT-identical to the same sources added one call at a time (the grouping must not change a bit)."""
    ops = _ops(dtype, 64)
    dev = ops.device
    g = torch.Generator().manual_seed(nsrc)
    shapes = [(4096, 512), (4096, 512), (300, 512), (4096, 64), (77, 130), (4096, 512)]
    mus = {512: torch.randn(512, generator=g, dtype=dtype).to(dev), 64: torch.randn(64, generator=g, dtype=dtype).to(dev),
           130: torch.randn(130, generator=g, dtype=dtype).to(dev)}
    items = []
    for j in range(nsrc):
        rows, cols = shapes[j % len(shapes)]
        items.append((torch.randn(rows, cols, generator=g, dtype=dtype).to(dev), mus[cols], 0.1 * (j + 1) * (-1) ** j))
    ref = {c: m.double().clone() for c, m in mus.items()}
    scale = {c: 1.0 for c in mus}
    for G, m, a in items:
        ref[G.shape[1]] += a * G.double().sum(0)
        scale[G.shape[1]] += abs(a) * float(G.double().abs().sum(0).max())
    one = {c: m.clone() for c, m in mus.items()}
    for G, m, a in items:
        ops.colsum_accum(G, one[G.shape[1]], a)
    ops.colsum_accum_multi(items)                                   # (40 sources: two launches of 32 + 8)
    for c, m in mus.items():
        tol = (4e-7 if dtype == torch.float32 else 1e-15) * scale[c]
        assert float((m.double() - ref[c]).abs().max()) <= tol
        assert torch.equal(m, one[c])


@pytest.mark.parametrize("dtype", [torch.float32, "fp32-exact", torch.float64])
@pytest.mark.parametrize("rows,out_f,in_f", [(256, 64, 64), (512, 128, 64), (4096, 512, 512), (1024, 64, 192),
                                             (1000, 64, 128), (257, 64, 64), (300, 128, 64), (4100, 128, 128)])      # ragged row counts: zero-filled tail
@pytest.mark.parametrize("bias", [True, False])
def test_linear_wgrad_fused_kernel(dtype, rows, out_f, in_f, bias):
    """pn_linear_wgrad / pn_linear_wgrad_finish (csrc/pn_linear.hip; fp32 by exact three-way bf16 splitting on
    v_mfma_f32_32x32x16_bf16 -- the default -- or on v_mfma_f32_32x32x2_f32, fp64 on v_mfma_f64_16x16x4_f64): dW and db of a Linear layer from the cotangent G and the input X, accumulated over several
    (G, X, alpha) -- the stages and steps of a reverse sweep -- in the partial buffers, then added to mu.  Against float64
    (fp64: <= 1e-13 relative); bit-reproducible; the partial buffers come back zero; unsupported shapes are refused."""
    from pnode_amd import _lib
    exact = dtype == "fp32-exact"          # PN_WGRAD_EXACT_FP32: v_mfma_f32_32x32x2_f32 instead of the split-bf16 form (the default)
    dtype = torch.float32 if exact else dtype
    ops = _ops(dtype, 64)
    ops.wgrad_flags = _lib.PN_WGRAD_EXACT_FP32 if exact else 0
    dev = ops.device
    f32 = dtype == torch.float32
    # any number of rows from 256 on (eight K ranges of whole 32-row slabs; rows that do not exist are read as zeros)
    assert ops.linear_wgrad_supported(rows, out_f, in_f) and ops.linear_wgrad_supported(rows + 33, out_f, in_f)
    assert not ops.linear_wgrad_supported(255, out_f, in_f) and not ops.linear_wgrad_supported(0, out_f, in_f)
    assert not ops.linear_wgrad_supported(rows, out_f + 8, in_f) and not ops.linear_wgrad_supported(rows, out_f, in_f - 4)
    assert ops.linear_wgrad_supported(rows, 2048, 2048) and not ops.linear_wgrad_supported(rows, 4096, 2048)   # 8 x weight partials
    gen = torch.Generator().manual_seed(rows + out_f)
    pairs = [(torch.randn(rows, out_f, generator=gen, dtype=dtype).to(dev), torch.randn(rows, in_f, generator=gen, dtype=dtype).to(dev) - 0.3, a)
             for a in (0.5, -0.125, 1.0, 0.3)]
    mu_w0 = torch.randn(out_f, in_f, generator=gen, dtype=dtype).to(dev)
    mu_b0 = torch.randn(out_f, generator=gen, dtype=dtype).to(dev)

    def run():
        pw, pb = ops.linear_wgrad_buffers(out_f, in_f, bias)
        assert pw.dtype == dtype and (pb is None or pb.dtype == torch.float64)
        mu_w, mu_b = mu_w0.clone(), mu_b0.clone()
        for G, X, a in pairs:
            ops.linear_wgrad(G, X, a, pw, pb)
        ops.linear_wgrad_finish(out_f, in_f, pw, pb, mu_w, mu_b if bias else None)
        assert float(pw.abs().max()) == 0.0 and (pb is None or float(pb.abs().max()) == 0.0)
        return mu_w, mu_b
    mu_w, mu_b = run()
    ref_w, ref_b, sw, sb = mu_w0.double(), mu_b0.double(), 1.0, 1.0
    for G, X, a in pairs:
        ref_w = ref_w + a * (G.double().t() @ X.double())
        ref_b = ref_b + a * G.double().sum(0)
        sw += abs(a) * float((G.double().abs().t() @ X.double().abs()).max())
        sb += abs(a) * float(G.double().abs().sum(0).max())
    tol = 4e-7 if f32 else 2e-15
    assert float((mu_w.double() - ref_w).abs().max()) <= tol * sw
    if not f32:
        assert float((mu_w - ref_w).abs().max()) <= 1e-13 * float(ref_w.abs().max())
    if bias:
        assert float((mu_b.double() - ref_b).abs().max()) <= tol * sb
    else:
        assert torch.equal(mu_b, mu_b0)
    again_w, again_b = run()
    assert torch.equal(again_w, mu_w) and torch.equal(again_b, mu_b)


@pytest.mark.parametrize("rows", [4096, 1000, 289])
def test_linear_wgrad_the_128_tile_form_gives_the_bits_of_the_64_tile_form(rows):
    """csrc/pn_linear.hip, wgrad_body_x3_wide: a grouped launch whose 128 x 128 tiles fill whole rounds of the chip takes that
    form (sixteen waves, wave tiles of 64 x 32) by itself; PN_WGRAD_TILE_64 keeps the 64 x 64 one.  Every element of the dW
    partials and every db partial (per K range and 64-wide tile column) must be the same bits -- also with ragged rows, with and
    without bias, accumulated over two launches -- so that a result never depends on how a launch was tiled."""
    from pnode_amd import _lib
    ops = _ops(torch.float32, 64)
    dev = ops.device
    gen = torch.Generator().manual_seed(rows)
    shapes = [(512, 512, True), (512, 512, False), (256, 512, True), (512, 256, True), (128, 128, True), (512, 512, True), (384, 640, True), (512, 512, True)]
    layers = [(torch.randn(rows, o, generator=gen).to(dev) * 0.7, torch.randn(rows, i, generator=gen).to(dev) + 0.1, o, i, b) for o, i, b in shapes]
    res = {}
    for flags in (_lib.PN_WGRAD_TILE_64, 0):
        ops.wgrad_flags = flags
        bufs = [ops.linear_wgrad_buffers(o, i, b) for _, _, o, i, b in layers]
        for alpha in (0.5, -1.25):
            ops.linear_wgrad_group([(G, X, alpha * (k + 1), pw, pb) for k, ((G, X, _, _, _), (pw, pb)) in enumerate(zip(layers, bufs))])
        res[flags] = bufs
    ops.wgrad_flags = 0
    for (pw1, pb1), (pw2, pb2), (G, X, o, i, b) in zip(res[_lib.PN_WGRAD_TILE_64], res[0], layers):
        assert torch.equal(pw1, pw2), (o, i)
        assert pb1 is None or torch.equal(pb1, pb2), (o, i)
    # (and the numbers are right: the first layer against float64)
    G, X, o, i, _ = layers[0]
    want = (0.5 - 1.25) * (G.double().t() @ X.double())
    got = res[0][0][0].view(8, o, i).double().sum(0)
    assert float((got - want).abs().max()) <= 4e-6 * float(want.abs().max())
    wb = (0.5 - 1.25) * G.double().sum(0)
    gb = res[0][0][1].view(-1, o).sum(0)
    assert float((gb - wb).abs().max()) <= 1e-5 * float(wb.abs().max())


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_linear_wgrad_group_equals_the_pairs_one_by_one(dtype):
    """pn_linear_wgrad_group: the (cotangent, input) pairs of SEVERAL layers -- one stage VJP's -- in one launch; layers of
    different shapes, with and without bias, 1..10 pairs (two launches beyond PN_WGRAD_MAX_PAIRS = 8).  The same bits as
    pn_linear_wgrad pair by pair; a group that holds the same partial buffer twice is the caller's error (not checked here)."""
    ops = _ops(dtype, 64)
    dev = ops.device
    gen = torch.Generator().manual_seed(11)
    rows = 1024
    shapes = [(512, 512, True), (64, 192, False), (128, 64, True), (512, 64, True), (64, 64, False),
              (192, 128, True), (64, 512, True), (256, 256, False), (128, 128, True), (64, 128, True)]
    layers = []
    for out_f, in_f, bias in shapes:
        G = torch.randn(rows, out_f, generator=gen, dtype=dtype).to(dev)
        X = torch.randn(rows, in_f, generator=gen, dtype=dtype).to(dev) + 0.2
        layers.append((G, X, out_f, in_f, bias))
    for n in (1, 4, 8, 10):
        res = {}
        for mode in ("single", "group"):
            bufs = [ops.linear_wgrad_buffers(o, i, b) for _, _, o, i, b in layers[:n]]
            for rep, alpha in enumerate((0.5, -1.25)):                         # two stages accumulate into the same partials
                items = [(G, X, alpha * (k + 1), pw, pb) for k, ((G, X, _, _, _), (pw, pb)) in enumerate(zip(layers[:n], bufs))]
                if mode == "single":
                    for it in items:
                        ops.linear_wgrad(*it)
                else:
                    ops.linear_wgrad_group(items)
            res[mode] = bufs
        for k, ((pw1, pb1), (pw2, pb2), (G, X, o, i, b)) in enumerate(zip(res["single"], res["group"], layers)):
            assert torch.equal(pw1, pw2) and (pb1 is None or torch.equal(pb1, pb2))
            want = (0.5 - 1.25) * (k + 1) * (G.double().t() @ X.double())
            got = pw2.view(8, o, i).double().sum(0)
            assert float((got - want).abs().max()) <= (4e-6 if dtype == torch.float32 else 1e-12) * float(want.abs().max())
