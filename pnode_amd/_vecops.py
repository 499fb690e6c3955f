"""The device entry points of ``include/pnode_amd.h`` as methods on tensors: ``HipVecOps`` (what ODEPetsc calls ``self._ops``;
the CPU-only test-suite injects ``tests/_cpu_vecops.py`` in its place) and the buffers of the device-resident GMRES.  Each method
names the C entry point it wraps; what that entry point replaces in PETSc is in the header."""
import contextlib
import ctypes
import warnings

import torch
import torch.nn as nn  # noqa: F401

from . import _lib, options
from ._lib import PnError, check  # noqa: F401


class HipVecOps(object):
    """Device entry points of the C ABI over flat torch tensors on one HIP device."""

    def __init__(self, device, dtype, n):
        if device.type != "cuda":
            raise RuntimeError(
                "pnode_amd runs on MI355X HIP devices only (got a %s tensor); there is no CPU path" % device.type)
        self.lib = _lib.load()
        self.device, self.dtype, self.n = device, dtype, n
        self.code = _lib.dtype_code(dtype)
        self.work = None
        self.dots_work = None
        self._err_host = self._err_dev = None
        self._pinned_stream = None
        self._seg_cache = {}
        self._ptr_buf = (ctypes.c_void_p * 16)()
        self._colsum_work = None

    def __del__(self):
        try:
            for h in (self._err_host, getattr(self, "_dots_host", None)):
                if h is not None and h.value:
                    self.lib.pn_pinned_free(h)
        except Exception:
            pass

    def stream(self):
        """The calling thread's current HIP stream (pinned for the duration of a sweep: looking it
        up costs more host time than a launch)."""
        st = self._pinned_stream           # (a c_void_p holding 0 -- the default stream -- is falsy: compare with None)
        return st if st is not None else ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def empty(self, *shape):
        return torch.empty(*shape, dtype=self.dtype, device=self.device)

    def _ptrs(self, tensors):
        """Device pointers of `tensors` as a C array.  One reusable array: the entry points copy what they need
        before they return, and building a ctypes array per launch costs more host time than the launch."""
        buf = self._ptr_buf
        k = 0
        for t in tensors:
            buf[k] = t.data_ptr()
            k += 1
        return buf

    @staticmethod
    def _dbl(vals):
        """C array of doubles; arrays prepared once per (tableau, step size) are passed through."""
        if isinstance(vals, ctypes.Array):
            return vals
        return (ctypes.c_double * len(vals))(*vals)

    dbl = _dbl

    def rk_stage(self, y, u, Ks, coefs):
        check(self.lib.pn_rk_stage(self.stream(), self.code, self.n, y.data_ptr(), u.data_ptr(),
                                   len(Ks), self._ptrs(Ks), self._dbl(coefs)))

    # the C++ step loops (pn_rk_attempt / pn_rk_adjoint_step) launch this library's HIP entry points themselves
    native_steps = True
    vec_ops = None

    def wrms_buffers(self):
        """(work area, pinned result block) of the error-norm kernel, made on first use."""
        if self.work is None:
            # zero-filled once: the first words are the kernel's arrival counter, which every launch leaves at zero
            self.work = torch.zeros(self.lib.pn_wrms_work_bytes(self.n) // 8 + 1, dtype=torch.float64, device=self.device)
            h, d = ctypes.c_void_p(), ctypes.c_void_p()
            # pinned block the kernel's workgroups store their partial sums into; read_enorm adds them on the host
            check(self.lib.pn_pinned_block(8 * self.lib.pn_wrms_partials(self.n), ctypes.byref(h), ctypes.byref(d)))
            self._err_host, self._err_dev = h, d
        return self.work.data_ptr(), self._err_dev

    def combine_wrms(self, unew, u, Ks, cb, ce, atol, rtol):
        self.wrms_buffers()
        check(self.lib.pn_rk_combine_wrms(self.stream(), self.code, self.n,
                                          None if unew is None else unew.data_ptr(), u.data_ptr(),
                                          len(Ks), self._ptrs(Ks), self._dbl(cb), self._dbl(ce),
                                          atol, rtol, self.work.data_ptr(), self._err_dev))

    def read_enorm(self):
        v = ctypes.c_double()
        check(self.lib.pn_stream_wait_wrms(self.stream(), self._err_host, self.n, ctypes.byref(v)))
        return v.value

    def adj_theta(self, w, lam, c_lam, dlams, coefs):
        check(self.lib.pn_adj_theta(self.stream(), self.code, self.n, w.data_ptr(),
                                    None if lam is None else lam.data_ptr(), c_lam,
                                    len(dlams), self._ptrs(dlams), self._dbl(coefs)))

    def adj_accum(self, lam_out, lam, dlams, coefs, forcing, w_next=None, c_next=0.0):
        check(self.lib.pn_adj_accum(self.stream(), self.code, self.n, lam_out.data_ptr(), lam.data_ptr(),
                                    len(dlams), self._ptrs(dlams), self._dbl(coefs),
                                    None if forcing is None else forcing.data_ptr(),
                                    None if w_next is None else w_next.data_ptr(), c_next))

    def _segments(self, offsets, lens):
        """ctypes copies of the (constant) parameter layout, built once per layout."""
        key = (id(offsets), id(lens), len(offsets))
        c = self._seg_cache.get(key)
        if c is None:
            n = len(offsets)
            c = ((ctypes.c_int64 * n)(*offsets), (ctypes.c_int64 * n)(*lens), offsets, lens)   # keep the lists alive
            self._seg_cache[key] = c
        return c[0], c[1]

    def param_accum(self, mu, alpha, grads, offsets, lens):
        n = len(grads)
        ptrs = (ctypes.c_void_p * n)(*[None if g is None else g.data_ptr() for g in grads])
        off, ln = self._segments(offsets, lens)
        check(self.lib.pn_param_accum(self.stream(), self.code, mu.data_ptr(), alpha, n, ptrs, off, ln))

    MAX_SOURCES = 32           # gradient sets per pn_param_accum_multi call (include/pnode_amd.h)

    def param_accum_multi(self, mu, alphas, grad_sets, offsets, lens):
        """mu += sum_j alphas[j]*grad_sets[j] (the stages of one or several time steps), added in the
        order j = 0, 1, ...; one call per MAX_SOURCES sets."""
        n = len(offsets)
        off, ln = self._segments(offsets, lens)
        for k in range(0, len(grad_sets), self.MAX_SOURCES):
            sets = grad_sets[k:k + self.MAX_SOURCES]
            ptrs = (ctypes.c_void_p * (n * len(sets)))(*[None if g is None else g.data_ptr() for gs in sets for g in gs])
            check(self.lib.pn_param_accum_multi(self.stream(), self.code, mu.data_ptr(), len(sets),
                                                self._dbl(alphas[k:k + self.MAX_SOURCES]), n, ptrs, off, ln))

    MAX_COLSUM_SOURCES = 32        # sources per pn_colsum_accum_multi call (include/pnode_amd.h)

    def colsum_accum_multi(self, items):
        """For (g, mu, alpha) in items, in order:  mu[c] += alpha * sum_r g[r, c]  (g: rows x cols, contiguous; mu: the slice of
        the flat parameter-sensitivity buffer that belongs to a bias) -- ONE pass over all the g's per <= 32 items."""
        for k in range(0, len(items), self.MAX_COLSUM_SOURCES):
            part = items[k:k + self.MAX_COLSUM_SOURCES]
            n = len(part)
            rows = (ctypes.c_int64 * n)(*[g.shape[0] for g, _, _ in part])
            cols = (ctypes.c_int64 * n)(*[g.shape[1] for g, _, _ in part])
            need = self.lib.pn_colsum_work_bytes(n, rows, cols) // 8 + 1
            w = self._colsum_work
            if w is None or w.numel() < need:
                w = self._colsum_work = torch.empty(need, dtype=torch.float64, device=self.device)
            gp = (ctypes.c_void_p * n)(*[g.data_ptr() for g, _, _ in part])
            mp = (ctypes.c_void_p * n)(*[m.data_ptr() for _, m, _ in part])
            al = (ctypes.c_double * n)(*[a for _, _, a in part])
            check(self.lib.pn_colsum_accum_multi(self.stream(), self.code, n, rows, cols, gp, mp, al, w.data_ptr()))

    def colsum_accum(self, g, mu, alpha):
        self.colsum_accum_multi([(g, mu, alpha)])

    # ---- the fused weight / bias sensitivity kernel of a Linear layer (csrc/pn_linear.hip)
    def linear_wgrad_supported(self, rows, out_f, in_f):
        return bool(self.lib.pn_linear_wgrad_supported(self.code, rows, out_f, in_f))

    def linear_wgrad_buffers(self, out_f, in_f, bias):
        """Zero-filled partial buffers (pw, pb) of one layer: they carry the sum over the stages and steps of a reverse sweep."""
        nb = ctypes.c_int64()
        nw = self.lib.pn_linear_wgrad_work_bytes(self.code, out_f, in_f, ctypes.byref(nb))
        pw = torch.zeros(nw // (4 if self.dtype == torch.float32 else 8), dtype=self.dtype, device=self.device)
        pb = torch.zeros(nb.value // 8, dtype=torch.float64, device=self.device) if bias else None
        return pw, pb

    def linear_wgrad(self, g, x, alpha, pw, pb):
        self.linear_wgrad_group([(g, x, alpha, pw, pb)])

    MAX_WGRAD_PAIRS = _lib.PN_WGRAD_MAX_PAIRS

    wgrad_flags = 0            # PN_WGRAD_EXACT_FP32 with -pn_linear_wgrad_exact 1

    def linear_wgrad_group(self, items, stream=None):
        """The pairs (g, x, alpha, pw, pb) of several layers -- one stage VJP's -- in ONE launch per <= 8 pairs (pn_linear_wgrad_group);
        all g have the same number of rows.  `stream`: a raw stream handle (default: the sweep's stream)."""
        st = self.stream() if stream is None else stream
        for k in range(0, len(items), self.MAX_WGRAD_PAIRS):
            part = items[k:k + self.MAX_WGRAD_PAIRS]
            arr = (_lib.WgradPair * len(part))()
            for q, (g, x, alpha, pw, pb) in zip(arr, part):
                q.g, q.x, q.pw, q.pb = g.data_ptr(), x.data_ptr(), pw.data_ptr(), (None if pb is None else pb.data_ptr())
                q.alpha, q.out_f, q.in_f = alpha, g.shape[1], x.shape[1]
            check(self.lib.pn_linear_wgrad_group(st, self.code, part[0][0].shape[0], len(part), arr, self.wgrad_flags))

    def linear_wgrad_finish(self, out_f, in_f, pw, pb, mu_w, mu_b):
        check(self.lib.pn_linear_wgrad_finish(self.stream(), self.code, out_f, in_f, pw.data_ptr(), None if pb is None else pb.data_ptr(),
                                              mu_w.data_ptr(), None if mu_b is None else mu_b.data_ptr()))

    def copy(self, y, x):
        check(self.lib.pn_copy(self.stream(), self.code, self.n, y.data_ptr(), x.data_ptr()))

    def lincomb(self, out, xs, cs):
        check(self.lib.pn_lincomb(self.stream(), self.code, self.n, out.data_ptr(), len(xs), self._ptrs(xs), self._dbl(cs)))

    def dots(self, x, ys):
        """[<x, y_j>] as Python floats; any number of vectors, ONE host synchronisation."""
        nmax = 64
        if self.dots_work is None:
            per = (self.lib.pn_dots_work_bytes(self.n) // 8 + 2) // 2 * 2          # every chunk's area stays 16-byte aligned
            self._dots_per = per
            self.dots_work = torch.zeros(per * (nmax // 8), dtype=torch.float64, device=self.device)   # arrival counters start at zero
            h, d = ctypes.c_void_p(), ctypes.c_void_p()
            check(self.lib.pn_pinned_block(8 * nmax, ctypes.byref(h), ctypes.byref(d)))
            self._dots_host, self._dots_dev = h, d
        if len(ys) > nmax:
            return self.dots(x, ys[:nmax]) + self.dots(x, ys[nmax:])
        st = self.stream()
        for c, k in enumerate(range(0, len(ys), 8)):
            chunk = ys[k:k + 8]
            check(self.lib.pn_dots(st, self.code, self.n, x.data_ptr(), len(chunk), self._ptrs(chunk),
                                   self.dots_work.data_ptr() + 8 * c * self._dots_per,
                                   ctypes.c_void_p(self._dots_dev.value + 8 * k)))
        vals = (ctypes.c_double * len(ys))()
        check(self.lib.pn_stream_wait_scalars(st, self._dots_host, len(ys), vals))
        return list(vals)

    # ---- device-resident GMRES (include/pnode_amd.h section 3c).  `reduce`: None, or a callable that sums a small
    # device tensor over the ranks in stream order (the products must be global before GMRES decides anything)
    MAX_KRYLOV_RESTART = 126

    def krylov_new(self, restart):
        return _KrylovBuffers(self, restart)

    def krylov_begin(self, kr, r, rtol, atol, maxit, first, reduce=None):
        st, lib = self.stream(), self.lib
        args = (st, self.code, self.n, kr.m, kr.state.data_ptr(), kr.status_dev, r.data_ptr(), kr.V.data_ptr(), kr.npad,
                kr.vin.data_ptr(), rtol, atol, int(min(maxit, 2 ** 62)), 1 if first else 0)
        if reduce is None:
            check(lib.pn_krylov_begin(*(args + (0,))))
        else:
            check(lib.pn_krylov_begin(*(args + (1,))))
            reduce(kr.products(1))
            check(lib.pn_krylov_begin(*(args + (2,))))

    def krylov_step(self, kr, k, reduce=None):
        st, lib = self.stream(), self.lib
        args = (st, self.code, self.n, kr.m, kr.state.data_ptr(), kr.status_dev, k, kr.w.data_ptr(), kr.V.data_ptr(), kr.npad,
                kr.vin.data_ptr())
        if reduce is None:
            check(lib.pn_krylov_step(*(args + (0,))))
        else:
            check(lib.pn_krylov_step(*(args + (1,))))
            reduce(kr.products(k + 2))
            check(lib.pn_krylov_step(*(args + (2,))))
            reduce(kr.products(k + 2))
            check(lib.pn_krylov_step(*(args + (3,))))

    def krylov_close(self, kr, x):
        check(self.lib.pn_krylov_close(self.stream(), self.code, self.n, kr.m, kr.state.data_ptr(), kr.status_dev,
                                       x.data_ptr(), kr.V.data_ptr(), kr.npad))

    def krylov_status(self, kr):
        """(stop, iterations of this cycle, iterations of the solve, residual-norm estimate) -- waits for the stream."""
        v = kr._vals
        check(self.lib.pn_stream_wait_scalars(self.stream(), kr.status_host, 8, v))
        kr.second_passes = int(v[7])                     # of this solve so far (diagnostic)
        return int(v[0]), int(v[1]), int(v[2]), v[3]


class _KrylovBuffers(object):
    """Device memory of one GMRES solver (pn_krylov_*, include/pnode_amd.h section 3c): the Krylov vectors, the
    operator's input and output buffers, the state block GMRES keeps its decisions in and the pinned status block."""

    def __init__(self, ops, restart):
        lib = ops.lib
        self.m = restart
        self.npad = (ops.n + 63) // 64 * 64
        self.V = ops.empty(restart + 1, self.npad)
        self.vin = ops.empty(self.npad)
        self.w = ops.empty(self.npad)
        nd = lib.pn_krylov_state_doubles(ops.n, restart)
        if nd <= 0:
            raise PnError("pn_krylov: restart length %d is outside 1..126" % restart)
        self.state = torch.zeros(nd, dtype=torch.float64, device=ops.device)      # arrival counter starts at zero
        self.hoff = lib.pn_krylov_products_offset(restart)
        h, d = ctypes.c_void_p(), ctypes.c_void_p()
        check(lib.pn_pinned_block(64, ctypes.byref(h), ctypes.byref(d)))
        self.status_host, self.status_dev = h, d
        self._lib = lib
        self._vals = (ctypes.c_double * 8)()

    def __del__(self):
        try:
            if self.status_host.value:
                self._lib.pn_pinned_free(self.status_host)
        except Exception:
            pass

    def products(self, count):
        """The Gram-Schmidt products of the pass in flight (for the sum over the ranks)."""
        return self.state[self.hoff: self.hoff + count]
