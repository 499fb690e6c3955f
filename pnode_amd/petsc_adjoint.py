"""Drop-in for ``pnode.petsc_adjoint`` on the explicit-RK path, MI355X-native.

Same surface as the reference (``/root/reference/pnode/petsc_adjoint.py``, "pa.py"):
``ODEPetsc.setupTS / odeint / odeint_adjoint`` and ``OdeintAdjointMethod`` with the same
arguments, argument meaning and error behaviour -- but no PETSc/petsc4py underneath.  The
PETSc TS / TSAdapt / TSAdjoint / TSTrajectory / Vec machinery the reference drives
(pa.py:370, 637-656, 766-775, 812-829, 875-878) is replaced by ``libpnode_amd.so``
(``include/pnode_amd.h``): hand-written gfx950 kernels for all state-vector arithmetic and a
C++ host engine for the stepper state machine and the checkpoint schedule.  What stays in
Python is what is Python in the reference too: the callback shells around the user's
``nn.Module`` (pa.py:393-412 ``evalRHSFunction``, 52-82 ``RHSJacShell.multTranspose``,
341-363 ``RHSJacPShell.multTranspose``) and the autograd entry (pa.py:903-947).

The product path has no CPU fallback: states must live on a HIP device and the shared
library must be present, otherwise an exception is raised.
"""
import contextlib
import ctypes
import sys
import warnings

import torch
import torch.nn as nn

from . import _lib, options
from ._lib import PnError, check  # noqa: F401
from .misc import flat_parameters
from ._sweepgraphs import SweepGraphs

__all__ = ["ODEPetsc", "OdeintAdjointMethod", "PnError", "PnUnpinnedWarning"]


class PnUnpinnedWarning(RuntimeWarning):
    """Issued once per process and piece when a piece of PETSc's behaviour that is restated here WITHOUT anything
    PETSc-produced to check it against ("parity unpinned") decides the numbers of a solve."""


_UNPINNED_WARNED = set()
_ENV_WARNED = [False]           # the "HIP runtime initialised before import" note of auto graph capture: once per process


def _warn_unpinned(key, what):
    if key in _UNPINNED_WARNED:
        return
    _UNPINNED_WARNED.add(key)
    warnings.warn("pnode_amd: %s -- restated from PETSc's documentation and the literature, not checked against anything "
                  "PETSc produced (no PETSc build and no PETSc-made log exists in the reference): the numbers claim PETSc's "
                  "semantics, not bit-parity with it.  See DESIGN.md section 3 (parity unpinned).  Said once per process."
                  % what, PnUnpinnedWarning, stacklevel=3)


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
        check(self.lib.pn_linear_wgrad(self.stream(), self.code, g.shape[0], g.shape[1], x.shape[1], g.data_ptr(), x.data_ptr(), alpha,
                                       pw.data_ptr(), None if pb is None else pb.data_ptr()))

    MAX_WGRAD_PAIRS = _lib.PN_WGRAD_MAX_PAIRS

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
            check(self.lib.pn_linear_wgrad_group(st, self.code, part[0][0].shape[0], len(part), arr))

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


_NO_STORES = {}


class _Trajectory(object):
    """HBM-resident checkpoint store: slots planned by the C++ scheduler (pn_traj_*), memory
    owned here as torch slabs.  A slot holds `vecs` state-sized vectors (1 = the state at the
    start of a step; s_eff = state + stage values in store-all mode), each padded to a
    multiple of 64 elements so every vector starts 256-byte aligned."""

    CHUNK_BYTES = 1 << 28

    def __init__(self, lib, ops, n, vecs, mode, max_slots):
        self.lib, self.ops, self.n, self.vecs = lib, ops, n, vecs
        self.npad = (n + 63) // 64 * 64
        self.handle = ctypes.c_void_p(lib.pn_traj_create())
        check(lib.pn_traj_begin(self.handle, mode, max_slots))
        if mode == _lib.PN_TRAJ_BUDGET and vecs > 1:
            check(lib.pn_traj_set_carry(self.handle, 1))     # slots hold stage values: place the checkpoints for that cost
        esize = 4 if ops.dtype == torch.float32 else 8
        slot_bytes = self.vecs * self.npad * esize
        if mode == _lib.PN_TRAJ_BUDGET:
            self.chunk_slots = max(1, int(max_slots))
        else:
            self.chunk_slots = max(1, min(64, self.CHUNK_BYTES // max(1, slot_bytes)))
        self.chunks = []
        self.plan_cap = max(64, min(int(max_slots), 4096)) if mode == _lib.PN_TRAJ_BUDGET else 64
        self._plan_buf = None
        self.stage_step = {}          # slot -> step whose stage values Y_1.. are stored behind the state

    def __del__(self):
        try:
            self.lib.pn_traj_destroy(self.handle)
        except Exception:
            pass

    def view(self, slot):
        """(vecs, npad) tensor of `slot`."""
        c, i = divmod(slot, self.chunk_slots)
        while c >= len(self.chunks):
            self.chunks.append(self.ops.empty(self.chunk_slots, self.vecs, self.npad))
        return self.chunks[c][i]

    def fwd_slot(self, step):
        return self.lib.pn_traj_fwd_slot(self.handle, step)

    def rev_plan(self, step, cap=0):
        cap = cap or self.plan_cap
        buf = self._plan_buf
        if buf is None or buf[0] != cap:
            buf = self._plan_buf = (cap, ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int(),
                                    (ctypes.c_int64 * cap)(), (ctypes.c_int64 * cap)())
        _, fs, fl, ns, ss, sl = buf
        check(self.lib.pn_traj_rev_plan(self.handle, step, ctypes.byref(fs), ctypes.byref(fl), ctypes.byref(ns),
                                        ss, sl, cap))
        return fs.value, fl.value, ({ss[k]: sl[k] for k in range(ns.value)} if ns.value else _NO_STORES)

    def rev_done(self, step):
        check(self.lib.pn_traj_rev_done(self.handle, step))

    def high_water(self):
        return self.lib.pn_traj_high_water(self.handle)

    # the HBM tier needs none of these (see _DiskTrajectory)
    def claim(self, slot):
        """`slot` is about to be (re)written with a new checkpoint: its buffer, whatever it held."""
        return self.view(slot)

    def seal(self, slot):
        pass

    def begin_reverse(self):
        pass

    on_disk = False


class _DiskTrajectory(_Trajectory):
    """``-ts_trajectory_type basic`` -- PETSc's default trajectory type, the one the reference runs with
    unless ``-ts_trajectory_type memory`` is given (examples-pnode/ode_demo_petsc.py:26): every checkpoint
    is a file under ``-ts_trajectory_dirname``.  The device keeps RING checkpoint buffers as a small
    least-recently-used cache; a checkpoint leaves for its file as soon as it is complete (``seal``:
    asynchronous copy + background write, pn_spill_put) and comes back when a sweep asks for it (``view``:
    pn_spill_get), the one before it being read ahead in the reverse sweep.  Works for every placement of
    the checkpoints -- every step, or the bounded set of ``-ts_trajectory_max_cps_ram`` whose slots are
    recycled (``claim``) -- with the same slots, kernels and results as the HBM tier."""

    RING = 4             # at most three buffers are in use at once (source and destination of a step, stage values)
    STAGING = 6          # pinned staging buffers (and STAGING/2 I/O threads)
    _seq = 0
    on_disk = True

    def __init__(self, lib, ops, n, vecs, mode, max_slots, dirname, keep_files):
        import os
        _Trajectory.__init__(self, lib, ops, n, vecs, mode, max_slots)
        self.ring = ops.empty(self.RING, vecs, self.npad)
        self.holds = [-1] * self.RING
        self.stamp = [0] * self.RING
        self.clock = 0
        self.sealed = set()
        self.reverse = False
        esize = 4 if ops.dtype == torch.float32 else 8
        os.makedirs(dirname, exist_ok=True)
        _DiskTrajectory._seq += 1
        self.dir = os.path.join(dirname, "pn-%d-%d" % (os.getpid(), _DiskTrajectory._seq))
        self.spill = ctypes.c_void_p(lib.pn_spill_create(self.dir.encode(), vecs * self.npad * esize, self.STAGING,
                                                         1 if ops.device.type == "cuda" else 0, 1 if keep_files else 0))
        if not self.spill:
            raise PnError(lib.pn_last_error().decode())

    def __del__(self):
        try:
            if self.spill:
                self.lib.pn_spill_destroy(self.spill)
        except Exception:
            pass
        if _Trajectory is not None:                # (None while the interpreter shuts down: module globals go first)
            _Trajectory.__del__(self)

    def _stream(self):
        return self.ops.stream() if hasattr(self.ops, "stream") else None

    def _buffer(self, slot, load):
        self.clock += 1
        if slot in self.holds:
            r = self.holds.index(slot)
        else:
            r = min(range(self.RING), key=lambda k: self.stamp[k])      # least recently used (complete checkpoints are
            if load and slot in self.sealed:                             # in their files already: nothing to write back)
                check(self.lib.pn_spill_get(self.spill, self._stream(), slot, self.ring[r].data_ptr()))
            self.holds[r] = slot
        self.stamp[r] = self.clock
        return r

    def view(self, slot):
        r = self._buffer(slot, True)
        if self.reverse:                                 # the sweep walks backwards: read the one before it ahead
            prev = slot - 1
            if prev >= 0 and prev in self.sealed and prev not in self.holds:
                check(self.lib.pn_spill_prefetch(self.spill, prev))
        return self.ring[r]

    def claim(self, slot):
        self.sealed.discard(slot)                        # what the file holds belongs to the checkpoint that had this slot before
        return self.ring[self._buffer(slot, False)]

    def seal(self, slot):
        """The checkpoint in `slot` is complete (again): off to its file.  Called after every change of a slot's contents."""
        if slot >= 0:
            if slot not in self.holds:
                raise PnError("trajectory disk tier: slot %d sealed without being resident" % slot)
            check(self.lib.pn_spill_put(self.spill, self._stream(), slot, self.ring[self.holds.index(slot)].data_ptr()))
            self.sealed.add(slot)

    def begin_reverse(self):
        self.reverse = True

    def stats(self):
        f, w, r, wt = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
        self.lib.pn_spill_stats(self.spill, ctypes.byref(f), ctypes.byref(w), ctypes.byref(r), ctypes.byref(wt))
        return {"files": f.value, "bytes_written": w.value, "bytes_read": r.value, "waits": wt.value}


class _TwoLevelTrajectory(_DiskTrajectory):
    """``-ts_trajectory_max_cps_ram R`` together with ``-ts_trajectory_max_cps_disk D`` (PETSc's two-level checkpointing,
    /root/reference/README.md:91-96): a bounded set of R + D checkpoints placed by the same scheduler, the first R slots in
    HBM, the other D in files behind the four-buffer device cache of the disk tier.  Same slots, kernels and results as
    a budget of R + D in HBM."""

    def __init__(self, lib, ops, n, vecs, mode, max_slots, dirname, keep_files, ram_slots):
        _DiskTrajectory.__init__(self, lib, ops, n, vecs, mode, max_slots, dirname, keep_files)
        self.ram = int(ram_slots)
        self.chunk_slots = max(1, self.ram)        # one HBM slab for the R resident slots (allocated on first use)

    def view(self, slot):
        return _Trajectory.view(self, slot) if slot < self.ram else _DiskTrajectory.view(self, slot)

    def claim(self, slot):
        return _Trajectory.view(self, slot) if slot < self.ram else _DiskTrajectory.claim(self, slot)

    def seal(self, slot):
        if slot >= self.ram:
            _DiskTrajectory.seal(self, slot)


def _mem_now(device):
    """(bytes allocated, bytes reserved) by PyTorch's caching allocator on `device`.  torch.cuda.memory_allocated()
    flattens the whole statistics dictionary in Python (~90 us per call, measured in the eager sweep's profile); the
    nested dictionary underneath costs a tenth of that."""
    try:
        st = torch._C._cuda_memoryStats(device.index if device.index is not None else torch.cuda.current_device())
        return st["allocated_bytes"]["all"]["current"], st["reserved_bytes"]["all"]["current"]
    except Exception:
        return torch.cuda.memory_allocated(device), torch.cuda.memory_reserved(device)


class ODEPetsc(SweepGraphs):
    """Explicit-RK neural-ODE solver with discrete adjoint (drop-in for pa.py:366-900).  How its sweeps are launched
    (eagerly, or replayed from hipGraphs) lives in the SweepGraphs mixin, pnode_amd/_sweepgraphs.py."""

    def __init__(self, backend=None):
        self._lib = _lib.load()
        self._backend_cls = backend if backend is not None else HipVecOps
        self._ts = ctypes.c_void_p(self._lib.pn_ts_create())
        self.n = 0
        self.tensor_size = None
        self.tensor_dtype = None
        self.device = None
        self.adj_u = []
        self.adj_p = []
        self.mass = None
        self.funcIM = None
        self.funcEX = None
        self.flat_params = None
        self.npIM = self.npEX = self.np = None
        self.imex = None
        self.use_dlpack = True
        self.linear_solver = None
        self.matrixfree_jacobian = True
        self.step_size = 0.01
        self.enable_adjoint = True
        self._traj = None
        self._tmode = None
        self._ops = None
        self._nsteps = 0
        self._tapes = None
        self._init_sweep_graphs()
        self._lin = None               # engine-side parameter sensitivities of func's nn.Linear layers (_lineargrad.py)
        self._lin_sig = None
        self._theta = None
        self._theta_method = None
        self._imex_built = False
        self._paramsI = self._paramsE = self._pnamesI = self._pnamesE = ()
        self._options_sig = None
        self._view = False
        self._rtapes = None
        self._trace = False
        self._pg_enabled = False
        self._pg = None
        self._pg_average = True
        self._pg_global_norm = True
        self.nfe_forward = 0      # f evaluations in forward sweeps (NFE-F of the reference's examples)
        self.nfe_backward = 0     # f evaluations (each followed by a VJP) in reverse sweeps (NFE-B)

    def __del__(self):
        try:
            if self._lin is not None:
                self._lin.remove()             # the forward hooks on func's Linear layers go with the solver
        except Exception:
            pass
        try:
            self._lib.pn_ts_destroy(self._ts)
        except Exception:
            pass

    # ------------------------------------------------------------------ introspection
    @property
    def num_steps(self):
        """Accepted time steps of the last forward solve."""
        return self._nsteps

    @property
    def num_rejections(self):
        """Rejected step attempts of the last forward solve (adaptive schemes)."""
        return int(self._lib.pn_ts_rejections(self._ts))

    def step_log(self):
        """[(t_n, h_n)] of the accepted steps of the last forward solve."""
        return [self._step_info(k) for k in range(self._nsteps)]

    # ------------------------------------------------------------------ multi-GPU
    def setProcessGroup(self, group=None, average=True, global_error_norm=True, enabled=True):
        """Shard the batch of trajectories over the ranks of a ``torch.distributed`` group
        (one process per GPU; backend "nccl" is RCCL over xGMI).  Every rank integrates its own
        contiguous batch shard with replicated parameters; the only data-path exchange is ONE
        all-reduce of the flat parameter-gradient buffer per backward.  ``average`` divides by
        the world size (loss = mean over the global batch).  With an adaptive method
        ``global_error_norm`` additionally all-reduces one scalar (sum of squares) per step
        attempt so that every rank takes the reference's step sequence, whose WRMS norm spans
        the whole flattened batch; turning it off gives per-shard controllers (not parity).
        The reference has no counterpart (it is single-process, pa.py:367 COMM_SELF)."""
        self._pg_enabled = enabled
        self._pg = group
        self._pg_average = average
        self._pg_global_norm = global_error_norm

    def _sharded(self):
        """True when a process group is attached: the collectives are then issued whatever the group's
        size (a one-rank group reduces to itself -- that is how the RCCL path is exercised on one GPU)."""
        import torch.distributed as dist
        return bool(self._pg_enabled and dist.is_available() and dist.is_initialized())

    def _world(self):
        import torch.distributed as dist
        return dist.get_world_size(self._pg) if self._sharded() else 1

    def _global_enorm(self, enorm):
        """sqrt(sum_r n_r*enorm_r^2 / sum_r n_r): the WRMS norm over the global batch."""
        import torch.distributed as dist
        if not self._sharded() or not self._pg_global_norm:
            return enorm
        v = torch.tensor([enorm * enorm * self.n, float(self.n)], dtype=torch.float64, device=self.device)
        dist.all_reduce(v, op=dist.ReduceOp.SUM, group=self._pg)
        s, n = v.tolist()
        return (s / n) ** 0.5

    def _allreduce_adj_p(self):
        import torch.distributed as dist
        if not self._sharded() or self.np == 0:
            return
        w = self._world()
        dist.all_reduce(self.adj_p_tensor, op=dist.ReduceOp.SUM, group=self._pg)
        if self._pg_average and w > 1:
            self.adj_p_tensor.mul_(1.0 / w)

    # ------------------------------------------------------------------ setup (pa.py:534-775)
    def setupTS(self, u_tensor, func, step_size=0.01, enable_adjoint=True, implicit_form=False,
                use_dlpack=True, method="dopri5", mass=None, imex_form=False, func2=None,
                batch_size=1, linear_solver="petsc", fixed_jacobian=False, matrixfree_jacobian=True,
                fixed_jacobian_across_solves=None):
        """Set up the solver before it is used.  Arguments as in pa.py:551-584.

        ``u_tensor`` only donates shape, dtype and device.  ``step_size`` is a float or a list
        (per-step sizes).  ``use_dlpack``, ``batch_size``, ``linear_solver``,
        ``fixed_jacobian`` and ``matrixfree_jacobian`` are accepted and have no effect on the
        explicit path (there is one zero-copy mode and no linear solve); on the IMEX path
        ``linear_solver="torch"`` selects the direct stage solve and ``fixed_jacobian=True`` keeps its
        factors across solves when funcIM has no trainable parameter.  As in the reference,
        ``method`` is applied only when shape, dtype or device change (pa.py:627-656), and
        command-line options (``pnode_amd.init(argv)``) override it (pa.py:775).
        """
        if imex_form and func2 is None:
            raise ValueError("func2 must be provided to enable imex_form=True")
        if fixed_jacobian_across_solves is not None:
            # examples-sinode/KS/KS.py:494,516 still pass the keyword's earlier name
            fixed_jacobian = bool(fixed_jacobian_across_solves)
        from .theta import THETA_METHODS
        theta_method = implicit_form and not imex_form and method in THETA_METHODS
        imex_method = bool(imex_form) and method == "imex"
        if (implicit_form or imex_form or method in ("beuler", "cn", "imex")) and not (theta_method or imex_method):
            raise NotImplementedError(
                "pnode_amd implements the explicit-RK path (euler/midpoint/rk2/bosh3/rk4/dopri5), the implicit "
                "theta methods (implicit_form=True with method 'beuler' or 'cn') and IMEX (imex_form=True with "
                "method 'imex'); other combinations are not built (DESIGN.md section 8)")
        self.imex = imex_form
        self.linear_solver = linear_solver
        self.fixed_jacobian = fixed_jacobian
        self.matrixfree_jacobian = matrixfree_jacobian
        tensor_dtype = u_tensor.dtype
        tensor_size = u_tensor.size()
        device = u_tensor.device
        n = u_tensor.numel()
        if self.funcIM is not func or (imex_form and self.funcEX is not func2) or bool(imex_form) != bool(self._imex_built):
            # pa.py:600-621 -- IMEX: func is treated implicitly, func2 explicitly, flat parameters
            # = [func's, func2's]; otherwise func2 is ignored
            self._imex_built = bool(imex_form)
            self.funcIM = func
            self.funcEX = func2 if imex_form else func
            def trainable(f):
                if not isinstance(f, nn.Module):
                    return (), ()
                return (tuple(p for p in f.parameters() if p.requires_grad),
                        tuple(n for n, p in f.named_parameters() if p.requires_grad))
            self._paramsI, self._pnamesI = trainable(self.funcIM)
            self._paramsE, self._pnamesE = trainable(self.funcEX)
            self._params = self._paramsI + self._paramsE if imex_form else self._paramsE
            # The reference routes dL/dtheta through a cat of parameter views (pa.py:618-620).
            # Here the parameters themselves are inputs of the autograd Function (its `*args`),
            # and flat_params is a detached copy kept for its size/order only: a live cat graph
            # pins the parameters' AccumulateGrad nodes to the stream it was built on, which
            # breaks hipGraph capture of the reverse sweep.
            with torch.no_grad():
                self.flat_params = flat_parameters(self._params)
            self.np = self.flat_params.numel()
            self.npIM = sum(p.numel() for p in self._paramsI) if imex_form else self.np
            self.npEX = self.np - self.npIM if imex_form else self.np
            self._poff, off = [], 0
            for p in self._params:
                self._poff.append(off)
                off += p.numel()
            self._plen = [p.numel() for p in self._params]
            nI = len(self._paramsI) if imex_form else 0
            self._poffI, self._plenI = self._poff[:nI], self._plen[:nI]
            self._poffE, self._plenE = self._poff[nI:], self._plen[nI:]
            self.adj_p_tensor = None
            self._reset_sweep_graphs(new_func=True)
        if self.mass is not mass:
            self.mass = mass
        if tensor_size != self.tensor_size or tensor_dtype != self.tensor_dtype or device != self.device:
            self._ops = self._backend_cls(device, tensor_dtype, n)
            self.tensor_size = tensor_size
            self.tensor_dtype = tensor_dtype
            self.device = device
            self.use_dlpack = use_dlpack
            self.n = n
            self._npad = (n + 63) // 64 * 64
            check(self._lib.pn_ts_set_rk_type(self._ts, self._lib.pn_method_to_rk_type(str(method).encode())))
            self._theta_method = method if theta_method else ("imex" if imex_method else None)   # on rebuild only
            self.adj_u_tensor = None
            self.adj_p_tensor = None
            self._traj = None
            self._work = {}
            self._reset_sweep_graphs()
        self.step_size = step_size
        self.enable_adjoint = enable_adjoint
        if not enable_adjoint:
            self._traj = None          # ts.removeTrajectory() (pa.py:773-774)
        # ts.setFromOptions() (pa.py:775).  Callers such as train-Cifar10.py:121-139 call setupTS on
        # every forward: when neither the tableau choice nor the options database changed since
        # the last call this is a dictionary comparison, not a re-parse.
        sig = (options.get_all(), self._theta_method, id(self._ops))
        if sig != self._options_sig:
            self._set_from_options()
            self._theta = None
            self._reset_sweep_graphs()  # captured sweeps belong to the scheme and modes they were captured with
            # -ts_type on the command line overrides the `method` keyword, as ts.setFromOptions() does
            # (README.md:89: "-ts_type cn will choose the Crank-Nicolson methods")
            ts_type = str(options.get_all().get("ts_type", ""))
            stepper = self._theta_method
            if ts_type in ("beuler", "cn", "theta") and stepper != "imex":
                stepper = ts_type
            elif ts_type == "rk" and stepper != "imex":
                stepper = None
            elif ts_type == "arkimex" and stepper != "imex":
                raise PnError("-ts_type arkimex needs the IMEX set-up (setupTS(..., imex_form=True, method='imex', func2=...))")
            elif ts_type in ("beuler", "cn", "theta", "rk") and stepper == "imex":
                raise PnError("-ts_type %s cannot override an IMEX set-up (two functions were given)" % ts_type)
            self._stepper_kind = stepper
            adapt_wanted = str(options.get_all().get("ts_adapt_type", "basic")) != "none"
            if stepper == "imex":
                from .arkimex import ArkimexStepper
                self._theta = ArkimexStepper(self, options.get_all())
                has_embed = self._theta.embedded() is not None
                check(self._lib.pn_ts_set_scheme(self._ts, self._theta.tab["adapt_order"], 1 if has_embed else 0))
                if adapt_wanted and not has_embed:
                    warnings.warn("pnode_amd: ARKIMEX type %s has no embedded weights here and takes the fixed steps of "
                                  "step_size; PETSc adapts unless -ts_adapt_type none is given (every IMEX run of the "
                                  "reference gives it).  Pass -ts_adapt_type none to state that explicitly, or use type "
                                  "3, 4, 5 or 1bee." % self._theta.name, RuntimeWarning)
                self._adaptive = bool(self._lib.pn_ts_is_adaptive(self._ts))
                if self._theta.name in ("l2", "2c", "2d", "2e"):
                    _warn_unpinned("arkimex_table_" + self._theta.name,
                                   "ARKIMEX type %s: the identification of this table with PETSc's type of that name "
                                   "(l2) / the free entries of its explicit part (2c, 2d, 2e) are" % self._theta.name)
                if self._adaptive:
                    _warn_unpinned("arkimex_adapt", "adaptive ARKIMEX steps (TSAdapt basic on the embedded pair; embedded "
                                   "weights, controller constants and step rejection are")
            elif stepper:
                from .theta import ThetaStepper
                self._theta = ThetaStepper(self, stepper, options.get_all())
                # TSCreate_Theta sets the TS's default adapt type to NONE: beuler / cn take the fixed steps of
                # step_size unless `-ts_adapt_type basic` is given explicitly (the reference's spiral_unstable.py
                # and ode_demo_petsc.py give no adapt option and run cn / beuler with fixed steps).  With it, PETSc
                # estimates the local truncation error from the last three solutions and the controller uses
                # order 2 (see ThetaStepper.error_norm).
                explicit_basic = str(options.get_all().get("ts_adapt_type", "")) == "basic"
                check(self._lib.pn_ts_set_scheme(self._ts, 2, 1 if explicit_basic else 0))
                self._adaptive = bool(self._lib.pn_ts_is_adaptive(self._ts))
                if self._adaptive:
                    _warn_unpinned("theta_adapt", "adaptive beuler / cn steps (-ts_adapt_type basic: the three-solution local "
                                   "truncation error estimate of TSEvaluateWLTE_Theta and its controller order are")
            else:
                check(self._lib.pn_ts_set_scheme(self._ts, 0, 0))          # the RK tableau drives the controller
            self._options_sig = sig
        self._setup_linear_grads()

    def _set_from_options(self):
        """ts.setFromOptions() (pa.py:775) for the option subset of this path."""
        db = options.get_all()
        self._monitor = "ts_monitor" in db
        self._view = "ts_view" in db
        if "log_view" in db and self.device.type == "cuda":
            from . import logview
            logview.enable()
        # not a PETSc option: bracket the sweeps with roctx ranges (visible to rocprofv3 --marker-trace)
        self._trace = self.device.type == "cuda" and options.truthy(db.get("pn_trace"), False) if "pn_trace" in db else False
        # not a PETSc option: ONE switch that takes back every default of this package that a user of the reference
        # could observe (all of them leave the gradients bit-identical):
        #   * stage tapes are not retained (-pn_trajectory_retain_graph 0): func is re-evaluated inside every stage VJP
        #     of the reverse sweep, as RHSJacShell.multTranspose does (pa.py:66-74), so funcs that count their calls
        #     (NFE-B of the reference's drivers, examples-pnode/spiral_unstable.py:326-347) read what they read there;
        #   * PETSc's trajectory default stays solution-only whatever fits in HBM (no automatic stage keeping), and a
        #     reversed step is recomputed WHOLE, last stage derivative included, as TSTrajectory's TSStep does;
        #   * the steps of an output interval are counted with the reference's +-1e-5 / 1e-3 window (-pn_span_count
        #     reference, pa.py:527-532);
        #   * the Newton-Krylov solves launch func eagerly (-pn_krylov_graph 0): its side effects happen at every call.
        # Each of these can still be set on its own; an explicit option wins over the switch.
        self._ref_defaults = options.truthy(db.get("pn_reference_defaults"), False) if "pn_reference_defaults" in db else False
        self._solution_only = options.truthy(db.get("ts_trajectory_solution_only"), True)
        # option not given: PETSc's default (states only, stages recomputed) unless everything fits easily, see _pick_traj_mode
        self._solution_only_auto = "ts_trajectory_solution_only" not in db and not self._ref_defaults
        ram = int(float(db["ts_trajectory_max_cps_ram"])) if db.get("ts_trajectory_max_cps_ram", "") != "" else 0
        disk = int(float(db["ts_trajectory_max_cps_disk"])) if db.get("ts_trajectory_max_cps_disk", "") != "" else 0
        if ram < 0 or disk < 0:
            raise PnError("-ts_trajectory_max_cps_ram / -ts_trajectory_max_cps_disk must not be negative")
        # PETSc's two levels: at most `ram` checkpoints in memory (HBM here) and `disk` more in files; one scheduler places the
        # ram + disk of them (_TwoLevelTrajectory).  Only -ts_trajectory_max_cps_disk: the bounded set lives in files.
        self._max_cps = ram + disk
        self._max_cps_ram, self._max_cps_disk = ram, disk
        # -ts_trajectory_type: "memory" = HBM (the default here; PETSc's default is "basic" = one file per checkpoint,
        # which is what "basic" selects here too); PETSc's other types are not built and are refused, not ignored
        ttype = str(db.get("ts_trajectory_type", "memory"))
        if ttype not in ("memory", "basic"):
            raise PnError("-ts_trajectory_type %s is not implemented by pnode_amd (memory: checkpoints in HBM; basic: one file "
                          "per checkpoint under -ts_trajectory_dirname)" % ttype)
        self._traj_disk = ttype == "basic" or disk > 0
        self._traj_all_on_disk = ttype == "basic"          # -ts_trajectory_type basic: every checkpoint is a file, whatever the budgets
        self._traj_dirname = str(db.get("ts_trajectory_dirname", "SA-data"))
        self._traj_keep = options.truthy(db.get("ts_trajectory_keep_files"), False) if "ts_trajectory_keep_files" in db else False
        # not a PETSc option.  With store-all checkpoints (-ts_trajectory_solution_only 0) the forward sweep can
        # also keep every stage's autograd tape, so that the reverse sweep runs only the backward half of each
        # stage VJP instead of re-evaluating f first (pa.py:66-68 re-evaluates).  Same bits either way.
        #   auto (default): tapes are kept while they fit in half of the HBM that is free when the sweep starts
        #   1: always   0: never (the reference's recompute)
        rg = str(db.get("pn_trajectory_retain_graph", "0" if self._ref_defaults else "auto"))
        self._retain_graph = 2 if rg == "auto" else (1 if options.truthy(rg, False) else 0)
        # not a PETSc option: how the steps of an output interval are counted (see _span_post_step)
        self._span_count_reference = str(db.get("pn_span_count", "reference" if self._ref_defaults else "exact")) == "reference"
        self._accum_mode = str(db.get("pn_param_accum", "batch"))
        if self._accum_mode not in ("batch", "step", "stage"):
            raise PnError("-pn_param_accum must be batch, step or stage")
        self._accum_sources = max(1, min(32, int(float(db.get("pn_param_accum_sources", 32)))))
        # not a PETSc option: who walks the tableau.  native (default): one C++ entry point per step attempt / reversed step
        # (pn_rk_attempt, pn_rk_adjoint_step -- PETSc's C loops behind ts.solve / ts.adjointSolve, pa.py:829, 878) that calls
        # back only for func and its VJP; python: the stage loop of rounds 1-3, one ctypes call per launch.  Same launches,
        # same coefficients, same bits.
        sl = str(db.get("pn_step_loop", "native"))
        if sl not in ("native", "python"):
            raise PnError("-pn_step_loop must be native or python")
        self._native = sl == "native" and bool(getattr(self._ops, "native_steps", False))
        # not a PETSc option: after GRAPH_WARMUP_CALLS eager calls with the same shapes/times,
        # capture the whole forward sweep and the whole reverse sweep as two hipGraphs and
        # replay them (fixed-step only; func must be capturable: no host-side data dependence)
        #   auto (the default on a HIP device; off under -pn_reference_defaults): explicit fixed-step RK sweeps only, and only
        #        when it is safe and pays -- see _graph_entry / _auto_capture_forward: plain call counters of func keep counting
        #        (their increments are learnt in the warm-up calls), any other Python-side change during a sweep keeps the solver
        #        eager, the first replay of each sweep is compared with the eager sweep of the same call, and replay must not
        #        be slower than the eager launches it replaces
        #   1: always (every capturable stepper; a failure to capture warns and falls back)      0: never
        gco = str(db.get("pn_graph_capture", "0" if self._ref_defaults else "auto"))
        self._graph_mode = 2 if gco == "auto" else (1 if options.truthy(gco, False) else 0)
        self._graph_status = "eager (-pn_graph_capture 0)" if self._graph_mode == 0 else "eager (warming up)"
        self._auto_veto = None
        # not a PETSc option: in auto mode every N-th replayed call of a captured pair is ALSO run eagerly and compared, as
        # the call that captured it was (0: never).  Catches state of func that the capture guard cannot see.
        self._revalidate_every = int(float(db.get("pn_graph_revalidate", self.GRAPH_REVALIDATE_EVERY)))
        if self._revalidate_every < 0:
            raise PnError("-pn_graph_revalidate must not be negative")
        for key, val in db.items():
            if key.startswith("ts_trajectory") or key in ("ts_monitor", "ts_view") or key.startswith("pn_"):
                continue
            if key == "ts_type" and str(val) in ("beuler", "cn", "theta", "arkimex"):
                continue
            if key.startswith("ts_arkimex") or key.startswith("ts_theta"):
                continue
            if key.startswith("ts_"):
                try:
                    check(self._lib.pn_ts_set_option(self._ts, key.encode(), str(val).encode()))
                except PnError as exc:
                    raise PnError("PETSc option -%s %s is not implemented by pnode_amd (%s). Implemented: -ts_type, -ts_rk_type, "
                                  "-ts_adapt_type none|basic, -ts_rtol, -ts_atol, -ts_max_steps, -ts_max_reject, -ts_adapt_safety, "
                                  "-ts_adapt_reject_safety, -ts_adapt_clip, -ts_adapt_dt_min, -ts_adapt_dt_max, -ts_arkimex_type, "
                                  "-ts_trajectory_*, -ts_monitor, -ts_view, -snes_*, -ksp_*, -log_view; an option that could change "
                                  "the numbers is refused rather than ignored" % (key, val, exc)) from None
        tab = _lib.Tableau()
        check(self._lib.pn_ts_get_tableau(self._ts, ctypes.byref(tab)))
        s = tab.s
        self._s = s
        self._fsal = bool(tab.fsal)
        self._s_eff = s - 1 if self._fsal else s      # stages whose adjoint is not structurally zero
        self._A = [[tab.A[i][j] for j in range(s)] for i in range(s)]
        self._b = [tab.b[j] for j in range(s)]
        self._c = [tab.c[j] for j in range(s)]
        self._e = [tab.bembed[j] - tab.b[j] for j in range(s)]
        self._plans = {}
        self._adaptive = bool(self._lib.pn_ts_is_adaptive(self._ts))
        a, r = ctypes.c_double(), ctypes.c_double()
        check(self._lib.pn_ts_get_tolerances(self._ts, ctypes.byref(a), ctypes.byref(r)))
        self._atol, self._rtol = a.value, r.value
        # budgeted checkpoints with -ts_trajectory_solution_only 0: a checkpoint holds the stage values
        # of its step as well (PETSc's checkpoints do), so reversing a checkpointed step recomputes nothing
        self._budget_stages = self._max_cps > 0 and not self._solution_only
        if self._max_cps > 0:
            self._traj_mode = _lib.PN_TRAJ_BUDGET
        elif self._solution_only:
            self._traj_mode = _lib.PN_TRAJ_SOLUTION
        else:
            self._traj_mode = _lib.PN_TRAJ_ALL

    def _new_trajectory(self, vecs, mode):
        """TSTrajectory of the coming forward sweep: HBM slabs, or files for -ts_trajectory_type basic (every placement:
        all steps, or the bounded set of -ts_trajectory_max_cps_ram)."""
        if self._max_cps_disk > 0 and self._max_cps_ram > 0 and mode == _lib.PN_TRAJ_BUDGET and not self._traj_all_on_disk:
            return _TwoLevelTrajectory(self._lib, self._ops, self.n, vecs, mode, self._max_cps, self._traj_dirname, self._traj_keep,
                                       self._max_cps_ram)
        if self._traj_disk:
            return _DiskTrajectory(self._lib, self._ops, self.n, vecs, mode, self._max_cps, self._traj_dirname, self._traj_keep)
        return _Trajectory(self._lib, self._ops, self.n, vecs, mode, self._max_cps)

    # ------------------------------------------------------------------ helpers
    def _flat(self, t):
        return t.reshape(-1)

    def _buf(self, name):
        b = self._work.get(name)
        if b is None:
            b = self._ops.empty(self._npad)
            self._work[name] = b
        return b

    def _shaped(self, flat):
        return flat[: self.n].view(self.tensor_size)

    def _func_with_grad(self, t, y, which="EX"):
        """f(t, y) recorded by autograd; returns (output, parameter tensors to differentiate
        with respect to).  While a hipGraph is being captured the parameters are replaced by
        fresh detached aliases (same storage): the real parameters' AccumulateGrad nodes live
        on the stream of the enclosing autograd graph and a gradient edge to them would make
        autograd synchronise the capture stream with that stream."""
        fn, params, names = ((self.funcIM, self._paramsI, self._pnamesI) if which == "IM"
                             else (self.funcEX, self._paramsE, self._pnamesE))
        lin = self._lin if (which == "EX" and self._lin is not None and self._lin.active) else None
        capturing = self.device.type == "cuda" and params and torch.cuda.is_current_stream_capturing()
        seen = tuple(p.detach().requires_grad_(True) for p in params) if capturing else params
        if lin is not None:
            lin.begin()                # func's Linear layers hook their outputs: dW / db are accumulated by the engine
        out = None
        try:
            if capturing:
                out = torch.func.functional_call(fn, dict(zip(names, seen)), (t, y))
            else:
                out = fn(t, y)
        finally:
            if lin is not None and out is None:
                lin.abort()
        # the structural check of THIS evaluation (pnode_amd/_lineargrad.py): a handled weight or bias that func also used
        # outside its layer's call leaves the whole evaluation to autograd, as the reference does with every evaluation
        if lin is None or not lin.end(out, [seen[k] for k in lin.handled]):
            return out, seen
        return out, tuple(seen[k] for k in lin.rest)

    def _call_func(self, t, y_flat, tape=None):
        """evalRHSFunction (pa.py:393-412): K = f(t, Y); no copy of the result.  With `tape`
        (a list) the evaluation is recorded by autograd and (input, output) is appended."""
        y = self._shaped(y_flat)
        if tape is not None:
            with torch.enable_grad():
                y = y.detach().requires_grad_(True)
                k, wrt = self._func_with_grad(t, y)
            tape.append((y, k, wrt))
        else:
            k = self.funcEX(t, y)
        if k.dtype != self.tensor_dtype or k.device != self.device or k.numel() != self.n:
            raise ValueError("func must return a tensor with the state's shape, dtype and device")
        if not k.is_contiguous():
            k = k.contiguous()
        if k.untyped_storage().data_ptr() == y_flat.untyped_storage().data_ptr():
            k = k.clone()      # func returned (a view of) its input; the input buffer is recycled
        self.nfe_forward += 1
        return k.detach().reshape(-1)

    def _rk_step(self, t, h, u, K0, unew, stage_dest, want_err, tapes=None, t_first=None):
        """One explicit RK step attempt from the flat state `u` (TSStep_RK's body).

        `t_first`: time at which the first stage derivative is evaluated when it is not handed in
        (see `_first_stage_time`).

        stage_dest(i) -> flat buffer for stage value Y_i, 1 <= i < s (FSAL: Y_{s-1} is `unew`).
        Returns the stage derivatives K (K[s-1] is the FSAL derivative of the next step).
        `tapes` (list of s entries, filled here) receives the autograd tape of each stage.
        """
        ops, s, A, b = self._ops, self._s, self._A, self._b
        if self._native:
            return self._rk_step_native(t, h, u, K0, unew, stage_dest, want_err, tapes, t_first)
        plan = self._stage_plan(h)
        K = [None] * s
        for i in range(s):
            if i == 0:
                y = u
            else:
                y = unew if (self._fsal and i == s - 1) else stage_dest(i)
                idx, coef = plan[i]
                ops.rk_stage(y, u, [K[j] for j in idx], coef)
            if i == 0 and K0 is not None:
                K[0] = K0
            elif tapes is not None:
                rec = []
                K[i] = self._call_func(t + self._c[i] * h, y, rec)
                tapes[i] = rec[0]
            else:
                K[i] = self._call_func(t_first if (i == 0 and t_first is not None) else t + self._c[i] * h, y)
        if want_err:
            idx = [j for j in range(s) if self._e[j] != 0.0 or (not self._fsal and b[j] != 0.0)]
            ops.combine_wrms(None if self._fsal else unew, unew if self._fsal else u, [K[j] for j in idx],
                             [h * b[j] for j in idx], [h * self._e[j] for j in idx], self._atol, self._rtol)
        elif not self._fsal:
            idx, coef = plan[s]
            ops.rk_stage(unew, u, [K[j] for j in idx], coef)
        return K

    # ---- the C++ step loops (include/pnode_amd.h section 3a) and their two callbacks
    def _make_callbacks(self):
        import weakref
        ref = weakref.ref(self)

        def stage_cb(user, i, t):
            o = ref()
            try:
                tens, tapes, K = o._cbs
                if tapes is not None:
                    rec = []
                    k = o._call_func(t, tens[i], rec)
                    tapes[i] = rec[0]
                else:
                    k = o._call_func(t, tens[i])
                K[i] = k                                # keeps the derivative alive; the loop gets its address
                return k.data_ptr()
            except BaseException as exc:                # (an exception must not propagate through the C frame)
                o._cb_exc = exc
                return 0

        def vjp_cb(user, i, t, cot_in_w, scale):
            o = ref()
            try:
                Y, tapes, dlam, t0 = o._rcbs
                if i == 0 and t0 is not None:
                    t = t0                              # first-same-as-last: where the forward sweep evaluated this stage
                w = o.adj_u_flat if not cot_in_w else o._buf("w_a" if cot_in_w == 1 else "w_b")
                gy, gp = o._vjp(t, Y[i], w, tapes[i] if tapes else None, alpha=scale, last=(i == 0))
                if tapes:
                    tapes[i] = None                     # release the stage's activations as soon as they are used
                if gy is not None and gy.data_ptr() == w.data_ptr():
                    gy = gy.clone()                     # f returned its cotangent unchanged (identity-like f)
                dlam[i] = gy
                if o.np > 0 and any(g is not None for g in gp):
                    if o._accum_mode == "stage":
                        o._ops.param_accum(o.adj_p_tensor, scale, gp, o._poff, o._plen)
                    else:
                        o._pend_a.append(scale)
                        o._pend_g.append(gp)
                return 0 if gy is None else gy.data_ptr()
            except BaseException as exc:
                o._cb_exc = exc
                return -1

        self._stage_cb_c = _lib.STAGE_CB(stage_cb)
        self._vjp_cb_c = _lib.VJP_CB(vjp_cb)
        self._ystage = (ctypes.c_void_p * _lib.PN_MAX_STAGES)()
        self._kout = (ctypes.c_void_p * _lib.PN_MAX_STAGES)()
        self._ytens = [None] * _lib.PN_MAX_STAGES
        self._cb_exc = None

    def _raise_from_loop(self, rc):
        exc, self._cb_exc = self._cb_exc, None
        if exc is not None:
            raise exc
        check(rc)

    def _rk_step_native(self, t, h, u, K0, unew, stage_dest, want_err, tapes, t_first):
        ops, s = self._ops, self._s
        if getattr(self, "_stage_cb_c", None) is None:
            self._make_callbacks()
        ys, tens = self._ystage, self._ytens
        tens[0] = u
        for i in range(1, s):
            y = unew if (self._fsal and i == s - 1) else stage_dest(i)
            tens[i] = y
            ys[i] = y.data_ptr()
        K = [None] * s
        K[0] = K0
        self._cbs = (tens, tapes, K)
        work, res = ops.wrms_buffers() if want_err else (None, None)
        rc = self._lib.pn_rk_attempt(ops.stream(), ops.code, self.n, self._ts, ops.vec_ops, t, h, u.data_ptr(), unew.data_ptr(), ys,
                                     None if K0 is None else K0.data_ptr(),
                                     1 if (K0 is None and t_first is not None) else 0, 0.0 if t_first is None else t_first,
                                     self._stage_cb_c, None, 1 if want_err else 0, work, res, self._kout)
        self._cbs = None
        if rc:
            self._raise_from_loop(rc)
        return K

    def _stage_plan(self, h):
        """Per stage i: (indices j of the non-zero a_ij, the coefficients h*a_ij as a C array); entry s: the same for
        the weights b.  Built once per step size (fixed-step sweeps use one; adaptive ones a few dozen)."""
        plan = self._plans.get(h)
        if plan is None:
            if len(self._plans) >= 256:
                self._plans.clear()
            mk = getattr(self._ops, "dbl", list)
            s, A, b = self._s, self._A, self._b
            plan = []
            for i in range(s):
                idx = [j for j in range(i) if A[i][j] != 0.0]
                plan.append((idx, mk([h * A[i][j] for j in idx])))
            idx = [j for j in range(s) if b[j] != 0.0]
            plan.append((idx, mk([h * b[j] for j in idx])))
            self._plans[h] = plan
        return plan

    # ------------------------------------------------------------------ forward (pa.py:777-869)
    def odeint(self, u0, t):
        """Solve du/dt = func(t, u), u(t[0]) = u0; returns the states at the times `t`
        (first dimension), or, when `t` has one element, integrates [0, t[0]] (pa.py:818-820)."""
        return self._odeint(u0, t, self.enable_adjoint)

    def _device_guard(self):
        """The device entry points launch on the calling thread's current HIP device: make it the
        solver's device for the duration of a sweep (a no-op context for the CPU test stand-in)."""
        if self.device is not None and self.device.type == "cuda":
            return self._sweep_context()
        return contextlib.nullcontext()

    @contextlib.contextmanager
    def _sweep_context(self):
        with torch.cuda.device(self.device):
            ops = self._ops
            prev = getattr(ops, "_pinned_stream", None)
            if ops is not None and hasattr(ops, "_pinned_stream"):
                ops._pinned_stream = ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
            try:
                yield
            finally:
                if ops is not None and hasattr(ops, "_pinned_stream"):
                    ops._pinned_stream = prev

    def _odeint(self, u0, t, save):
        with self._device_guard():
            if not self._trace:
                return self._odeint_impl(u0, t, save)
            torch.cuda.nvtx.range_push("pnode_amd.forward_sweep")      # roctx range on ROCm
            try:
                return self._odeint_impl(u0, t, save)
            finally:
                torch.cuda.nvtx.range_pop()

    def _odeint_impl(self, u0, t, save):
        if self._ops is None:
            raise RuntimeError("setupTS must be called before odeint")
        if u0.size() != self.tensor_size or u0.dtype != self.tensor_dtype or u0.device != self.device:
            raise ValueError("u0 does not match the tensor given to setupTS (shape, dtype, device)")
        if self._theta is not None:
            return self._theta.odeint(u0, t, save)
        lib, ops, ts = self._lib, self._ops, self._ts
        self.sol_times = t.detach().cpu().to(dtype=torch.float64)
        T = int(t.shape[0])
        times = self.sol_times.tolist()
        dt0 = float(self.step_size[0] if isinstance(self.step_size, list) else self.step_size)
        check(lib.pn_ts_begin(ts, 0.0, dt0, T, (ctypes.c_double * T)(*times)))
        self._span_begin(T)
        solution = ops.empty((T,) + tuple(self.tensor_size))
        sol_flat = solution.view(T, -1)
        u0f = u0.detach().contiguous().reshape(-1)

        # where the state at the start of step k lives
        self._tmode = self._pick_traj_mode(self._s_eff) if save else self._traj_mode
        if save:
            vecs = self._s_eff if (self._tmode == _lib.PN_TRAJ_ALL or self._budget_stages) else 1
            self._traj = self._new_trajectory(vecs, self._tmode)
            traj = self._traj
            if self._tmode == _lib.PN_TRAJ_BUDGET and not self._adaptive and not isinstance(self.step_size, list):
                total = lib.pn_ts_count_fixed_steps(ts)         # fixed step: the sweep length is known
                if total > 0:
                    check(lib.pn_traj_set_total(traj.handle, total))
        else:
            traj = self._traj = None
        store_stages = save and self._tmode == _lib.PN_TRAJ_ALL
        keep_tape = store_stages and self._retain_graph != 0
        tape_budget = None
        if keep_tape and self._retain_graph == 2:
            tape_budget = self._tape_budget()
            keep_tape = tape_budget is not None and tape_budget > 0
        self._tapes = {} if keep_tape else None
        tape_fsal = None
        pingpong = [self._buf("u_a"), self._buf("u_b")]
        pp = 0

        cur_slot = -1                # trajectory slot `cur` lives in, -1 when it is a ping-pong buffer

        def state_home(step):
            nonlocal pp, home_slot
            if traj is not None:
                slot = traj.fwd_slot(step)
                if slot >= 0:
                    home_slot = slot
                    traj.stage_step.pop(slot, None)  # a recycled slot no longer holds the old step's stages
                    return traj.claim(slot)         # (vecs, npad)
            home_slot = -1
            pp ^= 1
            return pingpong[pp].view(1, -1)

        home_slot = -1
        cur = state_home(0)
        cur_slot = home_slot
        ops.copy(cur[0], u0f)
        if T > 1:
            ops.copy(sol_flat[0], u0f)
        K_fsal = None
        tt, hh = ctypes.c_double(), ctypes.c_double()
        acc, hit, done = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(0)
        finished = not (times[-1] > (0.0 if T == 1 else times[0]))
        if self._monitor:
            print("%d TS dt %g time %g" % (0, dt0, 0.0 if T == 1 else times[0]))
        while not finished:
            step = lib.pn_ts_steps(ts)
            nxt = state_home(step + 1)
            nxt_slot = home_slot
            K0 = K_fsal
            tape0 = tape_fsal
            while True:
                check(lib.pn_ts_attempt(ts, ctypes.byref(tt), ctypes.byref(hh)))
                tn, h = tt.value, hh.value
                if store_stages or (self._budget_stages and cur_slot >= 0):
                    dest = lambda i, c=cur: c[i]
                else:
                    dest = lambda i: self._buf("y_scratch")
                tapes = [tape0] + [None] * (self._s - 1) if keep_tape else None
                K = self._rk_step(tn, h, cur[0], K0, nxt[0], dest, self._adaptive, tapes)
                if keep_tape:
                    tape0 = tapes[0]
                enorm = self._global_enorm(ops.read_enorm()) if self._adaptive else -1.0
                check(lib.pn_ts_judge(ts, enorm, ctypes.byref(acc), ctypes.byref(hit), ctypes.byref(done)))
                if acc.value:
                    break
                K0 = K[0]            # f(t_n, u_n) does not depend on h
            K_fsal = K[self._s - 1] if self._fsal else None
            if keep_tape:
                self._tapes[step] = tapes[: self._s_eff]
                tape_fsal = tapes[self._s - 1] if self._fsal else None
                if tape_budget is not None and tape_budget != float("inf"):
                    if step == 0:                 # one measurement: what a step's tapes (and its slot) take
                        per_step = max(_mem_now(self.device)[0] - self._tape_mem0, 1)
                        tape_steps = int(tape_budget // per_step) - 1
                    if step + 1 >= tape_steps:
                        keep_tape, tape_fsal = False, None       # later steps re-evaluate f in the reverse sweep
                        self._tape_all_fit = False
            if self._budget_stages and cur_slot >= 0 and save:
                traj.stage_step[cur_slot] = step
            if traj is not None and cur_slot >= 0:
                traj.seal(cur_slot)          # the step's checkpoint is complete (a no-op on the HBM tier)
            cur = nxt
            cur_slot = nxt_slot
            stepno = step + 1
            tnew = lib.pn_ts_time(ts)
            self._span_post_step(T, times, hit.value, done.value, stepno, tnew, cur[0], sol_flat)
            if self._monitor:
                print("%d TS dt %g time %g" % (stepno, h, tnew))
            finished = bool(done.value)
        self._nsteps = lib.pn_ts_steps(ts)
        if self._view:
            self._ts_view()
        if T == 1:
            ops.copy(sol_flat[0], cur[0])
        else:
            self._span_end(T)
        return solution

    # ------------------------------------------------------------------ time span (pa.py:518-532, 822-868)
    def _span_begin(self, T):
        self.cur_sol_steps = [0] * T      # steps taken from the previous output time to this one
        self.cur_sol_index = 1
        self._span_hits = 1               # output times whose solution has been kept (t[0] is u0)
        self._span_delta = 1e-5 if self.tensor_dtype == torch.double else 1e-3

    def _span_post_step(self, T, times, hit, done, stepno, tnew, cur, sol_flat):
        """What happens after an accepted step of a multi-output solve.

        * The output itself: the reference reads PETSc's ``getTimeSpanSolutions()`` (pa.py:845), i.e.
          the state of exactly the step that landed on t[i].  Here: ``pn_ts_judge`` reports that step
          (`hit` = i) and the state is copied out then.
        * ``tspanPostStep`` (pa.py:518-532): a ``step_size`` list sets the next step; the steps of
          each output interval are counted for the reverse sweep.  The reference advances its
          interval counter when ``|t - t[i]| < 1e-5`` (fp64) / ``1e-3`` (fp32), which is one step
          early whenever the step is shorter than that window: its backward pass then injects
          dL/dy(t[i]) one step off and never reverses the sweep's first step.  The default here
          counts with the exact hit (the discrete adjoint of what the forward sweep computed);
          ``-pn_span_count reference`` counts as the reference does (identical whenever every step
          is longer than the window)."""
        if T <= 1:
            return
        if hit >= 0:
            self._ops.copy(sol_flat[hit], cur)
            self._span_hits += 1
        if self.cur_sol_index < T:
            if isinstance(self.step_size, list) and stepno < len(self.step_size) and not done:
                check(self._lib.pn_ts_override_next_dt(self._ts, float(self.step_size[stepno])))
            self.cur_sol_steps[self.cur_sol_index] += 1
            if self._span_count_reference:
                if abs(tnew - times[self.cur_sol_index]) < self._span_delta:
                    self.cur_sol_index += 1
            elif hit >= 0:
                self.cur_sol_index = hit + 1

    def _span_end(self, T):
        if self.cur_sol_index != T or self._span_hits != T:
            raise Exception("TSSolve fails to step on all the specified points")

    def _pick_traj_mode(self, vecs_all):
        """Trajectory mode of the solve that was just begun (pn_ts_begin done).  When
        -ts_trajectory_solution_only is not given PETSc keeps the states only and recomputes a step's
        stages when it is reversed.  Every mode replays the same arithmetic -- gradients are identical bit
        for bit -- so on a 288 GB part the stage values are kept as well whenever the step count is known
        (fixed step) and the whole trajectory fits in a quarter of the HBM that is free right now: the
        reverse sweep then recomputes nothing.  Give the option (0 or 1) to decide yourself."""
        mode = self._traj_mode
        if (mode != _lib.PN_TRAJ_SOLUTION or not self._solution_only_auto or self.device.type != "cuda"
                or isinstance(self.step_size, list)):
            return mode
        if torch.cuda.is_current_stream_capturing():        # no driver query while capturing: as the last eager solve
            return getattr(self, "_tmode_auto", mode)
        total = self._lib.pn_ts_count_fixed_steps(self._ts)
        self._tmode_auto = mode
        if total > 0:
            esize = 4 if self.tensor_dtype == torch.float32 else 8
            need = (total + 1) * vecs_all * self._npad * esize
            free, _ = torch.cuda.mem_get_info(self.device)
            allocated, reserved = _mem_now(self.device)
            if need <= 0.25 * (free + max(reserved - allocated, 0)):
                self._tmode_auto = _lib.PN_TRAJ_ALL
        return self._tmode_auto

    def _tape_budget(self):
        """Bytes the retained tapes of this sweep may take in `auto` mode: half of the HBM that is free now
        (driver-free + cached-but-unused blocks of PyTorch's allocator); None on the CPU test stand-in.
        While a hipGraph is being captured no driver query is made: the sweep keeps what the eager
        warm-up call before it kept."""
        if self.device.type != "cuda":
            return None
        if torch.cuda.is_current_stream_capturing():
            return float("inf") if getattr(self, "_tape_all_fit", False) else None
        self._tape_mem0, reserved = _mem_now(self.device)
        free, _ = torch.cuda.mem_get_info(self.device)
        self._tape_all_fit = True
        return 0.5 * (free + max(reserved - self._tape_mem0, 0))

    # ------------------------------------------------------------------ reverse (pa.py:871-890)
    def _step_info(self, k):
        tt, hh = ctypes.c_double(), ctypes.c_double()
        check(self._lib.pn_ts_step_log(self._ts, k, ctypes.byref(tt), ctypes.byref(hh)))
        return tt.value, hh.value

    def _ts_view(self):
        """-ts_view: the solver's settings and counters after a solve (PETSc prints its TS object there)."""
        modes = {_lib.PN_TRAJ_ALL: "every step, with stage values", _lib.PN_TRAJ_SOLUTION: "every step, solution only",
                 _lib.PN_TRAJ_BUDGET: "at most %d checkpoints (%s)" % (self._max_cps, "with stage values" if self._budget_stages else "solution only")}
        tab = _lib.Tableau()
        check(self._lib.pn_ts_get_tableau(self._ts, ctypes.byref(tab)))
        print("TS Object (pnode_amd): type rk, order %d, %d stages%s%s"
              % (tab.order, tab.s, ", first same as last" if tab.fsal else "", ", embedded error estimate" if tab.has_embed else ""))
        print("  adapt: %s%s" % ("basic, atol %g rtol %g" % (self._atol, self._rtol) if self._adaptive else "none (fixed steps)",
                                 "; final time matched exactly (MATCHSTEP)"))
        print("  state: %s %s on %s;  trainable parameters: %d" % (tuple(self.tensor_size), str(self.tensor_dtype).replace("torch.", ""), self.device, self.np))
        print("  launches: %s;  step loop: %s" % (self._graph_status, "C++ (pn_rk_attempt / pn_rk_adjoint_step)" if self._native else "Python"))
        print("  total number of time steps=%d, rejected=%d;  trajectory: %s"
              % (self._nsteps, self._lib.pn_ts_rejections(self._ts), modes[self._tmode] if self._traj is not None else "not saved"))

    def _first_stage_time(self, k):
        """Time argument of f for the first stage of step k when it is RE-computed from a checkpoint.
        In the original sweep of a first-same-as-last tableau that derivative was the previous step's
        last stage, evaluated at t_{k-1} + c_{s-1} h_{k-1}; that is not t_k to the last bit (5dp's
        c_{s-1} is the row sum 0.9999999999999998; matched output times are set exactly), and a
        time-dependent f would see it.  Same expression here, so that every checkpoint mode
        reproduces the store-all sweep bit for bit."""
        if self._fsal and k > 0 and not self._ref_defaults:
            tp, hp = self._step_info(k - 1)
            return tp + self._c[self._s - 1] * hp
        # (-pn_reference_defaults: PETSc's TSTrajectory restarts the stepper at a restored checkpoint, so the first stage is
        # re-evaluated -- and its Jacobian taken, TSAdjointStep_RK -- at t_k; for a time-dependent f under a first-same-as-last
        # tableau that is the forward sweep's derivative only up to the last bits of the time argument, as with the reference)
        return None

    def _stages_of(self, step):
        """Stage values Y_0..Y_{s_eff-1} of `step` as flat tensors: read from the store-all
        trajectory, or recomputed from the nearest kept state (TSTrajectoryGet)."""
        traj, ops = self._traj, self._ops
        s_eff = self._s_eff
        if self._tmode == _lib.PN_TRAJ_ALL:
            fs, fl, _ = traj.rev_plan(step)
            v = traj.view(fl)
            return [v[i] for i in range(s_eff)]
        fs, fl, stores = traj.rev_plan(step)
        keep = self._budget_stages
        if keep and fs == step and traj.stage_step.get(fl) == step:
            v = traj.view(fl)              # the checkpoint of this very step holds its stage values
            return [v[i] for i in range(s_eff)]
        slot_view = traj.view(fl)
        cur, cur_slot = slot_view[0], fl
        K_fsal = None
        k = fs
        pp = 0
        while k < step:                   # re-advance k -> k+1, keeping what the plan asks for
            tn, h = self._step_info(k)
            if (k + 1) in stores:
                nxt_slot = stores[k + 1]
                nxt_view = traj.claim(nxt_slot)
                nxt = nxt_view[0]
                traj.stage_step.pop(nxt_slot, None)
            else:
                pp ^= 1
                nxt_slot, nxt_view = -1, None
                nxt = self._buf("r_a" if pp else "r_b")
            if keep and cur_slot >= 0:
                dest = lambda i, c=slot_view: c[i]          # stage values of step k go behind its checkpoint
            else:
                dest = lambda i: self._buf("y_scratch")
            K = self._rk_step(tn, h, cur, K_fsal, nxt, dest, False,
                              t_first=self._first_stage_time(k) if K_fsal is None else None)
            if keep and cur_slot >= 0:
                traj.stage_step[cur_slot] = k
                traj.seal(cur_slot)                  # (disk tier) the checkpoint now carries its stage values
            if nxt_slot >= 0 and not keep:
                traj.seal(nxt_slot)                  # (disk tier) a new state-only checkpoint is complete
            K_fsal = K[self._s - 1] if self._fsal else None
            cur, cur_slot, slot_view = nxt, nxt_slot, nxt_view
            k += 1
        # stage values of `step` itself (its own derivatives K_0..K_{s_eff-2} are needed)
        tn, h = self._step_info(step)
        Y = [cur]
        K = [K_fsal]
        # The derivatives K_0..K_{s_eff-2} evaluated here are evaluations of f at exactly the points the stage VJPs of this
        # step differentiate f at: unless tapes are switched off (-pn_trajectory_retain_graph 0, -pn_reference_defaults) they
        # are recorded by autograd and the VJPs of those stages run their backward half only -- (s_eff - 1) evaluations of f
        # fewer per reversed step in every mode that recomputes stage values (solution-only, checkpoint budgets); same bits.
        rt = [None] * self._s if self._retain_graph != 0 else None
        self._rtapes = rt
        for i in range(1, s_eff):
            if K[i - 1] is None:
                t_eval = self._first_stage_time(step) if i == 1 else None
                tt = tn + self._c[i - 1] * h if t_eval is None else t_eval
                if rt is not None:
                    rec = []
                    K[i - 1] = self._call_func(tt, Y[i - 1], rec)
                    rt[i - 1] = rec[0]
                else:
                    K[i - 1] = self._call_func(tt, Y[i - 1])
            y = self._buf("ys%d" % i)
            idx = [j for j in range(i) if self._A[i][j] != 0.0]
            ops.rk_stage(y, cur, [K[j] for j in idx], [h * self._A[i][j] for j in idx])
            Y.append(y)
            K.append(None)
        if self._ref_defaults:
            # -pn_reference_defaults: PETSc's TSTrajectory re-runs the WHOLE step (TSStep) to get the stage values back,
            # i.e. it also evaluates the stage derivatives nothing in the reverse sweep reads.  Evaluated here too (and
            # dropped), so that a func that counts its calls sees s evaluations per recomputed step.
            if K[s_eff - 1] is None:
                K[s_eff - 1] = self._call_func(tn + self._c[s_eff - 1] * h, Y[s_eff - 1])
            if self._fsal:
                i = self._s - 1
                y = self._buf("y_scratch")
                idx = [j for j in range(i) if self._A[i][j] != 0.0]
                ops.rk_stage(y, cur, [K[j] for j in idx], [h * self._A[i][j] for j in idx])
                self._call_func(tn + self._c[i] * h, y)
        return Y

    def _vjp(self, t, y_flat, w_flat, tape=None, which="EX", alpha=None, last=False):
        """RHSJacShell.multTranspose + RHSJacPShell.multTranspose (pa.py:52-82, 341-363): one
        forward of f with grad and one backward with the cotangent `w`; returns
        (J^T w as a flat tensor or None, list of parameter cotangents over ALL parameters of that f).  With a `tape`
        (input, output) recorded in the forward sweep only the backward runs.  `alpha`: the scale the caller will give the
        parameter cotangents when it adds them to mu -- the explicit RK path passes it so that the sensitivities of func's
        nn.Linear layers can be accumulated during the backward pass itself (pnode_amd/_lineargrad.py); those entries of
        the returned list are then None.  `last`: this is the last stage VJP of a reversed step (lambda is rewritten next)."""
        lin = self._lin if (which == "EX" and self._lin is not None) else None
        all_params = self._paramsI if which == "IM" else self._paramsE
        if tape is not None:
            y, out, wrt = tape
        else:
            self.nfe_backward += 1
        with torch.enable_grad() if tape is None else contextlib.nullcontext():
            if tape is None:
                y = self._shaped(y_flat).detach().requires_grad_(True)
                out, wrt = self._func_with_grad(t, y, which)
            cot = self._shaped(w_flat).view(out.shape)
            hooked = lin is not None and len(wrt) != len(all_params)      # this evaluation left the Linear layers to the hooks
            if hooked and not lin.disabled and alpha is not None:
                capturing = self.device.type == "cuda" and torch.cuda.is_current_stream_capturing()
                if not lin.checked and not capturing:
                    ok, worst = lin.self_check(self, out, y, all_params, cot)
                    if not ok:
                        lin.disabled = True
                        lin.why = "its result differed from autograd's at the self-check (relative %.1e)" % worst
                        lin.remove_hooks_only()
                        warnings.warn("pnode_amd: the engine-side accumulation of the nn.Linear layers' parameter sensitivities is "
                                      "switched off for this solver: its result differs from autograd's (relative %.1e) -- a "
                                      "weight or bias of such a layer is also used somewhere else in func.  Results are autograd's; "
                                      "-pn_linear_param_grads 0 silences this." % worst, RuntimeWarning)
                if not lin.disabled:
                    lin.alpha, lin.target = float(alpha), self.adj_p_tensor
                    lin.cot_storage = w_flat.untyped_storage().data_ptr()
                    try:
                        grads = torch.autograd.grad(out, (y,) + wrt, cot, allow_unused=True)
                    finally:
                        lin.alpha = None
                    grads = (grads[0],) + tuple(lin.expand(grads[1:], len(all_params)))
                    hooked = None
            if hooked:
                # evaluated with the hooks on, differentiated without them (the self-check failed, or a caller that adds the
                # parameter cotangents itself): autograd differentiates with respect to every parameter
                if self.device.type == "cuda" and torch.cuda.is_current_stream_capturing():
                    raise PnError("pnode_amd: a stage evaluation recorded for the engine-side Linear accumulation cannot be "
                                  "differentiated by autograd alone inside a hipGraph capture")
                lin.muted = True
                try:
                    grads = torch.autograd.grad(out, (y,) + tuple(all_params), cot, allow_unused=True)
                finally:
                    lin.muted = False
            elif hooked is False:
                grads = torch.autograd.grad(out, (y,) + wrt, cot, allow_unused=True)
            if lin is not None:
                # the stage's queued (cotangent, input) pairs: one grouped launch of the fused kernel, beside the next stage; the
                # launches of earlier stages are waited for (every stage VJP, also one autograd did alone: the buffers turn)
                lin.flush(self, lam=self.adj_u_flat if last else None)
        gy = grads[0]
        if gy is not None:
            if gy.dtype != self.tensor_dtype:
                gy = gy.to(self.tensor_dtype)
            gy = gy.contiguous().reshape(-1)
        gp = []
        dt = self.tensor_dtype
        # Deferred accumulation (-pn_param_accum batch|step) reads these gradients launches later, after the
        # cotangent buffer (w_a, or lambda itself for a folded stage) has been rewritten in place.  Autograd hands
        # the cotangent, or ANY view of it, straight through for f = ... + p, cat([z[:2] + b1, ...]), stack((.. + p0, ..)):
        # every gradient that shares the cotangent's storage is copied, whatever its size.
        wst = None if self._accum_mode == "stage" else w_flat.untyped_storage().data_ptr()
        for g in grads[1:]:
            if g is not None:
                if g.dtype != dt or not g.is_contiguous():
                    g = g.to(dt).contiguous()
                if wst is not None and g.untyped_storage().data_ptr() == wst:
                    g = g.clone()
            gp.append(g)
        return gy, gp

    def _adjoint_steps(self, nsteps, forcing):
        """TSAdjointSolve over `nsteps` steps, newest first (TSAdjointStep_RK per step), then
        add `forcing` (dL/dy at the span point reached; pa.py:938) fused into the last update.

        Per step [t_n, t_n+H] with stage values Y_i, incoming lambda and mu:
            for i = s-1 .. 0:   w_i = H*(b_i*lambda + sum_{j>i} a_ji*dlam_j)
                                (dlam_i, dmu_i) = VJP of f at Y_i with cotangent w_i
            mu     <- mu + sum_i dmu_i      (stages added in the order s-1..0: one multi-tensor launch per
                                             stage, or per time step with -pn_param_accum step; same rounding)
            lambda <- lambda + sum_i dlam_i
        (the scale PETSc applies after MatMultTranspose is applied to the cotangent instead).
        A stage whose cotangent is a pure multiple of lambda -- the last non-trivial stage of
        every tableau -- is differentiated with lambda itself and the scalar is folded into
        the coefficients of everything that consumes its result: no kernel, no extra vector."""
        if self._theta is not None:
            return self._theta.adjoint_steps(nsteps, forcing)
        ops, s_eff, A, b = self._ops, self._s_eff, self._A, self._b
        lam = self.adj_u_flat
        if nsteps == 0 and forcing is not None:
            ops.adj_accum(lam, lam, [], [], forcing)
        # two cotangent buffers in turn while the weight-sensitivity products of a stage run beside the next stage on a second
        # stream (pnode_amd/_lineargrad.py): the product of stage i reads stage i's cotangent while stage i-1's is written
        two_w = self._lin is not None and self._lin.side_on
        for r in range(nsteps):
            step = self._rev_next
            tn, H = self._step_info(step)
            if self._lin is not None and self._tmode != _lib.PN_TRAJ_ALL:
                self._lin.join()             # a product still running may read stage values the recomputation below rewrites
            Y = self._stages_of(step)
            tapes = self._tapes.pop(step, None) if self._tapes else None
            if tapes is None and self._rtapes is not None:
                tapes = self._rtapes             # recorded while the stage values were recomputed (_stages_of)
            self._rtapes = None
            dlam = [None] * self._s          # raw VJP results
            if self._native:
                if getattr(self, "_vjp_cb_c", None) is None:
                    self._make_callbacks()
                self._rcbs = (Y, tapes, dlam, self._first_stage_time(step))
                fo = forcing if r == nsteps - 1 else None
                rc = self._lib.pn_rk_adjoint_step(ops.stream(), ops.code, self.n, self._ts, ops.vec_ops, tn, H, lam.data_ptr(),
                                                  self._buf("w_a").data_ptr(), self._buf("w_b").data_ptr() if two_w else None,
                                                  self._vjp_cb_c, None,
                                                  None if fo is None else fo.data_ptr())
                self._rcbs = None
                if rc:
                    self._raise_from_loop(rc)
                if self._pend_g and (self._accum_mode == "step" or len(self._pend_g) + s_eff > self._accum_cap):
                    self._flush_param_accum()
                elif self._pend_bias and self._accum_mode == "step":
                    self._flush_bias_accum()
                self._traj.rev_done(step)
                self._rev_next = step - 1
                continue
            scale = [1.0] * self._s          # true dlam_i = scale[i] * dlam[i]
            pend_a, pend_g = self._pend_a, self._pend_g      # parameter gradients waiting to be added to mu
            nw = 0
            for i in range(s_eff - 1, -1, -1):
                js = [j for j in range(i + 1, s_eff) if A[j][i] != 0.0 and dlam[j] is not None]
                if b[i] == 0.0 and not js:
                    continue                   # structurally zero cotangent
                if not js:
                    w, scale[i] = lam, H * b[i]
                else:
                    w = self._buf("w_b" if (two_w and nw % 2) else "w_a")
                    nw += 1
                    ops.adj_theta(w, lam if b[i] != 0.0 else None, H * b[i],
                                  [dlam[j] for j in js], [H * A[j][i] * scale[j] for j in js])
                # (stage 0 of a first-same-as-last tableau was evaluated at the previous step's last stage time, which is
                # t_n only to the last bit: the VJP differentiates f THERE, with and without a tape -- the exact discrete
                # adjoint, the same bits in every checkpoint mode for a time-dependent f; PETSc passes t_n)
                t0 = self._first_stage_time(step) if i == 0 else None
                gy, gp = self._vjp(tn + self._c[i] * H if t0 is None else t0, Y[i], w, tapes[i] if tapes else None, alpha=scale[i], last=(i == 0))
                if tapes:
                    tapes[i] = None            # release the stage's activations as soon as they are used
                if gy is not None and gy.data_ptr() == w.data_ptr():
                    gy = gy.clone()            # f returned its cotangent unchanged (identity-like f)
                dlam[i] = gy
                if self.np > 0 and any(g is not None for g in gp):
                    if self._accum_mode == "stage":
                        ops.param_accum(self.adj_p_tensor, scale[i], gp, self._poff, self._plen)
                    else:
                        pend_a.append(scale[i])
                        pend_g.append(gp)
            if pend_g and (self._accum_mode == "step" or len(pend_g) + s_eff > self._accum_cap):
                self._flush_param_accum()      # mu += sum_j scale_j * dmu_j, oldest first: one launch
            elif self._pend_bias and self._accum_mode == "step":
                self._flush_bias_accum()
            idx = [i for i in range(s_eff) if dlam[i] is not None]
            ops.adj_accum(lam, lam, [dlam[i] for i in idx], [scale[i] for i in idx],
                          forcing if r == nsteps - 1 else None)
            self._traj.rev_done(step)
            self._rev_next = step - 1

    def _add_param_grads(self, alpha, gp, first=0, stable=True, cotangent=None):
        """mu[parameters first .. first+len(gp)) += alpha * gp for the implicit / IMEX steppers: one launch per call with
        -pn_param_accum stage, else queued for the batched launch of _flush_param_accum (same order, same rounding).
        `stable` False: the gradients sit in buffers that are rewritten before a deferred launch would read them (the
        outputs of a replayed graph): what is queued is added first, then these, at once.  `cotangent`: the buffer the
        gradients were computed FROM when they did not come through _vjp -- a gradient that is a view of it is copied."""
        if not any(g is not None for g in gp):
            return
        n_all = len(self._poff)
        full = first == 0 and len(gp) == n_all
        if self._accum_mode == "stage" or not stable:
            self._flush_param_accum()
            if full:
                off, ln = self._poff, self._plen
            elif first == 0:
                off, ln = self._poffI, self._plenI
            else:
                off, ln = self._poffE, self._plenE
            self._ops.param_accum(self.adj_p_tensor, alpha, list(gp), off, ln)
            return
        if cotangent is not None:
            st = cotangent.untyped_storage().data_ptr()
            gp = [g.clone() if (g is not None and g.untyped_storage().data_ptr() == st) else g for g in gp]
        self._pend_a.append(alpha)
        self._pend_g.append(list(gp) if full else [None] * first + list(gp) + [None] * (n_all - first - len(gp)))
        if len(self._pend_g) >= self._accum_cap:
            self._flush_param_accum()

    def _colsum_accum(self, g2, mu_slice, alpha):
        """mu_slice += alpha * column sums of g2 (rows x cols): the sensitivity of a bias.  Queued like the parameter
        cotangents of autograd (-pn_param_accum batch|step: the cotangent tensors stay alive, at most 1 GiB of them, and up to
        32 are summed by ONE pn_colsum_accum_multi pass; stage: at once) -- same bits whatever the grouping."""
        g2 = g2.contiguous()
        self._pend_bias.append((g2, mu_slice, float(alpha)))
        self._pend_bias_bytes += g2.numel() * g2.element_size()
        if self._accum_mode == "stage" or len(self._pend_bias) >= 32 or self._pend_bias_bytes >= (1 << 30):
            self._flush_bias_accum()

    def _flush_bias_accum(self):
        if self._pend_bias:
            fn = getattr(self._ops, "colsum_accum_multi", None)
            if fn is not None and self._pend_bias[0][0].device.type == "cuda":
                fn(self._pend_bias)
            else:                                        # the CPU test stand-in: same order, double sums
                for g2, mu_slice, alpha in self._pend_bias:
                    mu_slice.add_(g2.double().sum(0).to(mu_slice.dtype), alpha=alpha)
            self._pend_bias = []
            self._pend_bias_bytes = 0

    @property
    def linear_param_grads(self):
        """How the parameter sensitivities of func's nn.Linear layers are formed: "engine (N parameters)" or "autograd (why)"."""
        lin = self._lin
        if lin is None:
            return "autograd (no eligible nn.Linear layer, a theta stepper, or -pn_linear_param_grads 0)"
        if lin.disabled:
            return "autograd (%s)" % lin.why
        note = "; fused dW + db MFMA kernel on %d layers" % len(lin.partials) if lin.partials else ""
        if lin.n_autograd:
            # the structural check (LinearParamGrads.end): evaluations in which a handled parameter was also used outside its layer
            note += "; %d of %d recorded evaluations of func left to autograd (a handled weight or bias is also used outside its layer there)" \
                    % (lin.n_autograd, lin.n_autograd + lin.n_clean)
        return "engine (%d of %d parameter tensors%s)" % (len(lin.handled), len(self._paramsE), note)

    def _setup_linear_grads(self):
        """(Re)install the engine-side accumulation of func's nn.Linear layers (pnode_amd/_lineargrad.py): explicit RK path
        only; -pn_linear_param_grads auto|gemm|0 (not a PETSc option)."""
        opt = str(options.get_all().get("pn_linear_param_grads", "auto"))
        gemm = opt == "gemm"             # the library GEMM + pn_colsum_accum_multi for every layer (no fused MFMA kernel)
        on = opt in ("auto", "gemm") or options.truthy(opt, False)
        # explicit RK (func), and ARKIMEX's explicitly treated func2: the only grad-enabled evaluations of that function are the
        # solver's own taped stage evaluations and stage VJPs.  Not the theta methods: their Newton-Krylov solves differentiate
        # func in ways of their own (double VJPs, captured linearisations)
        side = str(options.get_all().get("pn_linear_side_stream", "0"))
        # -pn_linear_side_stream 1 | same-priority (default 0): the products on a second stream beside the next stage's backward
        # pass -- the explicit RK sweep only (its cotangent buffers are doubled for it); ARKIMEX's stage vectors are rewritten on a
        # schedule of their own.  Measured at BASELINE's target configuration (profiles/r06_side_stream.txt): +1.5 % time-steps/s,
        # the same bits; the dX GEMMs of the next stage take 36 us beside the product against 19.4 alone -- the two share the
        # matrix pipes -- and every kernel's own duration stops being a statement about that kernel, so it is not the default.
        side_on = (side == "same-priority" or options.truthy(side, False)) and self._stepper_kind is None and self.device.type == "cuda"
        sig = (id(self.funcEX), on, self._stepper_kind in (None, "imex"), tuple(id(p) for p in self._paramsE), gemm, side_on)
        if sig == self._lin_sig:
            return
        self._lin_sig = sig
        if self._lin is not None:
            self._lin.remove()
            self._lin = None
        if on and sig[2] and self._paramsE:
            from ._lineargrad import LinearParamGrads
            lin = LinearParamGrads(self)
            lin.fused = not gemm
            lin.side_on = side_on
            lin.side_priority = side != "same-priority"
            if lin.install(self.funcEX, self._paramsE, self._poffE if self._stepper_kind == "imex" else self._poff):
                self._lin = lin

    def _flush_param_accum(self):
        self._flush_bias_accum()
        if self._pend_g:
            self._ops.param_accum_multi(self.adj_p_tensor, self._pend_a, self._pend_g, self._poff, self._plen)
            del self._pend_a[:], self._pend_g[:]

    def petsc_adjointsolve(self, t, i=1):
        """Reverse one output interval (pa.py:871-890): all steps when `t` has one element,
        else the ``cur_sol_steps[i]`` steps that led to output time i."""
        if t.shape[0] == 1:
            self._adjoint_steps(self._nsteps, None)
        else:
            self._adjoint_steps(self.cur_sol_steps[i], None)
        self._flush_param_accum()
        self._finish_linear_accum()
        return self._shaped(self.adj_u_flat), self.adj_p_tensor

    def _begin_adjoint(self, seed):
        if self._traj is None:
            raise RuntimeError("adjoint requested but no trajectory was saved "
                               "(setupTS(enable_adjoint=True) and a differentiable input are required)")
        if self.adj_u_tensor is None:
            self.adj_u_tensor = self._ops.empty(self._npad)
        if self.adj_p_tensor is None or self.adj_p_tensor.numel() != self.np:
            self.adj_p_tensor = self._ops.empty(max(self.np, 1))[: self.np]
        self.adj_u_flat = self.adj_u_tensor
        self._ops.copy(self.adj_u_flat, seed)
        self.adj_p_tensor.zero_()
        self._traj.begin_reverse()
        self._rev_next = self._nsteps - 1
        self._pend_a, self._pend_g = [], []
        self._pend_bias, self._pend_bias_bytes = [], 0
        if self._lin is not None:
            self._lin.reset()              # (partial sums a sweep that raised may have left behind)
        # pending stage results are kept alive until they are added: bound them to 1 GiB
        esize = 4 if self.tensor_dtype == torch.float32 else 8
        self._accum_cap = max(1, min(self._accum_sources, (1 << 30) // max(self.np * esize, 1)))

    def _reverse_sweep(self, g, T):
        """The body of OdeintAdjointMethod.backward (pa.py:924-944) on the (T, n) cotangent."""
        with self._device_guard():
            if self._trace:
                torch.cuda.nvtx.range_push("pnode_amd.reverse_sweep")
                try:
                    return self._reverse_sweep_impl(g, T)
                finally:
                    torch.cuda.nvtx.range_pop()
            return self._reverse_sweep_impl(g, T)

    def _reverse_sweep_impl(self, g, T):
        self._begin_adjoint(g[T - 1])
        if T == 1:
            self._adjoint_steps(self._nsteps, None)
        for i in range(T - 1, 0, -1):
            self._adjoint_steps(self.cur_sol_steps[i], g[i - 1])
        self._flush_param_accum()
        self._finish_linear_accum()

    def _finish_linear_accum(self):
        """End of a reverse sweep: the partial sums of the fused Linear-sensitivity kernel go into mu (pn_linear_wgrad_finish)."""
        if self._lin is not None:
            self._lin.finish(self, self.adj_p_tensor)

    # ------------------------------------------------------------------ autograd entry (pa.py:892-900)
    def odeint_adjoint(self, y0, t):
        if not isinstance(self.funcIM, nn.Module):
            raise ValueError("func is required to be an instance of nn.Module.")
        # inside Function.forward grad mode is always off, so note here whether a backward can follow
        self._grad_mode = torch.is_grad_enabled()
        return OdeintAdjointMethod.apply(y0, t, self.flat_params, self, *self._params)


class OdeintAdjointMethod(torch.autograd.Function):
    """pa.py:903-947.  Forward: solve under no_grad.  Backward: seed lambda with dL/dy(t_T),
    reverse the output intervals newest first, adding dL/dy(t_{i-1}) after each one."""

    @staticmethod
    def forward(ctx, y0, t, flat_params, ode, *args):
        ctx.ode = ode
        need = ode.enable_adjoint and ode._grad_mode and (ctx.needs_input_grad[0] or any(ctx.needs_input_grad[4:]))
        with torch.no_grad():
            ans, e, warm = ode._sweep_forward(y0, t, need)
        ctx.graph_entry = e
        ctx.warm_entry = warm
        if "pnode_amd.logview" in sys.modules:
            sys.modules["pnode_amd.logview"].note_forward(ode)
        ctx.save_for_backward(t, flat_params, ans)
        return ans

    @staticmethod
    def backward(ctx, *grad_output):
        t, flat_params, ans = ctx.saved_tensors
        ode = ctx.ode
        T = ans.shape[0]
        g = grad_output[0]
        if g.dtype != ode.tensor_dtype:
            g = g.to(ode.tensor_dtype)
        g = g.contiguous().view(T, -1)
        with torch.no_grad():
            ode._sweep_backward(ctx.graph_entry, getattr(ctx, "warm_entry", None), g, T)
            ode._allreduce_adj_p()
            if "pnode_amd.logview" in sys.modules:
                sys.modules["pnode_amd.logview"].note_backward(ode)
            adj_u = ode._shaped(ode.adj_u_flat).detach().clone()
            adj_p = ode.adj_p_tensor.detach().clone()
            gparams = tuple(adj_p[o:o + l].view_as(p).to(p.dtype) for p, o, l in zip(ode._params, ode._poff, ode._plen))
        return (adj_u, None, None, None) + gparams
