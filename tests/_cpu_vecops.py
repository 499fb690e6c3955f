"""TEST SCAFFOLDING -- a CPU stand-in for ``pnode_amd.petsc_adjoint.HipVecOps``.

It exists so that the host side of the product (the C++ stepper/controller/scheduler behind
the C ABI and the Python orchestration in ``pnode_amd/petsc_adjoint.py``) can be exercised in
the CPU-only container against the oracle.  It is injected through ``ODEPetsc(backend=...)``
from tests only; the product never selects it (the default backend refuses CPU tensors).
The arithmetic order mirrors the kernels: ((c0*x0) + c1*x1) + c2*x2 ...
"""
import torch

from oracle import ts_oracle


class CpuVecOps(object):
    def __init__(self, device, dtype, n):
        assert device.type == "cpu"
        self.device, self.dtype, self.n = device, dtype, n
        self._enorm = -1.0
        self.calls = {"rk_stage": 0, "combine_wrms": 0, "adj_theta": 0, "adj_accum": 0, "param_accum": 0, "copy": 0}

    # ---- the C++ step loops on this backend: the product's pn_rk_attempt / pn_rk_adjoint_step walk the tableau and call the
    # four vector operations through a table of function pointers (pn_vec_ops, include/pnode_amd.h section 3a); here the
    # table points at this stand-in, so that the loops themselves run in the CPU-only container against the oracle
    native_steps = True
    code = property(lambda self: 1 if self.dtype == torch.float64 else 0)

    def stream(self):
        return None

    def wrms_buffers(self):
        return None, None

    def _tensor_at(self, address):
        import ctypes
        import numpy as np
        if not address:
            return None
        ct = ctypes.c_double if self.dtype == torch.float64 else ctypes.c_float
        return torch.from_numpy(np.ctypeslib.as_array(ctypes.cast(address, ctypes.POINTER(ct)), shape=(self.n,)))

    @property
    def vec_ops(self):
        import ctypes
        from pnode_amd import _lib
        if getattr(self, "_vec_ops", None) is None:
            at = self._tensor_at

            def guard(fn):
                def run(*a):
                    try:
                        fn(*a)
                        return 0
                    except BaseException as exc:           # surfaces as a failed launch
                        self.loop_error = exc
                        return 1
                return run

            def rk_stage(st, dt, n, y, u, nk, K, coef):
                self.rk_stage(at(y), at(u), [at(K[j]) for j in range(nk)], [coef[j] for j in range(nk)])

            def combine(st, dt, n, unew, u, nk, K, cb, ce, atol, rtol, work, res):
                self.combine_wrms(at(unew), at(u), [at(K[j]) for j in range(nk)], [cb[j] for j in range(nk)],
                                  [ce[j] for j in range(nk)], atol, rtol)

            def adj_theta(st, dt, n, w, lam, c_lam, nk, dl, coef):
                self.adj_theta(at(w), at(lam), c_lam, [at(dl[j]) for j in range(nk)], [coef[j] for j in range(nk)])

            def adj_accum(st, dt, n, out, lam, nk, dl, coef, forcing, w_next, c_next):
                self.adj_accum(at(out), at(lam), [at(dl[j]) for j in range(nk)], [coef[j] for j in range(nk)], at(forcing),
                               at(w_next), c_next)
            self._vec_ops_fns = (_lib.RK_STAGE_FN(guard(rk_stage)), _lib.RK_COMBINE_WRMS_FN(guard(combine)),
                                 _lib.ADJ_THETA_FN(guard(adj_theta)), _lib.ADJ_ACCUM_FN(guard(adj_accum)))
            self._vec_ops_struct = _lib.VecOps(*self._vec_ops_fns)
            self._vec_ops = ctypes.cast(ctypes.pointer(self._vec_ops_struct), ctypes.c_void_p)
        return self._vec_ops

    def empty(self, *shape):
        # NaN-filled so that reads of never-written memory are caught
        return torch.full(*shape, float("nan"), dtype=self.dtype) if len(shape) == 1 and isinstance(shape[0], tuple) \
            else torch.full(shape, float("nan"), dtype=self.dtype)

    def _put(self, dst, val):
        """Write like a raw-pointer kernel does: invisible to autograd's version counters."""
        dst.detach().numpy()[: self.n] = val.detach().numpy()[: self.n]

    def _lin(self, xs, cs):
        n = self.n
        acc = cs[0] * xs[0][:n]
        for x, c in zip(xs[1:], cs[1:]):
            acc = acc + c * x[:n]
        return acc

    def rk_stage(self, y, u, Ks, coefs):
        self.calls["rk_stage"] += 1
        self._put(y, self._lin([u] + list(Ks), [1.0] + list(coefs)))

    def combine_wrms(self, unew, u, Ks, cb, ce, atol, rtol):
        self.calls["combine_wrms"] += 1
        n = self.n
        if unew is not None:
            un = self._lin([u] + list(Ks), [1.0] + list(cb))
            self._put(unew, un)
        else:
            un = u[:n]
        err = self._lin(list(Ks), list(ce)) if Ks else torch.zeros(n, dtype=self.dtype)
        uh = (un + err).to(self.dtype)
        self._enorm = ts_oracle.wrms(un.numpy(), uh.numpy(), atol, rtol)

    def read_enorm(self):
        return self._enorm

    def adj_theta(self, w, lam, c_lam, dlams, coefs):
        self.calls["adj_theta"] += 1
        xs = ([lam] if lam is not None else []) + list(dlams)
        cs = ([c_lam] if lam is not None else []) + list(coefs)
        self._put(w, self._lin(xs, cs))

    def adj_accum(self, lam_out, lam, dlams, coefs, forcing, w_next=None, c_next=0.0):
        self.calls["adj_accum"] += 1
        xs = [lam] + list(dlams) + ([forcing] if forcing is not None else [])
        out = self._lin(xs, [1.0] + list(coefs) + ([1.0] if forcing is not None else []))
        self._put(lam_out, out)
        if w_next is not None:
            self._put(w_next, c_next * out)

    def param_accum(self, mu, alpha, grads, offsets, lens):
        self.calls["param_accum"] += 1
        for g, o, l in zip(grads, offsets, lens):
            if g is not None:
                mu.detach().numpy()[o:o + l] += (alpha * g.reshape(-1)).detach().numpy()

    def param_accum_multi(self, mu, alphas, grad_sets, offsets, lens):
        self.calls["param_accum"] += 1
        for alpha, grads in zip(alphas, grad_sets):
            for g, o, l in zip(grads, offsets, lens):
                if g is not None:
                    mu.detach().numpy()[o:o + l] += (alpha * g.reshape(-1)).detach().numpy()

    def lincomb(self, out, xs, cs):
        self.calls["lincomb"] = self.calls.get("lincomb", 0) + 1
        self._put(out, self._lin(list(xs), list(cs)))

    def dots(self, x, ys):
        self.calls["dots"] = self.calls.get("dots", 0) + 1
        n = self.n
        return [float(torch.dot(x[:n].double(), y[:n].double())) for y in ys]

    def copy(self, y, x):
        self.calls["copy"] += 1
        self._put(y, x)

    # ---- device-resident GMRES (pn_krylov_*): a Python restatement of the device state machine of
    # pnode_amd/csrc/pn_krylov.hip, decision for decision, so that the host flow that drives it (chunks of iterations
    # enqueued ahead, launches past convergence that must be no-ops, restart, the deferred decisions of a sharded solve)
    # runs in the CPU-only container.  The kernels themselves are covered by the -m gpu tests.
    MAX_KRYLOV_RESTART = 126

    class _Kr(object):
        pass

    def krylov_new(self, restart):
        kr = self._Kr()
        kr.m = restart
        kr.V = torch.full((restart + 1, self.n), float("nan"), dtype=self.dtype)
        kr.vin = torch.full((self.n,), float("nan"), dtype=self.dtype)
        kr.w = torch.full((self.n,), float("nan"), dtype=self.dtype)
        kr.S = dict(stop=0, kdone=0, cur=-1, phase=0, total=0, closed=0, apply=0, res=0.0, hk1=0.0, tol=0.0, nt=0, brk=0)
        kr.h = torch.zeros(restart + 2, dtype=torch.float64)          # products of the pass in flight
        kr.d, kr.c = [0.0] * (restart + 2), [0.0] * (restart + 2)
        kr.H = [[0.0] * (restart + 1) for _ in range(restart)]
        kr.cs, kr.sn, kr.g = [0.0] * restart, [0.0] * restart, [0.0] * (restart + 1)
        kr.launches = 0
        kr.noops = 0
        return kr

    def _kr_dot(self, a, b):
        return float(torch.dot(a[: self.n].double(), b[: self.n].double()))

    def _kr_update(self, kr, out, w, nt, out2=None):
        T = self.dtype
        acc = torch.tensor(kr.c[0], dtype=T) * w[: self.n]
        for j in range(nt):
            acc = acc + torch.tensor(kr.c[1 + j], dtype=T) * kr.V[j][: self.n]
        self._put(out, acc)
        if out2 is not None:
            self._put(out2, acc)

    def _kr_finish_column(self, kr, k, hk1, refined):
        import math
        S, m = kr.S, kr.m
        col = kr.H[k]
        for j in range(k + 1):
            col[j] = kr.d[j] + float(kr.h[j]) if refined else float(kr.h[j])
        col[k + 1] = hk1
        for i in range(k):
            t = kr.cs[i] * col[i] + kr.sn[i] * col[i + 1]
            col[i + 1] = -kr.sn[i] * col[i] + kr.cs[i] * col[i + 1]
            col[i] = t
        a, b = col[k], col[k + 1]
        r = math.hypot(a, b)
        kr.cs[k], kr.sn[k] = (1.0, 0.0) if r == 0.0 else (a / r, b / r)
        col[k], col[k + 1] = r, 0.0
        kr.g[k + 1] = -kr.sn[k] * kr.g[k]
        kr.g[k] = kr.cs[k] * kr.g[k]
        S["res"], S["hk1"], S["kdone"] = abs(kr.g[k + 1]), hk1, k + 1
        S["total"] += 1
        res = S["res"]
        S["stop"] = 4 if res != res else 1 if res <= S["tol"] else 2 if hk1 == 0.0 else 3 if S["total"] >= S["maxit"] else 0

    def krylov_begin(self, kr, r, rtol, atol, maxit, first, reduce=None):
        S = kr.S
        kr.launches += 1
        kr.h[0] = self._kr_dot(r, r)
        if reduce is not None:
            reduce(kr.h[:1])
        rr = float(kr.h[0])
        beta = max(rr, 0.0) ** 0.5
        S["beta"] = beta
        if first:
            S.update(rtol=rtol, atol=atol, maxit=maxit, bnorm=beta, tol=max(rtol * beta, atol), total=0, brk=0, res=beta)
            stop = 1 if (beta == 0.0 or beta <= atol) else 0
        else:
            stop = 1 if beta <= S["tol"] else 0
            if stop:
                S["res"] = beta
            if not stop and S["total"] >= S["maxit"]:
                stop = 3
        if rr != rr:
            stop = 4
        S.update(stop=stop, kdone=0, cur=-1, phase=0, closed=0, apply=0, nt=0)
        if first:
            kr.second_passes = 0
        if not stop:
            kr.g = [beta] + [0.0] * kr.m
            kr.c[0] = 1.0 / beta
            self._kr_update(kr, kr.V[0], r, 0, kr.vin)

    def krylov_step(self, kr, k, reduce=None):
        S, m = kr.S, kr.m
        kr.launches += 1
        if k < 0:                                          # "the iteration the state says is due" (graph-replayed steps)
            k = S["kdone"]
        if not (S["stop"] == 0 and S["kdone"] == k and S["closed"] == 0 and k < m):
            kr.noops += 1
            if reduce is not None:                         # the collectives are issued whatever the device decides
                reduce(kr.h[: k + 2])
                reduce(kr.h[: k + 2])
            return
        w = kr.w
        for j in range(k + 1):
            kr.h[j] = self._kr_dot(w, kr.V[j])
        kr.h[k + 1] = self._kr_dot(w, w)
        if reduce is not None:
            reduce(kr.h[: k + 2])
        ww = float(kr.h[k + 1])
        ssq = 0.0
        for j in range(k + 1):
            ssq += float(kr.h[j]) * float(kr.h[j])
        rest = ww - ssq
        S["cur"] = k
        if rest > 0.25 * ww and rest > 0.0:
            hk1 = rest ** 0.5
            kr.c[0] = 1.0 / hk1
            for j in range(k + 1):
                kr.c[1 + j] = -float(kr.h[j]) / hk1
            S["nt"], S["phase"] = k + 1, 1
            self._kr_finish_column(kr, k, hk1, False)
            if S["stop"] == 0:
                self._kr_update(kr, kr.V[k + 1], w, k + 1, kr.vin)
            if reduce is not None:
                reduce(kr.h[: k + 2])
            return
        if ww != ww:
            S["phase"], S["stop"] = 0, 4
            if reduce is not None:
                reduce(kr.h[: k + 2])
            return
        for j in range(k + 1):
            kr.d[j] = float(kr.h[j])
            kr.c[1 + j] = -float(kr.h[j])
        kr.c[0] = 1.0
        S["nt"], S["phase"] = k + 1, 2
        kr.second_passes = getattr(kr, "second_passes", 0) + 1
        self._kr_update(kr, w, w, k + 1)
        for j in range(k + 1):
            kr.h[j] = self._kr_dot(w, kr.V[j])
        kr.h[k + 1] = self._kr_dot(w, w)
        if reduce is not None:
            reduce(kr.h[: k + 2])
        ww = float(kr.h[k + 1])
        ssq = 0.0
        for j in range(k + 1):
            ssq += float(kr.h[j]) * float(kr.h[j])
        hk1 = max(ww - ssq, 0.0) ** 0.5
        if hk1 > 0.0:
            kr.c[0] = 1.0 / hk1
            for j in range(k + 1):
                kr.c[1 + j] = -float(kr.h[j]) / hk1
        S["nt"], S["phase"] = k + 1, 3
        self._kr_finish_column(kr, k, hk1, True)
        if ww != ww:
            S["stop"] = 4
        if hk1 > 0.0 and S["stop"] == 0:
            self._kr_update(kr, kr.V[k + 1], w, k + 1, kr.vin)

    def krylov_close(self, kr, x):
        S, m = kr.S, kr.m
        kr.launches += 1
        S["apply"] = 0
        kd = S["kdone"]
        if S["closed"] or not (S["stop"] != 0 or kd >= m):
            return
        S["closed"] = 1
        if kd > 0 and S["stop"] != 4:
            y = [0.0] * kd
            for i in range(kd - 1, -1, -1):
                s = kr.g[i]
                for j in range(i + 1, kd):
                    s -= kr.H[j][i] * y[j]
                if kr.H[i][i] == 0.0:
                    S["brk"], S["stop"] = 1, 5
                    return
                y[i] = s / kr.H[i][i]
            kr.c[0] = 1.0
            kr.c[1: 1 + kd] = y
            self._kr_update(kr, x, x, kd)

    def krylov_status(self, kr):
        S = kr.S
        return int(S["stop"]), int(S["kdone"]), int(S["total"]), S["res"]
