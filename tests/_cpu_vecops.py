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
