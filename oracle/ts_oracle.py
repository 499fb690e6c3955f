"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product.

CPU restatement of the reference's explicit-RK path: ``ODEPetsc.setupTS / odeint /
odeint_adjoint`` (reference ``pnode/petsc_adjoint.py``, "pa.py" below) on top of the C
restatement of PETSc's TS pieces in ``petsc_ts_restated.c``.  The control structure is the
reference's: a C time-stepping loop that calls back into Python once per RK stage
(forward) and twice per stage (adjoint), tensors crossing the boundary zero-copy.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module.  Pinning: see ``tests/test_oracle_pins.py`` -- the reference's own
known-answer constants (``tests/test_pnode.py:183-201``) and fp64 autograd through the
unrolled steps (``oracle/autograd_rk.py``).  The adaptive step-size sequence cannot be
checked against a real PETSc build here: **parity unpinned** for that part.
"""
import ctypes
import os
import subprocess

import numpy as np
import torch
import torch.nn as nn

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    """Compile the C restatement (gcc) into oracle/libpnoracle.so."""
    subprocess.run(["make", "-s", "-C", _HERE], check=True)


def _lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libpnoracle.so")
        if not os.path.exists(so):
            build()
        _LIB = ctypes.CDLL(so)
    return _LIB


# pa.py:641-650 -- the reference's method -> TSRK type map.  Any other string leaves the TS
# on PETSc's default tableau (3bs); tests/test_pnode.py:189 relies on that with "rk3".
METHOD_TO_RK = {
    "euler": "1fe",
    "rk2": "2b",
    "fixed_bosh3": "3bs",
    "bosh3": "3bs",
    "rk4": "4",
    "fixed_dopri5": "5dp",
    "dopri5": "5dp",
    "midpoint": "midpoint",  # extension named by BASELINE.json's north_star (not in pa.py)
}
PETSC_DEFAULT_RK = "3bs"


class MiniTS:
    """ctypes face of the C mini-TS for one scalar width."""

    def __init__(self, n, np_, dtype):
        L = _lib()
        self.sfx = "_f64" if dtype == torch.float64 else "_f32"
        self.real = ctypes.c_double if dtype == torch.float64 else ctypes.c_float
        self.npdtype = np.float64 if dtype == torch.float64 else np.float32
        self.n, self.np = n, np_
        P = ctypes.POINTER(self.real)
        self.RHS = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_double, P, P)
        self.JAC = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_double, P)
        self.JT = ctypes.CFUNCTYPE(None, ctypes.c_void_p, P, P)
        self.PS = ctypes.CFUNCTYPE(None, ctypes.c_void_p)
        f = self._f
        f("ots_create").restype = ctypes.c_void_p
        f("ots_create").argtypes = [ctypes.c_long, ctypes.c_long]
        self.h = ctypes.c_void_p(f("ots_create")(n, np_))
        for name, res, args in [
            ("ots_destroy", None, [ctypes.c_void_p]),
            ("ots_set_rk_type", ctypes.c_int, [ctypes.c_void_p, ctypes.c_char_p]),
            ("ots_set_callbacks", None, [ctypes.c_void_p, ctypes.c_void_p, self.RHS, self.JAC, self.JT, self.JT, self.PS]),
            ("ots_set_poststep", None, [ctypes.c_void_p, self.PS]),
            ("ots_set_adapt", None, [ctypes.c_void_p, ctypes.c_int]),
            ("ots_set_tolerances", None, [ctypes.c_void_p, ctypes.c_double, ctypes.c_double]),
            ("ots_set_max_steps", None, [ctypes.c_void_p, ctypes.c_long]),
            ("ots_set_max_reject", None, [ctypes.c_void_p, ctypes.c_int]),
            ("ots_set_monitor", None, [ctypes.c_void_p, ctypes.c_int]),
            ("ots_set_rollback_exact", None, [ctypes.c_void_p, ctypes.c_int]),
            ("ots_set_time_step", None, [ctypes.c_void_p, ctypes.c_double]),
            ("ots_get_time_step", ctypes.c_double, [ctypes.c_void_p]),
            ("ots_set_time", None, [ctypes.c_void_p, ctypes.c_double]),
            ("ots_get_time", ctypes.c_double, [ctypes.c_void_p]),
            ("ots_set_max_time", None, [ctypes.c_void_p, ctypes.c_double]),
            ("ots_set_step_number", None, [ctypes.c_void_p, ctypes.c_long]),
            ("ots_get_step_number", ctypes.c_long, [ctypes.c_void_p]),
            ("ots_get_rejections", ctypes.c_long, [ctypes.c_void_p]),
            ("ots_get_nfe", ctypes.c_long, [ctypes.c_void_p]),
            ("ots_get_reason", ctypes.c_int, [ctypes.c_void_p]),
            ("ots_set_save_trajectory", None, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]),
            ("ots_set_cost_gradients", None, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
            ("ots_get_traj_len", ctypes.c_long, [ctypes.c_void_p]),
            ("ots_get_traj_times", None, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
            ("ots_set_time_span", None, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]),
            ("ots_get_span_solution", P, [ctypes.c_void_p, ctypes.c_int]),
            ("ots_get_span_count", ctypes.c_int, [ctypes.c_void_p]),
            ("ots_solve", ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
            ("ots_adjoint_set_steps", None, [ctypes.c_void_p, ctypes.c_long]),
            ("ots_adjoint_solve", ctypes.c_int, [ctypes.c_void_p]),
        ]:
            fn = f(name)
            fn.restype, fn.argtypes = res, args

    def _f(self, name):
        return getattr(_lib(), name + self.sfx)

    def call(self, name, *args):
        return self._f(name)(self.h, *args)

    def as_tensor(self, ptr, n):
        return torch.from_numpy(np.ctypeslib.as_array(ptr, shape=(n,)))

    def __del__(self):
        try:
            self._f("ots_destroy")(self.h)
        except Exception:
            pass


def tableau_info(name):
    """(s, order, fsal, has_embed, A[7x7], b, bembed, c) of the oracle's tableau `name`."""
    L = _lib()
    s, order, fsal, emb = (ctypes.c_int() for _ in range(4))
    A = np.zeros(49)
    b, be, c = np.zeros(7), np.zeros(7), np.zeros(7)
    fn = L.ots_tableau_info_f64
    fn.restype = ctypes.c_int
    rc = fn(name.encode(), ctypes.byref(s), ctypes.byref(order), ctypes.byref(fsal), ctypes.byref(emb),
            A.ctypes.data_as(ctypes.c_void_p), b.ctypes.data_as(ctypes.c_void_p),
            be.ctypes.data_as(ctypes.c_void_p), c.ctypes.data_as(ctypes.c_void_p))
    if rc:
        raise KeyError(name)
    S = s.value
    return dict(s=S, order=order.value, fsal=bool(fsal.value), has_embed=bool(emb.value),
                A=A.reshape(7, 7)[:S, :S].copy(), b=b[:S].copy(), bembed=be[:S].copy(), c=c[:S].copy())


def wrms(u, y, atol, rtol):
    """TSErrorWeightedNorm (NORM_2) of the oracle on numpy arrays."""
    L = _lib()
    u = np.ascontiguousarray(u)
    y = np.ascontiguousarray(y, dtype=u.dtype)
    fn = L.ots_wrms_f64 if u.dtype == np.float64 else L.ots_wrms_f32
    fn.restype = ctypes.c_double
    fn.argtypes = [ctypes.c_long, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_double, ctypes.c_double]
    return fn(u.size, u.ctypes.data, y.ctypes.data, atol, rtol)


def _flatten(seq):
    """pnode/misc.py:4-6"""
    flat = [p.contiguous().view(-1) for p in seq]
    return torch.cat(flat) if len(flat) > 0 else torch.tensor([])


def _flatten_none_to_zeros(seq, like):
    """pnode/misc.py:9-14"""
    flat = [p.contiguous().view(-1) if p is not None else torch.zeros_like(q).view(-1) for p, q in zip(seq, like)]
    return torch.cat(flat) if len(flat) > 0 else torch.tensor([])


class ODEPetscOracle(object):
    """Restatement of ``ODEPetsc`` (pa.py:366-900), explicit-RK branch, CPU tensors only.

    `options` stands in for the PETSc options database filled by ``petsc4py.init(argv)``
    (ode_demo_petsc.py:63-66) and read by ``ts.setFromOptions()`` (pa.py:775); keys are the
    PETSc spellings without the dash, e.g. ``{"ts_adapt_type": "none"}``.
    """

    def __init__(self, options=None):
        self.options = dict(options or {})
        self.ts = None
        self.n = 0
        self.tensor_size = None
        self.tensor_dtype = None
        self.device = None
        self.funcIM = None
        self.funcEX = None
        self.flat_params = None
        self.np = None
        self.nfe_rhs = 0

    # ---- callbacks (pa.py:393-412, 443-457, 52-82, 341-363)
    def _evalRHSFunction(self, ctx, t, U, F):
        u = self.mts.as_tensor(U, self.n).view(self.tensor_size)
        f = self.mts.as_tensor(F, self.n).view(self.tensor_size)
        with torch.no_grad():
            f.copy_(self.funcEX(t, u))

    def _evalRHSJacobian(self, ctx, t, U):
        self.t = t
        self.cached_u_tensor = self.mts.as_tensor(U, self.n).view(self.tensor_size)

    def _jacT(self, ctx, X, Y):
        x = self.mts.as_tensor(X, self.n).view(self.tensor_size)
        y = self.mts.as_tensor(Y, self.n).view(self.tensor_size)
        f_params = tuple(p for p in self.funcEX.parameters() if p.requires_grad)
        with torch.enable_grad():
            u = self.cached_u_tensor.detach().requires_grad_(True)
            out = self.funcEX(self.t, u)
            vjp_u, *self.vjp_params = torch.autograd.grad(out, (u,) + f_params, x, allow_unused=True)
        if vjp_u is None:
            vjp_u = torch.zeros_like(y)
        y.copy_(vjp_u)

    def _jacPT(self, ctx, X, Y):
        y = self.mts.as_tensor(Y, self.np)
        f_params = tuple(p for p in self.funcEX.parameters() if p.requires_grad)
        y.copy_(_flatten_none_to_zeros(self.vjp_params, f_params))

    def _tspanPostStep(self, ctx):
        """pa.py:518-532"""
        stepno = self.mts.call("ots_get_step_number")
        t = self.mts.call("ots_get_time")
        if self.cur_sol_index < len(self.sol_times):
            if isinstance(self.step_size, list):
                if stepno < len(self.step_size):
                    self.mts.call("ots_set_time_step", float(self.step_size[stepno]))
            self.cur_sol_steps[self.cur_sol_index] += 1
            delta = 1e-5 if self.tensor_dtype == torch.double else 1e-3
            if abs(t - float(self.sol_times[self.cur_sol_index])) < delta:
                self.cur_sol_index += 1

    # ---- pa.py:534-775
    def setupTS(self, u_tensor, func, step_size=0.01, enable_adjoint=True, implicit_form=False,
                use_dlpack=True, method="dopri5", mass=None, imex_form=False, func2=None,
                batch_size=1, linear_solver="petsc", fixed_jacobian=False, matrixfree_jacobian=True):
        if imex_form and func2 is None:
            raise ValueError("func2 must be provided to enable imex_form=True")
        if implicit_form or imex_form:
            raise NotImplementedError("oracle covers the explicit-RK path only")
        if u_tensor.device.type != "cpu":
            raise ValueError("the oracle runs on CPU tensors")
        if self.funcIM is not func:
            self.funcIM = func
            self.funcEX = func
            self.flat_params = _flatten(p for p in func.parameters() if p.requires_grad) if isinstance(func, nn.Module) else torch.tensor([])
            self.np = self.flat_params.numel()
        if (u_tensor.size() != self.tensor_size or u_tensor.dtype != self.tensor_dtype or u_tensor.device != self.device
                or self.mts.np != self.np):
            self.tensor_size = u_tensor.size()
            self.tensor_dtype = u_tensor.dtype
            self.device = u_tensor.device
            self.n = u_tensor.numel()
            self.mts = MiniTS(self.n, self.np, self.tensor_dtype)
            self.mts.call("ots_set_rk_type", METHOD_TO_RK.get(method, PETSC_DEFAULT_RK).encode())
            self._cbs = (self.mts.RHS(self._evalRHSFunction), self.mts.JAC(self._evalRHSJacobian),
                         self.mts.JT(self._jacT), self.mts.JT(self._jacPT), self.mts.PS(self._tspanPostStep))
            self._nops = self.mts.PS(0)
            self.mts.call("ots_set_callbacks", None, self._cbs[0], self._cbs[1], self._cbs[2], self._cbs[3], self._nops)
            if enable_adjoint:
                self.adj_u_tensor = u_tensor.detach().clone().contiguous()
                self.adj_p_tensor = self.flat_params.detach().clone().to(self.tensor_dtype).contiguous()
                self.mts.call("ots_set_cost_gradients", self.adj_u_tensor.data_ptr(),
                              self.adj_p_tensor.data_ptr() if self.np > 0 else None)
        self.step_size = step_size
        if not isinstance(step_size, list):
            self.mts.call("ots_set_time_step", float(step_size))
        # ts.setSaveTrajectory / removeTrajectory (pa.py:771-774), then setFromOptions (775)
        o = self.options
        sol_only = int(str(o.get("ts_trajectory_solution_only", 1)).lower() not in ("0", "false", "no"))
        self.mts.call("ots_set_save_trajectory", 1 if enable_adjoint else 0, sol_only)
        if "ts_rk_type" in o:
            if self.mts.call("ots_set_rk_type", str(o["ts_rk_type"]).encode()):
                raise ValueError("unknown -ts_rk_type %r" % o["ts_rk_type"])
        self.mts.call("ots_set_adapt", 0 if str(o.get("ts_adapt_type", "basic")) == "none" else 1)
        self.mts.call("ots_set_tolerances", float(o.get("ts_atol", 1e-4)), float(o.get("ts_rtol", 1e-4)))
        if "ts_max_steps" in o:
            self.mts.call("ots_set_max_steps", int(o["ts_max_steps"]))
        self.mts.call("ots_set_max_reject", int(o.get("ts_max_reject", 10)))
        self.mts.call("ots_set_monitor", 1 if "ts_monitor" in o else 0)
        # not a PETSc option: undo rejected steps by restoring u_n instead of TSRollBack_RK's
        # subtraction (the product's behaviour; see petsc_ts_restated.c ots_set_rollback_exact)
        self.mts.call("ots_set_rollback_exact", int(o.get("oracle_exact_rollback", 0)))

    # ---- pa.py:777-869
    def odeint(self, u0, t):
        self.u0 = u0.detach().clone().contiguous()
        m = self.mts
        self.sol_times = t.cpu().to(dtype=torch.float64)
        m.call("ots_set_time_step", float(self.step_size if not isinstance(self.step_size, list) else self.step_size[0]))
        T = t.shape[0]
        if T == 1:
            m.call("ots_set_time", 0.0)
            m.call("ots_set_max_time", float(self.sol_times[0]))
            m.call("ots_set_poststep", self._nops)
        else:
            times = np.ascontiguousarray(self.sol_times.numpy())
            m.call("ots_set_time_span", T, times.ctypes.data)
            self.cur_sol_steps = [0] * T
            self.cur_sol_index = 1
            m.call("ots_set_poststep", self._cbs[4])
        m.call("ots_set_step_number", 0)
        rc = m.call("ots_solve", self.u0.data_ptr())
        if rc:
            raise RuntimeError("oracle TSSolve diverged, reason %d" % m.call("ots_get_reason"))
        self.nsteps = m.call("ots_get_step_number")
        if T == 1:
            return torch.stack([self.u0.clone()], dim=0)
        sols = [m.as_tensor(m.call("ots_get_span_solution", i), self.n).clone().view(self.tensor_size) for i in range(T)]
        m.call("ots_set_poststep", self._nops)
        if self.cur_sol_index != len(self.sol_times):
            raise Exception("TSSolve fails to step on all the specified points")
        return torch.stack(sols, dim=0)

    def step_log(self):
        """(t_end[k], h[k]) of the accepted steps of the last forward solve, rejections."""
        m = self.mts
        L = m.call("ots_get_traj_len")
        te, h = np.zeros(L), np.zeros(L)
        m.call("ots_get_traj_times", te.ctypes.data, h.ctypes.data)
        return te[1:], h[1:], m.call("ots_get_rejections")

    # ---- pa.py:871-890
    def petsc_adjointsolve(self, t, i=1):
        m = self.mts
        if t.shape[0] == 1:
            # the reference uses round(|t/dt|) (pa.py:875), right only for uniform steps;
            # the oracle reverses the steps actually taken
            m.call("ots_adjoint_set_steps", self.nsteps)
        else:
            m.call("ots_adjoint_set_steps", self.cur_sol_steps[i])
        rc = m.call("ots_adjoint_solve")
        if rc:
            raise RuntimeError("oracle TSAdjointSolve failed (%d)" % rc)
        return self.adj_u_tensor, self.adj_p_tensor

    # ---- pa.py:892-900
    def odeint_adjoint(self, y0, t):
        if not isinstance(self.funcIM, nn.Module):
            raise ValueError("func is required to be an instance of nn.Module.")
        return _OracleAdjoint.apply(y0, t, self.flat_params, self)


class _OracleAdjoint(torch.autograd.Function):
    """pa.py:903-947"""

    @staticmethod
    def forward(ctx, y0, t, flat_params, ode):
        ctx.ode = ode
        with torch.no_grad():
            ans = ode.odeint(y0, t)
        ctx.save_for_backward(t, flat_params, ans)
        return ans

    @staticmethod
    def backward(ctx, *grad_output):
        t, flat_params, ans = ctx.saved_tensors
        ode = ctx.ode
        T = ans.shape[0]
        with torch.no_grad():
            ode.adj_u_tensor.copy_(grad_output[0][-1])
            ode.adj_p_tensor.zero_()
            if T == 1:
                adj_u, adj_p = ode.petsc_adjointsolve(t)
            for i in range(T - 1, 0, -1):
                adj_u, adj_p = ode.petsc_adjointsolve(t, i)
                adj_u.add_(grad_output[0][i - 1])
            adj_u = adj_u.detach().clone()
            adj_p = adj_p.detach().clone().to(flat_params.dtype)
        return adj_u, None, adj_p, None
