"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product.

CPU restatement of the reference's implicit one-step path: ``setupTS(..., implicit_form=True,
method="beuler"|"cn")`` (reference ``pnode/petsc_adjoint.py`` ("pa.py") 651-654, 666-680:
TS type BE / CN with ``evalIFunction`` F(t,u,udot) = M udot - f(t,u), pa.py:414-441) and its
discrete adjoint (``TSAdjointStep_Theta``, driven by pa.py:875-878).

PETSc's TSTHETA as restated here (M = mass matrix or identity, shift = 1/(theta*h)):
  * ``beuler``: theta = 1, stage form:   M (X - u_n) = theta*h f(t_n + theta*h, X),
                u_{n+1} = u_n + (X - u_n)/theta
  * ``cn``:     theta = 1/2, endpoint form: M (u_{n+1} - u_n) = theta*h f(t_{n+1}, u_{n+1})
                + (1-theta)*h f(t_n, u_n)
The nonlinear stage equation is solved here by Newton with an exact dense Jacobian to round-off
(PETSc uses Newton-Krylov with loose default tolerances: the product is compared with this
oracle at the tolerance it is asked to solve to).  Discrete adjoint of one step, nu solving the
transposed stage system:
  stage form:     (M - theta*h J_X)^T nu = lambda_{n+1}/theta ;
                  lambda_n = (1 - 1/theta) lambda_{n+1} + M^T nu ;  mu += theta*h (df/dp)_X^T nu
  endpoint form:  (M - theta*h J_{n+1})^T nu = lambda_{n+1} ;
                  lambda_n = M^T nu + (1-theta)*h J_n^T nu ;
                  mu += theta*h (df/dp)_{n+1}^T nu + (1-theta)*h (df/dp)_n^T nu
Pinned by tests/test_oracle_pins.py: the reference's CN known answer (reference
tests/test_pnode.py:133-152: 1.85e-6 / 3.36e-6) and autograd through the converged Newton
iteration.  The reference's tests and examples run these with ``-ts_adapt_type none``; for adaptive runs
``lte_norm`` restates PETSc's error estimate (parity unpinned) and ``plan=`` makes the solve follow a given
accepted-step sequence.
"""
import torch

from .ts_oracle import ODEPetscOracle

THETA_METHODS = {"beuler": (1.0, False), "cn": (0.5, True)}


def step_plan(t, step_size, dtype=torch.float64):
    """Accepted (t_n, h_n) sequence and steps per output interval for a fixed-step solve: the
    exact-final-time / time-span logic does not depend on the scheme, so the explicit oracle's
    state machine is driven with f = 0."""
    class Zero(torch.nn.Module):
        def forward(self, tt, y):
            return torch.zeros_like(y)
    o = ODEPetscOracle({"ts_adapt_type": "none"})
    y0 = torch.zeros(1, dtype=dtype)
    o.setupTS(y0, Zero(), step_size=step_size, method="euler", enable_adjoint=True)
    with torch.no_grad():
        o.odeint(y0, t)
    te, h, _ = o.step_log()
    per = list(o.cur_sol_steps) if t.shape[0] > 1 else [len(h)]
    return [(float(te[k] - h[k]), float(h[k])) for k in range(len(h))], per


def _flat_f(func, t, shape):
    return lambda v: func(t, v.view(shape)).reshape(-1)


def _jac(func, t, u):
    """Dense df/du at (t,u), flattened."""
    n = u.numel()
    return torch.autograd.functional.jacobian(_flat_f(func, t, u.shape), u.reshape(-1).detach()).reshape(n, n)


def theta_step(func, t, h, u, theta, endpoint, mass=None, tol=1e-15, max_it=50):
    """One step; returns (u_next, X) with X the stage value (== u_next in endpoint form)."""
    n = u.numel()
    M = torch.eye(n, dtype=u.dtype) if mass is None else mass
    uf = u.reshape(-1)
    ts = t + h if endpoint else t + theta * h
    rhs0 = (1.0 - theta) * h * func(t, u).reshape(-1) if endpoint else torch.zeros_like(uf)
    x = uf.clone()
    for _ in range(max_it):
        r = M @ (x - uf) - theta * h * func(ts, x.view(u.shape)).reshape(-1) - rhs0
        J = M - theta * h * _jac(func, ts, x.view(u.shape))
        dx = torch.linalg.solve(J, -r)
        x = x + dx
        if dx.norm() <= tol * (1.0 + x.norm()):
            break
    X = x.view(u.shape)
    unew = X if endpoint else (uf + (x - uf) / theta).view(u.shape)
    return unew, X


def lte_norm(x, x0, xprev, h, h_prev, atol=1e-4, rtol=1e-4):
    """PETSc's local-truncation-error estimate for the theta methods (TSEvaluateWLTE_Theta, restated from memory of
    theta.c -- PARITY UNPINNED): with a = 1 + h_prev/h, Y = X + X/a - X0/(a-1) + Xprev/(a(a-1)) (a scaled second backward
    difference of the last three solutions on the non-uniform grid) and the WRMS norm of X - Y; the controller uses
    order 2 for it.  Not available in the first step (returns -1: the step is accepted with its size unchanged)."""
    from .ts_oracle import wrms
    if xprev is None:
        return -1.0
    a = 1.0 + h_prev / h
    y = x + x / a - x0 / (a - 1.0) + xprev / (a * (a - 1.0))
    return wrms(x.detach().reshape(-1).numpy(), y.detach().reshape(-1).numpy(), atol, rtol)


def solve_theta(func, u0, t, step_size, method, mass=None, plan=None):
    """Forward solve with exact Newton; returns (solutions at t (T,...), trajectory list).  `plan` = ([(t_n, h_n)],
    steps per output interval) makes it follow a given accepted-step sequence (an adaptive solve's)."""
    theta, endpoint = THETA_METHODS[method] if isinstance(method, str) else method     # or a (theta, endpoint) pair: -ts_type theta
    plan, per = step_plan(t, step_size) if plan is None else plan
    T = t.shape[0]
    u = u0.detach().clone()
    traj = []
    sols = [u.clone()] if T > 1 else []
    k = 0
    with torch.no_grad():
        for seg in range(1, T) if T > 1 else [0]:
            for _ in range(per[seg]):
                tn, h = plan[k]
                unew, X = theta_step(func, tn, h, u, theta, endpoint, mass)
                traj.append((tn, h, u, X))
                u = unew
                k += 1
            sols.append(u.clone())
    return torch.stack(sols, dim=0), traj, per


def adjoint_theta(func, params, traj, per, grad_out, method, mass=None):
    """Discrete adjoint over the stored trajectory.  grad_out: (T, ...) cotangent of the outputs.
    Returns (dL/du0, [dL/dp ...])."""
    theta, endpoint = THETA_METHODS[method] if isinstance(method, str) else method     # or a (theta, endpoint) pair: -ts_type theta
    T = grad_out.shape[0]
    lam = grad_out[-1].reshape(-1).clone()
    mu = [torch.zeros_like(p) for p in params]
    n = lam.numel()
    M = torch.eye(n, dtype=lam.dtype) if mass is None else mass

    def vjp(tt, y, w):
        with torch.enable_grad():
            yy = y.detach().requires_grad_(True)
            out = func(tt, yy)
            g = torch.autograd.grad(out, (yy,) + tuple(params), w.view(out.shape), allow_unused=True)
        gy = g[0] if g[0] is not None else torch.zeros_like(y)
        return gy.reshape(-1), [torch.zeros_like(p) if x is None else x for x, p in zip(g[1:], params)]

    k = len(traj)
    segs = range(T - 1, 0, -1) if T > 1 else [0]
    for seg in segs:
        for _ in range(per[seg]):
            k -= 1
            tn, h, u, X = traj[k]
            ts = tn + h if endpoint else tn + theta * h
            A = (M - theta * h * _jac(func, ts, X)).T
            if endpoint:
                nu = torch.linalg.solve(A, lam)
                gX, gpX = vjp(ts, X, nu)            # J_{n+1}^T nu is only needed for mu here
                gU, gpU = vjp(tn, u, nu)
                lam = M.T @ nu + (1.0 - theta) * h * gU
                for m, a, b in zip(mu, gpX, gpU):
                    m += theta * h * a + (1.0 - theta) * h * b
            else:
                nu = torch.linalg.solve(A, lam / theta)
                _, gpX = vjp(ts, X, nu)
                lam = (1.0 - 1.0 / theta) * lam + M.T @ nu
                for m, a in zip(mu, gpX):
                    m += theta * h * a
        if T > 1:
            lam = lam + grad_out[seg - 1].reshape(-1)
    return lam.view(grad_out.shape[1:]), mu


class ThetaSolve(torch.autograd.Function):
    """odeint_adjoint for the theta methods (oracle)."""

    @staticmethod
    def forward(ctx, u0, t, step_size, method, func, mass, *params):
        plan = None
        if isinstance(method, tuple) and len(method) == 2 and isinstance(method[1], tuple) and isinstance(method[1][0], list):
            method, plan = method
        sol, traj, per = solve_theta(func, u0, t, step_size, method, mass, plan)
        ctx.stuff = (func, params, traj, per, method, mass)
        return sol

    @staticmethod
    def backward(ctx, g):
        func, params, traj, per, method, mass = ctx.stuff
        with torch.no_grad():
            gu, gp = adjoint_theta(func, params, traj, per, g, method, mass)
        return (gu, None, None, None, None, None) + tuple(gp)


def odeint_adjoint_theta(func, u0, t, step_size, method, mass=None, plan=None):
    params = tuple(p for p in func.parameters() if p.requires_grad)
    return ThetaSolve.apply(u0, t, step_size, method if plan is None else (method, plan), func, mass, *params)


def odeint_unrolled_theta(func, u0, t, step_size, method, newton_its=12, mass=None):
    """Second checker: the same scheme with the Newton iteration written as differentiable torch
    ops (fixed number of iterations from a converged start), differentiated by autograd."""
    theta, endpoint = THETA_METHODS[method] if isinstance(method, str) else method     # or a (theta, endpoint) pair: -ts_type theta
    plan, per = step_plan(t, step_size)
    T = t.shape[0]
    n = u0.numel()
    M = torch.eye(n, dtype=u0.dtype) if mass is None else mass
    u = u0
    outs = [u] if T > 1 else []
    k = 0
    for seg in range(1, T) if T > 1 else [0]:
        for _ in range(per[seg]):
            tn, h = plan[k]
            ts = tn + h if endpoint else tn + theta * h
            uf = u.reshape(-1)
            rhs0 = (1.0 - theta) * h * func(tn, u).reshape(-1) if endpoint else 0.0
            with torch.no_grad():
                x0 = theta_step(func, tn, h, u.detach(), theta, endpoint, mass)[1].reshape(-1)
            x = x0
            for _ in range(newton_its):     # differentiable Newton steps around the converged point
                r = M @ (x - uf) - theta * h * func(ts, x.view(u.shape)).reshape(-1) - rhs0
                J = M - theta * h * _jac(func, ts, x.detach().view(u.shape))
                x = x - torch.linalg.solve(J, r)
            u = x.view(u.shape) if endpoint else (uf + (x - uf) / theta).view(u.shape)
            k += 1
        outs.append(u)
    return torch.stack(outs, dim=0)
