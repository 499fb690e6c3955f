"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product.

Second, independent checker: explicit RK written as plain differentiable torch ops and
differentiated by ``torch.autograd`` through the unrolled steps.  The discrete adjoint of
an explicit RK scheme (what PETSc ``TSAdjoint`` computes for the reference,
``pnode/petsc_adjoint.py:875-878``) equals this gradient to round-off, for a FIXED
sequence of accepted step sizes (the controller is not differentiated).

Used to pin ``oracle/ts_oracle.py``'s adjoint recurrence and to produce the committed
gradient goldens (``tests/golden/make_golden.py``).
"""
import torch

from .ts_oracle import METHOD_TO_RK, PETSC_DEFAULT_RK, tableau_info


def rk_step(func, tab, t, h, u):
    """One explicit RK step u -> u + h sum_j b_j K_j with K_i = f(t + c_i h, Y_i)."""
    K = []
    for i in range(tab["s"]):
        Y = u
        for j in range(i):
            if tab["A"][i, j] != 0.0:
                Y = Y + (h * tab["A"][i, j]) * K[j]
        K.append(func(t + tab["c"][i] * h, Y))
    un = u
    for j in range(tab["s"]):
        if tab["b"][j] != 0.0:
            un = un + (h * tab["b"][j]) * K[j]
    return un


def odeint_unrolled(func, u0, t_end, h_seq, save_after, method="rk4", t0=0.0):
    """Integrate with the given accepted-step sequence.

    t_end[k], h_seq[k]: end time and size of step k (as logged by the TS oracle).
    save_after: list of step counts; the state after that many steps is an output
    (0 = the initial state).  Returns the stacked outputs, differentiable.
    """
    tab = tableau_info(METHOD_TO_RK.get(method, PETSC_DEFAULT_RK) if method in METHOD_TO_RK or not _is_rk_name(method) else method)
    outs = []
    u = u0
    if 0 in save_after:
        outs.append(u)
    for k, h in enumerate(h_seq):
        tn = float(t_end[k]) - float(h)
        u = rk_step(func, tab, tn, float(h), u)
        if (k + 1) in save_after:
            outs.append(u)
    return torch.stack(outs, dim=0)


def _is_rk_name(name):
    try:
        tableau_info(name)
        return True
    except KeyError:
        return False
