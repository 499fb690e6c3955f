"""Regenerates the committed fixtures in tests/golden/.  Run from the repo root:

    python tests/golden/make_golden.py

What is pinned and where it comes from
  rober.json            (also: the implicit CN / backward-Euler runs of oracle/theta_oracle.py on the
                        same inputs, with the reference's CN assertion, tests/test_pnode.py:151-152)
                        inputs of the reference's explicit-RK test (tests/test_pnode.py:15-23,
                        59-96: ROBER kinetics, SciPy-BDF truth, variable step list) and the
                        reference's asserted constants (tests/test_pnode.py:200-201), plus the
                        tighter values the oracle produces for the same run (loss, std, dL/dk).
  spiral_autograd.npz   fp64 autograd-through-unrolled-RK results (oracle/autograd_rk.py) for
                        every tableau on the spiral MLP, batch 20x1x2, 10 output times.
  dopri5_steps.json     SELF-golden: accepted (t, h) sequence + rejection count of the oracle's
                        adaptive controller on y' = y^3 A.  PETSc itself is not available, so
                        this pins regressions only ("parity unpinned").
Nothing of the reference's source is copied; the reference's Python cannot be imported here
(petsc4py missing), so no fixture is produced by running it.
"""
import json
import os
import sys

import numpy as np
import torch
import torch.nn as nn
from scipy.integrate import solve_ivp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle.autograd_rk import odeint_unrolled  # noqa: E402
from oracle.ts_oracle import ODEPetscOracle  # noqa: E402
from problems import SpiralFunc, SpiralTruth, flat_grads  # noqa: E402


def rober():
    t = torch.cat((torch.tensor([0], dtype=torch.float64), torch.logspace(start=-5, end=-3, steps=3, dtype=torch.float64)))
    step_size = (t[1:] - t[:-1]).tolist()

    def fun(_, s):
        k1, k2, k3 = 0.04, 3e7, 1e4
        return np.array([-k1 * s[0] + k3 * s[1] * s[2], k1 * s[0] - k3 * s[1] * s[2] - k2 * s[1] ** 2, k2 * s[1] ** 2])

    def jac(_, s):
        k1, k2, k3 = 0.04, 3e7, 1e4
        return np.array([[-k1, k3 * s[2], k3 * s[1]], [k1, -2 * k2 * s[1] - k3 * s[2], -k3 * s[1]], [0, 2 * k2 * s[1], 0]])

    path = solve_ivp(fun=fun, jac=jac, t_span=[0, 1.1e-3], y0=[1.0, 0.0, 0.0], t_eval=t.numpy(), method="BDF",
                     rtol=1e-11, atol=1e-14)
    true_y = torch.from_numpy(path["y"].T)

    class Lambda(nn.Module):
        def __init__(self):
            super().__init__()
            self.k = nn.Parameter(torch.tensor([0.05, 4e7, 2e4], dtype=torch.float64))

        def forward(self, t, y):
            k1, k2, k3 = self.k[0], self.k[1], self.k[2]
            return torch.stack((-k1 * y[0] + k3 * y[1] * y[2], k1 * y[0] - k3 * y[1] * y[2] - k2 * y[1] ** 2,
                                k2 * y[1] ** 2), -1)

    out = {"t": t.tolist(), "step_size": step_size, "true_y": true_y.tolist(),
           "reference_asserts": {"loss": 1.85e-6, "std": 3.21e-6, "abs_tol": 1e-6,
                                 "source": "tests/test_pnode.py:200-201 (method='rk3' -> PETSc default 3bs)"}}
    for name, method in [("explicit_3bs", "rk3"), ("explicit_rk4", "rk4"), ("explicit_5dp", "dopri5")]:
        f = Lambda()
        ode = ODEPetscOracle({"ts_adapt_type": "none", "ts_trajectory_type": "memory"})
        ode.setupTS(true_y[0], f, step_size=step_size, method=method, enable_adjoint=True)
        pred = ode.odeint_adjoint(true_y[0], t)
        loss = torch.mean(torch.abs(pred - true_y))
        loss.backward()
        std = torch.std(torch.abs(pred - true_y))
        out[name] = {"loss": loss.item(), "std": std.item(), "grad_k": f.k.grad.tolist(), "pred": pred.tolist()}
    # implicit theta methods (reference tests/test_pnode.py:133-152: CN, 1.85e-6 / 3.36e-6)
    from oracle.theta_oracle import odeint_adjoint_theta
    out["reference_asserts_cn"] = {"loss": 1.85e-6, "std": 3.36e-6, "abs_tol": 1e-6,
                                   "source": "tests/test_pnode.py:151-152 (method='cn', implicit_form=True)"}
    for name, method in [("implicit_cn", "cn"), ("implicit_beuler", "beuler")]:
        f = Lambda()
        pred = odeint_adjoint_theta(f, true_y[0], t, step_size, method)
        loss = torch.mean(torch.abs(pred - true_y))
        loss.backward()
        std = torch.std(torch.abs(pred - true_y))
        out[name] = {"loss": loss.item(), "std": std.item(), "grad_k": f.k.grad.tolist(), "pred": pred.tolist()}
    # IMEX (reference tests/test_pnode.py:155-180: ARKIMEX default type, 3.11e-6 / 5.65e-6, abs tol 3e-6)
    from oracle.arkimex_oracle import odeint_adjoint_arkimex
    from problems import RoberEX, RoberIM
    out["reference_asserts_imex"] = {"loss": 3.11e-6, "std": 5.65e-6, "abs_tol": 3e-6,
                                     "source": "tests/test_pnode.py:179-180 (method='imex', default ARKIMEX type 3)"}
    for name in ("3", "ars122", "a2", "ars443"):
        fI, fE = RoberIM(), RoberEX()
        pred = odeint_adjoint_arkimex(fI, fE, true_y[0], t, step_size, name)
        loss = torch.mean(torch.abs(pred - true_y))
        loss.backward()
        std = torch.std(torch.abs(pred - true_y))
        out["imex_" + name] = {"loss": loss.item(), "std": std.item(),
                               "grad": torch.cat([fI.k1.grad, fI.k3.grad, fE.k2.grad]).tolist(), "pred": pred.tolist()}
    json.dump(out, open(os.path.join(HERE, "rober.json"), "w"), indent=1)
    print("rober:", {k: (v["loss"], v["std"]) for k, v in out.items() if k.startswith("explicit")})


def spiral_autograd():
    torch.manual_seed(0)
    y0 = torch.randn(20, 1, 2, dtype=torch.float64)
    t = torch.linspace(0.0, 0.225, 10, dtype=torch.float64)
    target = torch.randn(10, 20, 1, 2, dtype=torch.float64)
    f = SpiralFunc()
    data = {"y0": y0.numpy(), "t": t.numpy(), "target": target.numpy(),
            "theta": torch.cat([p.detach().reshape(-1) for p in f.parameters()]).numpy()}
    h = 0.025
    for method in ["euler", "midpoint", "rk2", "bosh3", "rk4", "dopri5"]:
        f.zero_grad()
        y = y0.clone().requires_grad_(True)
        te = [h * (k + 1) for k in range(9)]
        pred = odeint_unrolled(f, y, te, [h] * 9, list(range(10)), method=method)
        torch.mean(torch.abs(pred - target)).backward()
        data[method + "_ans"] = pred.detach().numpy()
        data[method + "_gy0"] = y.grad.numpy()
        data[method + "_gtheta"] = flat_grads(f).numpy()
    np.savez_compressed(os.path.join(HERE, "spiral_autograd.npz"), **data)
    print("spiral_autograd: ok")


def dopri5_steps():
    y0 = torch.tensor([[2.0, 0.0], [1.0, 1.0], [-1.5, 0.5]], dtype=torch.float64)
    t = torch.tensor([0.0, 1.0, 2.5, 6.0], dtype=torch.float64)
    # step_size 0.5 makes the first attempt blow up (|u| ~ 1e25): PETSc-style rollback by
    # subtraction then corrupts u_n, so this fixture is produced with the exact restore the
    # product uses; step_size 0.2 rejects without blowing up and both rollbacks agree.
    out = {"y0": y0.tolist(), "t": t.tolist(), "rtol": 1e-4, "atol": 1e-4}
    for method, h0, exact in [("dopri5", 0.5, 1), ("bosh3", 0.5, 1), ("dopri5", 0.2, 0), ("bosh3", 0.2, 0)]:
        f = SpiralTruth()
        ode = ODEPetscOracle({"oracle_exact_rollback": exact})
        ode.setupTS(y0, f, step_size=h0, method=method)
        y = y0.clone().requires_grad_(True)
        pred = ode.odeint_adjoint(y, t)
        pred.abs().mean().backward()
        te, h, rej = ode.step_log()
        out["%s_h%g" % (method, h0)] = {"step_size": h0, "exact_rollback": exact, "t_end": te.tolist(), "h": h.tolist(), "rejections": int(rej), "per_interval": ode.cur_sol_steps,
                       "ans": pred.tolist(), "gy0": y.grad.tolist(), "gA": f.A.grad.tolist()}
        print(method, h0, "steps", len(h), "rejections", rej, "per interval", ode.cur_sol_steps)
    json.dump(out, open(os.path.join(HERE, "dopri5_steps.json"), "w"), indent=1)


if __name__ == "__main__":
    rober()
    spiral_autograd()
    dopri5_steps()
