"""CPU-side checks of the product's host logic: the C++ stepper / controller / checkpoint
scheduler behind the C ABI and the Python orchestration of pnode_amd.petsc_adjoint, driven
with the test-only CPU stand-in for the device ops (tests/_cpu_vecops.py) and compared with
the oracle.  The device kernels themselves are covered by the -m gpu tests."""
import ctypes
import json
import os
import random

import numpy as np
import pytest
import torch
import torch.nn as nn

from _cpu_vecops import CpuVecOps
from oracle.ts_oracle import ODEPetscOracle
from pnode_amd import _lib, options, petsc_adjoint
from problems import SpiralFunc, SpiralTruth, TimeDependent, flat_grads, rel_err

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _pair(make_func, y0, t, target, method, opts, step_size=0.025):
    f_ref = make_func()
    ref = ODEPetscOracle(dict(opts, oracle_exact_rollback=1))
    ref.setupTS(y0, f_ref, step_size=step_size, method=method)
    yr = y0.clone().requires_grad_(True)
    pr = ref.odeint_adjoint(yr, t)
    torch.mean(torch.abs(pr - target)).backward()
    options.clear()
    for k, v in opts.items():
        options.set_option(k, v)
    f = make_func()
    ode = petsc_adjoint.ODEPetsc(backend=CpuVecOps)
    ode.setupTS(y0, f, step_size=step_size, method=method)
    y = y0.clone().requires_grad_(True)
    p = ode.odeint_adjoint(y, t)
    torch.mean(torch.abs(p - target)).backward()
    return (pr, yr.grad, flat_grads(f_ref), ref), (p, y.grad, flat_grads(f), ode)


CASES = [
    ("rk4", {"ts_adapt_type": "none"}),
    ("rk4", {"ts_adapt_type": "none", "ts_trajectory_solution_only": 0}),
    ("rk4", {"ts_adapt_type": "none", "ts_trajectory_max_cps_ram": 3}),
    ("euler", {}), ("midpoint", {}), ("rk2", {}), ("bosh3", {}), ("dopri5", {}),
    ("dopri5", {"ts_trajectory_solution_only": 0}),
    ("dopri5", {"ts_trajectory_max_cps_ram": 2}),
    ("rk3", {"ts_adapt_type": "none"}),
    ("euler", {"ts_rk_type": "5f"}),
    ("rk4", {"ts_rk_type": "2a", "ts_rtol": 1e-6, "ts_atol": 1e-6}),
]


@pytest.mark.parametrize("method,opts", CASES)
def test_spiral_batch_against_oracle(method, opts):
    torch.manual_seed(0)
    y0 = torch.randn(20, 1, 2, dtype=torch.float64)
    t = torch.linspace(0.0, 25.0, 1001, dtype=torch.float64)[:10]
    target = torch.randn(10, 20, 1, 2, dtype=torch.float64)
    a, b = _pair(SpiralFunc, y0, t, target, method, opts)
    assert rel_err(b[0], a[0]) < 1e-13 and rel_err(b[1], a[1]) < 1e-12 and rel_err(b[2], a[2]) < 1e-12
    assert b[3].cur_sol_steps == a[3].cur_sol_steps
    te, h, rej = a[3].step_log()
    assert b[3]._nsteps == len(h)
    for k in range(len(h)):
        assert b[3]._step_info(k)[1] == pytest.approx(h[k], rel=1e-10)


@pytest.mark.parametrize("key", ["dopri5_h0.5", "bosh3_h0.5", "dopri5_h0.2", "bosh3_h0.2"])
@pytest.mark.parametrize("opts", [{}, {"ts_trajectory_max_cps_ram": 4}, {"ts_trajectory_solution_only": 0}])
def test_adaptive_controller_with_rejections(key, opts):
    gold = json.load(open(os.path.join(GOLD, "dopri5_steps.json")))
    G = gold[key]
    y0 = torch.tensor(gold["y0"], dtype=torch.float64)
    t = torch.tensor(gold["t"], dtype=torch.float64)
    for k, v in opts.items():
        options.set_option(k, v)
    f = SpiralTruth()
    ode = petsc_adjoint.ODEPetsc(backend=CpuVecOps)
    ode.setupTS(y0, f, step_size=G["step_size"], method=key.split("_")[0])
    y = y0.clone().requires_grad_(True)
    pred = ode.odeint_adjoint(y, t)
    pred.abs().mean().backward()
    assert ode._nsteps == len(G["h"]) and ode.cur_sol_steps == G["per_interval"]
    assert ode._lib.pn_ts_rejections(ode._ts) == G["rejections"]
    hs = [ode._step_info(k)[1] for k in range(ode._nsteps)]
    assert np.allclose(hs, G["h"], rtol=1e-10)
    tol = 1e-9 if G["exact_rollback"] else 1e-7        # PETSc-style rollback noise in the fixture
    assert rel_err(pred, torch.tensor(G["ans"], dtype=torch.float64)) < tol
    assert rel_err(y.grad, torch.tensor(G["gy0"], dtype=torch.float64)) < tol
    assert rel_err(f.A.grad, torch.tensor(G["gA"], dtype=torch.float64)) < tol
    if "ts_trajectory_max_cps_ram" in opts:
        assert ode._traj.high_water() <= 4


@pytest.mark.parametrize("method", ["rk4", "dopri5", "midpoint", "bosh3"])
def test_single_end_time_time_dependent_unused_parameter(method):
    torch.manual_seed(1)
    y0 = torch.randn(7, 5, dtype=torch.float64)
    t = torch.tensor([1.0], dtype=torch.float64)
    target = torch.randn(1, 7, 5, dtype=torch.float64)
    a, b = _pair(lambda: TimeDependent(5), y0, t, target, method, {"ts_adapt_type": "none"}, step_size=0.1)
    assert b[3]._nsteps == 10
    assert rel_err(b[0], a[0]) < 1e-13 and rel_err(b[1], a[1]) < 1e-12 and rel_err(b[2], a[2]) < 1e-12


def test_match_step_rules_on_a_grid_the_step_does_not_divide():
    """t = linspace(0,25,1001)[:6] has spacing 0.025 exactly; linspace(0,25,1000) has 0.025025:
    the 1% stretch / halving / restore-after-span rules must give the oracle's sequence."""
    torch.manual_seed(0)
    y0 = torch.randn(4, 2, dtype=torch.float64)
    for t in [torch.linspace(0.0, 25.0, 1000, dtype=torch.float64)[:6],
              torch.tensor([0.0, 0.03, 0.1, 0.1 + 1e-3, 0.26], dtype=torch.float64)]:
        target = torch.zeros(len(t), 4, 2, dtype=torch.float64)
        a, b = _pair(SpiralFunc, y0, t, target, "rk4", {"ts_adapt_type": "none"})
        te, h, _ = a[3].step_log()
        hs = [b[3]._step_info(k)[1] for k in range(b[3]._nsteps)]
        assert len(hs) == len(h) and np.allclose(hs, h, rtol=1e-12)
        assert b[3].cur_sol_steps == a[3].cur_sol_steps
        assert rel_err(b[0], a[0]) < 1e-13 and rel_err(b[2], a[2]) < 1e-12


def test_step_size_list_and_the_references_failure_mode():
    """pa.py:523-525: a list gives the size of each step; pa.py:867-868: a missed output time
    raises the reference's bare Exception."""
    gold = json.load(open(os.path.join(GOLD, "rober.json")))
    t = torch.tensor(gold["t"], dtype=torch.float64)
    y0 = torch.tensor(gold["true_y"][0], dtype=torch.float64)

    class F(nn.Module):
        def __init__(self):
            super().__init__()
            self.a = nn.Parameter(torch.tensor(-1.0, dtype=torch.float64))

        def forward(self, t, y):
            return self.a * y

    options.set_option("ts_adapt_type", "none")
    ode = petsc_adjoint.ODEPetsc(backend=CpuVecOps)
    ode.setupTS(y0, F(), step_size=gold["step_size"], method="rk4")
    out = ode.odeint_adjoint(y0, t)
    assert ode._nsteps == 3 and ode.cur_sol_steps == [0, 1, 1, 1]
    assert [ode._step_info(k)[1] for k in range(3)] == pytest.approx(gold["step_size"], rel=1e-12)
    assert out.shape == (4, 3)
    options.set_option("ts_max_steps", "2")
    ode2 = petsc_adjoint.ODEPetsc(backend=CpuVecOps)
    ode2.setupTS(y0, F(), step_size=gold["step_size"], method="rk4")
    with pytest.raises(Exception, match="fails to step on all the specified points"):
        ode2.odeint(y0, t)


def test_error_behaviour_mirrors_the_reference():
    y0 = torch.zeros(3, dtype=torch.float64)
    ode = petsc_adjoint.ODEPetsc(backend=CpuVecOps)
    with pytest.raises(ValueError, match="func2 must be provided"):          # pa.py:585-586
        ode.setupTS(y0, nn.Linear(3, 3), imex_form=True)
    with pytest.raises(NotImplementedError):
        ode.setupTS(y0, nn.Linear(3, 3), implicit_form=True, method="dopri5")
    ode.setupTS(y0, lambda t, y: -y, step_size=0.1, method="euler")
    with pytest.raises(ValueError, match="instance of nn.Module"):           # pa.py:896-897
        ode.odeint_adjoint(y0, torch.tensor([1.0]))
    out = ode.odeint(y0 + 1.0, torch.tensor([1.0]))                           # plain callables may be solved
    assert out.shape == (1, 3) and torch.allclose(out, torch.full((1, 3), 0.9 ** 10, dtype=torch.float64))
    options.set_option("ts_rk_type", "9z")
    with pytest.raises(_lib.PnError, match="unknown RK type"):
        ode.setupTS(y0, nn.Linear(3, 3).double())


def test_method_is_only_applied_when_the_shape_changes_like_the_reference():
    """pa.py:627-656: a second setupTS with another method on the same shape keeps the tableau."""
    options.set_option("ts_adapt_type", "none")
    y0 = torch.ones(3, dtype=torch.float64)
    f = TimeDependent(3)
    ode = petsc_adjoint.ODEPetsc(backend=CpuVecOps)
    ode.setupTS(y0, f, step_size=0.1, method="euler")
    assert ode._s == 1
    ode.setupTS(y0, f, step_size=0.1, method="rk4")
    assert ode._s == 1
    ode.setupTS(torch.ones(4, dtype=torch.float64), TimeDependent(4), step_size=0.1, method="rk4")
    assert ode._s == 4


def test_no_trajectory_without_adjoint_or_under_no_grad():
    options.set_option("ts_adapt_type", "none")
    f = SpiralFunc()
    y0 = torch.randn(5, 2, dtype=torch.float64)
    ode = petsc_adjoint.ODEPetsc(backend=CpuVecOps)
    ode.setupTS(y0, f, step_size=0.05, method="rk4", enable_adjoint=False)
    ode.odeint(y0, torch.tensor([0.5]))
    assert ode._traj is None and f.nfe == 40
    ode.setupTS(y0, f, step_size=0.05, method="rk4", enable_adjoint=True)
    with torch.no_grad():
        ode.odeint_adjoint(y0, torch.tensor([0.0, 0.5]))       # ode_demo_petsc.py:283-293 pattern
    assert ode._traj is None
    y = y0.clone().requires_grad_(True)
    ode.odeint_adjoint(y, torch.tensor([0.0, 0.5])).sum().backward()
    # solution-only default: the stage values of a reversed step are recomputed WITH autograd's tape (3 evaluations, counted
    # as forward evaluations), so only the last stage's VJP evaluates f again: NFE-B = 1 per step (-pn_reference_defaults: 4)
    assert y.grad is not None and ode.nfe_backward == 10


@pytest.mark.parametrize("method,adapt", [("rk4", "none"), ("dopri5", "basic"), ("bosh3", "basic"), ("midpoint", "none")])
def test_retain_graph_mode_is_bitwise_identical_and_skips_the_recompute(method, adapt):
    """-pn_trajectory_retain_graph 1 (extension): same numbers, no forward of f in the reverse sweep."""
    torch.manual_seed(0)
    y0 = torch.randn(20, 1, 2, dtype=torch.float64)
    t = torch.linspace(0.0, 25.0, 1001, dtype=torch.float64)[:10]
    target = torch.randn(10, 20, 1, 2, dtype=torch.float64)
    res = []
    for retain in (0, 1):
        options.clear()
        options.set_option("ts_adapt_type", adapt)
        options.set_option("ts_trajectory_solution_only", 0)
        if retain:
            options.set_option("pn_trajectory_retain_graph", 1)
        f = SpiralFunc()
        ode = petsc_adjoint.ODEPetsc(backend=CpuVecOps)
        ode.setupTS(y0, f, step_size=0.025, method=method)
        y = y0.clone().requires_grad_(True)
        p = ode.odeint_adjoint(y, t)
        torch.mean(torch.abs(p - target)).backward()
        res.append((p.detach(), y.grad.clone(), flat_grads(f).clone(), ode.nfe_backward, f.nfe, ode._tapes))
    a, b = res
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    assert a[3] > 0 and b[3] == 0 and b[4] < a[4]
    assert b[5] == {}                      # every tape was consumed and released


class FastClock(nn.Module):
    """f depends on t so sharply that a one-ulp change of t changes its value."""

    def __init__(self, d):
        super().__init__()
        g = torch.Generator().manual_seed(11)
        self.W = nn.Parameter(torch.randn(d, d, generator=g, dtype=torch.float64) * 0.3)

    def forward(self, t, y):
        import math
        return torch.tanh(y @ self.W) * math.cos(2.0e3 * t)


@pytest.mark.parametrize("method", ["dopri5", "bosh3"])
def test_recomputed_first_stage_uses_the_time_of_the_original_sweep(method):
    """First-same-as-last tableaus: the first stage derivative of step n is the last one of step n-1,
    evaluated at t_{n-1} + c_s h_{n-1} -- not t_n to the last bit (5dp's c_s is the row sum
    0.9999999999999998, and matched output times are set exactly).  A sweep that recomputes a step from a
    checkpoint evaluates it at that same time, so that an explicitly time-dependent f gives the store-all
    bits in every checkpoint mode."""
    torch.manual_seed(5)
    y0 = torch.randn(7, 9, dtype=torch.float64) * 0.5
    t = torch.tensor([0.0, 0.15428970145112675, 0.20295411058710836, 0.23250107590483735], dtype=torch.float64)
    target = torch.randn(4, 7, 9, dtype=torch.float64)
    res = []
    for extra in ({"ts_trajectory_solution_only": 0}, {"ts_trajectory_solution_only": 1},
                  {"ts_trajectory_max_cps_ram": 2, "ts_trajectory_solution_only": 1},
                  {"ts_trajectory_max_cps_ram": 2, "ts_trajectory_solution_only": 0}):
        options.clear()
        options.set_option("ts_rtol", 1e-5)
        options.set_option("ts_atol", 1e-5)
        for k, v in extra.items():
            options.set_option(k, v)
        f = FastClock(9)
        ode = petsc_adjoint.ODEPetsc(backend=CpuVecOps)
        ode.setupTS(y0, f, step_size=0.07, method=method)
        y = y0.clone().requires_grad_(True)
        p = ode.odeint_adjoint(y, t)
        torch.mean(torch.abs(p - target)).backward()
        res.append((p.detach().clone(), y.grad.clone(), flat_grads(f).clone()))
    for got in res[1:]:
        assert torch.equal(got[0], res[0][0]) and torch.equal(got[1], res[0][1]) and torch.equal(got[2], res[0][2])


def test_ts_type_on_the_command_line_overrides_the_method_keyword():
    """README.md:89 of the reference: "-ts_type cn will choose the Crank-Nicolson methods" -- the option
    database wins over setupTS's `method` (ts.setFromOptions, pa.py:775), in both directions; -ts_type theta
    takes -ts_theta_theta / -ts_theta_endpoint (PETSc's defaults 0.5 / off = implicit midpoint)."""
    from oracle.theta_oracle import odeint_adjoint_theta
    torch.manual_seed(0)
    y0 = torch.randn(5, 3, dtype=torch.float64)
    t = torch.tensor([0.0, 0.2, 0.5], dtype=torch.float64)
    tight = {"ts_adapt_type": "none", "snes_rtol": 1e-14, "snes_stol": 1e-15, "snes_atol": 1e-14, "ksp_rtol": 1e-13}

    def run(method, implicit_form, extra):
        options.clear()
        for k, v in dict(tight, **extra).items():
            options.set_option(k, v)
        f = TimeDependent(3)
        ode = petsc_adjoint.ODEPetsc(backend=CpuVecOps)
        ode.setupTS(y0, f, step_size=0.1, method=method, implicit_form=implicit_form)
        y = y0.clone().requires_grad_(True)
        p = ode.odeint_adjoint(y, t)
        p.pow(2).sum().backward()
        return p.detach().clone(), y.grad.clone(), flat_grads(f).clone(), ode

    a = run("cn", True, {})
    b = run("rk4", False, {"ts_type": "cn"})                      # explicit set-up, CN from the command line
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]) and b[3]._theta is not None
    c = run("cn", True, {"ts_type": "rk", "ts_rk_type": "4"})     # and back
    d = run("rk4", False, {})
    assert torch.equal(c[0], d[0]) and torch.equal(c[2], d[2]) and c[3]._theta is None
    for th, endpoint in ((0.5, False), (0.7, False), (0.5, True)):
        extra = {"ts_type": "theta", "ts_theta_theta": th}
        if endpoint:
            extra["ts_theta_endpoint"] = ""
        e = run("rk4", False, extra)
        f2 = TimeDependent(3)
        y2 = y0.clone().requires_grad_(True)
        p2 = odeint_adjoint_theta(f2, y2, t, 0.1, (th, endpoint))
        p2.pow(2).sum().backward()
        assert rel_err(e[0], p2) < 1e-12 and rel_err(e[1], y2.grad) < 1e-10 and rel_err(e[2], flat_grads(f2)) < 1e-10
    cn_by_theta = run("rk4", False, {"ts_type": "theta", "ts_theta_theta": 0.5, "ts_theta_endpoint": ""})
    assert torch.equal(cn_by_theta[0], a[0])


def test_unmodified_driver_preamble_runs_with_the_compat_shim(tmp_path):
    """The preamble of the reference's drivers (ode_demo_petsc.py:60-73, tests/test_pnode.py:24-35), verbatim in
    spirit: argparse leftovers go to petsc4py.init, PETSc.ScalarType is checked, pnode.petsc_adjoint is imported.
    With <repo>/compat on PYTHONPATH it runs without an edit and the options reach the solver."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import sys, argparse
parser = argparse.ArgumentParser(); parser.add_argument("--double_prec", action="store_true")
args, unknown = parser.parse_known_args()
import numpy as np
import petsc4py
sys.argv = [sys.argv[0]] + unknown
petsc4py.init(sys.argv)
from petsc4py import PETSc
assert PETSc.ScalarType == np.float64
from pnode import petsc_adjoint
import pnode_amd
db = pnode_amd.options.get_all()
assert db == {"ts_adapt_type": "none", "ts_trajectory_type": "memory", "ts_rk_type": "4"}, db
ode = petsc_adjoint.ODEPetsc()
print("DRIVER-OK", type(ode).__name__)
"""
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(root, "compat"), root, os.environ.get("PYTHONPATH", "")]))
    r = subprocess.run([sys.executable, "-c", code, "--double_prec", "-ts_adapt_type", "none", "-ts_trajectory_type", "memory",
                        "-ts_rk_type", "4"], capture_output=True, text=True, timeout=300, env=env, cwd=str(tmp_path))
    assert r.returncode == 0 and "DRIVER-OK ODEPetsc" in r.stdout, r.stdout + r.stderr


def test_setupTS_signature_is_the_references():
    """pa.py:534-550: same positional order, names and defaults (plus one trailing alias that one of the
    reference's own drivers still passes, KS.py:494)."""
    import inspect
    sig = inspect.signature(petsc_adjoint.ODEPetsc.setupTS)
    got = [(n, p.default) for n, p in sig.parameters.items() if n != "self"]
    assert got[:14] == [("u_tensor", inspect.Parameter.empty), ("func", inspect.Parameter.empty), ("step_size", 0.01),
                        ("enable_adjoint", True), ("implicit_form", False), ("use_dlpack", True), ("method", "dopri5"),
                        ("mass", None), ("imex_form", False), ("func2", None), ("batch_size", 1),
                        ("linear_solver", "petsc"), ("fixed_jacobian", False), ("matrixfree_jacobian", True)]
    assert [n for n, _ in got[14:]] == ["fixed_jacobian_across_solves"]
    assert list(inspect.signature(petsc_adjoint.ODEPetsc.odeint).parameters) == ["self", "u0", "t"]
    assert list(inspect.signature(petsc_adjoint.ODEPetsc.odeint_adjoint).parameters) == ["self", "y0", "t"]


def test_ts_view_prints_the_solver_state(capsys):
    options.set_option("ts_view", "")
    y0 = torch.randn(5, 2, dtype=torch.float64)
    ode = petsc_adjoint.ODEPetsc(backend=CpuVecOps)
    ode.setupTS(y0, SpiralTruth(), step_size=0.1, method="dopri5")
    ode.odeint_adjoint(y0.clone().requires_grad_(True), torch.tensor([0.5])).sum().backward()
    out = capsys.readouterr().out
    assert "TS Object (pnode_amd): type rk, order 5, 7 stages, first same as last, embedded error estimate" in out
    assert "adapt: basic, atol 0.0001 rtol 0.0001" in out and "total number of time steps=%d, rejected=%d" % (ode.num_steps, ode.num_rejections) in out
    assert "trajectory: every step, solution only" in out


def test_nfe_counts():
    """NFE-F / NFE-B as the reference's examples report them (spiral_unstable.py:326-347)."""
    options.set_option("ts_adapt_type", "none")
    y0 = torch.randn(5, 2, dtype=torch.float64)
    # store-all: every stage VJP re-evaluates f (no tapes on the CPU stand-in); solution-only: the three evaluations that
    # recompute a reversed step's stage values are taped and serve the VJPs of those stages, the last stage evaluates again;
    # with -pn_trajectory_retain_graph 0 (the reference's way) every VJP evaluates f: 4 per step
    for so, retain, expect_fwd, expect_bwd in [(0, "auto", 4 * 10, 4 * 10), (1, "auto", 4 * 10 + 3 * 10, 1 * 10), (1, 0, 4 * 10 + 3 * 10, 4 * 10)]:
        options.set_option("ts_trajectory_solution_only", so)
        options.set_option("pn_trajectory_retain_graph", retain)
        f = SpiralFunc()
        ode = petsc_adjoint.ODEPetsc(backend=CpuVecOps)
        ode.setupTS(y0, f, step_size=0.05, method="rk4")
        y = y0.clone().requires_grad_(True)
        ode.odeint_adjoint(y, torch.tensor([0.5])).sum().backward()
        assert ode.nfe_forward == expect_fwd and ode.nfe_backward == expect_bwd


@pytest.mark.parametrize("method,adapt,step", [("rk4", "none", 0.05), ("bosh3", "none", 0.05), ("dopri5", "basic", 0.1)])
def test_budgeted_checkpoints_hold_stage_values_when_asked_to(method, adapt, step):
    """-ts_trajectory_max_cps_ram c with -ts_trajectory_solution_only 0: a checkpoint carries the stage
    values of its step (as PETSc's do), so reversing a checkpointed step recomputes nothing.  Same bits
    as store-all for every budget; func evaluations never exceed the state-only budget's; with room
    for every step the sweep costs exactly what store-all costs."""
    torch.manual_seed(2)
    y0 = torch.randn(6, 2, dtype=torch.float64) * 0.5
    t = torch.tensor([0.0, 0.35, 0.9], dtype=torch.float64)
    target = torch.randn(3, 6, 2, dtype=torch.float64)

    def run(extra):
        options.clear()
        options.set_option("ts_adapt_type", adapt)
        if adapt != "none":
            options.set_option("ts_rtol", 1e-8)
            options.set_option("ts_atol", 1e-8)
        for k, v in extra.items():
            options.set_option(k, v)
        f = SpiralFunc() if method != "dopri5" else SpiralTruth()
        ode = petsc_adjoint.ODEPetsc(backend=CpuVecOps)
        ode.setupTS(y0, f, step_size=step, method=method)
        y = y0.clone().requires_grad_(True)
        p = ode.odeint_adjoint(y, t)
        torch.mean(torch.abs(p - target)).backward()
        return p.detach().clone(), y.grad.clone(), flat_grads(f).clone(), ode.nfe_forward, ode._nsteps

    ref = run({"ts_trajectory_solution_only": 0})
    nsteps = ref[4]
    assert nsteps >= 6
    for c in (1, 2, 3, 5, nsteps // 2, nsteps + 1, nsteps + 7):     # (adaptive sweeps also park the end state)
        with_stages = run({"ts_trajectory_solution_only": 0, "ts_trajectory_max_cps_ram": c})
        state_only = run({"ts_trajectory_solution_only": 1, "ts_trajectory_max_cps_ram": c})
        for got in (with_stages, state_only):
            assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1]) and torch.equal(got[2], ref[2]), c
        assert with_stages[3] <= state_only[3], c
        if c > nsteps:
            assert with_stages[3] == ref[3]                     # nothing recomputed at all
            assert state_only[3] > ref[3]                       # state-only checkpoints recompute every step's stages
    if adapt != "none":          # with rejected attempts in the sweep (default tolerances, first step far too long)
        def run_rej(extra):
            options.clear()
            for k, v in extra.items():
                options.set_option(k, v)
            f = SpiralTruth()
            ode = petsc_adjoint.ODEPetsc(backend=CpuVecOps)
            yy = torch.tensor([[2.0, 0.0], [1.5, 0.5], [1.0, -1.0]], dtype=torch.float64)
            ode.setupTS(yy, f, step_size=0.5, method=method)
            y = yy.clone().requires_grad_(True)
            p = ode.odeint_adjoint(y, torch.tensor([0.0, 1.0, 2.5], dtype=torch.float64))
            p.abs().mean().backward()
            return p.detach().clone(), y.grad.clone(), flat_grads(f).clone(), ode.num_rejections
        r0 = run_rej({"ts_trajectory_solution_only": 0})
        assert r0[3] > 0
        for c in (1, 2, 4, 7):
            got = run_rej({"ts_trajectory_solution_only": 0, "ts_trajectory_max_cps_ram": c})
            assert torch.equal(got[0], r0[0]) and torch.equal(got[1], r0[1]) and torch.equal(got[2], r0[2]) and got[3] == r0[3]
    few, many = run({"ts_trajectory_solution_only": 0, "ts_trajectory_max_cps_ram": 2})[3], \
        run({"ts_trajectory_solution_only": 0, "ts_trajectory_max_cps_ram": nsteps // 2})[3]
    assert many < few


# ---------------------------------------------------------------- implicit theta methods
@pytest.mark.parametrize("method", ["cn", "beuler"])
def test_theta_reference_known_answer_with_petsc_default_tolerances(method):
    """The reference's implicit test (tests/test_pnode.py:133-152) through the product's host
    logic (Newton + GMRES with PETSc's default tolerances, CPU stand-in for the kernels)."""
    from test_oracle_pins import Rober
    gold = json.load(open(os.path.join(GOLD, "rober.json")))
    t = torch.tensor(gold["t"], dtype=torch.float64)
    true_y = torch.tensor(gold["true_y"], dtype=torch.float64)
    for k, v in {"ts_adapt_type": "none", "ts_trajectory_type": "memory"}.items():
        options.set_option(k, v)
    f = Rober()
    ode = petsc_adjoint.ODEPetsc(backend=CpuVecOps)
    ode.setupTS(true_y[0], f, step_size=gold["step_size"], method=method, enable_adjoint=True, implicit_form=True)
    pred = ode.odeint_adjoint(true_y[0], t)
    loss = torch.mean(torch.abs(pred - true_y))
    loss.backward()
    std = torch.std(torch.abs(pred - true_y))
    G = gold["implicit_cn" if method == "cn" else "implicit_beuler"]
    if method == "cn":
        assert loss.item() == pytest.approx(1.85e-6, abs=1e-6) and std.item() == pytest.approx(3.36e-6, abs=1e-6)
    assert loss.item() == pytest.approx(G["loss"], rel=1e-6)
    assert rel_err(f.k.grad, torch.tensor(G["grad_k"], dtype=torch.float64)) < 1e-5      # ksp_rtol = 1e-5
    assert ode._theta.newton_its >= 3 and ode._theta.linear_its >= ode._theta.newton_its


@pytest.mark.parametrize("method", ["cn", "beuler"])
@pytest.mark.parametrize("with_mass", [False, True])
def test_theta_matches_exact_newton_oracle_at_tight_tolerances(method, with_mass):
    from oracle.theta_oracle import odeint_adjoint_theta
    torch.manual_seed(0)
    y0 = torch.randn(5, 3, dtype=torch.float64)
    t = torch.tensor([0.0, 0.3, 0.5, 1.0], dtype=torch.float64)
    target = torch.randn(4, 5, 3, dtype=torch.float64)
    # the reference applies the mass matrix as torch.matmul(mass, udot) on the state tensor (pa.py:430)
    Ms = torch.eye(5, dtype=torch.float64) + 0.1 * torch.randn(5, 5, dtype=torch.float64) if with_mass else None
    Mfull = torch.kron(Ms, torch.eye(3, dtype=torch.float64)) if with_mass else None
    for k, v in {"ts_adapt_type": "none", "snes_rtol": 1e-14, "snes_stol": 1e-15, "ksp_rtol": 1e-13}.items():
        options.set_option(k, v)
    f = TimeDependent(3)
    ode = petsc_adjoint.ODEPetsc(backend=CpuVecOps)
    ode.setupTS(y0, f, step_size=0.1, method=method, implicit_form=True, mass=Ms)
    y = y0.clone().requires_grad_(True)
    p = ode.odeint_adjoint(y, t)
    torch.mean(torch.abs(p - target)).backward()
    f2 = TimeDependent(3)
    y2 = y0.clone().requires_grad_(True)
    p2 = odeint_adjoint_theta(f2, y2, t, 0.1, method, mass=Mfull)
    torch.mean(torch.abs(p2 - target)).backward()
    assert ode._nsteps == 10 and ode.cur_sol_steps == [0, 3, 2, 5]
    assert rel_err(p, p2) < 1e-13 and rel_err(y.grad, y2.grad) < 1e-11 and rel_err(flat_grads(f), flat_grads(f2)) < 1e-11


def test_theta_ksponly_is_exact_for_linear_dynamics():
    """-snes_type ksponly (one Newton step, Burgers/run_a100_512.sh) solves a linear f exactly."""
    class Lin(nn.Module):
        def __init__(self):
            super().__init__()
            self.A = nn.Parameter(torch.tensor([[-1.0, 2.0], [-2.0, -0.5]], dtype=torch.float64))

        def forward(self, t, y):
            return y @ self.A.T

    for k, v in {"ts_adapt_type": "none", "snes_type": "ksponly", "ksp_rtol": 1e-14}.items():
        options.set_option(k, v)
    y0 = torch.tensor([[1.0, 0.5], [0.2, -1.0]], dtype=torch.float64)
    ode = petsc_adjoint.ODEPetsc(backend=CpuVecOps)
    ode.setupTS(y0, Lin(), step_size=0.1, method="beuler", implicit_form=True)
    out = ode.odeint_adjoint(y0, torch.tensor([0.3], dtype=torch.float64))
    A = torch.tensor([[-1.0, 2.0], [-2.0, -0.5]], dtype=torch.float64)
    step = torch.linalg.inv(torch.eye(2, dtype=torch.float64) - 0.1 * A)
    want = y0 @ torch.linalg.matrix_power(step, 3).T
    assert torch.allclose(out[0], want, rtol=1e-12, atol=1e-14) and ode._theta.newton_its == 3


def test_imex_reference_known_answer_with_petsc_default_tolerances():
    """The reference's IMEX test (tests/test_pnode.py:155-180) through the product's host logic."""
    from problems import RoberEX, RoberIM
    gold = json.load(open(os.path.join(GOLD, "rober.json")))
    t = torch.tensor(gold["t"], dtype=torch.float64)
    true_y = torch.tensor(gold["true_y"], dtype=torch.float64)
    options.set_option("ts_adapt_type", "none")
    fI, fE = RoberIM(), RoberEX()
    ode = petsc_adjoint.ODEPetsc(backend=CpuVecOps)
    ode.setupTS(true_y[0], fI, step_size=gold["step_size"], method="imex", enable_adjoint=True,
                implicit_form=True, imex_form=True, func2=fE)
    pred = ode.odeint_adjoint(true_y[0], t)
    loss = torch.mean(torch.abs(pred - true_y))
    loss.backward()
    std = torch.std(torch.abs(pred - true_y))
    assert loss.item() == pytest.approx(3.11e-6, abs=3e-6) and std.item() == pytest.approx(5.65e-6, abs=3e-6)
    assert loss.item() == pytest.approx(gold["imex_3"]["loss"], rel=1e-6)
    g = torch.cat([fI.k1.grad, fI.k3.grad, fE.k2.grad])            # flat order: implicit part first (pa.py:603-614)
    assert rel_err(g, torch.tensor(gold["imex_3"]["grad"], dtype=torch.float64)) < 1e-5
    assert ode.npIM == 2 and ode.npEX == 1 and ode.np == 3


@pytest.mark.parametrize("name", ["3", "4", "5", "l2", "ars122", "a2", "ars443", "1bee", "2c", "2d", "2e", "prssp2", "bpr3"])
@pytest.mark.parametrize("linear_solver", ["petsc", "torch"])
def test_imex_matches_oracle_on_a_burgers_like_split(name, linear_solver):
    """Stiff linear row-wise implicit part + nonlinear MLP explicit part; Newton-GMRES
    (linear_solver='petsc') and the direct LU of the single-sample Jacobian (linear_solver='torch',
    torch_linearsolve.py) must both give the oracle's numbers."""
    from oracle.arkimex_oracle import odeint_adjoint_arkimex
    from problems import DiffusionIM, ReactionEX
    torch.manual_seed(0)
    y0 = torch.randn(3, 6, dtype=torch.float64)
    t = torch.tensor([0.0, 0.1, 0.25], dtype=torch.float64)
    target = torch.randn(3, 3, 6, dtype=torch.float64)
    for k, v in {"ts_adapt_type": "none", "ts_arkimex_type": name, "snes_rtol": 1e-14, "snes_stol": 1e-15,
                 "ksp_rtol": 1e-13}.items():
        options.set_option(k, v)
    if linear_solver == "torch":
        options.set_option("snes_type", "ksponly")                   # Burgers/run_a100_512.sh
    fI, fE = DiffusionIM(6), ReactionEX(6)
    ode = petsc_adjoint.ODEPetsc(backend=CpuVecOps)
    ode.setupTS(y0, fI, step_size=0.05, method="imex", implicit_form=True, imex_form=True, func2=fE,
                batch_size=3, linear_solver=linear_solver, matrixfree_jacobian=False)
    y = y0.clone().requires_grad_(True)
    p = ode.odeint_adjoint(y, t)
    torch.mean(torch.abs(p - target)).backward()
    fI2, fE2 = DiffusionIM(6), ReactionEX(6)
    y2 = y0.clone().requires_grad_(True)
    p2 = odeint_adjoint_arkimex(fI2, fE2, y2, t, 0.05, name)
    torch.mean(torch.abs(p2 - target)).backward()
    assert ode._nsteps == 5 and ode.cur_sol_steps == [0, 2, 3]
    assert rel_err(p, p2) < 1e-12 and rel_err(y.grad, y2.grad) < 1e-10
    assert rel_err(flat_grads(fI), flat_grads(fI2)) < 1e-10 and rel_err(flat_grads(fE), flat_grads(fE2)) < 1e-10
    if linear_solver == "torch":
        assert ode._theta.linear_its == 0 and ode._theta.newton_its > 0      # no Krylov iteration at all


@pytest.mark.parametrize("method", ["rk4", "dopri5", "bosh3"])
def test_param_accum_per_step_equals_per_stage_bitwise(method):
    """-pn_param_accum step: the parameter sensitivities of all stages of a time step are added in
    one call (pn_param_accum_multi), in the same order as the per-stage calls: identical bits, and one
    call per time step instead of one per stage."""
    torch.manual_seed(3)
    y0 = torch.randn(6, 2, dtype=torch.float64)
    t = torch.tensor([0.0, 0.1, 0.25], dtype=torch.float64)
    res = {}
    for mode in ("stage", "step"):
        options.clear()
        options.set_option("pn_linear_param_grads", 0)      # (these tests count pn_param_accum calls: autograd's gradients)
        options.set_option("ts_adapt_type", "none")
        options.set_option("pn_param_accum", mode)
        f = SpiralFunc(torch.float64)
        ode = petsc_adjoint.ODEPetsc(backend=CpuVecOps)
        ode.setupTS(y0, f, step_size=0.05, method=method)
        y = y0.clone().requires_grad_(True)
        ode.odeint_adjoint(y, t).abs().mean().backward()
        res[mode] = (flat_grads(f).clone(), y.grad.clone(), ode._ops.calls["param_accum"], ode._nsteps, ode._s_eff)
    assert torch.equal(res["stage"][0], res["step"][0]) and torch.equal(res["stage"][1], res["step"][1])
    nsteps, s_eff = res["step"][3], res["step"][4]
    assert res["step"][2] == nsteps and res["stage"][2] == nsteps * s_eff


@pytest.mark.parametrize("method", ["beuler", "cn"])
@pytest.mark.parametrize("rowwise", [False, True])
def test_singular_mass_matrix_index1_dae(method, rowwise):
    """setupTS(..., implicit_form=True, mass=M) with a SINGULAR M (the reference's pendulum_DAE.py use):
    forward and discrete adjoint against autograd through the unrolled scheme.  `mass` acts on the
    state as in the reference (pa.py:426-431: numel x numel on the flat state, or matmul(M, state));
    a (d x d) matrix acting on every batch row is accepted as an extension and gives the same numbers."""
    from oracle.theta_oracle import odeint_unrolled_theta
    from problems import SemiExplicitDAE
    torch.manual_seed(1)
    f0 = SemiExplicitDAE()
    u0 = f0.consistent(torch.randn(4, 3, dtype=torch.float64))
    M5 = SemiExplicitDAE.mass()
    Mflat = torch.kron(torch.eye(4, dtype=torch.float64), M5)
    t = torch.tensor([0.0, 0.2, 0.5], dtype=torch.float64)
    target = torch.randn(3, 4, 5, dtype=torch.float64)
    for k, v in {"ts_adapt_type": "none", "snes_rtol": 1e-14, "snes_stol": 1e-15, "snes_atol": 1e-14, "ksp_rtol": 1e-13}.items():
        options.set_option(k, v)
    f = SemiExplicitDAE()
    ode = petsc_adjoint.ODEPetsc(backend=CpuVecOps)
    ode.setupTS(u0, f, step_size=0.1, method=method, implicit_form=True, mass=M5 if rowwise else Mflat)
    u = u0.clone().requires_grad_(True)
    p = ode.odeint_adjoint(u, t)
    torch.mean(torch.abs(p - target)).backward()
    f2 = SemiExplicitDAE()
    u2 = u0.clone().requires_grad_(True)
    p2 = odeint_unrolled_theta(f2, u2, t, 0.1, method, newton_its=1, mass=Mflat)
    torch.mean(torch.abs(p2 - target)).backward()
    assert rel_err(p, p2) < 1e-13 and rel_err(u.grad, u2.grad) < 1e-11 and rel_err(flat_grads(f), flat_grads(f2)) < 1e-11
    # the algebraic constraint holds along the trajectory
    assert (p[..., :3] @ f.C.detach() - p[..., 3:]).abs().max() < 1e-12
    with pytest.raises(ValueError, match="mass is 7x7"):
        bad = petsc_adjoint.ODEPetsc(backend=CpuVecOps)
        bad.setupTS(u0, f, step_size=0.1, method=method, implicit_form=True, mass=torch.eye(7, dtype=torch.float64))
        bad.odeint_adjoint(u0.clone().requires_grad_(True), t)


def test_fixed_jacobian_keeps_the_factors_only_for_a_parameter_free_implicit_part(monkeypatch):
    """fixed_jacobian=True (pa.py:582: "the Jacobian is constant across ODE solves"): the LU factors
    are computed once when funcIM has no trainable parameter, and at every solve (as the reference
    does, pa.py:792-799) otherwise; results are the same either way."""
    from pnode_amd import arkimex
    from problems import DiffusionIM, ReactionEX
    calls = []
    orig = arkimex.ArkimexStepper._jacobian
    monkeypatch.setattr(arkimex.ArkimexStepper, "_jacobian", lambda self, t, u: (calls.append(1), orig(self, t, u))[1])
    y0 = torch.randn(3, 6, dtype=torch.float64)
    t = torch.tensor([0.0, 0.1], dtype=torch.float64)
    res = {}
    for frozen in (False, True):
        for fixed in (False, True):
            options.clear()
            for k, v in {"ts_adapt_type": "none", "snes_type": "ksponly", "pn_affine_vjp": 0}.items():     # (the shortcut that
                options.set_option(k, v)                     # fixed_jacobian also allows has its own test, below)
            fI, fE = DiffusionIM(6), ReactionEX(6)
            fI.nu.requires_grad_(not frozen)
            ode = petsc_adjoint.ODEPetsc(backend=CpuVecOps)
            ode.setupTS(y0, fI, step_size=0.05, method="imex", implicit_form=True, imex_form=True, func2=fE,
                        batch_size=3, linear_solver="torch", fixed_jacobian=fixed, matrixfree_jacobian=False)
            del calls[:]
            for _ in range(3):
                y = y0.clone().requires_grad_(True)
                out = ode.odeint_adjoint(y, t)
                out.abs().mean().backward()
            res[(frozen, fixed)] = (out.detach().clone(), y.grad.clone(), len(calls))
    assert res[(True, True)][2] == 1 and res[(True, False)][2] == 3
    assert res[(False, True)][2] == 3 and res[(False, False)][2] == 3
    for frozen in (False, True):
        assert torch.equal(res[(frozen, True)][0], res[(frozen, False)][0])
        assert torch.equal(res[(frozen, True)][1], res[(frozen, False)][1])


@pytest.mark.parametrize("name,adapt", [("3", "none"), ("l2", "none"), ("5", "none"), ("3", "basic")])
def test_imex_keeps_the_stage_tapes_of_the_explicit_part_for_the_reverse_sweep(name, adapt):
    """Round 5: with store-all trajectories the ARKIMEX forward sweep keeps the autograd tape of every stage evaluation of
    funcEX (-pn_trajectory_retain_graph 1; `auto` on a HIP device while they fit), as the explicit RK path does: the reverse
    sweep runs only the backward half of those VJPs -- the reference re-evaluates func inside every multTranspose
    (pa.py:66-68).  Same bits; funcEX is not called in the reverse sweep; rejected attempts of an adaptive solve leave no tape."""
    from problems import DiffusionIM, ReactionEX

    class CountedEX(ReactionEX):
        calls = 0

        def forward(self, t, y):
            type(self).calls += 1
            return super().forward(t, y)
    torch.manual_seed(4)
    y0 = torch.randn(3, 6, dtype=torch.float64) * (1.5 if adapt == "basic" else 1.0)
    t = torch.tensor([0.0, 0.1, 0.3], dtype=torch.float64)
    w = torch.randn(3, 3, 6, dtype=torch.float64)
    res = {}
    for retain in (0, 1):
        options.clear()
        opts = {"ts_adapt_type": adapt, "ts_arkimex_type": name, "ts_trajectory_solution_only": 0, "pn_trajectory_retain_graph": retain,
                "snes_rtol": 1e-12, "ksp_rtol": 1e-12}
        if adapt == "basic":
            opts.update({"ts_rtol": 1e-6, "ts_atol": 1e-6})
        for k, v in opts.items():
            options.set_option(k, v)
        torch.manual_seed(9)
        fI, fE = DiffusionIM(6), CountedEX(6)
        ode = petsc_adjoint.ODEPetsc(backend=CpuVecOps)
        ode.setupTS(y0, fI, step_size=0.25 if adapt == "basic" else 0.05, method="imex", implicit_form=True, imex_form=True, func2=fE, batch_size=3)
        options.clear()
        CountedEX.calls = 0
        y = y0.clone().requires_grad_(True)
        out = ode.odeint_adjoint(y, t)
        fwd = CountedEX.calls
        assert (ode._tapes is not None and len(ode._tapes) == ode._nsteps) if retain else ode._tapes is None
        (out * w).sum().backward()
        res[retain] = (out.detach(), y.grad.clone(), flat_grads(fI).clone(), flat_grads(fE).clone(), fwd, CountedEX.calls - fwd,
                       ode._nsteps, ode.num_rejections)
        assert not ode._tapes                                # all consumed
    a, b = res[0], res[1]
    assert all(torch.equal(x, z) for x, z in zip(a[:4], b[:4])) and a[6:] == b[6:]
    assert a[4] == b[4] and a[5] > 0 and b[5] == 0
    if adapt == "basic":
        assert a[7] > 0


@pytest.mark.parametrize("name", ["3", "l2", "4"])
def test_affine_implicit_part_with_a_declared_constant_jacobian_is_differentiated_by_one_product(name):
    """Round 5, BASELINE config 5's shape (fixed linear funcIM, reference examples-sinode/Burgers/Burgers.py:170-195 with
    fixed_linear=True; `fixed_jacobian` of pa.py:582): when the user declares the Jacobian constant, funcIM has no trainable
    parameter and passes the affinity check against the kept one-sample Jacobian, the reverse sweep forms J^T w with a dense
    product instead of calling funcIM's autograd, and evaluates funcIM as Y J^T + funcIM(t, one zero row) -- same states and
    gradients to round-off, funcIM called on one row, in the forward sweep only.
    A funcIM that is NOT affine (the declaration is wrong), one with a trainable parameter, `-pn_affine_vjp 0` and
    `-pn_reference_defaults 1` keep the autograd path, bit for bit."""
    import torch.nn as nn
    from problems import AdvectionDiffusionIM, ReactionEX

    class Counted(AdvectionDiffusionIM):                      # nonsymmetric J: J and J^T are told apart
        calls = 0

        def forward(self, t, y):
            type(self).calls += 1
            return super().forward(t, y) + 0.3 * torch.sin(torch.tensor(t, dtype=y.dtype))     # affine: a forcing c(t)

    class Bent(Counted):
        def forward(self, t, y):
            return super().forward(t, y) + 1e-3 * y ** 2      # not affine: the check must refuse

    torch.manual_seed(3)
    n = 8
    y0 = torch.randn(5, n, dtype=torch.float64)
    t = torch.tensor([0.0, 0.1, 0.25], dtype=torch.float64)
    w = torch.randn(3, 5, n, dtype=torch.float64)

    def run(cls, opts, fixed=True, trainable=False):
        options.clear()
        for k, v in dict({"ts_adapt_type": "none", "snes_type": "ksponly", "ts_arkimex_type": name,
                          "ts_trajectory_solution_only": 0}, **opts).items():      # (stage values kept: nothing is re-solved)
            options.set_option(k, v)
        torch.manual_seed(11)
        fI, fE = cls(n), ReactionEX(n)
        for p in fI.parameters():
            p.requires_grad_(trainable)
        ode = petsc_adjoint.ODEPetsc(backend=CpuVecOps)
        ode.setupTS(y0, fI, step_size=0.05, method="imex", implicit_form=True, imex_form=True, func2=fE, batch_size=5,
                    linear_solver="torch", fixed_jacobian=fixed, matrixfree_jacobian=False)
        options.clear()
        cls.calls = 0
        y = y0.clone().requires_grad_(True)
        out = ode.odeint_adjoint(y, t)
        fwd_calls = cls.calls
        (out * w).sum().backward()
        return out.detach(), y.grad.clone(), flat_grads(fE).clone(), ode._theta._affine, fwd_calls, cls.calls - fwd_calls

    base = run(Counted, {"pn_affine_vjp": 0})
    fast = run(Counted, {})
    assert base[3] is False and fast[3] is True
    assert rel_err(fast[0], base[0]) < 1e-14 and rel_err(fast[1], base[1]) < 1e-13 and rel_err(fast[2], base[2]) < 1e-13
    # funcIM itself: the same number of calls in the forward sweep (+ the six of the check, once), each of them on ONE row
    # (f(t, Y) = Y J^T + f(t, 0-row)); none in the reverse sweep
    assert base[5] > 0 and fast[5] == 0 and fast[4] == base[4] + 6
    for other in (run(Bent, {}), run(Counted, {}, fixed=False), run(Counted, {}, trainable=True), run(Counted, {"pn_reference_defaults": 1})):
        assert other[3] is False and other[5] > 0
    bent0 = run(Bent, {"pn_affine_vjp": 0})
    bent = run(Bent, {})
    assert torch.equal(bent[1], bent0[1]) and torch.equal(bent[2], bent0[2])


@pytest.mark.parametrize("name", ["3", "l2"])
def test_imex_direct_solve_with_a_nonsymmetric_implicit_operator(name):
    """linear_solver="torch": rows are solved against the LU of shift*I - J with lu_solve(left=False), the forward
    stage solve needing the adjoint flag and the discrete adjoint the plain one (the reference's PCShell has them the
    other way round, torch_linearsolve.py:24-35, over PETSc's column-major storage).  Only a nonsymmetric J tells the
    two apart: upwind advection + diffusion against the oracle's dense exact solves."""
    from oracle.arkimex_oracle import odeint_adjoint_arkimex
    from problems import AdvectionDiffusionIM, ReactionEX
    torch.manual_seed(0)
    n = 8
    y0 = torch.randn(3, n, dtype=torch.float64)
    t = torch.tensor([0.0, 0.1, 0.25], dtype=torch.float64)
    target = torch.randn(3, 3, n, dtype=torch.float64)
    for k, v in {"ts_adapt_type": "none", "ts_arkimex_type": name, "snes_type": "ksponly"}.items():
        options.set_option(k, v)
    fI, fE = AdvectionDiffusionIM(n), ReactionEX(n)
    ode = petsc_adjoint.ODEPetsc(backend=CpuVecOps)
    ode.setupTS(y0, fI, step_size=0.05, method="imex", implicit_form=True, imex_form=True, func2=fE, batch_size=3,
                linear_solver="torch", matrixfree_jacobian=False)
    y = y0.clone().requires_grad_(True)
    p = ode.odeint_adjoint(y, t)
    torch.mean(torch.abs(p - target)).backward()
    fI2, fE2 = AdvectionDiffusionIM(n), ReactionEX(n)
    y2 = y0.clone().requires_grad_(True)
    p2 = odeint_adjoint_arkimex(fI2, fE2, y2, t, 0.05, name)
    torch.mean(torch.abs(p2 - target)).backward()
    g = torch.cat([q.grad.reshape(-1) for q in list(fI.parameters()) + list(fE.parameters())])
    g2 = torch.cat([q.grad.reshape(-1) for q in list(fI2.parameters()) + list(fE2.parameters())])
    assert rel_err(p, p2) < 1e-13 and rel_err(y.grad, y2.grad) < 1e-12 and rel_err(g, g2) < 1e-12


@pytest.mark.parametrize("method", ["cn", "beuler"])
@pytest.mark.parametrize("with_mass", [False, True])
def test_theta_methods_with_the_direct_linear_solver(method, with_mass):
    """implicit_form=True, linear_solver="torch" (torch_linearsolve.py): stage systems solved with the LU of
    shift*M - J, J = d func/du of one sample, frozen for the solve, rows of the batch as right-hand sides -- forward
    stage solves and the transposed ones of the discrete adjoint.  For a func that is linear in u (nonsymmetric
    advection-diffusion, trainable speeds) that is exact, with one Newton step (-snes_type ksponly): forward and
    gradients equal the oracle's exact-Newton solve; no Krylov iteration happens."""
    from oracle.theta_oracle import odeint_adjoint_theta
    from problems import AdvectionDiffusionIM
    torch.manual_seed(0)
    n, B = 8, 3
    y0 = torch.randn(B, n, dtype=torch.float64)
    t = torch.tensor([0.0, 0.1, 0.25], dtype=torch.float64)
    target = torch.randn(3, B, n, dtype=torch.float64)
    M = (torch.eye(n, dtype=torch.float64) + 0.2 * torch.randn(n, n, dtype=torch.float64)) if with_mass else None
    for k, v in {"ts_adapt_type": "none", "snes_type": "ksponly"}.items():
        options.set_option(k, v)
    f = AdvectionDiffusionIM(n)
    ode = petsc_adjoint.ODEPetsc(backend=CpuVecOps)
    ode.setupTS(y0, f, step_size=0.05, method=method, implicit_form=True, mass=M, batch_size=B, linear_solver="torch",
                matrixfree_jacobian=False)
    y = y0.clone().requires_grad_(True)
    p = ode.odeint_adjoint(y, t)
    torch.mean(torch.abs(p - target)).backward()
    f2 = AdvectionDiffusionIM(n)
    y2 = y0.clone().requires_grad_(True)
    Mfull = None if M is None else torch.kron(torch.eye(B, dtype=torch.float64), M)
    p2 = odeint_adjoint_theta(f2, y2, t, 0.05, method, mass=Mfull)
    torch.mean(torch.abs(p2 - target)).backward()
    assert rel_err(p, p2) < 1e-12 and rel_err(y.grad, y2.grad) < 1e-11 and rel_err(flat_grads(f), flat_grads(f2)) < 1e-11
    assert ode._theta.linear_its == 0 and ode._theta.newton_its > 0


def test_direct_theta_solver_on_a_nonlinear_func_is_a_modified_newton():
    """With a nonlinear func the frozen one-sample Jacobian makes the stage solve a modified Newton iteration
    (as in the reference, pa.py:474-508): it converges to the same stage values as the matrix-free exact Newton."""
    torch.manual_seed(1)
    y0 = torch.randn(4, 3, dtype=torch.float64) * 0.3
    t = torch.tensor([0.0, 0.3], dtype=torch.float64)
    sols = {}
    for ls in ("petsc", "torch"):
        options.clear()
        for k, v in {"ts_adapt_type": "none", "snes_rtol": 1e-13, "snes_stol": 1e-14, "snes_atol": 1e-13, "ksp_rtol": 1e-13,
                     "snes_max_it": 200}.items():
            options.set_option(k, v)
        f = TimeDependent(3)
        ode = petsc_adjoint.ODEPetsc(backend=CpuVecOps)
        ode.setupTS(y0, f, step_size=0.1, method="cn", implicit_form=True, batch_size=4, linear_solver=ls)
        with torch.no_grad():
            sols[ls] = (ode.odeint_adjoint(y0, t).clone(), ode._theta.newton_its, ode._theta.linear_its)
    assert rel_err(sols["torch"][0], sols["petsc"][0]) < 1e-11
    assert sols["torch"][2] == 0 and sols["torch"][1] >= sols["petsc"][1]


def test_imex_without_adapt_none_adapts_or_warns():
    """PETSc's ARKIMEX adapts its steps unless -ts_adapt_type none is given (every IMEX run of the reference gives it).
    The default type (3) has embedded weights and adapts here too; a type without them takes fixed steps and says so
    (test_arkimex_without_embedded_weights_warns_and_takes_fixed_steps); the theta methods take fixed steps unless
    -ts_adapt_type basic is given explicitly (TSCreate_Theta's default adapt type is none)."""
    import warnings as _w
    ode = petsc_adjoint.ODEPetsc(backend=CpuVecOps)
    with _w.catch_warnings():
        _w.simplefilter("error")
        _w.simplefilter("ignore", petsc_adjoint.PnUnpinnedWarning)
        ode.setupTS(torch.zeros(3, dtype=torch.float64), nn.Linear(3, 3).double(), method="imex",
                    implicit_form=True, imex_form=True, func2=nn.Linear(3, 3).double())
    assert ode._adaptive
    ode = petsc_adjoint.ODEPetsc(backend=CpuVecOps)
    with _w.catch_warnings():
        _w.simplefilter("error")
        ode.setupTS(torch.zeros(3, dtype=torch.float64), nn.Linear(3, 3).double(), method="cn", implicit_form=True)
    assert not ode._adaptive                 # theta methods: fixed steps unless -ts_adapt_type basic is explicit


def test_imex_unavailable_tableaus_and_missing_func2():
    options.set_option("ts_arkimex_type", "dirk75")        # not an ARKIMEX type of PETSc's either
    ode = petsc_adjoint.ODEPetsc(backend=CpuVecOps)
    with pytest.raises(_lib.PnError, match="not available"):
        ode.setupTS(torch.zeros(3, dtype=torch.float64), nn.Linear(3, 3).double(), method="imex",
                    implicit_form=True, imex_form=True, func2=nn.Linear(3, 3).double())
    options.clear()
    with pytest.raises(ValueError, match="func2 must be provided"):
        ode.setupTS(torch.zeros(3, dtype=torch.float64), nn.Linear(3, 3).double(), method="imex", imex_form=True)


def test_gmres_core_against_dense_solve():
    """pn_gmres_* (Hessenberg + Givens) driven with numpy Arnoldi on a random matrix."""
    lib = _lib.load()
    rng = np.random.default_rng(3)
    n = 12
    A = np.eye(n) * 3 + rng.standard_normal((n, n)) * 0.5
    b = rng.standard_normal(n)
    g = ctypes.c_void_p(lib.pn_gmres_create(n))
    beta = np.linalg.norm(b)
    V = [b / beta]
    _lib.check(lib.pn_gmres_begin(g, beta))
    res = ctypes.c_double()
    for k in range(n):
        w = A @ V[k]
        h = [float(w @ v) for v in V]
        for hj, v in zip(h, V):
            w = w - hj * v
        hk1 = float(np.linalg.norm(w))
        _lib.check(lib.pn_gmres_column(g, k, (ctypes.c_double * (k + 2))(*(h + [hk1])), ctypes.byref(res)))
        x_k = None
        y = (ctypes.c_double * (k + 1))()
        _lib.check(lib.pn_gmres_solve(g, k, y))
        x_k = sum(yi * vi for yi, vi in zip(y, V))
        assert np.linalg.norm(b - A @ x_k) == pytest.approx(res.value, rel=1e-8, abs=1e-12)
        if hk1 < 1e-13:
            break
        V.append(w / hk1)
    assert np.allclose(x_k, np.linalg.solve(A, b), rtol=1e-9)
    lib.pn_gmres_destroy(g)


# ---------------------------------------------------------------- controller unit tests
def _ts(**opt):
    lib = _lib.load()
    ts = ctypes.c_void_p(lib.pn_ts_create())
    for k, v in opt.items():
        _lib.check(lib.pn_ts_set_option(ts, k.encode(), str(v).encode()))
    return lib, ts


def test_controller_formula_and_clipping():
    lib, ts = _ts(ts_rk_type="5dp")
    span = (ctypes.c_double * 1)(100.0)
    _lib.check(lib.pn_ts_begin(ts, 0.0, 0.1, 1, span))
    acc, hit, done = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    t, h = ctypes.c_double(), ctypes.c_double()
    _lib.check(lib.pn_ts_judge(ts, 0.5, ctypes.byref(acc), ctypes.byref(hit), ctypes.byref(done)))
    lib.pn_ts_attempt(ts, ctypes.byref(t), ctypes.byref(h))
    assert acc.value == 1 and t.value == pytest.approx(0.1) and h.value == pytest.approx(0.1 * 0.9 * 0.5 ** (-1 / 5))
    h0 = h.value
    _lib.check(lib.pn_ts_judge(ts, 32.0, ctypes.byref(acc), ctypes.byref(hit), ctypes.byref(done)))   # reject
    lib.pn_ts_attempt(ts, ctypes.byref(t), ctypes.byref(h))
    assert acc.value == 0 and t.value == pytest.approx(0.1) and h.value == pytest.approx(h0 * 0.9 * 32 ** (-0.2))
    h1 = h.value
    _lib.check(lib.pn_ts_judge(ts, 2.0, ctypes.byref(acc), ctypes.byref(hit), ctypes.byref(done)))    # 2nd reject
    lib.pn_ts_attempt(ts, ctypes.byref(t), ctypes.byref(h))
    assert acc.value == 0 and h.value == pytest.approx(h1 * 0.45 * 2 ** (-0.2))                         # safety halved
    h2 = h.value
    _lib.check(lib.pn_ts_judge(ts, 0.0, ctypes.byref(acc), ctypes.byref(hit), ctypes.byref(done)))    # e = 0 -> x10
    lib.pn_ts_attempt(ts, ctypes.byref(t), ctypes.byref(h))
    assert acc.value == 1 and h.value == pytest.approx(10 * h2)
    _lib.check(lib.pn_ts_judge(ts, 1e12, ctypes.byref(acc), ctypes.byref(hit), ctypes.byref(done)))   # clip at 0.1
    lib.pn_ts_attempt(ts, ctypes.byref(t), ctypes.byref(h))
    assert acc.value == 0 and h.value == pytest.approx(h2)
    assert lib.pn_ts_rejections(ts) == 3 and lib.pn_ts_steps(ts) == 2
    assert lib.pn_ts_judge(ts, float("nan"), ctypes.byref(acc), ctypes.byref(hit), ctypes.byref(done)) != 0
    assert b"not-a-number" in lib.pn_last_error()
    lib.pn_ts_destroy(ts)


def test_max_reject_guard():
    lib, ts = _ts(ts_rk_type="3bs", ts_max_reject=3)
    span = (ctypes.c_double * 1)(1.0)
    _lib.check(lib.pn_ts_begin(ts, 0.0, 0.1, 1, span))
    acc, hit, done = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    rcs = [lib.pn_ts_judge(ts, 5.0, ctypes.byref(acc), ctypes.byref(hit), ctypes.byref(done)) for _ in range(4)]
    assert rcs == [0, 0, 0, 1] and b"ts_max_reject" in lib.pn_last_error()
    lib.pn_ts_destroy(ts)


def test_fixed_step_hits_every_span_point_exactly():
    lib, ts = _ts(ts_adapt_type="none", ts_rk_type="4")
    times = [0.0, 0.1, 0.25, 0.3, 1.0]
    span = (ctypes.c_double * 5)(*times)
    _lib.check(lib.pn_ts_begin(ts, 0.0, 0.04, 5, span))
    acc, hit, done = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(0)
    hits = []
    while not done.value:
        _lib.check(lib.pn_ts_judge(ts, -1.0, ctypes.byref(acc), ctypes.byref(hit), ctypes.byref(done)))
        if hit.value >= 0:
            hits.append((hit.value, lib.pn_ts_time(ts)))
    assert [i for i, _ in hits] == [1, 2, 3, 4]
    assert [tt for _, tt in hits] == times[1:]            # bit-exact landing
    lib.pn_ts_destroy(ts)


# ---------------------------------------------------------------- checkpoint scheduler
def _simulate(mode, budget, nsteps, known_total=False, carry=False):
    """Drive the scheduler like a forward + reverse sweep; returns (#re-advanced steps, high water), or with `carry`
    (checkpoints hold their step's stage values, written whenever a sweep steps on from a kept state)
    (#re-advanced steps + #reversed steps whose stage values had to be computed, high water)."""
    lib = _lib.load()
    tj = ctypes.c_void_p(lib.pn_traj_create())
    _lib.check(lib.pn_traj_begin(tj, mode, budget))
    if carry:
        _lib.check(lib.pn_traj_set_carry(tj, 1))
    if known_total:
        _lib.check(lib.pn_traj_set_total(tj, nsteps))
    content = {}                                  # slot -> step whose start state it holds
    has_stages = {}                               # slot -> step whose stage values it holds
    for step in range(nsteps + 1):                # the state after the last step also gets a home
        slot = lib.pn_traj_fwd_slot(tj, step)
        if slot >= 0:
            content[slot] = step
            has_stages.pop(slot, None)
            if step < nsteps:
                has_stages[slot] = step           # the sweep steps on from it
        if mode == _lib.PN_TRAJ_BUDGET:
            assert lib.pn_traj_slots_in_use(tj) <= budget
    readv = 0
    stage_work = 0
    cap = 64
    fs, fl, ns = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int()
    ss, sl = (ctypes.c_int64 * cap)(), (ctypes.c_int64 * cap)()
    for step in range(nsteps - 1, -1, -1):
        _lib.check(lib.pn_traj_rev_plan(tj, step, ctypes.byref(fs), ctypes.byref(fl), ctypes.byref(ns), ss, sl, cap))
        assert fs.value <= step and content[fl.value] == fs.value, "plan points at a slot with other content"
        readv += step - fs.value
        if fs.value < step:
            has_stages[fl.value] = fs.value       # the re-advance steps on from the checkpoint it starts at ...
        prev = fs.value
        for k in range(ns.value):
            assert prev < ss[k] < step
            prev = ss[k]
            content[sl[k]] = ss[k]
            has_stages[sl[k]] = ss[k]             # ... and from every state it stores on the way
        if not (fs.value == step and has_stages.get(fl.value) == step):
            stage_work += 1                       # the reversed step's stage values have to be computed
        if mode == _lib.PN_TRAJ_BUDGET:
            assert lib.pn_traj_slots_in_use(tj) <= budget
        _lib.check(lib.pn_traj_rev_done(tj, step))
    hw = lib.pn_traj_high_water(tj)
    lib.pn_traj_destroy(tj)
    return (readv + stage_work if carry else readv), hw


def test_scheduler_unbounded_modes_never_recompute():
    for mode in (_lib.PN_TRAJ_ALL, _lib.PN_TRAJ_SOLUTION):
        readv, hw = _simulate(mode, 0, 137)
        assert readv == 0 and hw == 138


@pytest.mark.parametrize("nsteps", [1, 2, 9, 100, 1000])
@pytest.mark.parametrize("budget", [1, 2, 3, 7, 50])
def test_scheduler_budget_is_respected_and_every_step_is_reachable(nsteps, budget):
    readv, hw = _simulate(_lib.PN_TRAJ_BUDGET, budget, nsteps)
    assert hw <= budget
    if budget >= nsteps + 1:
        assert readv == 0
    if budget == 1:
        assert readv == nsteps * (nsteps - 1) // 2          # everything from step 0
    if budget >= 3 and nsteps >= 100:
        assert readv < nsteps * nsteps // (2 * (budget - 1))  # far better than restart-from-0


def _optimal_tables():
    """Brute-force dynamic programme of the re-advance count (independent of the C++ one)."""
    from functools import lru_cache

    @lru_cache(None)
    def cost(l, c):
        if l <= 1:
            return 0
        if c == 1:
            return l * (l - 1) // 2
        return min(m + cost(l - m, c - 1) + cost(m, c) for m in range(1, l))

    @lru_cache(None)
    def first(l, c):
        if l <= 1:
            return 0
        if c == 1:
            return l * (l - 1) // 2
        return min(first(l - m, c - 1) + cost(m, c) for m in range(1, l))
    return cost, first


def _optimal_tables_priced():
    """Brute force of the cost model for checkpoints that carry their step's stage values (CAMS-type): unit = one step of
    stage computation; a reversed step costs one unit unless its checkpoint holds its stages, which every sweep writes
    for free when it steps on from a kept state.  Independent of the C++ tables (full minimisation over every split)."""
    from functools import lru_cache

    @lru_cache(None)
    def cost(l, c):          # segment whose start checkpoint has just been stepped on from
        if l <= 1:
            return 0
        if c == 1:
            return l * (l - 1) // 2 + (l - 1)
        return min(m + (1 if l - m == 1 else cost(l - m, c - 1)) + cost(m, c) for m in range(1, l))

    @lru_cache(None)
    def first(l, c):
        if l <= 1:
            return 0
        if c == 1:
            return cost(l, 1)
        return min(first(l - m, c - 1) + cost(m, c) for m in range(1, l))
    return cost, first


@pytest.mark.parametrize("nsteps,budget", [(2, 1), (3, 2), (10, 3), (30, 2), (41, 4), (60, 5), (100, 3), (100, 10), (100, 50), (97, 7), (120, 13)])
def test_scheduler_with_stage_carrying_checkpoints_is_optimal_for_the_priced_cost(nsteps, budget):
    """-ts_trajectory_max_cps_ram with -ts_trajectory_solution_only 0 (checkpoints hold stage values, as PETSc's do): the
    placement minimises re-advanced steps PLUS the stage computations of the reversed steps (CAMS-type cost), and the work
    of the simulated sweep equals the brute-force optimum of that cost; it is never more than what the revolve-type
    placement costs under the same accounting."""
    _, first = _optimal_tables_priced()
    work, hw = _simulate(_lib.PN_TRAJ_BUDGET, budget, nsteps, known_total=True, carry=True)
    assert hw <= budget and work == first(nsteps, budget), (work, first(nsteps, budget))
    online, _ = _simulate(_lib.PN_TRAJ_BUDGET, budget, nsteps, known_total=False, carry=True)
    assert work <= online


@pytest.mark.parametrize("nsteps,budget", [(2, 1), (10, 3), (30, 2), (41, 4), (60, 5), (100, 3), (100, 10), (100, 50), (97, 7)])
def test_scheduler_is_optimal_when_the_sweep_length_is_known(nsteps, budget):
    """Fixed-step solves announce their length (pn_traj_set_total): the number of re-advanced
    steps equals the brute-force optimum of the checkpointing recursion (revolve-type schedule)."""
    _, first = _optimal_tables()
    readv, hw = _simulate(_lib.PN_TRAJ_BUDGET, budget, nsteps, known_total=True)
    assert hw <= budget and readv == first(nsteps, budget)
    online, _ = _simulate(_lib.PN_TRAJ_BUDGET, budget, nsteps, known_total=False)
    assert readv <= online <= 1.6 * readv + 2          # adaptive sweeps (unknown length) stay close


def test_fixed_step_solves_announce_their_length():
    lib, ts = _ts(ts_adapt_type="none", ts_rk_type="4")
    span = (ctypes.c_double * 3)(0.0, 0.26, 1.0)
    _lib.check(lib.pn_ts_begin(ts, 0.0, 0.1, 3, span))
    n = lib.pn_ts_count_fixed_steps(ts)
    acc, hit, done = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(0)
    k = 0
    while not done.value:
        _lib.check(lib.pn_ts_judge(ts, -1.0, ctypes.byref(acc), ctypes.byref(hit), ctypes.byref(done)))
        k += 1
    assert n == k == lib.pn_ts_steps(ts)
    lib.pn_ts_destroy(ts)
    lib, ts = _ts(ts_rk_type="5dp")
    _lib.check(lib.pn_ts_begin(ts, 0.0, 0.1, 3, span))
    assert lib.pn_ts_count_fixed_steps(ts) == -1
    lib.pn_ts_destroy(ts)


def test_scheduler_randomised():
    rng = random.Random(0)
    for _ in range(60):
        _simulate(_lib.PN_TRAJ_BUDGET, rng.randint(1, 12), rng.randint(1, 300), known_total=rng.random() < 0.5)
    for _ in range(60):
        _simulate(_lib.PN_TRAJ_BUDGET, rng.randint(1, 12), rng.randint(1, 300), known_total=rng.random() < 0.5, carry=True)
    _simulate(_lib.PN_TRAJ_BUDGET, 70, 9000, known_total=True)      # beyond the DP limits: fallback paths


def test_options_parser():
    assert options.parse(["-ts_adapt_type", "none", "-ts_monitor", "-ts_rtol", "1e-6", "-ts_trajectory_max_cps_ram", "50",
                          "positional", "-neg", "-3"]) == {
        "ts_adapt_type": "none", "ts_monitor": "", "ts_rtol": "1e-6", "ts_trajectory_max_cps_ram": "50", "neg": "-3"}
    options.init(["prog", "-ts_rk_type", "4", "-ts_trajectory_solution_only", "0"])
    assert options.get_all() == {"ts_rk_type": "4", "ts_trajectory_solution_only": "0"}
    assert options.truthy("") and options.truthy("1") and not options.truthy("0") and options.truthy(None, True)


@pytest.mark.parametrize("method", ["rk4", "dopri5"])
@pytest.mark.parametrize("sources", [32, 5, 1])
def test_param_accum_batched_over_time_steps_equals_per_stage_bitwise(method, sources):
    """Default mode (-pn_param_accum batch): stage results of several time steps wait and are added
    by ONE pn_param_accum_multi call per <= -pn_param_accum_sources results, oldest first: the bits of
    the per-stage calls, with ceil(nsteps*s_eff/cap)-ish calls instead of nsteps*s_eff."""
    torch.manual_seed(3)
    y0 = torch.randn(6, 2, dtype=torch.float64)
    t = torch.tensor([0.0, 0.1, 0.25, 0.5], dtype=torch.float64)
    res = {}
    for mode in ("stage", "batch"):
        options.clear()
        options.set_option("pn_linear_param_grads", 0)      # (these tests count pn_param_accum calls: autograd's gradients)
        options.set_option("ts_adapt_type", "none")
        options.set_option("pn_param_accum", mode)
        options.set_option("pn_param_accum_sources", sources)
        f = SpiralFunc(torch.float64)
        ode = petsc_adjoint.ODEPetsc(backend=CpuVecOps)
        ode.setupTS(y0, f, step_size=0.05, method=method)
        y = y0.clone().requires_grad_(True)
        ode.odeint_adjoint(y, t).abs().mean().backward()
        res[mode] = (flat_grads(f).clone(), y.grad.clone(), ode._ops.calls["param_accum"], ode._nsteps, ode._s_eff)
    assert torch.equal(res["stage"][0], res["batch"][0]) and torch.equal(res["stage"][1], res["batch"][1])
    nsteps, s_eff = res["batch"][3], res["batch"][4]
    per_flush = max(s_eff, (sources // s_eff) * s_eff) if sources >= s_eff else s_eff
    assert res["batch"][2] == -(-nsteps * s_eff // per_flush)
    assert res["stage"][2] == nsteps * s_eff


class _PassThrough(nn.Module):
    """f = tanh(y) * a + b with b of the state's shape: autograd returns the cotangent buffer itself as
    dL/db -- a batched accumulation must not read it after the buffer has been rewritten."""

    def __init__(self, shape):
        super().__init__()
        self.a = nn.Parameter(torch.full((1,), 0.7, dtype=torch.float64))
        self.b = nn.Parameter(torch.linspace(-0.3, 0.3, int(np.prod(shape)), dtype=torch.float64).reshape(shape).clone())

    def forward(self, t, y):
        return torch.tanh(y) * self.a + self.b


@pytest.mark.parametrize("mode", ["batch", "step", "stage"])
def test_parameter_gradient_that_aliases_the_cotangent_buffer(mode):
    torch.manual_seed(5)
    y0 = torch.randn(4, 3, dtype=torch.float64)
    t = torch.tensor([0.0, 0.2, 0.4], dtype=torch.float64)
    opts = {"ts_adapt_type": "none", "pn_param_accum": mode}
    target = torch.randn(3, 4, 3, dtype=torch.float64)
    a, b = _pair(lambda: _PassThrough((4, 3)), y0, t, target, "rk4", opts, step_size=0.05)
    assert rel_err(b[1], a[1]) < 1e-12 and rel_err(b[2], a[2]) < 1e-12


class _SlicedBias(nn.Module):
    """f = cat([tanh(z[:, :2]) + b1, a * tanh(z[:, 2:]) + b2]): autograd returns VIEWS of the cotangent buffer (smaller
    than the state) as dL/db1, dL/db2 (the alias check used to look at full-size tensors only)."""

    def __init__(self):
        super().__init__()
        self.a = nn.Parameter(torch.full((1,), 0.7, dtype=torch.float64))
        self.b1 = nn.Parameter(torch.linspace(-0.3, 0.3, 8, dtype=torch.float64).reshape(4, 2).clone())
        self.b2 = nn.Parameter(torch.linspace(0.2, -0.1, 4, dtype=torch.float64).reshape(4, 1).clone())

    def forward(self, t, z):
        return torch.cat([torch.tanh(z[:, :2]) + self.b1, self.a * torch.tanh(z[:, 2:]) + self.b2], -1)


class _StackedScalars(nn.Module):
    """The pattern of the reference's ROBER-type models: a 1-D state, f = stack((g0 + p0, g1 + p1, g2)) with 0-dim
    parameters: dL/dp0 is a 0-dim select view of the cotangent."""

    def __init__(self):
        super().__init__()
        self.p0 = nn.Parameter(torch.tensor(0.3, dtype=torch.float64))
        self.p1 = nn.Parameter(torch.tensor(-0.2, dtype=torch.float64))
        self.k = nn.Parameter(torch.tensor(0.9, dtype=torch.float64))

    def forward(self, t, y):
        return torch.stack((-self.k * y[0] + y[1] * y[2] + self.p0, self.k * y[0] - y[1] ** 2 + self.p1, torch.sin(y[2])), -1)


@pytest.mark.parametrize("mode", ["batch", "step", "stage"])
@pytest.mark.parametrize("which", ["cat", "stack"])
def test_parameter_gradient_that_is_a_small_view_of_the_cotangent_buffer(mode, which):
    torch.manual_seed(6)
    opts = {"ts_adapt_type": "none", "pn_param_accum": mode}
    if which == "cat":
        y0 = torch.randn(4, 3, dtype=torch.float64)
        make = _SlicedBias
    else:
        y0 = torch.tensor([0.4, -0.3, 0.8], dtype=torch.float64)
        make = _StackedScalars
    t = torch.tensor([0.0, 0.25, 0.5], dtype=torch.float64)
    target = torch.randn((3,) + tuple(y0.shape), dtype=torch.float64)
    a, b = _pair(make, y0, t, target, "rk4", opts, step_size=0.05)
    assert rel_err(b[1], a[1]) < 1e-12 and rel_err(b[2], a[2]) < 1e-12


def test_outputs_with_steps_shorter_than_the_span_window_fp32():
    """with h <= the reference's hit window (1e-3 in fp32) the |t - t_i| < window rule fires one step
    early.  The reference takes its OUTPUTS from PETSc's exact span solutions (pa.py:845) and only its
    step counts from that rule (pa.py:526-532).  Here: outputs are the states of the steps that land on
    t_i in either counting mode; -pn_span_count reference reproduces the reference's counts (= the oracle's),
    the default counts exactly."""
    torch.manual_seed(0)
    y0 = torch.randn(5, 2, dtype=torch.float32)
    t = torch.linspace(0.0, 0.2, 21, dtype=torch.float32)
    f_ref = SpiralFunc(torch.float32)
    ref = ODEPetscOracle({"ts_adapt_type": "none"})
    ref.setupTS(y0, f_ref, step_size=1e-3, method="euler")
    with torch.no_grad():
        pr = ref.odeint_adjoint(y0, t)
    # unrolled euler, output every 10 steps
    f = SpiralFunc(torch.float32)
    with torch.no_grad():
        u, outs = y0.clone(), [y0.clone()]
        for k in range(200):
            u = u + torch.tensor(1e-3, dtype=torch.float32) * f(0.0, u)
            if (k + 1) % 10 == 0:
                outs.append(u.clone())
        exact = torch.stack(outs)
    assert rel_err(pr, exact) < 1e-6
    for mode in ("exact", "reference"):
        options.clear()
        options.set_option("ts_adapt_type", "none")
        options.set_option("pn_span_count", mode)
        ode = petsc_adjoint.ODEPetsc(backend=CpuVecOps)
        ode.setupTS(y0, SpiralFunc(torch.float32), step_size=1e-3, method="euler")
        with torch.no_grad():
            p = ode.odeint_adjoint(y0, t)
        assert rel_err(p, pr) < 1e-6, mode
        assert ode._nsteps == 200
        if mode == "exact":
            assert ode.cur_sol_steps == [0] + [10] * 20
        else:
            assert ode.cur_sol_steps == ref.cur_sol_steps
            assert ode.cur_sol_steps != [0] + [10] * 20  # forcing terms land one step off in the reference's backward


def test_exact_span_counting_gives_the_discrete_adjoint_when_steps_are_short():
    torch.manual_seed(1)
    y0 = torch.randn(3, 2, dtype=torch.float64)
    t = torch.tensor([0.0, 2e-5, 4e-5], dtype=torch.float64)          # h = 4e-6 < the 1e-5 fp64 window
    target = torch.randn(3, 3, 2, dtype=torch.float64)
    options.set_option("ts_adapt_type", "none")
    f = SpiralFunc()
    ode = petsc_adjoint.ODEPetsc(backend=CpuVecOps)
    ode.setupTS(y0, f, step_size=4e-6, method="rk4")
    y = y0.clone().requires_grad_(True)
    p = ode.odeint_adjoint(y, t)
    (p * target).sum().backward()
    assert ode.cur_sol_steps == [0, 5, 5]
    from oracle.autograd_rk import odeint_unrolled
    f2 = SpiralFunc()
    y2 = y0.clone().requires_grad_(True)
    hs = [4e-6] * 10
    p2 = odeint_unrolled(f2, y2, [4e-6 * (k + 1) for k in range(10)], hs, [0, 5, 10], method="rk4")
    (p2 * target).sum().backward()
    assert rel_err(p, p2) < 1e-13 and rel_err(y.grad, y2.grad) < 1e-12 and rel_err(flat_grads(f), flat_grads(f2)) < 1e-12


@pytest.mark.parametrize("method", ["cn", "beuler", "imex3", "imexl2", "imex_torch"])
def test_checkpoint_modes_bitwise_identical_for_implicit_and_imex_steppers(method):
    """PETSc applies TSTrajectory to every TS type (pa.py:771-775): -ts_trajectory_solution_only and
    -ts_trajectory_max_cps_ram must work for cn / beuler / ARKIMEX as they do for RK.  Round 1 kept every step and
    stage of these steppers in a Python list whatever the options said.  All modes replay the same solves, so the
    gradients agree bit for bit; the budgets are respected."""
    from problems import DiffusionIM, ReactionEX
    torch.manual_seed(7)
    y0 = torch.randn(3, 6, dtype=torch.float64)
    t = torch.tensor([0.0, 0.1, 0.35], dtype=torch.float64)
    res = []
    variants = [{"ts_trajectory_solution_only": 0}, {}, {"ts_trajectory_solution_only": 1}, {"ts_trajectory_max_cps_ram": 1},
                {"ts_trajectory_max_cps_ram": 3}, {"ts_trajectory_max_cps_ram": 2, "ts_trajectory_solution_only": 0}]
    for opts in variants:
        options.clear()
        options.set_option("ts_adapt_type", "none")
        for k, v in opts.items():
            options.set_option(k, v)
        ode = petsc_adjoint.ODEPetsc(backend=CpuVecOps)
        if method.startswith("imex"):
            if method == "imex_torch":
                options.set_option("snes_type", "ksponly")
            else:
                options.set_option("ts_arkimex_type", method[4:])
            fI, fE = DiffusionIM(6), ReactionEX(6)
            ode.setupTS(y0, fI, step_size=0.05, method="imex", implicit_form=True, imex_form=True, func2=fE, batch_size=3,
                        linear_solver="torch" if method == "imex_torch" else "petsc", matrixfree_jacobian=method != "imex_torch")
            params = list(fI.parameters()) + list(fE.parameters())
        else:
            f = SpiralFunc()
            y0 = y0[:, :2].contiguous() if y0.shape[1] != 2 else y0
            ode.setupTS(y0, f, step_size=0.05, method=method, implicit_form=True)
            params = list(f.parameters())
        y = y0.clone().requires_grad_(True)
        p = ode.odeint_adjoint(y, t)
        p.abs().mean().backward()
        assert ode._nsteps == 7
        res.append((p.detach().clone(), y.grad.clone(), torch.cat([q.grad.reshape(-1) for q in params]).clone(),
                    ode._traj.high_water(), ode.nfe_forward))
    for r in res[1:]:
        assert torch.equal(r[0], res[0][0]) and torch.equal(r[1], res[0][1]) and torch.equal(r[2], res[0][2])
    assert res[3][3] <= 1 and res[4][3] <= 3 and res[5][3] <= 2
    assert res[1][4] > res[0][4]            # solution-only re-solves the stages of every reversed step


@pytest.mark.parametrize("method", ["rk4", "dopri5", "cn", "imex"])
@pytest.mark.parametrize("solution_only", [0, 1])
def test_disk_tier_of_the_trajectory_equals_the_memory_tier_bitwise(tmp_path, method, solution_only):
    """-ts_trajectory_type basic (PETSc's default type, the reference's default: ode_demo_petsc.py:26) keeps
    every checkpoint in a file under -ts_trajectory_dirname (C++ engine pn_spill_*: staging buffers, I/O
    thread, read-ahead in the reverse sweep).  Same slots, same arithmetic: the gradients equal the HBM
    tier's bit for bit; the files exist while the trajectory is alive and are removed with it."""
    import gc
    from problems import DiffusionIM, ReactionEX
    torch.manual_seed(11)
    y0 = torch.randn(4, 2 if method != "imex" else 6, dtype=torch.float64)
    t = torch.tensor([0.0, 0.2, 0.5], dtype=torch.float64)
    d = str(tmp_path / "ckpt")
    res = {}
    for ttype in ("memory", "basic"):
        options.clear()
        if method != "dopri5":
            options.set_option("ts_adapt_type", "none")
        options.set_option("ts_trajectory_solution_only", solution_only)
        options.set_option("ts_trajectory_type", ttype)
        options.set_option("ts_trajectory_dirname", d)
        ode = petsc_adjoint.ODEPetsc(backend=CpuVecOps)
        if method == "imex":
            fI, fE = DiffusionIM(6), ReactionEX(6)
            ode.setupTS(y0, fI, step_size=0.05, method="imex", implicit_form=True, imex_form=True, func2=fE, batch_size=4)
            params = list(fI.parameters()) + list(fE.parameters())
        else:
            f = SpiralFunc() if method != "dopri5" else SpiralTruth()
            ode.setupTS(y0, f, step_size=0.05 if method != "dopri5" else 0.3, method=method, implicit_form=method == "cn")
            params = list(f.parameters())
        y = y0.clone().requires_grad_(True)
        p = ode.odeint_adjoint(y, t)
        if ttype == "basic":
            assert ode._traj.on_disk and os.path.isdir(ode._traj.dir)
            st = ode._traj.stats() if ode._nsteps <= 4 else None
            ode._lib.pn_spill_prefetch(ode._traj.spill, 0)          # harmless extra read-ahead
            nfiles_dir = ode._traj.dir
        p.abs().mean().backward()
        res[ttype] = (p.detach().clone(), y.grad.clone(), torch.cat([q.grad.reshape(-1) for q in params]).clone(), ode._nsteps)
        if ttype == "basic":
            s = ode._traj.stats()
            assert s["bytes_written"] > 0 and s["bytes_read"] > 0 and s["files"] >= ode._nsteps - _DISK_RING
            del p, y
            ode._traj = None
            if ode._theta is not None:
                ode._theta.traj = None
            gc.collect()
            assert not os.path.exists(nfiles_dir)                    # files and the sweep's directory are gone
    a, b = res["memory"], res["basic"]
    assert a[3] == b[3] and a[3] >= 6
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])


_DISK_RING = 4


def test_disk_tier_keep_files_and_unknown_types(tmp_path):
    torch.manual_seed(2)
    y0 = torch.randn(3, 2, dtype=torch.float64)
    t = torch.tensor([0.4], dtype=torch.float64)
    d = str(tmp_path / "kept")
    options.set_option("ts_adapt_type", "none")
    options.set_option("ts_trajectory_type", "basic")
    options.set_option("ts_trajectory_dirname", d)
    options.set_option("ts_trajectory_keep_files", 1)
    ode = petsc_adjoint.ODEPetsc(backend=CpuVecOps)
    ode.setupTS(y0, SpiralFunc(), step_size=0.05, method="rk4")
    y = y0.clone().requires_grad_(True)
    ode.odeint_adjoint(y, t).sum().backward()
    sub = ode._traj.dir
    ode._traj = None
    import gc
    gc.collect()
    kept = sorted(os.listdir(sub))
    assert kept and all(k.startswith("SA-") and k.endswith(".bin") for k in kept)
    assert os.path.getsize(os.path.join(sub, kept[0])) == 64 * 8          # one padded state vector (64 doubles)
    options.set_option("ts_trajectory_type", "visualization")
    with pytest.raises(petsc_adjoint.PnError, match="not implemented"):
        petsc_adjoint.ODEPetsc(backend=CpuVecOps).setupTS(y0, SpiralFunc(), step_size=0.05, method="rk4")


@pytest.mark.parametrize("method", ["rk4", "dopri5", "cn", "imex"])
@pytest.mark.parametrize("budget,solution_only", [(1, 1), (2, 1), (3, 0), (7, 1), (50, 0)])
def test_disk_tier_for_the_bounded_checkpoint_set_equals_the_hbm_budget_bitwise(tmp_path, method, budget, solution_only):
    """-ts_trajectory_type basic together with -ts_trajectory_max_cps_ram N (README.md:91-96) keeps the
    bounded set in FILES: slots are recycled (a recycled slot's file is rewritten), the device holds a four-buffer
    least-recently-used cache.  Gradients equal those of the same budget in HBM bit for bit, at most N checkpoints
    ever exist, and with checkpoints that carry stage values the files are rewritten when the stages are added."""
    from problems import DiffusionIM, ReactionEX
    torch.manual_seed(13)
    y0 = torch.randn(4, 2 if method != "imex" else 6, dtype=torch.float64)
    t = torch.tensor([0.0, 0.3, 0.75], dtype=torch.float64)
    res = {}
    for ttype in ("memory", "basic"):
        options.clear()
        if method != "dopri5":
            options.set_option("ts_adapt_type", "none")
        for k, v in {"ts_trajectory_solution_only": solution_only, "ts_trajectory_type": ttype, "ts_trajectory_max_cps_ram": budget,
                     "ts_trajectory_dirname": str(tmp_path / "ck")}.items():
            options.set_option(k, v)
        ode = petsc_adjoint.ODEPetsc(backend=CpuVecOps)
        if method == "imex":
            fI, fE = DiffusionIM(6), ReactionEX(6)
            ode.setupTS(y0, fI, step_size=0.05, method="imex", implicit_form=True, imex_form=True, func2=fE, batch_size=4)
            params = list(fI.parameters()) + list(fE.parameters())
        else:
            f = SpiralFunc() if method != "dopri5" else SpiralTruth()
            ode.setupTS(y0, f, step_size=0.05 if method != "dopri5" else 0.3, method=method, implicit_form=method == "cn")
            params = list(f.parameters())
        y = y0.clone().requires_grad_(True)
        p = ode.odeint_adjoint(y, t)
        p.abs().mean().backward()
        assert ode._traj.on_disk == (ttype == "basic") and ode._traj.high_water() <= budget
        if ttype == "basic":
            st = ode._traj.stats()
            assert 0 < st["files"] <= budget and st["bytes_written"] > 0
            if _DISK_RING < budget < ode._nsteps:              # more checkpoints than device buffers: some came back from files
                assert st["bytes_read"] > 0
        res[ttype] = (p.detach().clone(), y.grad.clone(), torch.cat([q.grad.reshape(-1) for q in params]).clone(), ode._nsteps)
    a, b = res["memory"], res["basic"]
    assert a[3] == b[3] and a[3] >= 6
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])


@pytest.mark.parametrize("name", ["3", "4", "5", "1bee"])
def test_adaptive_arkimex_follows_the_basic_controller_and_its_adjoint_equals_the_oracle(name):
    """Without -ts_adapt_type none PETSc adapts ARKIMEX steps with TSAdapt basic (pa.py:655-656, 775): error estimate =
    WRMS distance between the propagated and the embedded solution, h_new = h clip(0.9 e^(-1/order), 0.1, 10), reject when
    e > 1.  Round 1 warned and took fixed steps.  Checked here: (a) every accepted step's error norm, recomputed by the
    oracle from its own stages, is <= 1 and reproduces the next step size unless the span logic cut it; (b) at least one
    attempt is rejected from the coarse first step; (c) states and gradients equal the oracle's when it follows the same
    accepted steps (the discrete adjoint does not differentiate the controller)."""
    import warnings as _w
    from oracle.arkimex_oracle import odeint_adjoint_arkimex, solve_arkimex, step_error_norm, tableau
    from problems import DiffusionIM, ReactionEX
    torch.manual_seed(0)
    y0 = torch.randn(3, 6, dtype=torch.float64)
    t = torch.tensor([0.0, 0.4, 1.0], dtype=torch.float64)
    target = torch.randn(3, 3, 6, dtype=torch.float64)
    tol = {"3": 1e-5, "4": 1e-8, "5": 1e-9, "1bee": 1e-3}[name]
    for k, v in {"ts_arkimex_type": name, "ts_rtol": tol, "ts_atol": tol, "snes_rtol": 1e-13, "snes_stol": 1e-15, "ksp_rtol": 1e-13}.items():
        options.set_option(k, v)
    fI, fE = DiffusionIM(6), ReactionEX(6)
    ode = petsc_adjoint.ODEPetsc(backend=CpuVecOps)
    with _w.catch_warnings():
        _w.simplefilter("error")                      # no "takes fixed steps" warning for these types
        _w.simplefilter("ignore", petsc_adjoint.PnUnpinnedWarning)
        ode.setupTS(y0, fI, step_size=0.5, method="imex", implicit_form=True, imex_form=True, func2=fE, batch_size=3)
    assert ode._adaptive
    y = y0.clone().requires_grad_(True)
    p = ode.odeint_adjoint(y, t)
    torch.mean(torch.abs(p - target)).backward()
    log = ode.step_log()
    hs = [h for _, h in log]
    assert ode.num_rejections >= 1 and len(set(round(h, 12) for h in hs)) > 3
    assert sum(ode.cur_sol_steps) == len(hs) and abs(sum(hs) - 1.0) < 1e-12
    # (c) oracle on the same accepted steps
    plan = (list(log), list(ode.cur_sol_steps))
    fI2, fE2 = DiffusionIM(6), ReactionEX(6)
    y2 = y0.clone().requires_grad_(True)
    p2 = odeint_adjoint_arkimex(fI2, fE2, y2, t, 0.5, name, plan=plan)
    torch.mean(torch.abs(p2 - target)).backward()
    assert rel_err(p, p2) < 1e-9 and rel_err(y.grad, y2.grad) < 1e-8
    assert rel_err(flat_grads(fI), flat_grads(fI2)) < 1e-8 and rel_err(flat_grads(fE), flat_grads(fE2)) < 1e-8
    # (a) the controller's decisions, from the oracle's own stages
    _, traj, _ = solve_arkimex(fI2, fE2, y0, t, 0.5, name, plan=plan)
    order = 2 if name == "1bee" else tableau(name)["order"]      # the order PETSc registers the type with (1bee: 2)
    followed, wants = 0, []
    for k, (tn, h, u, _) in enumerate(traj):
        e = step_error_norm(fI2, fE2, tn, h, u, name, tol, tol)
        assert e <= 1.0 + 1e-6, (k, e)
        if k + 1 < len(traj):
            wants.append(h * min(max(0.9 * e ** (-1.0 / order), 0.1), 10.0))
            if abs(hs[k + 1] - wants[-1]) <= 1e-4 * wants[-1]:   # (the error estimate is a difference of nearly equal vectors)
                followed += 1                      # the controller's choice was taken as it is
            else:
                # cut by the exact-final-time / span logic (never lengthened), or the step that logic had cut is restored
                # after the output time was hit (then it is one of the controller's earlier choices)
                assert hs[k + 1] <= wants[-1] * (1 + 1e-4) or any(abs(hs[k + 1] - w) <= 1e-4 * w for w in wants)
    # order 1 grows by the full factor 10 after almost every step and is then rejected back: its accepted sizes are
    # the controller's choices only after rejections (covered by the <= check above)
    assert followed >= 2 or name == "1bee"


def test_arkimex_without_embedded_weights_warns_and_takes_fixed_steps():
    from problems import DiffusionIM, ReactionEX
    options.set_option("ts_arkimex_type", "l2")
    ode = petsc_adjoint.ODEPetsc(backend=CpuVecOps)
    with pytest.warns(RuntimeWarning, match="no embedded weights"):
        ode.setupTS(torch.zeros(3, 6, dtype=torch.float64), DiffusionIM(6), step_size=0.05, method="imex", implicit_form=True,
                    imex_form=True, func2=ReactionEX(6), batch_size=3)
    assert not ode._adaptive


@pytest.mark.parametrize("method", ["cn", "beuler"])
def test_adaptive_theta_methods_follow_the_basic_controller_and_their_adjoint_equals_the_oracle(method):
    """With an explicit -ts_adapt_type basic PETSc adapts beuler / cn too (pa.py:651-654, 775): no embedded pair, but a local
    truncation error estimate from the last three solutions (TSEvaluateWLTE_Theta: a scaled second backward difference on
    the non-uniform grid, controller order 2; restated from memory of theta.c -- parity unpinned).  Checked: the first step
    has no estimate and is accepted with its size unchanged; every later accepted step's estimate, recomputed by the oracle
    from its own states, is <= 1 and explains the next step size unless the span logic cut it; rejections happen; states
    and gradients equal the oracle's on the same accepted steps."""
    from oracle.theta_oracle import lte_norm, odeint_adjoint_theta, solve_theta
    torch.manual_seed(0)
    y0 = torch.randn(4, 2, dtype=torch.float64)
    t = torch.tensor([0.0, 0.6, 1.5], dtype=torch.float64)
    target = torch.randn(3, 4, 2, dtype=torch.float64)
    tol = 1e-4 if method == "cn" else 1e-3
    for k, v in {"ts_adapt_type": "basic", "ts_rtol": tol, "ts_atol": tol, "snes_rtol": 1e-13, "snes_stol": 1e-15,
                 "ksp_rtol": 1e-13}.items():
        options.set_option(k, v)
    f = SpiralFunc()
    ode = petsc_adjoint.ODEPetsc(backend=CpuVecOps)
    ode.setupTS(y0, f, step_size=0.02, method=method, implicit_form=True)
    assert ode._adaptive
    y = y0.clone().requires_grad_(True)
    p = ode.odeint_adjoint(y, t)
    torch.mean(torch.abs(p - target)).backward()
    log = ode.step_log()
    hs = [h for _, h in log]
    assert len(set(round(h, 12) for h in hs)) > 3 and abs(sum(hs) - 1.5) < 1e-12
    assert hs[1] == pytest.approx(hs[0], rel=1e-12)          # no estimate in the first step: size kept
    plan = (list(log), list(ode.cur_sol_steps))
    f2 = SpiralFunc()
    y2 = y0.clone().requires_grad_(True)
    p2 = odeint_adjoint_theta(f2, y2, t, 0.02, method, plan=plan)
    torch.mean(torch.abs(p2 - target)).backward()
    assert rel_err(p, p2) < 1e-9 and rel_err(y.grad, y2.grad) < 1e-8 and rel_err(flat_grads(f), flat_grads(f2)) < 1e-8
    _, traj, _ = solve_theta(f2, y0, t, 0.02, method, plan=plan)
    followed, wants = 0, []
    for k in range(1, len(traj)):
        tn, h, u, _ = traj[k]
        unew = traj[k + 1][2] if k + 1 < len(traj) else p2[-1].detach()
        e = lte_norm(unew, u, traj[k - 1][2], h, traj[k - 1][1], tol, tol)
        assert 0.0 <= e <= 1.0 + 1e-6, (k, e)
        if k + 1 < len(traj):
            wants.append(h * min(max(0.9 * e ** -0.5, 0.1), 10.0) if e > 0 else 10.0 * h)
            if abs(hs[k + 1] - wants[-1]) <= 1e-4 * wants[-1]:
                followed += 1
            else:
                assert hs[k + 1] <= wants[-1] * (1 + 1e-4) or any(abs(hs[k + 1] - w) <= 1e-4 * w for w in wants)
    assert followed >= 2


@pytest.mark.parametrize("method", ["cn", "beuler"])
def test_theta_methods_take_fixed_steps_unless_adapt_basic_is_given(method):
    """TSCreate_Theta makes TSADAPTNONE the default of the theta methods, so the reference's drivers that
    give no adapt option (spiral_unstable.py, ode_demo_petsc.py) run cn / beuler with fixed steps: step_size 0.05 to
    t = 1 is exactly 20 steps.  RK and ARKIMEX keep adapting by default."""
    torch.manual_seed(0)
    y0 = torch.randn(4, 2, dtype=torch.float64)
    f = SpiralFunc()
    ode = petsc_adjoint.ODEPetsc(backend=CpuVecOps)
    ode.setupTS(y0, f, step_size=0.05, method=method, implicit_form=True)
    assert not ode._adaptive
    with torch.no_grad():
        ode.odeint_adjoint(y0, torch.tensor([1.0], dtype=torch.float64))
    assert ode.num_steps == 20 and ode.num_rejections == 0
    assert all(h == pytest.approx(0.05, rel=1e-12) for _, h in ode.step_log())
    options.set_option("ts_adapt_type", "basic")
    ode2 = petsc_adjoint.ODEPetsc(backend=CpuVecOps)
    ode2.setupTS(y0, f, step_size=0.05, method=method, implicit_form=True)
    assert ode2._adaptive
    options.clear()
    ode3 = petsc_adjoint.ODEPetsc(backend=CpuVecOps)
    ode3.setupTS(y0, f, step_size=0.05, method="dopri5")
    assert ode3._adaptive


def test_checkpoint_placement_tables_are_built_once_per_process_not_once_per_solve():
    """the priced placement for 1000 steps x 200 slots costs 0.15-0.2 s of host time; a new pn_traj is made
    for every forward sweep, so the table must come from the process-wide cache from the second solve on."""
    lib = _lib.load()
    torch.manual_seed(0)
    y0 = torch.randn(3, 2, dtype=torch.float64)
    for k, v in {"ts_adapt_type": "none", "ts_trajectory_max_cps_ram": 7, "ts_trajectory_solution_only": 0}.items():
        options.set_option(k, v)
    f = SpiralFunc()
    ode = petsc_adjoint.ODEPetsc(backend=CpuVecOps)
    ode.setupTS(y0, f, step_size=0.01, method="rk4")
    grads = []
    builds = []
    for _ in range(3):
        f.zero_grad()
        y = y0.clone().requires_grad_(True)
        ode.odeint_adjoint(y, torch.tensor([0.61], dtype=torch.float64)).abs().mean().backward()
        grads.append(flat_grads(f).clone())
        builds.append(lib.pn_traj_dp_builds())
    assert builds[1] == builds[0] and builds[2] == builds[0]
    assert torch.equal(grads[0], grads[1]) and torch.equal(grads[0], grads[2])
    assert ode._traj.high_water() <= 7


def _theta_run(method, opts, n=6, B=5, nt=4, imex=False, seed=3):
    """One small implicit (or IMEX) solve + adjoint on the CPU stand-in; returns results and the stepper."""
    from problems import DiffusionIM, ReactionEX
    options.clear()
    for k, v in dict({"ts_adapt_type": "none"}, **opts).items():
        options.set_option(k, v)
    torch.manual_seed(seed)
    y0 = torch.randn(B, n, dtype=torch.float64)
    ode = petsc_adjoint.ODEPetsc(backend=CpuVecOps)
    if imex:
        fI, fE = DiffusionIM(n), ReactionEX(n)
        ode.setupTS(y0, fI, step_size=0.05, method="imex", implicit_form=True, imex_form=True, func2=fE, batch_size=B)
        mods = (fI, fE)
    else:
        f = TimeDependent(n)
        ode.setupTS(y0, f, step_size=0.05, method=method, implicit_form=True)
        mods = (f,)
    y = y0.clone().requires_grad_(True)
    out = ode.odeint_adjoint(y, torch.tensor([0.05 * nt], dtype=torch.float64))
    out.abs().mean().backward()
    options.clear()
    return out.detach(), y.grad.clone(), torch.cat([flat_grads(m) for m in mods]), ode._theta


@pytest.mark.parametrize("method,imex", [("cn", False), ("beuler", False), ("imex", True)])
@pytest.mark.parametrize("restart", [30, 3])
def test_device_resident_gmres_takes_the_decisions_of_the_host_loop(method, imex, restart):
    """Round 3: GMRES keeps its Hessenberg column, rotations, residual estimate and stop flag on the device
    (pn_krylov_*); the host enqueues chunks of iterations and reads the state once per chunk.  The decisions are
    those of round 2's host loop (-pn_krylov host): same Newton and GMRES iteration counts, same numbers to
    round-off -- also across restarts (restart length 3) -- with fewer host synchronisations than iterations, and
    launches past convergence are no-ops.  (CPU stand-in: tests/_cpu_vecops.py restates the device state machine;
    the kernels themselves are checked by the -m gpu tests.)"""
    base = {"ksp_gmres_restart": restart, "ksp_rtol": 1e-9, "snes_rtol": 1e-12}
    dev = _theta_run(method, dict(base, pn_krylov="device"), imex=imex)
    host = _theta_run(method, dict(base, pn_krylov="host"), imex=imex)
    assert (dev[3].newton_its, dev[3].linear_its) == (host[3].newton_its, host[3].linear_its)
    assert dev[3].linear_its > 10
    for a, b in zip(dev[:3], host[:3]):
        assert rel_err(a, b) < 1e-12
    kr = dev[3]._kr
    assert kr is not None and host[3]._kr is None
    if restart == 30:
        assert 0 < dev[3].host_syncs < dev[3].linear_its       # chunks, not single iterations
    assert kr.noops >= 0 and kr.launches > 0


def test_device_resident_gmres_launches_past_convergence_are_no_ops():
    """The first chunk is as long as the previous solve of the same kind was; when a solve converges earlier, the
    iterations already enqueued must leave the state and the solution alone."""
    out = _theta_run("cn", {"pn_krylov": "device", "ksp_rtol": 1e-10})
    st = out[3]
    kr = st._kr
    ref = _theta_run("cn", {"pn_krylov": "host", "ksp_rtol": 1e-10})
    for a, b in zip(out[:3], ref[:3]):
        assert rel_err(a, b) < 1e-12
    # force an over-long first chunk and solve one more system directly through the stepper's GMRES
    st._its_guess[(False, 0)] = 25
    ops = st.ode._ops
    n = st.ode.n
    torch.manual_seed(0)
    A = torch.eye(n, dtype=torch.float64) * 3.0 + 0.1 * torch.randn(n, n, dtype=torch.float64)
    rhs = torch.randn(n, dtype=torch.float64)
    x = torch.zeros(n, dtype=torch.float64)
    before = kr.noops
    its = st._gmres(lambda v: -(A @ v[:n]), 0.0, rhs, x, False)          # operator shift*M v - J v with shift 0: A v
    assert kr.noops - before == 25 - its and its < 25
    assert torch.allclose(A @ x, rhs, rtol=0, atol=1e-7 * float(rhs.norm()))


def _nfe_run(opts, method="rk4", nsteps=7):
    options.clear()
    for k, v in opts.items():
        options.set_option(k, v)
    torch.manual_seed(0)
    y0 = torch.randn(5, 2, dtype=torch.float64)
    f = SpiralFunc()
    ode = petsc_adjoint.ODEPetsc(backend=CpuVecOps)
    ode.setupTS(y0, f, step_size=0.05, method=method)
    y = y0.clone().requires_grad_(True)
    f.nfe = 0
    out = ode.odeint_adjoint(y, torch.tensor([0.05 * nsteps], dtype=torch.float64))
    nfe_f = f.nfe
    f.nfe = 0
    out.abs().mean().backward()
    nfe_b = f.nfe
    options.clear()
    return nfe_f, nfe_b, y.grad.clone(), flat_grads(f).clone(), ode


@pytest.mark.parametrize("method,s,fsal", [("rk4", 4, False), ("euler", 1, False), ("midpoint", 2, False), ("dopri5", 7, True)])
def test_reference_defaults_switch_restores_the_references_call_counts(method, s, fsal):
    """-pn_reference_defaults 1 is ONE switch for everything a user of the reference could observe
    differently by default.  A func that counts its calls (NFE-F / NFE-B of the reference's drivers,
    examples-pnode/spiral_unstable.py:326-347) then reads the reference's numbers: s evaluations per step forward
    (first-same-as-last: s - 1 after the first step); backward s per step with -ts_trajectory_solution_only 0 (the
    re-evaluation inside every stage VJP, pa.py:66-74; FSAL: s - 1, the last stage has no adjoint) and s more with
    PETSc's solution-only default (TSTrajectory re-runs the whole step).  Gradients are the same bits as with this
    package's own defaults."""
    n = 7
    adapt = {"ts_adapt_type": "none"}
    fwd = n * (s - 1) + 1 if fsal else n * s
    vjps = n * (s - 1) if fsal else n * s
    mine = _nfe_run(dict(adapt), method, n)
    ref_so = _nfe_run(dict(adapt, pn_reference_defaults=1), method, n)
    assert ref_so[0] == fwd and ref_so[1] == vjps + n * s
    assert ref_so[4]._tmode == _lib.PN_TRAJ_SOLUTION and ref_so[4]._retain_graph == 0 and ref_so[4]._span_count_reference
    ref_all = _nfe_run(dict(adapt, pn_reference_defaults=1, ts_trajectory_solution_only=0), method, n)
    assert ref_all[0] == fwd and ref_all[1] == vjps
    for r in (ref_so, ref_all):
        assert torch.equal(r[2], mine[2]) and torch.equal(r[3], mine[3])
    # an explicit option wins over the switch
    kept = _nfe_run(dict(adapt, pn_reference_defaults=1, ts_trajectory_solution_only=0, pn_trajectory_retain_graph=1), method, n)
    assert kept[1] == 0 and torch.equal(kept[2], mine[2])


def test_unpinned_pieces_say_so_once_per_process():
    """adaptive theta methods, adaptive ARKIMEX and the ARKIMEX tables l2 / 2c / 2d / 2e rest on
    restatements nothing PETSc-produced pins.  They say so -- once per process and piece -- with a
    RuntimeWarning subclass a caller can filter."""
    import warnings as _w
    from problems import DiffusionIM, ReactionEX
    saved = set(petsc_adjoint._UNPINNED_WARNED)
    petsc_adjoint._UNPINNED_WARNED.clear()
    try:
        y3 = torch.zeros(3, 6, dtype=torch.float64)

        def imex(name, adapt_none):
            options.clear()
            options.set_option("ts_arkimex_type", name)
            if adapt_none:
                options.set_option("ts_adapt_type", "none")
            ode = petsc_adjoint.ODEPetsc(backend=CpuVecOps)
            ode.setupTS(y3, DiffusionIM(6), step_size=0.05, method="imex", implicit_form=True, imex_form=True,
                        func2=ReactionEX(6), batch_size=3)

        with pytest.warns(petsc_adjoint.PnUnpinnedWarning, match="adaptive ARKIMEX"):
            imex("3", False)
        for name in ("l2", "2c", "2d", "2e"):
            with pytest.warns(petsc_adjoint.PnUnpinnedWarning, match="ARKIMEX type %s" % name):
                imex(name, True)
        options.clear()
        options.set_option("ts_adapt_type", "basic")
        with pytest.warns(petsc_adjoint.PnUnpinnedWarning, match="adaptive beuler / cn"):
            ode = petsc_adjoint.ODEPetsc(backend=CpuVecOps)
            ode.setupTS(y3, SpiralFunc(), step_size=0.05, method="cn", implicit_form=True)
        # once per process: nothing the second time; and nothing at all for pinned configurations
        with _w.catch_warnings():
            _w.simplefilter("error")
            imex("3", False)
            imex("l2", True)
            imex("4", True)
            options.clear()
            options.set_option("ts_adapt_type", "none")
            ode = petsc_adjoint.ODEPetsc(backend=CpuVecOps)
            ode.setupTS(y3, SpiralFunc(), step_size=0.05, method="cn", implicit_form=True)
            ode.setupTS(torch.zeros(3, 2, dtype=torch.float64), SpiralFunc(), step_size=0.05, method="rk4")
    finally:
        petsc_adjoint._UNPINNED_WARNED.clear()
        petsc_adjoint._UNPINNED_WARNED.update(saved)


def test_restart_lengths_beyond_the_device_table_use_the_host_loop():
    """The coefficient table of the device-resident GMRES lives in LDS (restart + 2 entries, restart <= 126): a longer
    restart (-ksp_gmres_restart 200) is served by round 2's host loop, same results."""
    a = _theta_run("cn", {"ksp_gmres_restart": 200, "ksp_rtol": 1e-9})
    b = _theta_run("cn", {"ksp_gmres_restart": 30, "ksp_rtol": 1e-9})
    assert a[3]._kr is None and b[3]._kr is not None and a[3].host_syncs == 0
    assert a[3].linear_its == b[3].linear_its            # these systems converge well inside 30 iterations
    for u, v in zip(a[:3], b[:3]):
        assert rel_err(u, v) < 1e-12


# ---------------------------------------------------------------- the C++ step loops (pn_rk_attempt / pn_rk_adjoint_step)
@pytest.mark.parametrize("method,opts", [
    ("rk4", {"ts_adapt_type": "none"}), ("rk4", {"ts_adapt_type": "none", "ts_trajectory_solution_only": 0}),
    ("rk4", {"ts_adapt_type": "none", "ts_trajectory_max_cps_ram": 3}), ("euler", {}), ("midpoint", {}), ("rk2", {}),
    ("bosh3", {}), ("dopri5", {}), ("dopri5", {"ts_trajectory_solution_only": 0}), ("dopri5", {"ts_trajectory_max_cps_ram": 2}),
    ("dopri5", {"ts_trajectory_max_cps_ram": 4, "ts_trajectory_solution_only": 0}), ("bosh3", {"pn_param_accum": "stage"}),
    ("rk4", {"ts_adapt_type": "none", "pn_reference_defaults": 1}),
])
def test_native_step_loops_take_the_steps_of_the_python_loop_bit_for_bit(method, opts):
    """-pn_step_loop native (default: ONE C++ entry point per step attempt and per reversed step, PETSc's C loops behind
    ts.solve / ts.adjointSolve, /root/reference/pnode/petsc_adjoint.py:829, 878; callbacks only for func and its VJP) against
    -pn_step_loop python (the stage loop of rounds 1-3): same launches in the same order with the same coefficients --
    states, gradients, step log and call counts identical; both equal the oracle (the other tests of this file run native)."""
    torch.manual_seed(0)
    y0 = torch.randn(5, 1, 2, dtype=torch.float64)
    t = torch.tensor([0.0, 0.3, 0.35, 1.0], dtype=torch.float64)
    w = torch.randn(4, 5, 1, 2, dtype=torch.float64)
    res = {}
    for loop in ("native", "python"):
        options.clear()
        for k, v in dict(opts, pn_step_loop=loop).items():
            options.set_option(k, v)
        f = SpiralFunc()
        ode = petsc_adjoint.ODEPetsc(backend=CpuVecOps)
        ode.setupTS(y0, f, step_size=0.1, method=method)
        assert ode._native == (loop == "native")
        y = y0.clone().requires_grad_(True)
        out = ode.odeint_adjoint(y, t)
        (out * w).sum().backward()
        res[loop] = (out.detach().clone(), y.grad.clone(), flat_grads(f).clone(), ode.step_log(), ode.num_rejections,
                     ode.nfe_forward, ode.nfe_backward, dict(ode._ops.calls))
    a, b = res["native"], res["python"]
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    assert a[3:] == b[3:]


def test_an_exception_in_func_passes_through_the_native_step_loops():
    """func raises inside the callback of pn_rk_attempt / pn_rk_adjoint_step: the C frame returns an error code and the
    Python side re-raises the ORIGINAL exception; the solver object stays usable."""
    class Flaky(nn.Module):
        def __init__(self):
            super().__init__()
            self.lin = nn.Linear(2, 2).double()
            self.fail_at, self.calls = None, 0

        def forward(self, t, y):
            self.calls += 1
            if self.fail_at is not None and self.calls == self.fail_at:
                raise KeyError("boom at call %d" % self.calls)
            return torch.tanh(self.lin(y))
    options.set_option("ts_adapt_type", "none")
    f = Flaky()
    y0 = torch.randn(3, 2, dtype=torch.float64)
    t = torch.tensor([0.5], dtype=torch.float64)
    ode = petsc_adjoint.ODEPetsc(backend=CpuVecOps)
    ode.setupTS(y0, f, step_size=0.1, method="rk4")
    assert ode._native
    f.fail_at = 7
    with pytest.raises(KeyError, match="boom at call 7"):
        ode.odeint_adjoint(y0.clone().requires_grad_(True), t)
    f.fail_at, f.calls = None, 0
    y = y0.clone().requires_grad_(True)
    out = ode.odeint_adjoint(y, t)
    f.fail_at = f.calls + 3                         # in the reverse sweep (solution-only: func is re-evaluated there)
    with pytest.raises(KeyError):
        out.sum().backward()
    f.fail_at = None
    y = y0.clone().requires_grad_(True)
    ode.odeint_adjoint(y, t).sum().backward()
    assert torch.isfinite(y.grad).all()


@pytest.mark.parametrize("method", ["rk4", "dopri5", "cn", "imex"])
@pytest.mark.parametrize("ram,disk,solution_only", [(1, 1, 1), (2, 3, 1), (3, 4, 0), (1, 6, 1), (5, 45, 0)])
def test_two_level_checkpointing_equals_the_same_total_budget_in_hbm_bitwise(tmp_path, method, ram, disk, solution_only):
    """PETSc's two-level checkpointing: -ts_trajectory_max_cps_ram R with -ts_trajectory_max_cps_disk D
    (/root/reference/README.md:91-96) -- one scheduler places R + D checkpoints, R of them stay in memory, D are files.
    Gradients equal those of a budget of R + D in memory bit for bit; at most R + D checkpoints exist; at most D files;
    only -ts_trajectory_max_cps_disk: the whole bounded set is on disk."""
    from problems import DiffusionIM, ReactionEX
    torch.manual_seed(13)
    y0 = torch.randn(4, 2 if method != "imex" else 6, dtype=torch.float64)
    t = torch.tensor([0.0, 0.3, 0.75], dtype=torch.float64)
    res = {}
    for tag, opts in (("hbm", {"ts_trajectory_max_cps_ram": ram + disk}),
                      ("two", {"ts_trajectory_max_cps_ram": ram, "ts_trajectory_max_cps_disk": disk}),
                      ("disk", {"ts_trajectory_max_cps_disk": ram + disk})):
        options.clear()
        if method != "dopri5":
            options.set_option("ts_adapt_type", "none")
        for k, v in dict(opts, ts_trajectory_solution_only=solution_only, ts_trajectory_dirname=str(tmp_path / "ck")).items():
            options.set_option(k, v)
        ode = petsc_adjoint.ODEPetsc(backend=CpuVecOps)
        if method == "imex":
            fI, fE = DiffusionIM(6), ReactionEX(6)
            ode.setupTS(y0, fI, step_size=0.05, method="imex", implicit_form=True, imex_form=True, func2=fE, batch_size=4)
            params = list(fI.parameters()) + list(fE.parameters())
        else:
            f = SpiralFunc() if method != "dopri5" else SpiralTruth()
            ode.setupTS(y0, f, step_size=0.05 if method != "dopri5" else 0.3, method=method, implicit_form=method == "cn")
            params = list(f.parameters())
        y = y0.clone().requires_grad_(True)
        p = ode.odeint_adjoint(y, t)
        p.abs().mean().backward()
        traj = ode._traj
        assert traj.high_water() <= ram + disk and traj.on_disk == (tag != "hbm")
        if tag == "two":
            assert type(traj).__name__ == "_TwoLevelTrajectory" and len(traj.chunks) <= 1
            st = traj.stats()
            assert st["files"] <= disk
            if traj.high_water() > ram:
                assert st["files"] > 0 and st["bytes_written"] > 0
        if tag == "disk":
            assert type(traj).__name__ == "_DiskTrajectory" and 0 < traj.stats()["files"] <= ram + disk
        res[tag] = (p.detach().clone(), y.grad.clone(), torch.cat([q.grad.reshape(-1) for q in params]).clone(), ode._nsteps)
    for tag in ("two", "disk"):
        a, b = res["hbm"], res[tag]
        assert a[3] == b[3] and a[3] >= 6
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]), tag


@pytest.mark.parametrize("method", ["cn", "beuler", "imex3", "imex_torch", "imexl2"])
def test_batched_parameter_accumulation_in_the_implicit_steppers_equals_per_stage_bitwise(method):
    """the theta and ARKIMEX steppers queue their stage results for pn_param_accum_multi like the explicit
    path does (-pn_param_accum batch, the default; step; and with few sources per launch) -- the parameter sensitivities are
    those of one launch per stage result (-pn_param_accum stage) bit for bit, with fewer launches."""
    from problems import DiffusionIM, ReactionEX
    torch.manual_seed(4)
    imex = method.startswith("imex")
    y0 = torch.randn(5, 6 if imex else 2, dtype=torch.float64)
    t = torch.tensor([0.0, 0.2, 0.45], dtype=torch.float64)
    w = torch.randn(3, 5, 6 if imex else 2, dtype=torch.float64)
    res = {}
    for mode, extra in (("stage", {}), ("step", {}), ("batch", {}), ("batch", {"pn_param_accum_sources": 3})):
        options.clear()
        for k, v in dict({"ts_adapt_type": "none", "pn_param_accum": mode}, **extra).items():
            options.set_option(k, v)
        if method in ("imex3", "imex_torch"):
            options.set_option("ts_arkimex_type", "3")
        if method == "imexl2":
            options.set_option("ts_arkimex_type", "l2")
        ode = petsc_adjoint.ODEPetsc(backend=CpuVecOps)
        if imex:
            fI, fE = DiffusionIM(6), ReactionEX(6)
            ode.setupTS(y0, fI, step_size=0.05, method="imex", implicit_form=True, imex_form=True, func2=fE, batch_size=5,
                        linear_solver="torch" if method == "imex_torch" else "petsc")
            params = list(fI.parameters()) + list(fE.parameters())
        else:
            f = SpiralFunc()
            ode.setupTS(y0, f, step_size=0.05, method=method, implicit_form=True)
            params = list(f.parameters())
        y = y0.clone().requires_grad_(True)
        (ode.odeint_adjoint(y, t) * w).sum().backward()
        res[(mode, tuple(extra))] = (y.grad.clone(), torch.cat([p.grad.reshape(-1) for p in params]).clone(), ode._ops.calls["param_accum"])
    ref = res[("stage", ())]
    for key, got in res.items():
        assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1]), key
    assert res[("batch", ())][2] < res[("step", ())][2] <= ref[2]          # (beuler: one stage result per step)
