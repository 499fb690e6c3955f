"""-m gpu: the HIP path behind ODEPetsc against the oracle on the same seeded inputs.

Tolerances: fp64 states <= 1e-11 relative (round-off of a different but fixed summation
order); fp32 states vs the fp64 oracle <= 1e-5 relative on gradients -- the bar of
BASELINE.json's north_star."""
import os

import pytest
import torch
import torch.nn as nn

from conftest import ROOT, require_gpu
from oracle.ts_oracle import ODEPetscOracle
from pnode_amd import options, petsc_adjoint
from problems import MLPFunc, SpiralFunc, SpiralTruth, TimeDependent, flat_grads, rel_err

pytestmark = pytest.mark.gpu


def _solve_pair(make_func, y0, t, target, method, opts, step_size=0.025, dtype=torch.float64, dev=None):
    f_ref = make_func(torch.float64)
    ref = ODEPetscOracle(opts)
    ref.setupTS(y0.double(), f_ref, step_size=step_size, method=method)
    yr = y0.double().clone().requires_grad_(True)
    pr = ref.odeint_adjoint(yr, t.double())
    torch.mean(torch.abs(pr - target.double())).backward()

    options.clear()
    for k, v in opts.items():
        options.set_option(k, v)
    f = make_func(dtype).to(dev)
    ode = petsc_adjoint.ODEPetsc()
    ode.setupTS(y0.to(dev, dtype), f, step_size=step_size, method=method)
    y = y0.to(dev, dtype).requires_grad_(True)
    p = ode.odeint_adjoint(y, t.to(dev))
    torch.mean(torch.abs(p - target.to(dev, dtype))).backward()
    return (pr, yr.grad, flat_grads(f_ref), ref), (p, y.grad, flat_grads(f), ode)


CASES = [
    ("rk4", {"ts_adapt_type": "none"}),
    ("rk4", {"ts_adapt_type": "none", "ts_trajectory_solution_only": 0}),
    ("rk4", {"ts_adapt_type": "none", "ts_trajectory_max_cps_ram": 3}),
    ("euler", {}),
    ("midpoint", {}),
    ("rk2", {}),
    ("bosh3", {}),
    ("dopri5", {}),
    ("dopri5", {"ts_trajectory_solution_only": 0}),
    ("dopri5", {"ts_trajectory_max_cps_ram": 2}),
    ("rk3", {"ts_adapt_type": "none"}),            # unknown name -> PETSc default 3bs
    ("euler", {"ts_rk_type": "5f"}),               # options database overrides `method`
]


@pytest.mark.parametrize("method,opts", CASES)
def test_spiral_batch_fp64(method, opts):
    dev = require_gpu()
    torch.manual_seed(0)
    y0 = torch.randn(20, 1, 2, dtype=torch.float64)
    t = torch.linspace(0.0, 25.0, 1001, dtype=torch.float64)[:10]
    target = torch.randn(10, 20, 1, 2, dtype=torch.float64)
    a, b = _solve_pair(lambda dt: SpiralFunc(dt), y0, t, target, method, opts, dev=dev)
    assert rel_err(b[0], a[0]) < 1e-11
    assert rel_err(b[1], a[1]) < 1e-11
    assert rel_err(b[2], a[2]) < 1e-11
    assert b[3].cur_sol_steps == a[3].cur_sol_steps


@pytest.mark.parametrize("method,h0,exact", [("dopri5", 0.5, 1), ("bosh3", 0.5, 1), ("dopri5", 0.2, 0), ("bosh3", 0.2, 0)])
def test_adaptive_with_rejections_matches_step_sequence(method, h0, exact):
    """Adaptive schemes on y' = y^3 A over sparse outputs: the controller rejects steps; the
    accepted (t,h) sequence and the rejection count must equal the oracle's and the committed
    self-golden.  h0 = 0.5 makes the first attempt blow up (|u| ~ 1e25): PETSc's rollback by
    subtraction then corrupts u_n, so the oracle is run with the exact restore the product
    uses (the product never overwrites u_n); h0 = 0.2 rejects without blowing up and the
    oracle runs the PETSc-style rollback."""
    import json
    import os
    dev = require_gpu()
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "dopri5_steps.json")))
    G = gold["%s_h%g" % (method, h0)]
    y0 = torch.tensor(gold["y0"], dtype=torch.float64)
    t = torch.tensor(gold["t"], dtype=torch.float64)
    target = torch.zeros(4, 3, 2, dtype=torch.float64)
    f_ref = SpiralTruth()
    ref = ODEPetscOracle({"oracle_exact_rollback": exact})
    ref.setupTS(y0, f_ref, step_size=h0, method=method)
    yr = y0.clone().requires_grad_(True)
    pr = ref.odeint_adjoint(yr, t)
    pr.abs().mean().backward()
    te, h, rej = ref.step_log()

    f = SpiralTruth().to(dev)
    ode = petsc_adjoint.ODEPetsc()
    ode.setupTS(y0.to(dev), f, step_size=h0, method=method)
    y = y0.to(dev).requires_grad_(True)
    p = ode.odeint_adjoint(y, t.to(dev))
    p.abs().mean().backward()
    assert ode._nsteps == len(h) == len(G["h"])
    assert ode._lib.pn_ts_rejections(ode._ts) == rej == G["rejections"] and rej > 0
    assert ode.cur_sol_steps == ref.cur_sol_steps == G["per_interval"]
    for k in range(len(h)):
        tk, hk = ode._step_info(k)
        assert hk == pytest.approx(h[k], rel=1e-9) and hk == pytest.approx(G["h"][k], rel=1e-9)
        assert tk + hk == pytest.approx(te[k], rel=1e-12)
    tol = 1e-9
    assert rel_err(p, pr) < tol and rel_err(p, torch.tensor(G["ans"], dtype=torch.float64)) < tol
    assert rel_err(y.grad, yr.grad) < tol and rel_err(y.grad, torch.tensor(G["gy0"], dtype=torch.float64)) < tol
    assert rel_err(f.A.grad, f_ref.A.grad) < tol and rel_err(f.A.grad, torch.tensor(G["gA"], dtype=torch.float64)) < tol


@pytest.mark.parametrize("method", ["rk4", "dopri5", "midpoint"])
def test_time_dependent_func_unused_params_single_end_time(method):
    """t=[T] mode (train-Cifar10.py:119), explicit use of t, a parameter with no gradient."""
    dev = require_gpu()
    torch.manual_seed(1)
    y0 = torch.randn(7, 5, dtype=torch.float64)
    t = torch.tensor([1.0], dtype=torch.float64)
    target = torch.randn(1, 7, 5, dtype=torch.float64)
    a, b = _solve_pair(lambda dt: TimeDependent(5, dt), y0, t, target, method, {"ts_adapt_type": "none"},
                       step_size=0.1, dev=dev)
    assert rel_err(b[0], a[0]) < 1e-11
    assert rel_err(b[1], a[1]) < 1e-11
    assert rel_err(b[2], a[2]) < 1e-11


def test_step_size_list_rober():
    """The reference's own test shape (tests/test_pnode.py:183-201): 1-D state of 3, variable
    step list, method 'rk3' -> 3bs; the known-answer constants are reproduced on the GPU."""
    import json
    import os
    import torch.nn as nn
    dev = require_gpu()
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "rober.json")))
    t = torch.tensor(gold["t"], dtype=torch.float64)
    true_y = torch.tensor(gold["true_y"], dtype=torch.float64)

    class Lambda(nn.Module):
        def __init__(self):
            super().__init__()
            self.k = nn.Parameter(torch.tensor([0.05, 4e7, 2e4], dtype=torch.float64))

        def forward(self, t, y):
            k1, k2, k3 = self.k[0], self.k[1], self.k[2]
            f1 = -k1 * y[0] + k3 * y[1] * y[2]
            f2 = k1 * y[0] - k3 * y[1] * y[2] - k2 * y[1] ** 2
            f3 = k2 * y[1] ** 2
            return torch.stack((f1, f2, f3), -1)

    options.set_option("ts_adapt_type", "none")
    options.set_option("ts_trajectory_type", "memory")
    f = Lambda().to(dev)
    ode = petsc_adjoint.ODEPetsc()
    ode.setupTS(true_y[0].to(dev), f, step_size=gold["step_size"], method="rk3", enable_adjoint=True)
    pred = ode.odeint_adjoint(true_y[0].to(dev), t.to(dev))
    loss = torch.mean(torch.abs(pred - true_y.to(dev)))
    loss.backward()
    std = torch.std(torch.abs(pred - true_y.to(dev)))
    assert loss.item() == pytest.approx(1.85e-6, abs=1e-6)          # the reference's assertion
    assert std.item() == pytest.approx(3.21e-6, abs=1e-6)
    assert loss.item() == pytest.approx(gold["explicit_3bs"]["loss"], rel=1e-9)
    assert std.item() == pytest.approx(gold["explicit_3bs"]["std"], rel=1e-9)
    assert rel_err(f.k.grad, torch.tensor(gold["explicit_3bs"]["grad_k"], dtype=torch.float64)) < 1e-9


def test_checkpoint_modes_bitwise_identical():
    """Gradients do not depend on the checkpoint schedule: store-all, solution-only and every
    budget replay the same kernels with the same step sizes."""
    dev = require_gpu()
    torch.manual_seed(2)
    y0 = torch.randn(64, 2)
    t = torch.tensor([0.0, 0.4, 1.0])
    res = []
    for opts in [{"ts_trajectory_solution_only": 0}, {}, {"ts_trajectory_max_cps_ram": 1},
                 {"ts_trajectory_max_cps_ram": 3}, {"ts_trajectory_max_cps_ram": 50}]:
        options.clear()
        options.set_option("ts_adapt_type", "none")
        for k, v in opts.items():
            options.set_option(k, v)
        f = SpiralFunc(torch.float32).to(dev)
        ode = petsc_adjoint.ODEPetsc()
        ode.setupTS(y0.to(dev), f, step_size=0.025, method="rk4")
        y = y0.to(dev).requires_grad_(True)
        p = ode.odeint_adjoint(y, t.to(dev))
        p.abs().mean().backward()
        res.append((p.detach().clone(), y.grad.clone(), flat_grads(f).clone(), ode._traj.high_water()))
    for r in res[1:]:
        assert torch.equal(r[0], res[0][0]) and torch.equal(r[1], res[0][1]) and torch.equal(r[2], res[0][2])
    assert res[2][3] <= 1 and res[3][3] <= 3 and res[4][3] <= 50


def test_target_config_fp32_gradient_accuracy():
    """BASELINE config C3a scaled to a batch the fp64 oracle finishes in seconds: 256 x 512
    state, 3x512 tanh MLP, rk4, 20 steps of 0.01; fp32 engine vs fp64 oracle <= 1e-5."""
    dev = require_gpu()
    torch.manual_seed(0)
    y0 = torch.randn(256, 512)
    t = torch.tensor([0.2])
    target = torch.randn(1, 256, 512)
    a, b = _solve_pair(lambda dt: MLPFunc(512, dt), y0, t, target, "rk4", {"ts_adapt_type": "none"},
                       step_size=0.01, dtype=torch.float32, dev=dev)
    assert b[3]._nsteps == 20
    assert rel_err(b[0], a[0]) < 1e-5
    assert rel_err(b[1], a[1]) < 1e-5
    assert rel_err(b[2], a[2]) < 1e-5


def test_target_config_full_batch_few_steps():
    """BASELINE config C3a at its full state size (4096 x 512 fp32, N = 2 097 152), 4 rk4 steps:
    the oracle (fp64) finishes this in seconds."""
    dev = require_gpu()
    torch.manual_seed(0)
    y0 = torch.randn(4096, 512)
    t = torch.tensor([0.04])
    target = torch.randn(1, 4096, 512)
    a, b = _solve_pair(lambda dt: MLPFunc(512, dt), y0, t, target, "rk4",
                       {"ts_adapt_type": "none", "ts_trajectory_solution_only": 0},
                       step_size=0.01, dtype=torch.float32, dev=dev)
    assert b[3]._nsteps == 4
    assert rel_err(b[0], a[0]) < 1e-5
    assert rel_err(b[1], a[1]) < 1e-5
    assert rel_err(b[2], a[2]) < 1e-5


def test_headline_config_full_length_fp32_engine_vs_fp64_engine_self_comparison():
    """ENGINE vs ENGINE (not the oracle): the whole headline solve (C3a: 4096 x 512, rk4, 100 steps) in fp32
    against the same HIP engine in fp64, all rows.  The oracle comparison over the full length is
    tests/test_gpu_configs.py::test_c3a_headline_100_steps_fp32_against_the_fp64_oracle_on_a_row_subset; the fp64
    engine itself is held to the fp64 oracle at 1e-11 by the tests above.  Bar: 1e-5 relative
    (BASELINE.json north_star); measured 8.5e-7 / 8.1e-7 / 5.4e-7."""
    dev = require_gpu()
    options.set_option("ts_adapt_type", "none")
    options.set_option("ts_trajectory_solution_only", "0")
    torch.manual_seed(0)
    y0 = torch.randn(4096, 512)
    target = torch.randn(1, 4096, 512)
    t = torch.tensor([1.0], dtype=torch.float64)
    res = {}
    for dt in (torch.float64, torch.float32):
        f = MLPFunc(512, dt).to(dev)
        ode = petsc_adjoint.ODEPetsc()
        ode.setupTS(y0.to(dev, dt), f, step_size=0.01, method="rk4")
        y = y0.to(dev, dt).requires_grad_(True)
        out = ode.odeint_adjoint(y, t.to(dev))
        torch.mean(torch.abs(out - target.to(dev, dt))).backward()
        assert ode._nsteps == 100
        res[dt] = (out.detach().double(), y.grad.double(), flat_grads(f).double())
        del ode, f
        torch.cuda.empty_cache()
    a, b = res[torch.float64], res[torch.float32]
    assert rel_err(b[0], a[0]) < 1e-5 and rel_err(b[1], a[1]) < 1e-5 and rel_err(b[2], a[2]) < 1e-5


def test_conv_dynamics_single_end_time():
    """A convolutional func on an image-shaped state (the shape family of BASELINE config C4,
    train-Cifar10.py:104-140: t=[1.0], rk4), scaled down; fp64 parity with the oracle."""
    import torch.nn as nn
    dev = require_gpu()

    class ConvFunc(nn.Module):
        def __init__(self, dtype):
            super().__init__()
            g = torch.Generator().manual_seed(5)
            self.c1 = nn.Conv2d(4, 8, 3, padding=1)
            self.c2 = nn.Conv2d(8, 4, 3, padding=1)
            for p in self.parameters():
                with torch.no_grad():
                    p.copy_(torch.randn(p.shape, generator=g) * 0.2)
            self.to(dtype)

        def forward(self, t, y):
            return self.c2(torch.relu(self.c1(y))) * (1.0 + t)

    torch.manual_seed(3)
    y0 = torch.randn(6, 4, 8, 8, dtype=torch.float64)
    t = torch.tensor([1.0], dtype=torch.float64)
    target = torch.randn(1, 6, 4, 8, 8, dtype=torch.float64)
    a, b = _solve_pair(lambda dt: ConvFunc(dt), y0, t, target, "rk4", {"ts_adapt_type": "none"},
                       step_size=0.25, dev=dev)
    assert b[3]._nsteps == 4
    assert rel_err(b[0], a[0]) < 1e-11 and rel_err(b[1], a[1]) < 1e-10 and rel_err(b[2], a[2]) < 1e-10


def test_retain_graph_mode_bitwise_identical_on_gpu():
    dev = require_gpu()
    torch.manual_seed(4)
    y0 = torch.randn(128, 64)
    t = torch.tensor([0.0, 0.1, 0.3])
    res = []
    for retain in (0, 1):
        options.clear()
        options.set_option("ts_adapt_type", "none")
        options.set_option("ts_trajectory_solution_only", 0)
        options.set_option("pn_trajectory_retain_graph", retain)        # the default is "auto" (tapes kept while they fit)
        f = MLPFunc(64, torch.float32, std=0.1).to(dev)
        ode = petsc_adjoint.ODEPetsc()
        ode.setupTS(y0.to(dev), f, step_size=0.02, method="rk4")
        y = y0.to(dev).requires_grad_(True)
        p = ode.odeint_adjoint(y, t.to(dev))
        p.abs().mean().backward()
        res.append((p.detach().clone(), y.grad.clone(), flat_grads(f).clone(), ode.nfe_backward))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][2], res[1][2])
    assert res[0][3] == 60 and res[1][3] == 0


@pytest.mark.parametrize("extra", [{}, {"pn_trajectory_retain_graph": 1}, {"ts_trajectory_solution_only": 1},
                                   {"ts_trajectory_max_cps_ram": 3}])
@pytest.mark.parametrize("times", [[0.3], [0.0, 0.1, 0.3]])
def test_hipgraph_capture_of_whole_sweeps_is_bitwise_identical(extra, times):
    """-pn_graph_capture 1: two eager warm-up calls, then the forward sweep and the reverse sweep
    are replayed from hipGraphs.  Same kernels, same order: results equal bit for bit, and they
    follow parameter updates and new inputs."""
    dev = require_gpu()
    torch.manual_seed(4)
    t = torch.tensor(times)

    def run(graph):
        options.clear()
        options.set_option("ts_adapt_type", "none")
        options.set_option("ts_trajectory_solution_only", 0)
        for k, v in extra.items():
            options.set_option(k, v)
        if graph:
            options.set_option("pn_graph_capture", 1)
        torch.manual_seed(7)
        f = MLPFunc(64, torch.float32, std=0.1).to(dev)
        ode = petsc_adjoint.ODEPetsc()
        y0 = torch.randn(128, 64, device=dev)
        ode.setupTS(y0, f, step_size=0.02, method="rk4")
        outs = []
        for it in range(5):
            yin = (y0 * (1.0 + 0.1 * it)).requires_grad_(True)       # new input every iteration
            for p in f.parameters():
                p.grad = None
            p_out = ode.odeint_adjoint(yin, t.to(dev))
            p_out.abs().mean().backward()
            outs.append((p_out.detach().clone(), yin.grad.clone(), flat_grads(f).clone()))
            with torch.no_grad():                                    # an "optimizer step", in place
                for p in f.parameters():
                    p.add_(p.grad, alpha=-0.01)
        return outs, ode

    eager, _ = run(False)
    graphed, ode = run(True)
    e = next(iter(ode._graphs.values()))
    assert e.g_f is not None and e.g_b is not None and e.calls == 2
    for a, b in zip(eager, graphed):
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])


@pytest.mark.parametrize("accum", ["stage", "step"])
def test_hipgraph_replay_survives_stream_syncs_at_headline_width(accum):
    """ROCm 7.2: with CLR's AQL packet capture on, a replayed graph that contains PyTorch's two-pass
    reduction (the bias gradients of Linear(512,512) at batch 4096) returns wrong numbers after any
    hipStreamSynchronize -- reproduced without pnode_amd by tools/graph_sum_repro2.py.  pnode_amd
    switches packet capture off at import (pnode_amd/__init__.py); this is the regression test:
    replays separated by device and stream synchronisations equal the eager solve bit for bit, for
    both ways of accumulating the parameter sensitivities."""
    import pnode_amd
    dev = require_gpu()
    assert pnode_amd.GRAPH_REPLAY_SAFE and os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE") == "0"
    torch.manual_seed(0)
    f = MLPFunc(512, torch.float32).to(dev)
    y0 = torch.randn(4096, 512, device=dev)
    t = torch.tensor([0.02])

    def make(graph):
        options.clear()
        options.set_option("ts_adapt_type", "none")
        options.set_option("ts_trajectory_solution_only", 0)
        options.set_option("pn_param_accum", accum)
        if graph:
            options.set_option("pn_graph_capture", 1)
        ode = petsc_adjoint.ODEPetsc()
        ode.setupTS(y0, f, step_size=0.01, method="rk4")
        options.clear()
        return ode

    def solve(ode):
        for p in f.parameters():
            p.grad = None
        y = y0.detach().requires_grad_(True)
        out = ode.odeint_adjoint(y, t)
        out.abs().mean().backward()
        return out.detach().clone(), y.grad.clone(), [p.grad.clone() for p in f.parameters()]

    ref = solve(make(False))
    torch.cuda.synchronize()
    ode = make(True)
    for it in range(6):
        got = solve(ode)
        if it % 2:
            torch.cuda.synchronize()
        else:
            torch.cuda.current_stream().synchronize()
        assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1]), it
        for k, (a, b2) in enumerate(zip(got[2], ref[2])):
            assert torch.equal(a, b2), (it, k)
    assert ode.graphs_captured


def test_all_launch_and_checkpoint_modes_agree_bitwise_at_headline_width():
    """Small-shape mode tests cannot see what depends on the shapes (GEMM algorithm choice, two-pass
    reductions, 256 MiB trajectory chunks): at batch 4096 x 512 every mode -- eager / hipGraph,
    store-all / solution-only / budgeted checkpoints, retained tapes, both accumulate modes -- must
    give the same bits, with synchronisations between the solves."""
    dev = require_gpu()
    torch.manual_seed(0)
    f = MLPFunc(512, torch.float32).to(dev)
    y0 = torch.randn(4096, 512, device=dev)
    t = torch.tensor([0.0, 0.03, 0.05])

    def run(extra, reps):
        options.clear()
        options.set_option("ts_adapt_type", "none")
        options.set_option("ts_trajectory_solution_only", 0)
        for k, v in extra.items():
            options.set_option(k, v)
        ode = petsc_adjoint.ODEPetsc()
        ode.setupTS(y0, f, step_size=0.01, method="rk4")
        options.clear()
        out = None
        for it in range(reps):
            for p in f.parameters():
                p.grad = None
            y = y0.detach().requires_grad_(True)
            sol = ode.odeint_adjoint(y, t)
            (sol[1].abs().mean() + sol[2].pow(2).mean()).backward()
            got = (sol.detach().clone(), y.grad.clone(), flat_grads(f).clone())
            torch.cuda.synchronize()
            if out is not None:
                assert all(torch.equal(a, b2) for a, b2 in zip(got, out)), (extra, it)
            out = got
        return out, ode

    ref, _ = run({"pn_graph_capture": 0}, 2)
    for extra in ({"pn_graph_capture": 1}, {"pn_graph_capture": 1, "pn_param_accum": "step"},
                  {"pn_graph_capture": 1, "pn_trajectory_retain_graph": 1}, {"pn_graph_capture": 0, "pn_trajectory_retain_graph": 1},
                  {"pn_graph_capture": 0, "ts_trajectory_solution_only": 1}, {"pn_graph_capture": 1, "ts_trajectory_solution_only": 1},
                  {"pn_graph_capture": 0, "ts_trajectory_max_cps_ram": 2}, {"pn_graph_capture": 1, "ts_trajectory_max_cps_ram": 2},
                  {}, {"ts_trajectory_max_cps_ram": 2}, {"pn_step_loop": "python"}, {"pn_graph_capture": 0, "pn_step_loop": "python"}):
        got, ode = run(extra, 5)
        assert ode._nsteps == 5 and ode.cur_sol_steps == [0, 3, 2]
        # (no launch option: -pn_graph_capture auto, the default since round 4, captures at the third call)
        assert bool(ode.graphs_captured) == (extra.get("pn_graph_capture", "auto") != 0), extra
        for a, b2 in zip(got, ref):
            assert torch.equal(a, b2), extra


def test_graph_capture_is_refused_when_the_runtime_was_initialised_first():
    """The packet-capture switch is read when the HIP runtime initialises.  A process that touches
    the GPU before importing pnode_amd (and has not exported the variable) gets a warning and eager
    launches instead of graphs that may replay wrongly."""
    import subprocess
    import sys
    require_gpu()
    code = r"""
import os, sys, warnings
os.environ.pop("DEBUG_CLR_GRAPH_PACKET_CAPTURE", None)
import torch
torch.zeros(1, device="cuda")
sys.path.insert(0, %r); sys.path.insert(0, %r)
import pnode_amd
from pnode_amd import options, petsc_adjoint
from problems import MLPFunc
assert not pnode_amd.GRAPH_REPLAY_SAFE
options.set_option("ts_adapt_type", "none"); options.set_option("pn_graph_capture", 1)
f = MLPFunc(64).cuda(); y0 = torch.randn(32, 64, device="cuda")
ode = petsc_adjoint.ODEPetsc(); ode.setupTS(y0, f, step_size=0.05, method="rk4")
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    for _ in range(4):
        y = y0.clone().requires_grad_(True); ode.odeint_adjoint(y, torch.tensor([0.2])).sum().backward()
assert not ode.graphs_captured
assert sum("pn_graph_capture ignored" in str(x.message) for x in w) == 1, [str(x.message) for x in w]
print("REFUSED-OK")
""" % (ROOT, os.path.join(ROOT, "tests"))
    env = {k: v for k, v in os.environ.items() if k != "DEBUG_CLR_GRAPH_PACKET_CAPTURE"}
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "REFUSED-OK" in r.stdout, r.stdout + r.stderr


def test_graph_replay_self_test_detects_the_defect():
    """Whether the HIP runtime was initialised before the switch was set cannot be queried (a profiler's
    tool library does it before Python starts), so the first capture is preceded by a self-test that runs
    the failing pattern itself.  It must pass in this (guarded) process and fail in a process where
    packet capture is left on."""
    import subprocess
    import sys
    from pnode_amd import _graphcheck
    dev = require_gpu()
    assert _graphcheck.replay_is_sound(dev)
    code = ("import sys, torch; sys.path.insert(0, %r); import pnode_amd; from pnode_amd import _graphcheck; "
            "print('SOUND', _graphcheck.replay_is_sound(torch.device('cuda:0')), pnode_amd.GRAPH_REPLAY_SAFE)" % ROOT)
    env = dict(os.environ, DEBUG_CLR_GRAPH_PACKET_CAPTURE="1")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "SOUND False False" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("shape", [(1,), (3,), (5, 1), (7, 9), (2, 3, 5), (1, 1031)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_ragged_state_sizes_through_the_whole_solver(shape, dtype):
    """numel not a multiple of the 16-byte vector width (tail path) and padded trajectory slots;
    adaptive dopri5 so that the error-norm kernel sees the ragged tail as well."""
    import math
    dev = require_gpu()
    n = math.prod(shape)

    def make(dt):
        return TimeDependent(shape[-1], dt)

    torch.manual_seed(n)
    y0 = torch.randn(*shape, dtype=torch.float64)
    t = torch.tensor([0.0, 0.3, 0.7], dtype=torch.float64)
    target = torch.randn(3, *shape, dtype=torch.float64)
    a, b = _solve_pair(make, y0, t, target, "dopri5", {"ts_rtol": 1e-6, "ts_atol": 1e-6, "oracle_exact_rollback": 1},
                       step_size=0.1, dtype=dtype, dev=dev)
    tol = 1e-10 if dtype == torch.float64 else 2e-4
    assert rel_err(b[0], a[0]) < tol and rel_err(b[1], a[1]) < tol and rel_err(b[2], a[2]) < tol
    if dtype == torch.float64:
        assert b[3]._nsteps == len(a[3].step_log()[1])


def test_parameter_free_dynamics_and_input_without_grad():
    """func with no trainable parameter (np = 0): only dL/dy0 flows; and an input that does not
    require grad with trainable parameters: only dL/dtheta flows."""
    import torch.nn as nn
    dev = require_gpu()

    class NoParam(nn.Module):
        def forward(self, t, y):
            return -y * y.abs()

    options.set_option("ts_adapt_type", "none")
    y0 = torch.rand(33, 3, dtype=torch.float64) + 0.5
    f_ref = NoParam()
    ref = ODEPetscOracle({"ts_adapt_type": "none"})
    ref.setupTS(y0, f_ref, step_size=0.05, method="rk4")
    yr = y0.clone().requires_grad_(True)
    ref.odeint_adjoint(yr, torch.tensor([0.5], dtype=torch.float64)).sum().backward()
    ode = petsc_adjoint.ODEPetsc()
    ode.setupTS(y0.to(dev), NoParam().to(dev), step_size=0.05, method="rk4")
    y = y0.to(dev).requires_grad_(True)
    ode.odeint_adjoint(y, torch.tensor([0.5], dtype=torch.float64)).sum().backward()
    assert ode.np == 0 and rel_err(y.grad, yr.grad) < 1e-12

    f = SpiralFunc(torch.float64).to(dev)
    ode2 = petsc_adjoint.ODEPetsc()
    yin = torch.randn(16, 2, dtype=torch.float64, device=dev)
    ode2.setupTS(yin, f, step_size=0.05, method="rk4")
    out = ode2.odeint_adjoint(yin, torch.tensor([0.0, 0.2], dtype=torch.float64))   # yin.requires_grad is False
    out.pow(2).sum().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in f.parameters())
    f_ref2 = SpiralFunc(torch.float64)
    ref2 = ODEPetscOracle({"ts_adapt_type": "none"})
    ref2.setupTS(yin.cpu(), f_ref2, step_size=0.05, method="rk4")
    ref2.odeint_adjoint(yin.cpu(), torch.tensor([0.0, 0.2], dtype=torch.float64)).pow(2).sum().backward()
    assert rel_err(flat_grads(f), flat_grads(f_ref2)) < 1e-11


def test_two_solver_objects_and_reuse_across_iterations():
    """Separate objects for training and validation (Burgers.py:348-350 pattern), each reused over
    several iterations with changing parameters; setupTS called again every iteration like
    train-Cifar10.py:121-139 does."""
    dev = require_gpu()
    options.set_option("ts_adapt_type", "none")
    torch.manual_seed(0)
    f = SpiralFunc(torch.float32).to(dev)
    train, val = petsc_adjoint.ODEPetsc(), petsc_adjoint.ODEPetsc()
    y0 = torch.randn(64, 2, device=dev)
    yv = torch.randn(8, 2, device=dev)
    t = torch.tensor([0.0, 0.25, 0.5])
    losses = []
    for it in range(4):
        train.setupTS(y0, f, step_size=0.05, method="rk4")
        for p in f.parameters():
            p.grad = None
        loss = train.odeint_adjoint(y0, t).abs().mean()
        loss.backward()
        with torch.no_grad():
            for p in f.parameters():
                p.add_(p.grad, alpha=-0.05)
            val.setupTS(yv, f, step_size=0.05, method="rk4", enable_adjoint=False)
            v = val.odeint_adjoint(yv, t)
        assert v.shape == (3, 8, 2) and torch.isfinite(v).all()
        losses.append(loss.item())
    assert losses[-1] < losses[0]            # gradient descent on mean|y(t)| makes progress


@pytest.mark.parametrize("method", ["cn", "beuler"])
def test_theta_methods_reference_known_answer_on_gpu(method):
    """The reference's implicit test (tests/test_pnode.py:133-152) on the HIP path, PETSc-default
    Newton/GMRES tolerances."""
    import json
    import os
    from test_oracle_pins import Rober
    dev = require_gpu()
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "rober.json")))
    t = torch.tensor(gold["t"], dtype=torch.float64)
    true_y = torch.tensor(gold["true_y"], dtype=torch.float64).to(dev)
    options.set_option("ts_adapt_type", "none")
    f = Rober().to(dev)
    ode = petsc_adjoint.ODEPetsc()
    ode.setupTS(true_y[0], f, step_size=gold["step_size"], method=method, enable_adjoint=True, implicit_form=True)
    pred = ode.odeint_adjoint(true_y[0], t.to(dev))
    loss = torch.mean(torch.abs(pred - true_y))
    loss.backward()
    std = torch.std(torch.abs(pred - true_y))
    G = gold["implicit_cn" if method == "cn" else "implicit_beuler"]
    if method == "cn":
        assert loss.item() == pytest.approx(1.85e-6, abs=1e-6) and std.item() == pytest.approx(3.36e-6, abs=1e-6)
    assert loss.item() == pytest.approx(G["loss"], rel=1e-6) and std.item() == pytest.approx(G["std"], rel=1e-6)
    assert rel_err(f.k.grad, torch.tensor(G["grad_k"], dtype=torch.float64)) < 1e-5


@pytest.mark.parametrize("method", ["cn", "beuler"])
@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-10), (torch.float32, 2e-4)])
def test_theta_methods_against_exact_newton_oracle(method, dtype, tol):
    from oracle.theta_oracle import odeint_adjoint_theta
    dev = require_gpu()
    torch.manual_seed(0)
    y0 = torch.randn(64, 6, dtype=torch.float64)
    t = torch.tensor([0.0, 0.3, 0.5, 1.0], dtype=torch.float64)
    target = torch.randn(4, 64, 6, dtype=torch.float64)
    tight = dtype == torch.float64
    for k, v in {"ts_adapt_type": "none", "snes_rtol": 1e-13 if tight else 1e-6, "snes_stol": 1e-14 if tight else 1e-7,
                 "ksp_rtol": 1e-12 if tight else 1e-6}.items():
        options.set_option(k, v)
    f = TimeDependent(6, dtype).to(dev)
    ode = petsc_adjoint.ODEPetsc()
    ode.setupTS(y0.to(dev, dtype), f, step_size=0.1, method=method, implicit_form=True)
    y = y0.to(dev, dtype).requires_grad_(True)
    p = ode.odeint_adjoint(y, t.to(dev))
    torch.mean(torch.abs(p - target.to(dev, dtype))).backward()
    # the dense-Jacobian oracle is O(n^3): same dynamics on a few rows (the func is row-wise)
    rows = slice(0, 4)
    f2 = TimeDependent(6)
    y2 = y0[rows].clone().requires_grad_(True)
    p2 = odeint_adjoint_theta(f2, y2, t, 0.1, method)
    assert rel_err(p[:, rows], p2) < tol
    (torch.abs(p2 - target[:, rows]).sum() / target.numel()).backward()
    assert rel_err(y.grad[rows], y2.grad) < tol * 10
    assert ode._nsteps == 10 and ode._theta.linear_its > 0


@pytest.mark.parametrize("method", ["beuler", "cn"])
def test_singular_mass_matrix_dae_on_gpu(method):
    """implicit_form=True with a singular mass matrix (index-1 DAE, pendulum_DAE.py's use) on the HIP
    path against the exact-Newton oracle; row-wise (d x d) mass on the batch."""
    from oracle.theta_oracle import odeint_adjoint_theta
    from problems import SemiExplicitDAE
    dev = require_gpu()
    torch.manual_seed(1)
    f0 = SemiExplicitDAE()
    u0 = f0.consistent(torch.randn(6, 3, dtype=torch.float64))
    t = torch.tensor([0.0, 0.2, 0.5], dtype=torch.float64)
    target = torch.randn(3, 6, 5, dtype=torch.float64)
    for k, v in {"ts_adapt_type": "none", "snes_rtol": 1e-14, "snes_stol": 1e-15, "snes_atol": 1e-14, "ksp_rtol": 1e-13}.items():
        options.set_option(k, v)
    f = SemiExplicitDAE().to(dev)
    ode = petsc_adjoint.ODEPetsc()
    ode.setupTS(u0.to(dev), f, step_size=0.1, method=method, implicit_form=True, mass=SemiExplicitDAE.mass().to(dev))
    u = u0.to(dev).requires_grad_(True)
    p = ode.odeint_adjoint(u, t.to(dev))
    torch.mean(torch.abs(p - target.to(dev))).backward()
    f2 = SemiExplicitDAE()
    u2 = u0.clone().requires_grad_(True)
    p2 = odeint_adjoint_theta(f2, u2, t, 0.1, method, mass=torch.kron(torch.eye(6, dtype=torch.float64), SemiExplicitDAE.mass()))
    torch.mean(torch.abs(p2 - target)).backward()
    assert rel_err(p, p2) < 1e-12 and rel_err(u.grad, u2.grad) < 1e-10 and rel_err(flat_grads(f), flat_grads(f2)) < 1e-10


def test_imex_reference_known_answer_on_gpu():
    """The reference's third integration test (tests/test_pnode.py:155-180) on the HIP path."""
    import json
    import os
    from problems import RoberEX, RoberIM
    dev = require_gpu()
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "rober.json")))
    t = torch.tensor(gold["t"], dtype=torch.float64)
    true_y = torch.tensor(gold["true_y"], dtype=torch.float64).to(dev)
    options.set_option("ts_adapt_type", "none")
    fI, fE = RoberIM().to(dev), RoberEX().to(dev)
    ode = petsc_adjoint.ODEPetsc()
    ode.setupTS(true_y[0], fI, step_size=gold["step_size"], method="imex", enable_adjoint=True,
                implicit_form=True, imex_form=True, func2=fE)
    pred = ode.odeint_adjoint(true_y[0], t.to(dev))
    loss = torch.mean(torch.abs(pred - true_y))
    loss.backward()
    std = torch.std(torch.abs(pred - true_y))
    assert loss.item() == pytest.approx(3.11e-6, abs=3e-6) and std.item() == pytest.approx(5.65e-6, abs=3e-6)
    assert loss.item() == pytest.approx(gold["imex_3"]["loss"], rel=1e-6)
    g = torch.cat([fI.k1.grad, fI.k3.grad, fE.k2.grad])
    assert rel_err(g, torch.tensor(gold["imex_3"]["grad"], dtype=torch.float64)) < 1e-5


@pytest.mark.parametrize("name", ["3", "4", "5", "l2", "ars443", "1bee", "2e", "prssp2", "bpr3"])
@pytest.mark.parametrize("linear_solver", ["petsc", "torch"])
def test_imex_burgers_like_split_on_gpu(name, linear_solver):
    from oracle.arkimex_oracle import odeint_adjoint_arkimex
    from problems import DiffusionIM, ReactionEX
    dev = require_gpu()
    torch.manual_seed(0)
    y0 = torch.randn(3, 6, dtype=torch.float64)
    t = torch.tensor([0.0, 0.1, 0.25], dtype=torch.float64)
    target = torch.randn(3, 3, 6, dtype=torch.float64)
    for k, v in {"ts_adapt_type": "none", "ts_arkimex_type": name, "snes_rtol": 1e-14, "snes_stol": 1e-15,
                 "ksp_rtol": 1e-13}.items():
        options.set_option(k, v)
    if linear_solver == "torch":
        options.set_option("snes_type", "ksponly")
    fI, fE = DiffusionIM(6).to(dev), ReactionEX(6).to(dev)
    ode = petsc_adjoint.ODEPetsc()
    ode.setupTS(y0.to(dev), fI, step_size=0.05, method="imex", implicit_form=True, imex_form=True, func2=fE,
                batch_size=3, linear_solver=linear_solver, matrixfree_jacobian=False)
    y = y0.to(dev).requires_grad_(True)
    p = ode.odeint_adjoint(y, t.to(dev))
    torch.mean(torch.abs(p - target.to(dev))).backward()
    fI2, fE2 = DiffusionIM(6), ReactionEX(6)
    y2 = y0.clone().requires_grad_(True)
    p2 = odeint_adjoint_arkimex(fI2, fE2, y2, t, 0.05, name)
    torch.mean(torch.abs(p2 - target)).backward()
    assert rel_err(p, p2) < 1e-11 and rel_err(y.grad, y2.grad) < 1e-9
    assert rel_err(flat_grads(fI), flat_grads(fI2)) < 1e-9 and rel_err(flat_grads(fE), flat_grads(fE2)) < 1e-9


@pytest.mark.parametrize("name", ["3", "l2"])
@pytest.mark.parametrize("times", [[0.25], [0.0, 0.1, 0.25]])
def test_imex_direct_solve_sweeps_replay_from_hipgraphs_bitwise(name, times):
    """-pn_graph_capture with ARKIMEX + -snes_type ksponly + linear_solver="torch" (the Burgers run
    script's combination): the sweeps have no host synchronisation and are captured; d funcIM/du and
    its LU factors are recomputed eagerly before every replay into the tensors the graphs read, so the
    replays follow in-place parameter updates (trainable viscosity in funcIM) bit for bit."""
    from problems import DiffusionIM, ReactionEX
    dev = require_gpu()
    t = torch.tensor(times, dtype=torch.float64)

    def run(graph):
        options.clear()
        for k, v in {"ts_adapt_type": "none", "ts_arkimex_type": name, "snes_type": "ksponly"}.items():
            options.set_option(k, v)
        if graph is not None:
            options.set_option("pn_graph_capture", graph)
        else:
            options.set_option("pn_graph_revalidate", 2)       # (the default mode: also re-validate an existing pair, at call 4)
        torch.manual_seed(5)
        fI, fE = DiffusionIM(16).to(dev), ReactionEX(16).to(dev)
        y0 = torch.randn(8, 16, dtype=torch.float64, device=dev)
        ode = petsc_adjoint.ODEPetsc()
        ode.setupTS(y0, fI, step_size=0.05, method="imex", implicit_form=True, imex_form=True, func2=fE,
                    batch_size=8, linear_solver="torch", matrixfree_jacobian=False)
        options.clear()
        outs = []
        params = list(fI.parameters()) + list(fE.parameters())
        for it in range(6):
            for p in params:
                p.grad = None
            yin = (y0 * (1.0 + 0.1 * it)).requires_grad_(True)
            sol = ode.odeint_adjoint(yin, t.to(dev))
            sol.abs().mean().backward()
            outs.append((sol.detach().clone(), yin.grad.clone(), torch.cat([p.grad.reshape(-1) for p in params]).clone()))
            torch.cuda.synchronize()
            with torch.no_grad():
                for p in params:
                    p.add_(p.grad, alpha=-0.05)
        return outs, ode

    eager, _ = run(0)
    graphed, ode = run(1)
    assert ode.graphs_captured and ode._theta.capturable()
    for it, (a, b2) in enumerate(zip(eager, graphed)):
        assert torch.equal(a[0], b2[0]) and torch.equal(a[1], b2[1]) and torch.equal(a[2], b2[2]), it
    assert not torch.equal(eager[0][2], eager[-1][2])
    # round 5: the same through the DEFAULT launch mode (auto): the call that captures the sweeps also runs them eagerly, in
    # the same solver object, and the first replays reproduce that twin bit for bit -- which they did not while the captured
    # pass emptied the eager solve's factor cache (pnode_amd/theta.py::odeint; the reverse sweep then refactored with the LAST
    # step's shift, one ulp off the first step's).  Reference: pnode/torch_linearsolve.py:15-35, pa.py:792-799.
    default, ode_d = run(None)
    assert ode_d.graph_status == "graph(auto)", ode_d.graph_status
    for it, (a, b2) in enumerate(zip(eager, default)):
        assert torch.equal(a[0], b2[0]) and torch.equal(a[1], b2[1]) and torch.equal(a[2], b2[2]), it


@pytest.mark.parametrize("method", ["cn", "beuler"])
def test_theta_direct_solve_sweeps_on_gpu_eager_and_replayed(method):
    """implicit_form=True + linear_solver="torch" + -snes_type ksponly on the HIP path: equals the oracle's exact
    solve for a func that is linear in u (nonsymmetric), and the hipGraph replay equals the eager sweep bit for bit
    while the parameters change in place."""
    from oracle.theta_oracle import odeint_adjoint_theta
    from problems import AdvectionDiffusionIM
    dev = require_gpu()
    n, B = 16, 6
    torch.manual_seed(0)
    y0 = torch.randn(B, n, dtype=torch.float64)
    t = torch.tensor([0.0, 0.1, 0.25], dtype=torch.float64)
    target = torch.randn(3, B, n, dtype=torch.float64)

    def run(graph):
        options.clear()
        for k, v in {"ts_adapt_type": "none", "snes_type": "ksponly"}.items():
            options.set_option(k, v)
        if graph:
            options.set_option("pn_graph_capture", 1)
        f = AdvectionDiffusionIM(n).to(dev)
        ode = petsc_adjoint.ODEPetsc()
        ode.setupTS(y0.to(dev), f, step_size=0.05, method=method, implicit_form=True, batch_size=B, linear_solver="torch")
        options.clear()
        outs = []
        for it in range(5):
            for p in f.parameters():
                p.grad = None
            y = y0.to(dev).requires_grad_(True)
            sol = ode.odeint_adjoint(y, t.to(dev))
            torch.mean(torch.abs(sol - target.to(dev))).backward()
            outs.append((sol.detach().clone(), y.grad.clone(), flat_grads(f).clone()))
            torch.cuda.synchronize()
            with torch.no_grad():
                for p in f.parameters():
                    p.add_(p.grad, alpha=-0.05)
        return outs, ode

    eager, _ = run(False)
    graphed, ode = run(True)
    assert ode.graphs_captured
    for a, b2 in zip(eager, graphed):
        assert torch.equal(a[0], b2[0]) and torch.equal(a[1], b2[1]) and torch.equal(a[2], b2[2])
    f2 = AdvectionDiffusionIM(n)
    y2 = y0.clone().requires_grad_(True)
    p2 = odeint_adjoint_theta(f2, y2, t, 0.05, method)
    torch.mean(torch.abs(p2 - target)).backward()
    assert rel_err(eager[0][0], p2) < 1e-12 and rel_err(eager[0][1], y2.grad) < 1e-10 and rel_err(eager[0][2], flat_grads(f2)) < 1e-10


def test_uncapturable_func_falls_back_to_eager_launches():
    """A func that synchronises with the host (`.item()`) cannot be captured: -pn_graph_capture then warns once,
    switches itself off for that solver and the sweeps run eagerly with the same results."""
    import warnings as _w
    import torch.nn as nn
    dev = require_gpu()

    class HostSync(nn.Module):
        def __init__(self):
            super().__init__()
            self.lin = nn.Linear(4, 4)

        def forward(self, t, y):
            scale = float(y.abs().max().item() > -1.0)         # host round trip inside func
            return torch.tanh(self.lin(y)) * scale

    torch.manual_seed(0)
    y0 = torch.randn(8, 4, device=dev)
    res = {}
    for graph in (False, True):
        options.clear()
        options.set_option("ts_adapt_type", "none")
        if graph:
            options.set_option("pn_graph_capture", 1)
        torch.manual_seed(1)
        f = HostSync().to(dev)
        ode = petsc_adjoint.ODEPetsc()
        ode.setupTS(y0, f, step_size=0.05, method="rk4")
        with _w.catch_warnings(record=True) as caught:
            _w.simplefilter("always")
            for _ in range(5):
                for p in f.parameters():
                    p.grad = None
                y = y0.clone().requires_grad_(True)
                out = ode.odeint_adjoint(y, torch.tensor([0.2]))
                out.sum().backward()
        res[graph] = (out.detach().clone(), y.grad.clone(), flat_grads(f).clone())
        if graph:
            assert not ode.graphs_captured and not ode._graph_mode
            assert sum("switched off for this solver" in str(c.message) for c in caught) == 1
    assert all(torch.equal(a, b2) for a, b2 in zip(res[True], res[False]))


def test_iterative_implicit_solves_are_never_captured():
    """Newton/GMRES iterations read norms on the host: -pn_graph_capture leaves them eager."""
    from problems import DiffusionIM, ReactionEX
    dev = require_gpu()
    for k, v in {"ts_adapt_type": "none", "pn_graph_capture": 1}.items():
        options.set_option(k, v)
    fI, fE = DiffusionIM(8).to(dev), ReactionEX(8).to(dev)
    y0 = torch.randn(4, 8, dtype=torch.float64, device=dev)
    ode = petsc_adjoint.ODEPetsc()
    ode.setupTS(y0, fI, step_size=0.05, method="imex", implicit_form=True, imex_form=True, func2=fE, batch_size=4)
    for _ in range(4):
        y = y0.clone().requires_grad_(True)
        ode.odeint_adjoint(y, torch.tensor([0.1], dtype=torch.float64)).sum().backward()
    assert not ode.graphs_captured and not ode._theta.capturable()


def test_log_view_prints_a_summary_at_exit():
    """-log_view (PETSc's spelling): per-entry-point launch counts and times at interpreter exit."""
    import subprocess
    import sys
    require_gpu()
    code = r"""
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import pnode_amd, torch
pnode_amd.init(["prog", "-ts_adapt_type", "none", "-log_view", "-ts_trajectory_solution_only", "1"])
from pnode_amd import petsc_adjoint
from problems import SpiralFunc
f = SpiralFunc(torch.float32).cuda(); y0 = torch.randn(16, 2, device="cuda")
ode = petsc_adjoint.ODEPetsc(); ode.setupTS(y0, f, step_size=0.05, method="rk4")
for _ in range(3):
    y = y0.clone().requires_grad_(True); ode.odeint_adjoint(y, torch.tensor([0.5])).sum().backward()
""" % (ROOT, os.path.join(ROOT, "tests"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "pnode_amd -log_view: 3 forward sweeps, 3 reverse sweeps, 30 accepted time steps, 0 rejected attempts" in r.stdout
    line = [l for l in r.stdout.splitlines() if l.startswith("pn_rk_stage")][0].split()
    # per time step: 3 stage + 1 combine launch forward, 3 stage launches recomputed in reverse (-ts_trajectory_solution_only 1;
    # without the option the stages of so short a trajectory are kept and nothing is recomputed)
    assert int(line[1]) == 3 * 10 * (4 + 3)


def test_no_grad_solve_and_nfe_counts():
    dev = require_gpu()
    options.set_option("ts_adapt_type", "none")
    f = SpiralFunc(torch.float32).to(dev)
    ode = petsc_adjoint.ODEPetsc()
    y0 = torch.randn(8, 2, device=dev)
    ode.setupTS(y0, f, step_size=0.05, method="rk4", enable_adjoint=False)
    with torch.no_grad():
        out = ode.odeint_adjoint(y0, torch.tensor([0.0, 0.5, 1.0]))
    assert out.shape == (3, 8, 2) and f.nfe == 4 * 20
    assert ode._traj is None
    with pytest.raises(ValueError):
        ode2 = petsc_adjoint.ODEPetsc()
        ode2.setupTS(y0, lambda t, y: y, step_size=0.1, method="euler")
        ode2.odeint_adjoint(y0, torch.tensor([1.0]))


def test_round_trip_at_full_size():
    """Size-independent property at BASELINE's full size (4096 x 512, fp32): integrating
    u' = -u forward and the adjoint of sum(u(T)) gives exp(-T)-like factors that are known in
    closed form for rk4: every component is multiplied by R(h)^n, R = 1 - h + h^2/2 - h^3/6 + h^4/24."""
    import torch.nn as nn
    dev = require_gpu()

    class Decay(nn.Module):
        def __init__(self):
            super().__init__()
            self.a = nn.Parameter(torch.tensor(-1.0))

        def forward(self, t, y):
            return self.a * y

    options.set_option("ts_adapt_type", "none")
    options.set_option("ts_trajectory_solution_only", "0")
    f = Decay().to(dev)
    y0 = torch.randn(4096, 512, device=dev)
    ode = petsc_adjoint.ODEPetsc()
    ode.setupTS(y0, f, step_size=0.1, method="rk4")
    y = y0.clone().requires_grad_(True)
    out = ode.odeint_adjoint(y, torch.tensor([1.0]))
    out.sum().backward()
    h = 0.1
    R = 1 - h + h ** 2 / 2 - h ** 3 / 6 + h ** 4 / 24
    assert torch.allclose(out[0], y0 * R ** 10, rtol=2e-6, atol=1e-7)
    assert torch.allclose(y.grad, torch.full_like(y0, R ** 10), rtol=2e-6)


@pytest.mark.parametrize("solution_only", [0, 1])
def test_disk_tier_on_gpu_equals_hbm_tier_bitwise_at_headline_width(tmp_path, solution_only):
    """-ts_trajectory_type basic on the HIP path: slots leave for their files through pinned staging buffers
    (hipMemcpyAsync on the solver's stream + I/O thread) and come back in the reverse sweep with read-ahead;
    4096 x 512 fp32 (32 MiB slots in store-all mode), 12 rk4 steps.  Same bits as the HBM tier."""
    dev = require_gpu()
    torch.manual_seed(0)
    y0 = torch.randn(4096, 512, device=dev)
    t = torch.tensor([0.0, 0.05, 0.12])
    res = {}
    for ttype in ("memory", "basic"):
        options.clear()
        for k, v in {"ts_adapt_type": "none", "ts_trajectory_solution_only": solution_only, "ts_trajectory_type": ttype,
                     "ts_trajectory_dirname": str(tmp_path / "ckpt")}.items():
            options.set_option(k, v)
        f = MLPFunc(512, torch.float32).to(dev)
        ode = petsc_adjoint.ODEPetsc()
        ode.setupTS(y0, f, step_size=0.01, method="rk4")
        y = y0.clone().requires_grad_(True)
        p = ode.odeint_adjoint(y, t)
        p.abs().mean().backward()
        torch.cuda.synchronize()
        assert ode._nsteps == 12 and ode._traj.on_disk == (ttype == "basic")
        if ttype == "basic":
            s = ode._traj.stats()
            assert s["files"] >= 8 and s["bytes_read"] >= 8 * ode._traj.vecs * 4096 * 512 * 4
        res[ttype] = (p.detach().clone(), y.grad.clone(), flat_grads(f).clone())
    a, b = res["memory"], res["basic"]
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])


@pytest.mark.parametrize("method,solution_only", [("rk4", 1), ("rk4", 0), ("dopri5", 1)])
def test_two_level_checkpointing_on_gpu_equals_the_same_budget_in_hbm_bitwise(tmp_path, method, solution_only):
    """-ts_trajectory_max_cps_ram 3 with -ts_trajectory_max_cps_disk 5 (PETSc's two-level checkpointing, README.md:91-96) on
    the HIP path at headline width (4096 x 512 fp32): three slots in HBM, five in files behind the device cache; the same
    bits as eight slots in HBM, and as eight slots on disk."""
    dev = require_gpu()
    torch.manual_seed(0)
    y0 = torch.randn(4096, 512, device=dev)
    t = torch.tensor([0.0, 0.11, 0.3]) if method == "rk4" else torch.tensor([0.0, 0.5, 1.2])
    res = {}
    for tag, opts in (("hbm", {"ts_trajectory_max_cps_ram": 8}), ("two", {"ts_trajectory_max_cps_ram": 3, "ts_trajectory_max_cps_disk": 5}),
                      ("disk", {"ts_trajectory_max_cps_disk": 8})):
        options.clear()
        base = {"ts_trajectory_solution_only": solution_only, "ts_trajectory_dirname": str(tmp_path / "ckpt")}
        if method == "rk4":
            base["ts_adapt_type"] = "none"
        for k, v in dict(base, **opts).items():
            options.set_option(k, v)
        from problems import SwitchedMLPFunc
        f = (MLPFunc(512, torch.float32, std=0.1) if method == "rk4" else SwitchedMLPFunc(512, torch.float32)).to(dev)
        ode = petsc_adjoint.ODEPetsc()
        ode.setupTS(y0, f, step_size=0.01, method=method)
        options.clear()
        y = y0.clone().requires_grad_(True)
        p = ode.odeint_adjoint(y, t)
        p.abs().mean().backward()
        torch.cuda.synchronize()
        assert ode._nsteps >= 9 and ode._traj.high_water() <= 8 and ode._traj.on_disk == (tag != "hbm")
        if tag == "two":
            st = ode._traj.stats()
            assert 0 < st["files"] <= 5 and st["bytes_written"] > 0 and len(ode._traj.chunks) == 1
        res[tag] = (p.detach().clone(), y.grad.clone(), flat_grads(f).clone(), ode._nsteps)
    for tag in ("two", "disk"):
        a, b = res["hbm"], res[tag]
        assert a[3] == b[3] and torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]), tag


@pytest.mark.parametrize("method", ["cn", "imex3", "imex_torch"])
def test_checkpoint_modes_bitwise_identical_for_implicit_and_imex_steppers_on_gpu(method):
    """-ts_trajectory_solution_only / -ts_trajectory_max_cps_ram / -ts_trajectory_type basic for the theta and
    ARKIMEX steppers on the HIP path: every mode re-solves the same stage equations, the gradients agree bit for bit."""
    from problems import DiffusionIM, ReactionEX
    dev = require_gpu()
    torch.manual_seed(7)
    y0 = torch.randn(16, 6, dtype=torch.float64, device=dev)
    t = torch.tensor([0.0, 0.1, 0.35], dtype=torch.float64)
    res = []
    for opts in [{"ts_trajectory_solution_only": 0}, {}, {"ts_trajectory_solution_only": 1}, {"ts_trajectory_max_cps_ram": 1},
                 {"ts_trajectory_max_cps_ram": 3}, {"ts_trajectory_max_cps_ram": 2, "ts_trajectory_solution_only": 0},
                 {"ts_trajectory_type": "basic", "ts_trajectory_solution_only": 0}, {"ts_trajectory_type": "basic", "ts_trajectory_solution_only": 1}]:
        options.clear()
        options.set_option("ts_adapt_type", "none")
        for k, v in opts.items():
            options.set_option(k, v)
        ode = petsc_adjoint.ODEPetsc()
        if method.startswith("imex"):
            if method == "imex_torch":
                options.set_option("snes_type", "ksponly")
            fI, fE = DiffusionIM(6).to(dev), ReactionEX(6).to(dev)
            ode.setupTS(y0, fI, step_size=0.05, method="imex", implicit_form=True, imex_form=True, func2=fE, batch_size=16,
                        linear_solver="torch" if method == "imex_torch" else "petsc", matrixfree_jacobian=method != "imex_torch")
            params = list(fI.parameters()) + list(fE.parameters())
        else:
            f = MLPFunc(6, torch.float64, std=0.3).to(dev)
            ode.setupTS(y0, f, step_size=0.05, method=method, implicit_form=True)
            params = list(f.parameters())
        y = y0.clone().requires_grad_(True)
        p = ode.odeint_adjoint(y, t.to(dev))
        p.abs().mean().backward()
        assert ode._nsteps == 7
        res.append((p.detach().clone(), y.grad.clone(), torch.cat([q.grad.reshape(-1) for q in params]).clone(), ode._traj.high_water()))
    for r in res[1:]:
        assert torch.equal(r[0], res[0][0]) and torch.equal(r[1], res[0][1]) and torch.equal(r[2], res[0][2])
    assert res[3][3] <= 1 and res[4][3] <= 3 and res[5][3] <= 2


@pytest.mark.parametrize("name", ["3", "5"])
def test_adaptive_arkimex_on_gpu_against_the_oracle_on_the_same_accepted_steps(name):
    """ARKIMEX with TSAdapt basic on the HIP path (error norm through the fused WRMS kernel): the accepted steps vary, at
    least one attempt is rejected, and states / gradients equal the oracle's when it follows the same accepted steps."""
    from oracle.arkimex_oracle import odeint_adjoint_arkimex
    from problems import DiffusionIM, ReactionEX
    dev = require_gpu()
    torch.manual_seed(0)
    y0 = torch.randn(3, 6, dtype=torch.float64)
    t = torch.tensor([0.0, 0.4, 1.0], dtype=torch.float64)
    target = torch.randn(3, 3, 6, dtype=torch.float64)
    tol = {"3": 1e-5, "5": 1e-9}[name]
    for k, v in {"ts_arkimex_type": name, "ts_rtol": tol, "ts_atol": tol, "snes_rtol": 1e-13, "snes_stol": 1e-15, "ksp_rtol": 1e-13}.items():
        options.set_option(k, v)
    fI, fE = DiffusionIM(6).to(dev), ReactionEX(6).to(dev)
    ode = petsc_adjoint.ODEPetsc()
    ode.setupTS(y0.to(dev), fI, step_size=0.5, method="imex", implicit_form=True, imex_form=True, func2=fE, batch_size=3)
    assert ode._adaptive
    y = y0.to(dev).requires_grad_(True)
    p = ode.odeint_adjoint(y, t.to(dev))
    torch.mean(torch.abs(p - target.to(dev))).backward()
    log = ode.step_log()
    assert ode.num_rejections >= 1 and len(set(round(h, 12) for _, h in log)) > 3
    fI2, fE2 = DiffusionIM(6), ReactionEX(6)
    y2 = y0.clone().requires_grad_(True)
    p2 = odeint_adjoint_arkimex(fI2, fE2, y2, t, 0.5, name, plan=(list(log), list(ode.cur_sol_steps)))
    torch.mean(torch.abs(p2 - target)).backward()
    assert rel_err(p, p2) < 1e-9 and rel_err(y.grad, y2.grad) < 1e-8
    assert rel_err(flat_grads(fI), flat_grads(fI2)) < 1e-8 and rel_err(flat_grads(fE), flat_grads(fE2)) < 1e-8


@pytest.mark.parametrize("method", ["cn", "beuler"])
def test_adaptive_theta_methods_on_gpu_against_the_oracle_on_the_same_accepted_steps(method):
    """beuler / cn with TSAdapt basic on the HIP path (three-solution truncation-error estimate through pn_lincomb + the
    fused WRMS kernel): step sizes vary, the first step keeps its size, states and gradients equal the oracle's when it
    follows the same accepted steps."""
    from oracle.theta_oracle import odeint_adjoint_theta
    dev = require_gpu()
    torch.manual_seed(0)
    y0 = torch.randn(4, 2, dtype=torch.float64)
    t = torch.tensor([0.0, 0.6, 1.5], dtype=torch.float64)
    target = torch.randn(3, 4, 2, dtype=torch.float64)
    tol = 1e-4 if method == "cn" else 1e-3
    for k, v in {"ts_adapt_type": "basic", "ts_rtol": tol, "ts_atol": tol, "snes_rtol": 1e-13, "snes_stol": 1e-15,
                 "ksp_rtol": 1e-13}.items():
        options.set_option(k, v)
    f = SpiralFunc().to(dev)
    ode = petsc_adjoint.ODEPetsc()
    ode.setupTS(y0.to(dev), f, step_size=0.02, method=method, implicit_form=True)
    assert ode._adaptive
    y = y0.to(dev).requires_grad_(True)
    p = ode.odeint_adjoint(y, t.to(dev))
    torch.mean(torch.abs(p - target.to(dev))).backward()
    log = ode.step_log()
    hs = [h for _, h in log]
    assert len(set(round(h, 12) for h in hs)) > 3 and hs[1] == pytest.approx(hs[0], rel=1e-12)
    f2 = SpiralFunc()
    y2 = y0.clone().requires_grad_(True)
    p2 = odeint_adjoint_theta(f2, y2, t, 0.02, method, plan=(list(log), list(ode.cur_sol_steps)))
    torch.mean(torch.abs(p2 - target)).backward()
    assert rel_err(p, p2) < 1e-9 and rel_err(y.grad, y2.grad) < 1e-8 and rel_err(flat_grads(f), flat_grads(f2)) < 1e-8


@pytest.mark.parametrize("method,s,fsal", [("rk4", 4, False), ("dopri5", 7, True)])
def test_reference_defaults_switch_on_gpu_restores_the_references_call_counts(method, s, fsal):
    """-pn_reference_defaults 1 on the HIP path: a func that counts its calls reads the reference's
    NFE-F / NFE-B (examples-pnode/spiral_unstable.py:326-347): s per step forward; backward s per step with
    -ts_trajectory_solution_only 0 and 2s with PETSc's solution-only default; this package's own defaults (tapes retained,
    stages kept) re-evaluate nothing.  Gradients are the same bits under every setting."""
    dev = require_gpu()
    n = 9

    def run(opts):
        options.clear()
        for k, v in dict({"ts_adapt_type": "none"}, **opts).items():
            options.set_option(k, v)
        torch.manual_seed(0)
        y0 = torch.randn(64, 2, dtype=torch.float64, device=dev)
        f = SpiralFunc().to(dev)
        ode = petsc_adjoint.ODEPetsc()
        ode.setupTS(y0, f, step_size=0.05, method=method)
        y = y0.clone().requires_grad_(True)
        f.nfe = 0
        out = ode.odeint_adjoint(y, torch.tensor([0.05 * n], dtype=torch.float64))
        nf, f.nfe = f.nfe, 0
        out.abs().mean().backward()
        options.clear()
        return nf, f.nfe, y.grad.clone(), flat_grads(f).clone()

    fwd = n * (s - 1) + 1 if fsal else n * s
    vjps = n * (s - 1) if fsal else n * s
    mine = run({})
    assert mine[0] == fwd and mine[1] == 0
    ref_so = run({"pn_reference_defaults": 1})
    assert (ref_so[0], ref_so[1]) == (fwd, vjps + n * s)
    ref_all = run({"pn_reference_defaults": 1, "ts_trajectory_solution_only": 0})
    assert (ref_all[0], ref_all[1]) == (fwd, vjps)
    for r in (ref_so, ref_all):
        assert torch.equal(r[2], mine[2]) and torch.equal(r[3], mine[3])


# ---------------------------------------------------------------- MATCHSTEP / time-span invariants on the device
@pytest.mark.parametrize("h,times,expected", [
    (0.3, [0.0, 0.7, 1.5], [0.3, 0.2, 0.2, 0.3, 0.25, 0.25]),
    (0.3, [0.0, 0.75, 1.5], [0.3, 0.225, 0.225, 0.3, 0.225, 0.225]),
    (0.3, [0.0, 0.8, 2.0], [0.3, 0.25, 0.25, 0.3, 0.3, 0.3, 0.3]),
    (0.07, [0.0, 0.31, 0.32, 1.0, 1.05, 2.0], None),
    (0.013, [0.0, 0.11, 0.4000000000000001, 0.41], None),
])
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_cut_steps_come_back_after_every_output_time_on_the_device(h, times, expected, dtype):
    """pa.py:640 (MATCHSTEP) + pa.py:812-827 (time span), fixed step: the step log of the HIP path obeys the invariants
    of tests/test_matchstep_properties.py (I1-I4), equals the independent statement of the rule and the oracle's log,
    and forward / gradients agree with the oracle over those steps."""
    import numpy as np
    from test_matchstep_properties import check_invariants, spec_sequence
    dev = require_gpu()
    torch.manual_seed(5)
    y0 = torch.randn(6, 2, dtype=torch.float64)
    t = torch.tensor(times, dtype=torch.float64)
    target = torch.randn(len(times), 6, 2, dtype=torch.float64)
    a, b = _solve_pair(SpiralFunc, y0, t, target, "rk4", {"ts_adapt_type": "none"}, step_size=h, dtype=dtype, dev=dev)
    ode = b[3]
    log = ode.step_log()
    hs = [x for _, x in log]
    if expected is not None:
        assert np.allclose(hs, expected, rtol=1e-12)
    per = ode.cur_sol_steps[1:]
    ends = [tt for tt, _ in log[1:]] + [times[-1]]
    hits, k = [], 0
    for i, cnt in enumerate(per):
        k += cnt
        hits.append((i + 1, ends[k - 1]))
    check_invariants(h, times, log, hits, per)
    flat = [x for seq in spec_sequence(h, times) for x in seq]
    assert len(flat) == len(hs) and np.allclose(flat, hs, rtol=1e-10, atol=1e-13)
    te, ho, _ = a[3].step_log()
    assert len(ho) == len(hs) and np.allclose(ho, hs, rtol=1e-13) and a[3].cur_sol_steps == ode.cur_sol_steps
    tol = 1e-11 if dtype == torch.float64 else 1e-5
    assert rel_err(b[0].cpu().double(), a[0]) < tol
    assert rel_err(b[1].cpu().double(), a[1]) < tol and rel_err(b[2].cpu().double(), a[2]) < tol


# ---------------------------------------------------------------- -pn_graph_capture auto: the default
def _auto_runs(make_func, opts, calls, dev, times=(0.3,), shape=(64, 16), method="rk4", step=0.05, after_call=None):
    """`calls` training-style calls (fresh cotangent each) with the given options; returns per-call (out, dy0, dtheta),
    the solver and the func."""
    import warnings as _w
    options.clear()
    for k, v in opts.items():
        options.set_option(k, v)
    torch.manual_seed(1)
    f = make_func().to(dev)
    ode = petsc_adjoint.ODEPetsc()
    torch.manual_seed(0)
    y0 = torch.randn(*shape, device=dev).to(next(f.parameters()).dtype)      # (the same numbers in either precision)
    ode.setupTS(y0, f, step_size=step, method=method)
    options.clear()
    res = []
    with _w.catch_warnings(record=True) as caught:
        _w.simplefilter("always")
        for it in range(calls):
            for p in f.parameters():
                p.grad = None
            y = (y0 + 0.01 * it).requires_grad_(True)
            t = torch.tensor([times[it % len(times)]])
            out = ode.odeint_adjoint(y, t)
            (out * (1.0 + 0.1 * it)).sum().backward()
            res.append((out.detach().clone(), y.grad.clone(), flat_grads(f).clone()))
            if after_call is not None:
                after_call(it, f, ode)
    return res, ode, f, [str(c.message) for c in caught]


def _same(a, b):
    return all(torch.equal(x, y) for ra, rb in zip(a, b) for x, y in zip(ra, rb))


def test_auto_graph_capture_is_the_default_and_bitwise_equal_to_eager():
    """A default-constructed ODEPetsc (no option given) on a HIP device: two eager calls, a call that runs the sweeps eagerly
    AND captures them (first replays checked bit for bit), replays from then on -- same bits as -pn_graph_capture 0 in every
    call, the same NFE counters; -pn_reference_defaults 1 keeps the eager launches of the reference
    (/root/reference/pnode/petsc_adjoint.py:829, 878: ts.solve / ts.adjointSolve launch everything every time)."""
    dev = require_gpu()
    mk = lambda: MLPFunc(16, torch.float32)
    base = {"ts_adapt_type": "none"}
    auto, ode_a, _, warns = _auto_runs(mk, base, 7, dev)
    eager, ode_e, _, _ = _auto_runs(mk, dict(base, pn_graph_capture=0), 7, dev)
    assert ode_a.graphs_captured and ode_a.graph_status == "graph(auto)", ode_a.graph_status
    assert not ode_e.graphs_captured and ode_e.graph_status.startswith("eager")
    assert _same(auto, eager)
    assert (ode_a.nfe_forward, ode_a.nfe_backward) == (ode_e.nfe_forward, ode_e.nfe_backward)
    assert not [w for w in warns if "hipGraph" in w]
    ref, ode_r, _, _ = _auto_runs(mk, dict(base, pn_reference_defaults=1), 4, dev)
    assert not ode_r.graphs_captured and _same(ref, eager[:4])
    # under torch.no_grad() (evaluation passes, ode_demo_petsc.py:283-293): forward-only graphs, same states
    y0 = torch.randn(64, 16, device=dev)
    outs = []
    with torch.no_grad():
        for it in range(5):
            outs.append(ode_a.odeint_adjoint(y0, torch.tensor([0.3])).clone())
    assert all(torch.equal(o, outs[0]) for o in outs)


def test_auto_graph_capture_with_batchnorm_in_train_mode_updates_the_statistics_once_per_call():
    """func with BatchNorm1d in TRAIN mode (the reference's ODE blocks, examples-pnode/models/sqnxt_PETSc.py:116-120): the
    running statistics and num_batches_tracked after every call -- the call that validates the capture included -- are the
    eager run's, bit for bit; switching the module to eval() afterwards is a different captured configuration."""
    import torch.nn as nn
    dev = require_gpu()

    class BNFunc(nn.Module):
        def __init__(self):
            super().__init__()
            self.l1, self.bn, self.l2 = nn.Linear(16, 16), nn.BatchNorm1d(16), nn.Linear(16, 16)

        def forward(self, t, y):
            return self.l2(torch.tanh(self.bn(self.l1(y))))

    stats = {}

    def record(tag):
        def cb(it, f, ode):
            stats.setdefault(tag, []).append((f.bn.running_mean.clone(), f.bn.running_var.clone(), int(f.bn.num_batches_tracked)))
            if it == 5:
                f.eval()
        return cb
    base = {"ts_adapt_type": "none", "ts_trajectory_solution_only": 0}
    auto, ode_a, fa, _ = _auto_runs(BNFunc, base, 10, dev, after_call=record("a"))
    eager, ode_e, fe, _ = _auto_runs(BNFunc, dict(base, pn_graph_capture=0), 10, dev, after_call=record("e"))
    assert ode_a.graph_status == "graph(auto)"
    assert _same(auto, eager)
    for (m1, v1, n1), (m2, v2, n2) in zip(stats["a"], stats["e"]):
        assert torch.equal(m1, m2) and torch.equal(v1, v2) and n1 == n2
    assert stats["a"][5][2] > stats["a"][0][2] and stats["a"][9][2] == stats["a"][5][2]      # eval(): statistics frozen
    assert len(ode_a._graphs) == 2                                                           # train-mode and eval-mode graphs


def test_auto_graph_capture_keeps_the_call_counters_of_func_counting():
    """A func with a Python-side call counter (the NFE of the reference's ODE blocks, examples-pnode/models/sqnxt_PETSc.py,
    spiral_unstable.py:326-347) stops counting under a plain replay.  auto mode learns the increment of every integer
    attribute per forward and per reverse sweep from the eager warm-up calls and applies it at every replay: the counter
    reads what it reads with eager launches, call by call, also when the user resets it.  A func that changes anything
    else on the Python side during a sweep (here: remembers the last time it saw) is left eager, silently.  The explicit
    -pn_graph_capture 1 does no bookkeeping: the count freezes -- the user asked for it."""
    import torch.nn as nn
    dev = require_gpu()

    class Counting(nn.Module):
        def __init__(self):
            super().__init__()
            self.lin = nn.Linear(16, 16)
            self.nfe = 0

        def forward(self, t, y):
            self.nfe += 1
            return torch.tanh(self.lin(y))

    counts = {}

    def record(tag):
        def cb(it, f, ode):
            counts.setdefault(tag, []).append(f.nfe)
            if it == 4:
                f.nfe = 0                                 # the reference's drivers reset it after printing
        return cb
    base = {"ts_adapt_type": "none"}                      # (solution-only default: func is re-evaluated in the reverse sweep too)
    auto, ode_a, fa, warns = _auto_runs(Counting, base, 8, dev, after_call=record("a"))
    eager, ode_e, fe, _ = _auto_runs(Counting, dict(base, pn_graph_capture=0), 8, dev, after_call=record("e"))
    assert ode_a.graphs_captured and ode_a.graph_status == "graph(auto)"
    assert counts["a"] == counts["e"] and counts["a"][3] > counts["a"][2] > 0 and _same(auto, eager) and not warns
    forced, ode_f, ff, _ = _auto_runs(Counting, dict(base, pn_graph_capture=1), 5, dev)
    per_call = counts["e"][0]
    assert ode_f.graphs_captured and ff.nfe == 3 * per_call and _same(forced, eager[:5])     # two eager calls + the capture
    assert (ode_f.nfe_forward, ode_f.nfe_backward) == (ode_a.nfe_forward * 5 // 8, ode_a.nfe_backward * 5 // 8)

    class Remembering(Counting):
        def forward(self, t, y):
            self.last_t = float(t)
            return torch.tanh(self.lin(y))
    rem, ode_r, _, warns = _auto_runs(Remembering, base, 5, dev)
    assert not ode_r.graphs_captured and "not a plain call counter" in ode_r.graph_status and not warns
    assert _same(rem, eager[:5])


def test_auto_graph_capture_with_changing_output_times_and_a_func_that_syncs():
    """Every distinct `t` is its own captured pair (a handful are kept, least recently created dropped); a func that
    synchronises with the host cannot be captured: one warning, eager launches, the same results."""
    import torch.nn as nn
    dev = require_gpu()
    mk = lambda: MLPFunc(16, torch.float32)
    base = {"ts_adapt_type": "none"}
    times = (0.3, 0.5, 0.3, 0.5, 0.2)                                  # three distinct times: each gets its pair of graphs
    auto, ode_a, _, _ = _auto_runs(mk, base, 25, dev, times=times)
    eager, ode_e, _, _ = _auto_runs(mk, dict(base, pn_graph_capture=0), 25, dev, times=times)
    assert _same(auto, eager) and ode_a.graphs_captured and len(ode_a._graphs) == 3
    times = (0.3, 0.5, 0.3, 0.5, 0.2, 0.7, 0.9, 1.1)                  # six: more than the cache keeps -- entries are dropped
    auto, ode_a, _, _ = _auto_runs(mk, base, 32, dev, times=times)     # before they are warm, the calls stay eager; same bits
    eager, ode_e, _, _ = _auto_runs(mk, dict(base, pn_graph_capture=0), 32, dev, times=times)
    assert _same(auto, eager) and len(ode_a._graphs) <= petsc_adjoint.ODEPetsc.GRAPH_CACHE_ENTRIES

    class HostSync(nn.Module):
        def __init__(self):
            super().__init__()
            self.lin = nn.Linear(16, 16)

        def forward(self, t, y):
            return torch.tanh(self.lin(y)) * float(y.abs().max().item() > -1.0)
    s_auto, ode_s, _, warns = _auto_runs(HostSync, base, 6, dev)
    s_eager, _, _, _ = _auto_runs(HostSync, dict(base, pn_graph_capture=0), 6, dev)
    assert _same(s_auto, s_eager) and not ode_s.graphs_captured and "capturing the forward sweep failed" in ode_s.graph_status
    assert sum("launched eagerly instead" in w for w in warns) == 1


def test_auto_graph_capture_rejects_a_capture_whose_first_replay_differs():
    """The first-replay check: a func that draws from Python's random module gives the eager sweep and the captured sweep of
    the validating call different numbers -- auto mode must notice, warn once, and stay eager (every call's result then
    being what eager launches compute)."""
    import random
    import torch.nn as nn
    dev = require_gpu()

    class Noisy(nn.Module):
        def __init__(self):
            super().__init__()
            self.lin = nn.Linear(16, 16)

        def forward(self, t, y):
            return torch.tanh(self.lin(y)) * (1.0 + 0.2 * random.random())
    random.seed(5)
    res, ode, _, warns = _auto_runs(Noisy, {"ts_adapt_type": "none"}, 6, dev)
    assert not ode.graphs_captured and "does not reproduce the eager sweep" in ode.graph_status
    assert sum("launched eagerly instead" in w for w in warns) == 1
    assert all(torch.isfinite(r[2]).all() for r in res)


def test_auto_graph_capture_gives_up_on_a_call_signature_that_is_never_differentiated():
    """Solving with gradients enabled and never calling backward (an evaluation pass without torch.no_grad()): the call that
    would validate the captured reverse sweep never comes.  After two forward sweeps that paid for an eager twin, that call
    signature stays with eager launches; the states are the eager ones throughout, and a backward that does come later works."""
    dev = require_gpu()
    options.clear()
    options.set_option("ts_adapt_type", "none")
    f = MLPFunc(16, torch.float32).to(dev)
    torch.manual_seed(0)
    y0 = torch.randn(64, 16, device=dev)
    t = torch.tensor([0.3])
    ode = petsc_adjoint.ODEPetsc()
    ode.setupTS(y0, f, step_size=0.05, method="rk4")
    options.set_option("pn_graph_capture", 0)
    ref = petsc_adjoint.ODEPetsc()
    ref.setupTS(y0, f, step_size=0.05, method="rk4")
    options.clear()
    yr = y0.clone().requires_grad_(True)
    out_r = ref.odeint_adjoint(yr, t)
    out_r.sum().backward()
    gref, pref = yr.grad.clone(), flat_grads(f).clone()
    for it in range(8):
        out = ode.odeint_adjoint(y0.clone().requires_grad_(True), t)
        assert torch.equal(out, out_r)
    e = next(iter(ode._graphs.values()))
    assert e.eager_only and not ode.graphs_captured
    for p in f.parameters():
        p.grad = None
    y = y0.clone().requires_grad_(True)
    ode.odeint_adjoint(y, t).sum().backward()
    assert torch.equal(y.grad, gref) and torch.equal(flat_grads(f), pref)


def test_auto_graph_capture_ignores_the_padding_of_the_state_buffers():
    """Found by tools/fuzz_modes.py: state vectors are padded to a multiple of 64 elements and the padding is never written;
    with not-a-number bits in it the first-replay check of the reverse sweep compared NaN with NaN and kept the solver eager.
    The check looks at the n elements of the state only."""
    dev = require_gpu()
    options.clear()
    options.set_option("ts_adapt_type", "none")
    f = MLPFunc(257, torch.float32).to(dev)
    torch.manual_seed(0)
    y0 = torch.randn(7, 257, device=dev)                  # 1799 elements: 57 of padding
    ode = petsc_adjoint.ODEPetsc()
    ode.setupTS(y0, f, step_size=0.05, method="rk4")
    options.clear()
    for it in range(4):
        for p in f.parameters():
            p.grad = None
        y = y0.clone().requires_grad_(True)
        ode.odeint_adjoint(y, torch.tensor([0.3])).sum().backward()
        if it == 0:
            ode.adj_u_tensor[ode.n:].fill_(float("nan"))   # what uninitialised memory may hold
    assert ode.graphs_captured and ode.graph_status == "graph(auto)", ode.graph_status


def test_auto_graph_capture_with_a_step_size_list_and_several_output_times():
    """pa.py:523-525 (a list gives the size of every step) with four output times: the default launch mode captures it
    (the list is part of the capture key) and stays bitwise equal to eager launches; a different list is a different capture."""
    dev = require_gpu()
    mk = lambda: MLPFunc(16, torch.float32)
    torch.manual_seed(0)
    y0 = torch.randn(32, 16, device=dev)
    t = torch.tensor([0.0, 0.1, 0.25, 0.45])
    res = {}
    for tag, opts in (("auto", {}), ("eager", {"pn_graph_capture": 0})):
        options.clear()
        for k, v in dict({"ts_adapt_type": "none"}, **opts).items():
            options.set_option(k, v)
        f = mk().to(dev)
        ode = petsc_adjoint.ODEPetsc()
        outs = []
        for steps in ([0.1, 0.15, 0.2], [0.1, 0.15, 0.2], [0.1, 0.15, 0.2], [0.1, 0.15, 0.2], [0.1, 0.15, 0.2], [0.05, 0.05, 0.15, 0.2]):
            ode.setupTS(y0, f, step_size=list(steps), method="rk4")
            for p in f.parameters():
                p.grad = None
            y = y0.clone().requires_grad_(True)
            out = ode.odeint_adjoint(y, t)
            (out * torch.arange(1.0, 5.0, device=dev).view(4, 1, 1)).sum().backward()
            outs.append((out.detach().clone(), y.grad.clone(), flat_grads(f).clone(), ode._nsteps))
        options.clear()
        res[tag] = (outs, ode)
    assert res["auto"][1].graphs_captured and len(res["auto"][1]._graphs) == 2
    for a, b in zip(res["auto"][0], res["eager"][0]):
        assert a[3] == b[3] and torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    assert res["auto"][0][0][3] == 3 and res["auto"][0][-1][3] == 4


# ---------------------------------------------------------------- round 5: engine-side accumulation of the Linear layers' sensitivities
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_linear_layer_sensitivities_accumulated_by_the_engine_equal_autograd_at_headline_width(dtype):
    """-pn_linear_param_grads (pnode_amd/_lineargrad.py; row a-9: RHSJacPShell.multTranspose, pa.py:341-363): dW by an accumulating
    GEMM into mu, db by pn_colsum_accum (`gemm`), or both by the fused MFMA kernel pn_linear_wgrad (the default where the shape
    allows), during the stage VJP's backward pass.  At config 3's width (batch 512 of the 4096
    rows): equal to the autograd path to fp round-off in eager launches and under the default launch mode (captured sweeps:
    hooks run at capture, kernels replay), in store-all + tapes, solution-only and budget modes -- and all engine-side runs
    agree with each other bit for bit."""
    dev = require_gpu()
    mk = lambda: MLPFunc(512, dtype)
    tol = 5e-6 if dtype == torch.float32 else 1e-12
    runs = {}
    for tag, opts in (("autograd", {"pn_linear_param_grads": 0, "pn_graph_capture": 0}),
                      ("engine-eager", {"pn_graph_capture": 0}),
                      ("engine-default", {}),
                      ("engine-gemm", {"pn_linear_param_grads": "gemm"}),
                      ("engine-solution-only", {"ts_trajectory_solution_only": 1}),
                      ("engine-budget", {"ts_trajectory_max_cps_ram": 3}),
                      # -pn_linear_side_stream 1: the grouped product of a stage on a second stream beside the next stage's
                      # backward pass (two cotangent buffers in turn, operands kept referenced until the main stream has waited)
                      ("engine-side-stream", {"pn_linear_side_stream": 1}),
                      ("engine-side-stream-eager", {"pn_linear_side_stream": 1, "pn_graph_capture": 0}),
                      ("engine-side-stream-python-loop", {"pn_linear_side_stream": "same-priority", "pn_graph_capture": 0, "pn_step_loop": "python"}),
                      ("engine-side-stream-budget", {"pn_linear_side_stream": 1, "ts_trajectory_max_cps_ram": 3}),
                      ("engine-side-stream-solution-only", {"pn_linear_side_stream": 1, "ts_trajectory_solution_only": 1, "pn_graph_capture": 0})):
        res, ode, f, warns = _auto_runs(mk, dict({"ts_adapt_type": "none"}, **opts), 5, dev, shape=(512, 512), step=0.05)
        runs[tag] = res
        assert ode.linear_param_grads.startswith("autograd" if tag == "autograd" else "engine (8 of 8"), ode.linear_param_grads
        if "side-stream" in tag:
            assert ode._lin.side_on and ode._lin.side is not None and not ode._lin.inflight
        if tag in ("engine-default", "engine-side-stream"):
            assert ode.graph_status == "graph(auto)", ode.graph_status
        if tag.startswith("engine"):      # the fused MFMA kernel (csrc/pn_linear.hip; fp32 and, since round 6, fp64) unless asked otherwise
            assert ("fused dW + db MFMA kernel on 4 layers" in ode.linear_param_grads) == (tag != "engine-gemm"), ode.linear_param_grads
        assert not [w for w in warns if "Linear" in w]
    for tag in runs:
        for a, b in zip(runs[tag], runs["autograd"]):
            assert torch.equal(a[0], b[0]) and rel_err(a[1], b[1]) < tol and rel_err(a[2], b[2]) < tol, tag
    for tag in runs:
        if tag.startswith("engine") and tag != "engine-gemm":
            assert _same(runs[tag], runs["engine-eager"]), tag


class _TimeGatedMLP(MLPFunc):
    """the weight of the third Linear layer is ALSO used functionally, at some stage times only
    (`early`: for t < 0.15, else for t >= 0.15).  The first VJP of a reverse sweep is at the latest time."""

    def __init__(self, d, dtype, early=True):
        super().__init__(d, dtype)
        self.early = early

    def forward(self, t, y):
        net = self.net
        h = net[3](net[2](net[1](net[0](y))))
        z = net[5](net[4](h))
        if (t < 0.15) == self.early:
            z = z + 0.5 * torch.nn.functional.linear(h, net[4].weight)
        return net[6](z)


@pytest.mark.parametrize("early", [True, False])
def test_a_time_gated_second_use_of_a_linear_weight_is_right_with_the_fused_kernel_active(early):
    """Row a-9's guarantee (pnode_amd/_lineargrad.py): every recorded evaluation of func is checked structurally; the
    evaluations in which net[4].weight is used a second time are differentiated by autograd with respect to every parameter
    (as the reference differentiates every evaluation, pa.py:66-74), the others by the hooks with the fused MFMA kernel on
    all four layers.  dL/dtheta equals the autograd path (-pn_linear_param_grads 0) to fp32 round-off in eager launches,
    under the default launch mode (captured sweeps: the check runs at capture), with re-validation, in solution-only and
    budget modes -- and all engine-side runs agree bit for bit."""
    dev = require_gpu()
    mk = lambda: _TimeGatedMLP(512, torch.float32, early)
    runs = {}
    for tag, opts in (("autograd", {"pn_linear_param_grads": 0, "pn_graph_capture": 0}),
                      ("engine-eager", {"pn_graph_capture": 0}),
                      ("engine-default", {}),
                      ("engine-revalidate", {"pn_graph_revalidate": 2}),
                      ("engine-gemm", {"pn_linear_param_grads": "gemm", "pn_graph_capture": 0}),
                      ("engine-solution-only", {"ts_trajectory_solution_only": 1}),
                      ("engine-budget", {"ts_trajectory_max_cps_ram": 3}),
                      ("engine-side-stream", {"pn_linear_side_stream": 1})):
        res, ode, f, warns = _auto_runs(mk, dict({"ts_adapt_type": "none"}, **opts), 6, dev, shape=(512, 512), step=0.05)
        runs[tag] = res
        assert not [w for w in warns if "Linear" in w or "differs" in w], warns
        if tag == "autograd":
            assert ode.linear_param_grads.startswith("autograd")
            continue
        assert ode.linear_param_grads.startswith("engine (8 of 8"), ode.linear_param_grads
        assert ode._lin.n_clean > 0 and ode._lin.n_autograd > 0, ode.linear_param_grads
        if tag != "engine-gemm":
            assert "fused dW + db MFMA kernel on 4 layers" in ode.linear_param_grads, ode.linear_param_grads
        if tag in ("engine-default", "engine-revalidate"):
            assert ode.graph_status == "graph(auto)", ode.graph_status
    for tag in runs:
        for a, b in zip(runs[tag], runs["autograd"]):
            assert torch.equal(a[0], b[0]) and rel_err(a[1], b[1]) < 5e-6 and rel_err(a[2], b[2]) < 5e-6, (tag, rel_err(a[2], b[2]))
    for tag in ("engine-default", "engine-revalidate", "engine-solution-only", "engine-budget", "engine-side-stream"):
        assert _same(runs[tag], runs["engine-eager"]), tag


class _FailingMLP(MLPFunc):
    """Raises in its `fail_at`-th evaluation (counted from the moment `fail_at` is set)."""

    def __init__(self, d, dtype):
        super().__init__(d, dtype)
        self.fail_at = None

    def forward(self, t, y):
        if self.fail_at is not None:
            self.fail_at -= 1
            if self.fail_at == 0:
                self.fail_at = None
                raise RuntimeError("func failed")
        return super().forward(t, y)


def test_fused_linear_sensitivities_leave_nothing_behind_when_a_reverse_sweep_raises():
    """The fused dW + db kernel keeps its sums in per-layer partial buffers until the reverse sweep ends
    (pnode_amd/_lineargrad.py).  A sweep that dies half way (func raises in a recomputed stage) must not leak its partial sums
    into the next backward pass: the next call equals a fresh solver's, bit for bit."""
    dev = require_gpu()
    opts = {"ts_adapt_type": "none", "pn_graph_capture": 0, "pn_trajectory_retain_graph": 0}
    options.clear()
    for k, v in opts.items():
        options.set_option(k, v)
    torch.manual_seed(1)
    f = _FailingMLP(64, torch.float32).to(dev)
    ode = petsc_adjoint.ODEPetsc()
    torch.manual_seed(0)
    y0 = torch.randn(256, 64, device=dev)
    ode.setupTS(y0, f, step_size=0.05, method="rk4")
    options.clear()

    def call(it):
        for p in f.parameters():
            p.grad = None
        y = (y0 + 0.01 * it).requires_grad_(True)
        out = ode.odeint_adjoint(y, torch.tensor([0.3]))
        (out * (1.0 + 0.1 * it)).sum().backward()
        return out.detach().clone(), y.grad.clone(), flat_grads(f).clone()
    good = [call(0), call(1)]
    assert "fused dW + db MFMA kernel on 4 layers" in ode.linear_param_grads
    f.fail_at = 6 * 4 + 7                                  # 6 steps x 4 stages forward, then the 7th recomputed stage of the reverse sweep
    with pytest.raises(RuntimeError, match="func failed"):
        call(2)
    assert any(st[2] for st in ode._lin.partials.values())             # the dead sweep did leave sums in the partial buffers
    after_failure = call(3)
    ref, _, _, _ = _auto_runs(lambda: MLPFunc(64, torch.float32), opts, 4, dev, shape=(256, 64), step=0.05)
    assert _same(good, ref[:2])
    assert all(torch.equal(a, b) for a, b in zip(after_failure, ref[3]))


def test_explicitly_captured_sweeps_replay_the_fused_linear_kernel():
    """-pn_graph_capture 1: the partial buffers of the fused dW + db kernel are allocated before the capture (never inside one: a
    layer met for the first time while capturing would take the library path for that capture), the kernel and its end-of-sweep
    pass are part of the captured reverse sweep.  Replays equal the eager launches bit for bit and the autograd path to round-off."""
    dev = require_gpu()
    mk = lambda: MLPFunc(64, torch.float32)
    cap, ode, _, _ = _auto_runs(mk, {"ts_adapt_type": "none", "pn_graph_capture": 1}, 4, dev, shape=(256, 64), step=0.05)
    eag, _, _, _ = _auto_runs(mk, {"ts_adapt_type": "none", "pn_graph_capture": 0}, 4, dev, shape=(256, 64), step=0.05)
    ref, _, _, _ = _auto_runs(mk, {"ts_adapt_type": "none", "pn_graph_capture": 0, "pn_linear_param_grads": 0}, 4, dev, shape=(256, 64), step=0.05)
    assert ode.graphs_captured and "fused dW + db MFMA kernel on 4 layers" in ode.linear_param_grads
    assert not any(st[2] for st in ode._lin.partials.values())         # every sweep ended with the finishing pass
    assert _same(cap, eag)
    for a, b in zip(cap, ref):
        assert torch.equal(a[0], b[0]) and rel_err(a[1], b[1]) < 5e-6 and rel_err(a[2], b[2]) < 5e-6


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_fused_linear_sensitivities_take_any_batch_size_from_256_rows(dtype):
    """A batch that is not a multiple of 256 rows (1000 here): the fused kernel's K ranges are rounded up to whole 32-row slabs and
    the rows that do not exist are read as zeros (its `_ragged` instantiations) -- still the fused path, equal to autograd to
    round-off, eager and captured sweeps the same bits; 200 rows (< 256) take the library path."""
    dev = require_gpu()
    mk = lambda: MLPFunc(128, dtype)
    tol = 5e-6 if dtype == torch.float32 else 1e-12
    base = {"ts_adapt_type": "none"}
    ref, ode_r, _, _ = _auto_runs(mk, dict(base, pn_graph_capture=0, pn_linear_param_grads=0), 4, dev, shape=(1000, 128), step=0.05)
    eag, ode, _, _ = _auto_runs(mk, dict(base, pn_graph_capture=0), 4, dev, shape=(1000, 128), step=0.05)
    gra, ode_g, _, _ = _auto_runs(mk, base, 4, dev, shape=(1000, 128), step=0.05)
    assert "fused dW + db MFMA kernel on 4 layers" in ode.linear_param_grads and ode_g.graph_status == "graph(auto)"
    for a, b in zip(eag, ref):
        assert torch.equal(a[0], b[0]) and rel_err(a[1], b[1]) < tol and rel_err(a[2], b[2]) < tol
    assert _same(gra, eag)
    small, ode_s, _, _ = _auto_runs(mk, dict(base, pn_graph_capture=0), 2, dev, shape=(200, 128), step=0.05)
    assert ode_s.linear_param_grads.startswith("engine (8 of 8") and "fused" not in ode_s.linear_param_grads


@pytest.mark.parametrize("rows", [1024, 1000])
def test_the_128_tile_form_of_the_fused_kernel_in_a_solve_equals_the_64_tile_form_bitwise(rows):
    """BASELINE's layer shapes (four nn.Linear(512, 512): the stage VJP's grouped launch fills two rounds of the chip with 128 x 128
    tiles and takes that form, csrc/pn_linear.hip wgrad_body_x3_wide) against -pn_linear_wgrad_tile64 1: the same gradients bit
    for bit, eager and captured, aligned and ragged batch sizes; and both equal to autograd to round-off."""
    dev = require_gpu()
    mk = lambda: MLPFunc(512, torch.float32)
    base = {"ts_adapt_type": "none"}
    ref, _, _, _ = _auto_runs(mk, dict(base, pn_graph_capture=0, pn_linear_param_grads=0), 2, dev, shape=(rows, 512), step=0.05)
    t64, ode64, _, _ = _auto_runs(mk, dict(base, pn_graph_capture=0, pn_linear_wgrad_tile64=1), 2, dev, shape=(rows, 512), step=0.05)
    eag, ode, _, _ = _auto_runs(mk, dict(base, pn_graph_capture=0), 2, dev, shape=(rows, 512), step=0.05)
    gra, ode_g, _, _ = _auto_runs(mk, base, 4, dev, shape=(rows, 512), step=0.05)
    assert "fused dW + db MFMA kernel on 4 layers" in ode.linear_param_grads and "fused dW + db MFMA kernel on 4 layers" in ode64.linear_param_grads
    assert ode._ops.wgrad_flags == 0 and ode64._ops.wgrad_flags == 2
    assert ode_g.graph_status == "graph(auto)" or "not faster" in ode_g.graph_status, ode_g.graph_status
    assert _same(eag, t64)
    assert _same(gra[:2], eag)
    for a, b in zip(eag, ref):
        assert torch.equal(a[0], b[0]) and rel_err(a[1], b[1]) < 5e-6 and rel_err(a[2], b[2]) < 5e-6


class _CastBack(nn.Module):
    """func's output back in the state's precision: what a func run under autocast has to do (the engine, like the reference's
    PETSc vectors, takes the state's dtype only)."""

    def __init__(self, net):
        super().__init__()
        self.net = net

    def forward(self, t, y):
        return self.net(t, y).float()


def test_a_func_run_under_autocast_gets_autograds_parameter_gradients_without_a_warning():
    """Under torch.autocast the Linear layers compute in bf16: autograd forms dW from the bf16 copies, the hooks would form it from
    the fp32 input -- more accurate, but not the derivative of what func computed (the self-check saw 2.8e-3 and switched the
    engine path off with a warning about a weight used twice).  Such evaluations are left to autograd per evaluation, quietly:
    the gradients of -pn_linear_param_grads 0, bit for bit, eager and captured."""
    dev = require_gpu()
    base = {"ts_adapt_type": "none"}

    def run(opts, calls):
        import warnings as _w
        options.clear()
        for k, v in opts.items():
            options.set_option(k, v)
        torch.manual_seed(0)
        f = _CastBack(MLPFunc(128, torch.float32)).to(dev)
        y0 = torch.randn(512, 128, device=dev)
        ode = petsc_adjoint.ODEPetsc()
        ode.setupTS(y0, f, step_size=0.05, method="rk4")
        options.clear()
        res = []
        with _w.catch_warnings(record=True) as caught:
            _w.simplefilter("always")
            for it in range(calls):
                for p in f.parameters():
                    p.grad = None
                y = (y0 + 0.01 * it).requires_grad_(True)
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    out = ode.odeint_adjoint(y, torch.tensor([0.3]))
                out.float().abs().mean().backward()
                res.append((out.detach().clone(), y.grad.clone(), flat_grads(f).clone()))
        return res, ode, [str(c.message) for c in caught if "pnode_amd" in str(c.message)]

    ref, _, _ = run(dict(base, pn_graph_capture=0, pn_linear_param_grads=0), 4)
    eag, ode, msgs = run(dict(base, pn_graph_capture=0), 4)
    gra, ode_g, msgs_g = run(base, 4)
    assert not msgs and not msgs_g, (msgs, msgs_g)
    assert ode.linear_param_grads.startswith("engine (8 of 8") and "left to autograd" in ode.linear_param_grads
    assert _same(eag, ref) and _same(gra, ref)
    assert ode_g.graph_status.startswith("graph(auto") or "not faster" in ode_g.graph_status, ode_g.graph_status


class _GainFirst(nn.Module):
    """A scalar parameter in front of the Linear layers: every later slice of mu starts 4 bytes off a 16-byte boundary."""

    def __init__(self, d, dtype):
        super().__init__()
        self.gain = nn.Parameter(torch.tensor(0.7, dtype=dtype))
        self.mlp = MLPFunc(d, dtype)

    def forward(self, t, y):
        return self.gain * self.mlp(t, y)


def test_fused_linear_sensitivities_with_unaligned_slices_of_mu():
    """pn_linear_wgrad_finish needs 16-byte aligned slices of mu; when a parameter in front of the layers shifts them the partial
    sums are added through torch, in the same order -- same gradients as the autograd path to round-off, graph and eager equal."""
    dev = require_gpu()
    mk = lambda: _GainFirst(64, torch.float32)
    base = {"ts_adapt_type": "none"}
    ref, _, _, _ = _auto_runs(mk, dict(base, pn_graph_capture=0, pn_linear_param_grads=0), 4, dev, shape=(256, 64), step=0.05)
    eag, ode, f, _ = _auto_runs(mk, dict(base, pn_graph_capture=0), 4, dev, shape=(256, 64), step=0.05)
    gra, ode_g, _, _ = _auto_runs(mk, base, 4, dev, shape=(256, 64), step=0.05)
    assert "engine (8 of 9 parameter tensors; fused dW + db MFMA kernel on 4 layers" in ode.linear_param_grads
    assert min(v[0] for v in ode._lin.slots.values()) == 1 and ode_g.graph_status == "graph(auto)"      # mu_W slices start at float 1, 4097, ...
    for a, b in zip(eag, ref):
        assert torch.equal(a[0], b[0]) and rel_err(a[1], b[1]) < 5e-6 and rel_err(a[2], b[2]) < 5e-6
    assert _same(gra, eag)
