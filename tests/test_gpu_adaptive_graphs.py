"""Adaptive solves with func's evaluations replayed from per-evaluation hipGraphs (pnode_amd/_stagegraphs.py).

The oracle of this file is the solver itself with ``-pn_graph_capture 0``: the replayed evaluations launch the kernels the eager
ones launch, on the same operands, so every call of a replaying solver must give the eager solver's bits -- whatever launch mode
the solver settles on (``auto`` keeps eager launches when replaying does not pay, and says so in ``graph_status``)."""
import math
import warnings

import pytest
import torch
import torch.nn as nn

from conftest import require_gpu
from pnode_amd import options, petsc_adjoint
from problems import MLPFunc, SwitchedMLPFunc

pytestmark = pytest.mark.gpu


def _solver(func, y0, method, opts, step=0.01):
    options.clear()
    for k, v in opts.items():
        options.set_option(k, v)
    ode = petsc_adjoint.ODEPetsc()
    ode.setupTS(y0, func, step_size=step, method=method, enable_adjoint=True)
    options.clear()
    return ode


def _solve(ode, func, y0, t, w=None):
    for p in func.parameters():
        p.grad = None
    y = y0.detach().clone().requires_grad_(True)
    out = ode.odeint_adjoint(y, t)
    (out.abs().mean() if w is None else (out * w).sum()).backward()
    return out.detach().clone(), y.grad.clone(), torch.cat([p.grad.reshape(-1) for p in func.parameters()])


def _same(a, b):
    return all(torch.equal(x, y) for x, y in zip(a, b))


def _settled(ode):
    s = ode.graph_status
    return s.startswith("graph(") or "not faster" in s


class TensorTimeFunc(nn.Module):
    """t enters in tensor arithmetic only: capturable whether t is a float or a 0-dim device tensor."""

    def __init__(self, d, dtype):
        super().__init__()
        self.a = nn.Linear(d, d).to(dtype)
        self.b = nn.Linear(d, d).to(dtype)
        self.calls = 0

    def forward(self, t, y):
        self.calls += 1
        return self.b(torch.tanh(self.a(y))) * (1.0 + 0.5 * t) - 0.1 * y * t


class HostTimeFunc(TensorTimeFunc):
    """Branches on t on the host: cannot be captured with a device-resident time."""

    def forward(self, t, y):
        z = self.b(torch.tanh(self.a(y)))
        return z * (2.0 if float(t) < 0.3 else 0.5)


@pytest.mark.parametrize("opts", [{"ts_trajectory_max_cps_ram": 6}, {}, {"ts_trajectory_solution_only": 0}, {"ts_trajectory_solution_only": 1},
                                  {"pn_param_accum": "stage"}, {"pn_linear_param_grads": "0"}],
                         ids=["budget6", "default", "store-all", "solution-only", "accum-stage", "autograd-param-grads"])
def test_replayed_evaluations_give_the_eager_bits_in_every_trajectory_mode(opts):
    dev = require_gpu()
    torch.manual_seed(1)
    f = SwitchedMLPFunc(256, torch.float32).to(dev)
    y0 = torch.randn(512, 256, device=dev) * 0.5
    t = torch.tensor([0.6])
    eager = _solver(f, y0, "dopri5", dict(opts, pn_graph_capture="0"))
    auto = _solver(f, y0, "dopri5", dict(opts))
    ref = _solve(eager, f, y0, t)
    assert eager.num_steps > 10 and eager.num_rejections > 0
    with warnings.catch_warnings():
        warnings.filterwarnings("error", message="pnode_amd")     # no veto warning: only "not faster" (silent) may keep it eager
        for k in range(6):
            got = _solve(auto, f, y0, t)
            assert _same(got, ref), (k, auto.graph_status)
            assert auto.num_steps == eager.num_steps and auto.num_rejections == eager.num_rejections
    assert _settled(auto), auto.graph_status
    # evaluations are counted as the eager solver counts them
    if "ts_trajectory_solution_only" not in opts or opts["ts_trajectory_solution_only"]:
        # (store-all: the eager solver keeps the forward sweep's tapes, a replaying one re-evaluates func in its stage VJPs)
        n_e = (eager.nfe_forward, eager.nfe_backward)
        assert (auto.nfe_forward, auto.nfe_backward) == (6 * n_e[0], 6 * n_e[1])


def test_explicit_mode_replays_without_validation_and_training_updates_are_seen():
    """-pn_graph_capture 1: units from the third call on; the parameters are updated in place between the calls (an optimizer
    step), the replayed kernels read the new values."""
    dev = require_gpu()
    torch.manual_seed(2)
    fe = SwitchedMLPFunc(128, torch.float32).to(dev)
    fg = SwitchedMLPFunc(128, torch.float32).to(dev)
    fg.load_state_dict(fe.state_dict())
    y0 = torch.randn(256, 128, device=dev) * 0.5
    t = torch.tensor([0.5])
    eager = _solver(fe, y0, "dopri5", {"pn_graph_capture": "0", "ts_trajectory_max_cps_ram": 5})
    graph = _solver(fg, y0, "dopri5", {"pn_graph_capture": "1", "ts_trajectory_max_cps_ram": 5})
    for k in range(6):
        a = _solve(eager, fe, y0, t)
        b = _solve(graph, fg, y0, t)
        assert _same(a, b), (k, graph.graph_status)
        with torch.no_grad():
            for p, q in zip(fe.parameters(), fg.parameters()):
                p.add_(p.grad, alpha=-0.05)
                q.add_(q.grad, alpha=-0.05)
    assert graph.graph_status.startswith("graph(per-evaluation"), graph.graph_status
    e = next(iter(graph._graphs.values()))
    assert e.sg.replayed > 0 and {k[0] for k in e.sg.units} >= {"F", "A", "B", "AB"}


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_time_as_a_device_scalar_several_output_times_and_call_counters(dtype):
    dev = require_gpu()
    torch.manual_seed(3)
    fe, fg = TensorTimeFunc(64, dtype).to(dev), TensorTimeFunc(64, dtype).to(dev)
    fg.load_state_dict(fe.state_dict())
    y0 = torch.randn(300, 64, device=dev, dtype=dtype)
    t = torch.tensor([0.0, 0.4, 0.7, 1.0])
    w = torch.randn(4, 300, 64, device=dev, dtype=dtype)
    eager = _solver(fe, y0, "dopri5", {"pn_graph_capture": "0", "ts_rtol": 1e-6, "ts_atol": 1e-6}, step=0.05)
    auto = _solver(fg, y0, "dopri5", {"ts_rtol": 1e-6, "ts_atol": 1e-6}, step=0.05)
    for k in range(6):
        a = _solve(eager, fe, y0, t, w)
        b = _solve(auto, fg, y0, t, w)
        assert _same(a, b), (k, auto.graph_status)
        assert fe.calls == fg.calls, (k, fe.calls, fg.calls)       # func's own counter keeps counting under replay
    assert _settled(auto), auto.graph_status


def test_small_layers_are_left_to_autograd_inside_the_replayed_evaluations():
    """The reference's spiral dynamics (Linear(2, 50) - Tanh - Linear(50, 2) on y^3: layers outside the fused kernel's shapes).  In an
    eager sweep the engine forms their sensitivities with the library GEMM, whose scale is a host scalar; inside a per-evaluation
    graph autograd differentiates such an evaluation with respect to every parameter and the scale is applied per replay
    (pn_param_accum): the same numbers to round-off -- `auto` validates that and says so in graph_status -- every call."""
    from problems import SpiralFunc
    dev = require_gpu()
    torch.manual_seed(6)
    fe, fg = SpiralFunc(torch.float32).to(dev), SpiralFunc(torch.float32).to(dev)
    fg.load_state_dict(fe.state_dict())
    y0 = torch.randn(1024, 2, device=dev)
    t = torch.tensor([0.0, 1.0, 2.5])
    w = torch.randn(3, 1024, 2, device=dev)
    eager = _solver(fe, y0, "dopri5", {"pn_graph_capture": "0"}, step=0.025)
    auto = _solver(fg, y0, "dopri5", {}, step=0.025)
    assert eager.linear_param_grads.startswith("engine (4 of 4")
    for k in range(6):
        a = _solve(eager, fe, y0, t, w)
        b = _solve(auto, fg, y0, t, w)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]), (k, auto.graph_status)          # the state and dL/dy0: the same kernels
        assert float((a[2] - b[2]).norm() / a[2].norm()) < 2e-6, (k, auto.graph_status)
    s = auto.graph_status
    assert (s.startswith("graph(auto; per-evaluation") and "replays within" in s) or "not faster" in s, s


class BatchNormFunc(nn.Module):
    """Train-mode BatchNorm: every evaluation updates the running statistics (buffers written in place by the kernels)."""

    def __init__(self, d, dtype):
        super().__init__()
        self.a = nn.Linear(d, d).to(dtype)
        self.bn = nn.BatchNorm1d(d).to(dtype)
        self.b = nn.Linear(d, d).to(dtype)

    def forward(self, t, y):
        return self.b(torch.tanh(self.bn(self.a(y)))) * 0.5


def test_buffers_written_by_func_end_every_call_as_the_eager_solver_leaves_them():
    """BatchNorm's running statistics after each call -- the validating calls run every sweep twice and put the buffers back in
    between -- and the gradients: the eager solver's."""
    dev = require_gpu()
    torch.manual_seed(5)
    fe, fg = BatchNormFunc(64, torch.float32).to(dev), BatchNormFunc(64, torch.float32).to(dev)
    fg.load_state_dict(fe.state_dict())
    y0 = torch.randn(256, 64, device=dev)
    t = torch.tensor([0.5])
    eager = _solver(fe, y0, "dopri5", {"pn_graph_capture": "0"}, step=0.05)
    auto = _solver(fg, y0, "dopri5", {}, step=0.05)
    for k in range(6):
        a = _solve(eager, fe, y0, t)
        b = _solve(auto, fg, y0, t)
        assert _same(a, b), (k, auto.graph_status)
        for (n1, b1), (n2, b2) in zip(fe.named_buffers(), fg.named_buffers()):
            assert torch.equal(b1, b2), (k, n1, auto.graph_status)
    assert _settled(auto), auto.graph_status


def test_a_func_that_needs_the_time_on_the_host_stays_eager_with_the_same_results():
    dev = require_gpu()
    torch.manual_seed(4)
    f = HostTimeFunc(64, torch.float32).to(dev)
    y0 = torch.randn(128, 64, device=dev)
    t = torch.tensor([0.6])
    eager = _solver(f, y0, "dopri5", {"pn_graph_capture": "0"}, step=0.05)
    auto = _solver(f, y0, "dopri5", {}, step=0.05)
    ref = _solve(eager, f, y0, t)
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        for k in range(5):
            assert _same(_solve(auto, f, y0, t), ref), k
    assert auto.graph_status.startswith("eager (auto: capturing an evaluation of func failed"), auto.graph_status
    assert any("launched eagerly" in str(x.message) for x in rec)


def test_c3b_stiff_at_full_size_replayed_reverse_sweep_equals_the_eager_one():
    """BASELINE config 3's shapes on the dynamics that adapt (test_gpu_configs.py::test_c3b_adaptive_workload_that_really_adapts_4096x512),
    shorter horizon: the replayed sweeps against the eager solver, bit for bit."""
    dev = require_gpu()
    torch.manual_seed(0)
    f = SwitchedMLPFunc(512, torch.float32).to(dev)
    y0 = torch.randn(4096, 512, device=dev)
    t = torch.tensor([1.0])
    opts = {"ts_trajectory_type": "memory", "ts_trajectory_max_cps_ram": 20}
    eager = _solver(f, y0, "dopri5", dict(opts, pn_graph_capture="0"))
    graph = _solver(f, y0, "dopri5", dict(opts, pn_graph_capture="1"))
    ref = _solve(eager, f, y0, t)
    assert eager.num_steps > 20 and eager._traj.high_water() == 20
    for k in range(4):
        assert _same(_solve(graph, f, y0, t), ref), k
    assert graph.graph_status.startswith("graph(per-evaluation")
    assert "fused dW + db MFMA kernel on 4 layers" in graph.linear_param_grads
