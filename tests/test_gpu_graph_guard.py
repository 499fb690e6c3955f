"""-m gpu: the default launch mode (``-pn_graph_capture auto``) must never replay stale Python-side state of func.

With the reference func runs at every stage of every solve (pa.py:393-412, 52-82), so whatever its callers do to the module
between two solves is seen.  Each test below does one of those things between calls and requires EVERY call to be bitwise
equal to ``-pn_graph_capture 0``:

* GRAND: ``self.odefunc.x0 = x0.clone().detach()`` before each forward
  (/root/reference/examples-sinode/grand/src/base_classes.py:58-60, read at function_laplacian_diffusion.py:59);
  ``edge_index`` / ``edge_weight`` re-assigned (grand/src/block_pnode.py:61-63);
* a float hyper-parameter annealed, a flag toggled, a buffer replaced by assignment, a host tensor changed in place;
* FFJORD: ``self._e = None`` before the solve, a fresh sample inside the first evaluation
  (/root/reference/examples-pnode/ffjord-pnode/lib/layers/odefunc.py:341-364);
* state no guard can see (a module-level global): caught by the periodic re-validation (``-pn_graph_revalidate``).
"""
import warnings

import pytest
import torch
import torch.nn as nn

from conftest import require_gpu
from pnode_amd import options, petsc_adjoint
from problems import flat_grads

pytestmark = pytest.mark.gpu


def _runs(make_func, opts, calls, dev, before_call=None, after_call=None, shape=(64, 16), seed_calls=False):
    """`calls` training-style calls; per call (states, dL/dy0, dL/dtheta); the solver, the func, the warnings."""
    options.clear()
    for k, v in opts.items():
        options.set_option(k, v)
    torch.manual_seed(1)
    f = make_func().to(dev)
    ode = petsc_adjoint.ODEPetsc()
    torch.manual_seed(0)
    y0 = torch.randn(*shape, device=dev)
    ode.setupTS(y0, f, step_size=0.05, method="rk4")
    options.clear()
    res = []
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        for it in range(calls):
            for p in f.parameters():
                p.grad = None
            y = (y0 + 0.01 * it).requires_grad_(True)
            if before_call is not None:
                before_call(it, f, y.detach())
            if seed_calls:
                torch.manual_seed(100 + it)
            out = ode.odeint_adjoint(y, torch.tensor([0.3]))
            (out * (1.0 + 0.1 * it)).sum().backward()
            res.append((out.detach().clone(), y.grad.clone(), flat_grads(f).clone()))
            if after_call is not None:
                after_call(it, f, ode)
    return res, ode, f, [str(c.message) for c in caught]


def _same(a, b):
    return len(a) == len(b) and all(torch.equal(x, y) for ra, rb in zip(a, b) for x, y in zip(ra, rb))


def _calls_equal(a, b):
    return [all(torch.equal(x, y) for x, y in zip(ra, rb)) for ra, rb in zip(a, b)]


BASE = {"ts_adapt_type": "none"}
EAGER = dict(BASE, pn_graph_capture=0)


class GrandLike(nn.Module):
    """du/dt = alpha*(A u - u) + beta*x0 with x0 handed over as an attribute before every solve."""

    def __init__(self):
        super().__init__()
        self.lin = nn.Linear(16, 16)
        self.alpha, self.beta = 0.8, 0.3
        self.x0 = None
        self.nfe = 0

    def forward(self, t, y):
        self.nfe += 1
        return self.alpha * (torch.tanh(self.lin(y)) - y) + self.beta * self.x0 * torch.cos(y)      # (x0 is saved for the VJP)


@pytest.mark.parametrize("keep_old", [False, True])
def test_tensor_attribute_reassigned_before_every_forward_is_fed_to_the_replay(keep_old):
    dev = require_gpu()
    kept = []

    def before(it, f, y):
        f.x0 = (y * (1.0 + 0.2 * it)).clone().detach()
        if keep_old:
            kept.append(f.x0)                      # every call sees a NEW address (nothing is freed for the allocator to reuse)
    auto, ode_a, fa, warns = _runs(GrandLike, BASE, 9, dev, before_call=before)
    kept = []
    eager, ode_e, fe, _ = _runs(GrandLike, EAGER, 9, dev, before_call=before)
    assert _calls_equal(auto, eager) == [True] * 9
    assert ode_a.graphs_captured and ode_a.graph_status == "graph(auto)", ode_a.graph_status
    assert fa.nfe == fe.nfe and not [w for w in warns if "hipGraph" in w]
    if keep_old:
        assert (0, "x0") in ode_a._volatile
        e = [e for e in ode_a._graphs.values() if e.g_f is not None][0]
        assert [n for _, n, _ in e.static_in] == ["x0"] and fa.x0 is not e.static_in[0][2]       # the user's tensor is back in place
    # the explicit mode goes through the same key and the same static copies
    forced, ode_f, _, _ = _runs(GrandLike, dict(BASE, pn_graph_capture=1), 7, dev, before_call=before)
    assert ode_f.graphs_captured and _calls_equal(forced, eager[:7]) == [True] * 7


def test_edge_lists_reassigned_in_a_container_stay_correct():
    """block_pnode.py:61-63 style: tensors re-assigned inside a list attribute are guarded by address (never fed): calls
    whose addresses were not seen before run eagerly."""
    dev = require_gpu()

    class Edgy(GrandLike):
        def forward(self, t, y):
            idx, w = self.edges
            return self.alpha * (torch.tanh(self.lin(y))[:, idx] * w - y)
    keep = []

    def before(it, f, y):
        g = torch.Generator().manual_seed(it)
        f.edges = [torch.randperm(16, generator=g).to(y.device), torch.rand(16, generator=g).to(y.device)]
        keep.append(f.edges)
    auto, ode_a, _, _ = _runs(Edgy, BASE, 7, dev, before_call=before)
    eager, _, _, _ = _runs(Edgy, EAGER, 7, dev, before_call=before)
    assert _calls_equal(auto, eager) == [True] * 7 and not ode_a.graphs_captured


def test_float_attribute_annealed_between_calls():
    dev = require_gpu()

    def every_call(it, f, y):
        f.x0 = torch.zeros_like(y) if f.x0 is None else f.x0
        f.alpha = 0.8 - 0.05 * it
    auto, ode_a, _, _ = _runs(GrandLike, BASE, 8, dev, before_call=every_call)
    eager, _, _, _ = _runs(GrandLike, EAGER, 8, dev, before_call=every_call)
    assert _calls_equal(auto, eager) == [True] * 8 and not ode_a.graphs_captured       # no value is seen three times

    def per_epoch(it, f, y):
        f.x0 = torch.zeros_like(y) if f.x0 is None else f.x0
        f.alpha = 0.8 if it < 6 else 0.4
    auto, ode_a, _, warns = _runs(GrandLike, BASE, 12, dev, before_call=per_epoch)
    eager, _, _, _ = _runs(GrandLike, EAGER, 12, dev, before_call=per_epoch)
    assert _calls_equal(auto, eager) == [True] * 12
    assert ode_a.graph_status == "graph(auto)" and sum(e.g_b is not None for e in ode_a._graphs.values()) == 2
    assert not [w for w in warns if "hipGraph" in w]


def test_flag_toggled_and_host_tensor_changed_in_place():
    dev = require_gpu()

    class Flagged(nn.Module):
        def __init__(self):
            super().__init__()
            self.lin = nn.Linear(16, 16)
            self.use_x = True
            self.scale = torch.tensor(1.0)                 # stays on the host (a plain attribute: .to(dev) does not move it)

        def forward(self, t, y):
            out = torch.tanh(self.lin(y)) * self.scale      # a 0-dim host tensor is a kernel ARGUMENT: baked in at capture
            return out - y if self.use_x else out

    def before(it, f, y):
        f.use_x = it % 2 == 0
        if it == 9:
            f.scale.fill_(0.5)
    auto, ode_a, _, _ = _runs(Flagged, BASE, 14, dev, before_call=before)
    eager, _, _, _ = _runs(Flagged, EAGER, 14, dev, before_call=before)
    assert _calls_equal(auto, eager) == [True] * 14
    assert sum(e.g_b is not None for e in ode_a._graphs.values()) >= 2                   # one pair per value of the flag


def test_buffer_replaced_by_assignment():
    dev = require_gpu()

    class Masked(nn.Module):
        def __init__(self):
            super().__init__()
            self.lin = nn.Linear(16, 16)
            self.register_buffer("mask", torch.ones(16))

        def forward(self, t, y):
            return torch.tanh(self.lin(y)) * self.mask
    old = []

    def before(it, f, y):
        if it in (5, 8):
            old.append(f.mask)                              # the captured address stays allocated -- and stale
            f.mask = torch.full((16,), 0.5 if it == 5 else 0.25, device=y.device)
    auto, ode_a, fa, _ = _runs(Masked, BASE, 12, dev, before_call=before)
    eager, _, _, _ = _runs(Masked, EAGER, 12, dev, before_call=before)
    assert "mask" in fa._buffers
    assert _calls_equal(auto, eager) == [True] * 12
    assert ode_a.graphs_captured, ode_a.graph_status


def test_ffjord_style_resampling_inside_the_first_evaluation_keeps_the_solver_eager():
    dev = require_gpu()

    class Hutch(nn.Module):
        def __init__(self):
            super().__init__()
            self.lin = nn.Linear(16, 16)
            self._e = None

        def forward(self, t, y):
            if self._e is None:
                self._e = torch.randn_like(y)
            return torch.tanh(self.lin(y)) + 0.1 * self._e

    def before(it, f, y):
        f._e = None                                         # odefunc.py:341-343 before_odeint
    auto, ode_a, _, warns = _runs(Hutch, BASE, 7, dev, before_call=before, seed_calls=True)
    eager, _, _, _ = _runs(Hutch, EAGER, 7, dev, before_call=before, seed_calls=True)
    assert _calls_equal(auto, eager) == [True] * 7
    assert not ode_a.graphs_captured and "not a plain call counter" in ode_a.graph_status and "_e" in ode_a.graph_status


GAIN = {"v": 1.0}


class UsesGlobal(nn.Module):
    def __init__(self):
        super().__init__()
        self.lin = nn.Linear(16, 16)
        self.nfe = 0

    def forward(self, t, y):
        self.nfe += 1
        return torch.tanh(self.lin(y)) * GAIN["v"]


def test_state_the_guard_cannot_see_is_caught_by_the_periodic_revalidation():
    dev = require_gpu()

    def before(it, f, y):
        GAIN["v"] = 1.0 if it < 6 else 0.5
    try:
        # every replayed call is validated: always right, one warning when the change is met, eager from then on
        auto, ode_a, fa, warns = _runs(UsesGlobal, dict(BASE, pn_graph_revalidate=1), 10, dev, before_call=before)
        eager, ode_e, fe, _ = _runs(UsesGlobal, EAGER, 10, dev, before_call=before)
        assert _calls_equal(auto, eager) == [True] * 10
        assert "no longer computes what was captured" in ode_a.graph_status
        assert sum("no longer computes what was captured" in w for w in warns) == 1
        assert fa.nfe == fe.nfe and (ode_a.nfe_forward, ode_a.nfe_backward) == (ode_e.nfe_forward, ode_e.nfe_backward)
        # every third: capture in call 2, replays 3 4 (5 validated) 6 7 (8 validated: caught) -- calls 6 and 7 are the blind window
        auto, ode_a, _, warns = _runs(UsesGlobal, dict(BASE, pn_graph_revalidate=3), 10, dev, before_call=before)
        eq = _calls_equal(auto, eager)
        assert eq[:6] == [True] * 6 and eq[8:10] == [True] * 2 and eq[6:8] == [False, False]
        assert sum("no longer computes what was captured" in w for w in warns) == 1
        # a change far below any tolerance: this func reproduced its eager twin BIT FOR BIT when it was captured, so re-validation
        # demands the same -- 1e-6 of relative change in the gain is caught, not waved through as launch-to-launch noise
        def tiny(it, f, y):
            GAIN["v"] = 1.0 if it < 6 else 1.000001
        auto, ode_a, _, warns = _runs(UsesGlobal, dict(BASE, pn_graph_revalidate=1), 10, dev, before_call=tiny)
        eager_t, _, _, _ = _runs(UsesGlobal, EAGER, 10, dev, before_call=tiny)
        assert not _calls_equal(eager_t, eager)[6] and _calls_equal(auto, eager_t) == [True] * 10
        assert "no longer computes what was captured" in ode_a.graph_status
        # switched off: stale for ever (what round 4 did for everything the key did not hold)
        auto, ode_a, _, _ = _runs(UsesGlobal, dict(BASE, pn_graph_revalidate=0), 10, dev, before_call=before)
        assert _calls_equal(auto, eager)[6:] == [False] * 4 and ode_a.graph_status == "graph(auto)"
    finally:
        GAIN["v"] = 1.0


def test_revalidation_of_a_sound_capture_is_silent_and_keeps_counters_and_statistics():
    """Re-validating calls run func's Python twice (eager twin + replay bookkeeping): call counters, NFE and BatchNorm
    statistics must read what they read with eager launches after every call."""
    dev = require_gpu()

    class BNCount(nn.Module):
        def __init__(self):
            super().__init__()
            self.l1, self.bn, self.l2 = nn.Linear(16, 16), nn.BatchNorm1d(16), nn.Linear(16, 16)
            self.nfe = 0

        def forward(self, t, y):
            self.nfe += 1
            return self.l2(torch.tanh(self.bn(self.l1(y))))
    log = {}

    def record(tag):
        def cb(it, f, ode):
            log.setdefault(tag, []).append((f.nfe, f.bn.running_mean.clone(), int(f.bn.num_batches_tracked),
                                            ode.nfe_forward, ode.nfe_backward))
        return cb
    auto, ode_a, _, warns = _runs(BNCount, dict(BASE, pn_graph_revalidate=2), 11, dev, after_call=record("a"))
    eager, _, _, _ = _runs(BNCount, EAGER, 11, dev, after_call=record("e"))
    assert _calls_equal(auto, eager) == [True] * 11 and ode_a.graph_status == "graph(auto)"
    for a, b in zip(log["a"], log["e"]):
        assert a[0] == b[0] and torch.equal(a[1], b[1]) and a[2:] == b[2:]
    assert not [w for w in warns if "hipGraph" in w]
    # forward-only calls (evaluation under no_grad) are re-validated too
    y0 = torch.randn(64, 16, device=dev)
    with torch.no_grad():
        outs = [ode_a.odeint_adjoint(y0, torch.tensor([0.3])).clone() for _ in range(8)]
    assert all(torch.equal(o, outs[0]) for o in outs) and ode_a.graph_status == "graph(auto)"
