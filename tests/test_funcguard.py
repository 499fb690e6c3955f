"""The guard that keeps ``-pn_graph_capture auto`` from replaying stale Python-side state of func
(pnode_amd/_funcguard.py).  Host logic only: what a snapshot sees, what the capture key keeps of it, and which changes it
tells apart.  The patterns are the ones the reference's callers use between two solves
(/root/reference/examples-sinode/grand/src/base_classes.py:58-60, block_pnode.py:61-63,
examples-pnode/ffjord-pnode/lib/layers/odefunc.py:341-364); the replay behaviour itself is tested on the GPU
(tests/test_gpu_graph_guard.py)."""
import types

import torch
import torch.nn as nn

from pnode_amd import _funcguard as fg


class Func(nn.Module):
    def __init__(self):
        super().__init__()
        self.lin = nn.Linear(4, 4)
        self.register_buffer("mask", torch.ones(4))
        self.alpha = 0.5
        self.nfe = 0
        self.x0 = torch.zeros(3, 4)
        self.opt = {"beta": 1.0, "names": ["a", "b"], "w": torch.ones(2)}
        self._e = None

    def forward(self, t, y):
        self.nfe += 1
        return self.alpha * self.lin(y) * self.mask


def key(f, counters=(), volatile=()):
    return fg.key_of(fg.snapshot((f, f)), set(counters), set(volatile))


def test_snapshot_is_stable_and_sees_every_kind_of_attribute():
    f = Func()
    s1, s2 = fg.snapshot((f, f)), fg.snapshot((f, f))
    assert s1 == s2 and hash(fg.key_of(s1, set(), set())) == hash(fg.key_of(s2, set(), set()))
    names = {(mi, p) for mi, p, _ in s1[1]}
    assert (0, "alpha") in names and (0, "nfe") in names and (0, "_e") in names and (0, ("opt", "beta")) in names
    assert (0, ("opt", "names", 1)) in names
    tnames = {(t[0], t[1], t[fg.T_KIND]) for t in s1[2]}
    assert (0, "x0", "a") in tnames and (0, "mask", "b") in tnames and (1, "weight", "p") in tnames and (0, ("opt", "w"), "a") in tnames
    assert len(s1[0]) == 2 and s1[0][0] == (id(f), True)


def test_changes_between_calls_change_the_key():
    f = Func()
    k0 = key(f)
    f.alpha = 0.4                                           # a float that is annealed
    k1 = key(f)
    assert k1 != k0
    f.alpha = 0.5
    assert key(f) == k0
    f.x0 = f.x0.clone()                                     # GRAND: a tensor attribute re-assigned before every forward
    assert key(f) != k0
    f2 = Func()
    k0 = key(f2)
    f2.mask = torch.zeros(4)                                # a buffer replaced by assignment (nn.Module keeps it in _buffers)
    assert "mask" in f2._buffers and key(f2) != k0
    f3 = Func()
    k0 = key(f3)
    f3._e = torch.randn(3)                                  # FFJORD: None -> a sampled tensor
    assert key(f3) != k0
    f4 = Func()
    k0 = key(f4)
    f4.opt["beta"] = 2.0                                    # a dictionary of hyper-parameters, changed in place
    assert key(f4) != k0
    f5 = Func()
    k0 = key(f5)
    f5.eval()
    assert key(f5) != k0
    f6 = Func()
    k0 = key(f6)
    f6.lin = nn.Linear(4, 4)                                # a sub-module replaced
    assert key(f6) != k0
    f7 = Func()
    k0 = key(f7)
    f7.lin.weight.requires_grad_(False)                     # a parameter frozen
    assert key(f7) != k0


def test_host_tensors_are_guarded_by_value_version_device_tensors_by_address_only():
    f = Func()
    f.scale = torch.tensor(2.0)                             # 0-dim host tensor: baked into kernel arguments at capture
    k0 = key(f)
    f.scale.fill_(3.0)
    assert key(f) != k0
    k0 = key(f)
    with torch.no_grad():
        f.lin.weight.mul_(2.0)                              # an optimizer step: same storage, the replay reads the new values
    k1 = key(f)
    # (parameters live on the host in this container: their version is part of the record here, on the device it is not)
    rec0 = [t for t in k0[2] if t[fg.T_KIND] == "p"][0]
    rec1 = [t for t in k1[2] if t[fg.T_KIND] == "p"][0]
    assert rec0[fg.T_PTR] == rec1[fg.T_PTR] and rec0[fg.T_SHAPE] == rec1[fg.T_SHAPE]


def test_counters_are_learnt_and_left_out_of_the_key():
    f = Func()
    before = fg.snapshot((f, f))
    f(0.0, torch.zeros(3, 4))
    f(0.0, torch.zeros(3, 4))
    after = fg.snapshot((f, f))
    d = fg.counter_deltas(before, after)
    assert d == [(0, "nfe", 2)]
    assert fg.key_of(before, {(0, "nfe")}, set()) == fg.key_of(after, {(0, "nfe")}, set())
    assert fg.key_of(before, set(), set()) != fg.key_of(after, set(), set())
    # anything else that moves during a sweep is not a counter
    for change in (lambda: setattr(f, "alpha", 0.1), lambda: setattr(f, "_e", torch.ones(2)), lambda: setattr(f, "flag", True),
                   lambda: setattr(f, "x0", f.x0.clone()), lambda: f.lin.train(False), lambda: f.opt.__setitem__("beta", 3.0)):
        b = fg.snapshot((f, f))
        change()
        assert fg.counter_deltas(b, fg.snapshot((f, f))) is None
    b = fg.snapshot((f, f))
    assert fg.counter_deltas(b, fg.snapshot((f, f))) == []


def test_moved_tensors_finds_reassigned_attributes_and_buffers_only():
    f = Func()
    dev = torch.device("cpu")
    s0 = fg.snapshot((f, f))
    assert fg.moved_tensors(None, s0, dev) == [] and fg.moved_tensors(s0, s0, dev) == []
    f.x0 = f.x0.clone()
    f.mask = torch.zeros(4)
    s1 = fg.snapshot((f, f))
    assert sorted(fg.moved_tensors(s0, s1, dev)) == [(0, "mask"), (0, "x0")]
    vol = set(fg.moved_tensors(s0, s1, dev))
    assert fg.key_of(s0, set(), vol) == fg.key_of(s1, set(), vol)          # fed by copy: the address is no configuration
    f.x0 = torch.zeros(5, 4)                                                # another shape is another capture
    s2 = fg.snapshot((f, f))
    assert fg.moved_tensors(s1, s2, dev) == [] and fg.key_of(s1, set(), vol) != fg.key_of(s2, set(), vol)
    f.opt["w"] = torch.ones(2)                                              # inside a container: guarded by address, not fed
    s3 = fg.snapshot((f, f))
    assert fg.moved_tensors(s2, s3, dev) == [] and fg.key_of(s2, set(), vol) != fg.key_of(s3, set(), vol)
    f.lin.weight = nn.Parameter(torch.zeros(4, 4))                          # a parameter is never fed by copy
    s4 = fg.snapshot((f, f))
    assert fg.moved_tensors(s3, s4, dev) == []
    assert fg.holder_of(f, "mask") is f._buffers and fg.holder_of(f, "x0") is f.__dict__


def test_containers_namespaces_arrays_and_foreign_objects():
    import numpy as np
    f = Func()
    f.args = types.SimpleNamespace(lr=0.1, tol=1e-3)
    f.arr = np.arange(4.0)
    f.big = np.zeros(1000)
    f.fn = torch.tanh
    f.layers = [nn.Linear(2, 2)]                            # modules kept outside _modules: identity only
    k0 = key(f)
    f.args.tol = 1e-4
    k1 = key(f)
    assert k1 != k0
    f.arr[1] = 7.0
    k2 = key(f)
    assert k2 != k1
    f.big[3] = 1.0                                          # large arrays: identity and shape only (re-validation's business)
    assert key(f) == k2
    f.fn = torch.sigmoid
    k3 = key(f)
    assert k3 != k2
    f.deep = [[[[1.0]]]]                                    # below three levels: the container's identity
    k4 = key(f)
    f.deep[0][0][0][0] = 2.0
    assert key(f) == k4
    f.deep = [[[[2.0]]]]
    assert key(f) != k4


def test_describe_change_names_the_attribute():
    f = Func()
    s0 = fg.snapshot((f, f))
    f.alpha = 0.25
    mods = fg.modules_of((f, f))
    assert "Func.alpha" in fg.describe_change(s0, fg.snapshot((f, f)), mods)
    s0 = fg.snapshot((f, f))
    f._e = torch.ones(2)
    msg = fg.describe_change(s0, fg.snapshot((f, f)), mods)
    assert "_e" in msg
    s0 = fg.snapshot((f, f))
    f.x0 = f.x0.clone()
    assert "x0" in fg.describe_change(s0, fg.snapshot((f, f)), mods)
    s0 = fg.snapshot((f, f))
    f.eval()
    assert "train()" in fg.describe_change(s0, fg.snapshot((f, f)), mods)


# ---------------------------------------------------------------- the capture key of the solver (host logic, no capture)
def _solver(f, y0):
    from _cpu_vecops import CpuVecOps
    from pnode_amd import options, petsc_adjoint
    options.clear()
    options.set_option("ts_adapt_type", "none")
    ode = petsc_adjoint.ODEPetsc(backend=CpuVecOps)
    ode.setupTS(y0, f, step_size=0.05, method="rk4")
    options.clear()
    return ode


def test_solver_capture_key_follows_func_state_counters_and_reassigned_tensors():
    """SweepGraphs._graph_lookup / _note_side_effects on the CPU stand-in (the key is built the same way on the device; the
    capture itself needs one): one entry per configuration of func, call counters learnt in a warm-up sweep leave the key,
    a tensor attribute that is re-assigned between calls becomes a fed input (its address leaves the key)."""
    f = Func()
    y0 = torch.randn(3, 4)
    t = torch.tensor([0.3])
    ode = _solver(f, y0)
    e1 = ode._graph_lookup(y0, t, True)
    assert ode._graph_lookup(y0, t, True) is e1 and len(ode._graphs) == 1
    # a warm-up sweep: func counts its calls -> learnt, and the entry is re-keyed without the counter's value
    before = ode._last_fp
    f(0.0, y0), f(0.0, y0)
    assert ode._note_side_effects(e1, "f", before) and e1.deltas_f[0][1:3] == ("nfe", 2) and (0, "nfe") in ode._counters
    assert ode._graph_lookup(y0, t, True) is e1                      # nfe is 2 now: same entry
    f.nfe = 0                                                          # the drivers reset it after printing
    assert ode._graph_lookup(y0, t, True) is e1
    # between calls: a float, then back
    f.alpha = 0.4
    e2 = ode._graph_lookup(y0, t, True)
    assert e2 is not e1
    f.alpha = 0.5
    assert ode._graph_lookup(y0, t, True) is e1
    # other output times, another gradient requirement: other entries
    assert ode._graph_lookup(y0, torch.tensor([0.4]), True) is not e1 and ode._graph_lookup(y0, t, False) is not e1
    # GRAND: x0 re-assigned before every call -> volatile after the first move; from then on ONE entry whatever the address
    f.x0 = f.x0.clone()
    ode._graph_lookup(y0, t, True)
    assert (0, "x0") in ode._volatile
    f.x0 = f.x0.clone()
    e3 = ode._graph_lookup(y0, t, True)
    keep = f.x0
    f.x0 = f.x0.clone()
    assert ode._graph_lookup(y0, t, True) is e3 and keep is not f.x0
    f.x0 = torch.zeros(5, 4)                                           # another shape: another entry
    assert ode._graph_lookup(y0, t, True) is not e3
    assert len(ode._graphs) <= ode.GRAPH_CACHE_ENTRIES


def test_state_that_changes_during_a_sweep_and_is_not_a_counter_vetoes_auto_mode():
    f = Func()
    y0 = torch.randn(3, 4)
    ode = _solver(f, y0)
    ode._graph_mode = 2
    e = ode._graph_lookup(y0, torch.tensor([0.3]), True)
    before = ode._last_fp
    f._e = torch.randn(3)                                              # FFJORD: sampled inside the first evaluation
    assert not ode._note_side_effects(e, "f", before)
    assert "not a plain call counter" in ode.graph_status and "_e" in ode.graph_status and ode._auto_veto
    # the explicit mode only learns counters (no bookkeeping at replay, no veto)
    f2 = Func()
    ode2 = _solver(f2, y0)
    ode2._graph_mode = 1
    e = ode2._graph_lookup(y0, torch.tensor([0.3]), True)
    before = ode2._last_fp
    f2(0.0, y0)
    f2.alpha = 0.1
    assert ode2._note_side_effects(e, "f", before, veto=False) and e.deltas_f is None and not ode2._auto_veto
    # a counter that does not count the same in two warm-up sweeps is no configuration either
    f3 = Func()
    ode3 = _solver(f3, y0)
    ode3._graph_mode = 2
    e = ode3._graph_lookup(y0, torch.tensor([0.3]), True)
    before = ode3._last_fp
    f3(0.0, y0)
    assert ode3._note_side_effects(e, "f", before)
    before = ode3._py_fingerprint()
    f3(0.0, y0), f3(0.0, y0)
    assert not ode3._note_side_effects(e, "f", before) and "does not count the same" in ode3.graph_status


def test_entries_survive_a_change_of_the_key_sets_and_alternating_configurations_give_up():
    f = Func()
    y0 = torch.randn(3, 4)
    t = torch.tensor([0.3])
    ode = _solver(f, y0)
    ode._graph_mode = 2
    e = ode._graph_lookup(y0, t, True)
    e.calls = 1
    f.x0 = f.x0.clone()                                    # second call: the attribute turns out to be re-assigned per call
    assert ode._graph_lookup(y0, t, True) is e and e.calls == 1 and (0, "x0") in ode._volatile      # the warm-up is not lost
    # ... but a pair captured BEFORE the tensor was known to move reads its old address: dropped, not re-keyed
    f.mask = torch.zeros(4)
    e.g_f = object()
    f.mask = torch.ones(4)
    ode._prev_fp = None                                     # (as if this were the call right after the capture)
    ode._graph_lookup(y0, t, True)
    f.mask = torch.zeros(4)
    e2 = ode._graph_lookup(y0, t, True)
    assert (0, "mask") in ode._volatile and e2 is not e and e not in ode._graphs.values()
    # more configurations alternating than the cache keeps, each captured and dropped again: auto gives up, with a warning
    import warnings
    f2 = Func()
    ode2 = _solver(f2, y0)
    ode2._graph_mode = 2
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        for it in range(60):
            f2.alpha = 0.1 * (it % 6)
            ent = ode2._graph_lookup(y0, t, True)
            ent.g_f = ent.g_f or object()
            if ode2._auto_veto:
                break
    assert ode2._auto_veto and "keep alternating" in ode2.graph_status and it < 40
    assert sum("keep alternating" in str(x.message) for x in w) == 1


def test_a_func_the_guard_cannot_describe_is_never_replayed():
    """Whatever makes the snapshot fail (here: a tensor-like attribute whose metadata raises) must end in eager launches
    with a reason, not in an exception out of odeint_adjoint and not in a guess."""
    import warnings
    from pnode_amd import _sweepgraphs

    class Odd(torch.Tensor):
        @property
        def shape(self):
            raise RuntimeError("ragged")
    f = Func()
    f.odd = torch.zeros(3).as_subclass(Odd)
    s = fg.snapshot((f, f))                                 # an exotic tensor: identity only, no failure
    assert any(x[1] == "odd" and x[2][0] == "tensor" for x in s[1])
    # a failure deeper down (simulated) vetoes the default mode
    y0 = torch.randn(3, 4)
    ode = _solver(Func(), y0)
    ode._graph_mode = 2
    ode.device = type("D", (), {"type": "cuda", "index": 0})()         # (pretend: _graph_entry only builds the key here)
    ode._py_fingerprint = lambda: (_ for _ in ()).throw(ValueError("cannot describe"))
    import pnode_amd
    safe = pnode_amd.GRAPH_REPLAY_SAFE
    pnode_amd.GRAPH_REPLAY_SAFE = True
    try:
        real = torch.cuda.is_current_stream_capturing
        torch.cuda.is_current_stream_capturing = lambda: False
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            assert ode._graph_entry(y0, torch.tensor([0.3]), True) is None
        assert "could not be inspected" in ode.graph_status and ode._auto_veto and len(w) == 1
    finally:
        torch.cuda.is_current_stream_capturing = real
        pnode_amd.GRAPH_REPLAY_SAFE = safe
