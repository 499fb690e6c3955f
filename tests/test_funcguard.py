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
