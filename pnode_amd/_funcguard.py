"""What a replayed sweep cannot see: the Python side of ``func``.

A hipGraph of a whole sweep (``pnode_amd/_sweepgraphs.py``) replays kernels with the addresses and the scalar arguments
they had when they were captured.  The reference calls ``func`` at every stage (pa.py:393-412 ``evalRHSFunction``,
52-82 ``RHSJacShell.multTranspose``) and therefore sees whatever its callers did to the module between two solves:

* ``self.odefunc.x0 = x0.clone().detach()`` before every forward (examples-sinode/grand/src/base_classes.py:58-60, read at
  function_laplacian_diffusion.py:59), ``edge_index`` / ``edge_weight`` re-assigned (grand/src/block_pnode.py:61-63);
* ``self._e = None`` before the solve and a fresh sample inside the first evaluation
  (examples-pnode/ffjord-pnode/lib/layers/odefunc.py:341-364);
* scalar hyper-parameters that are annealed, flags that are toggled.

A *snapshot* is everything of that kind that can be found from the modules of func without running it:

    structure  (id, training flag) of every sub-module, in ``modules()`` order
    scalars    every int / float / bool / str / None / dtype / device / Size reachable from a module's own attributes
               through lists, tuples and dicts (three levels), and the identity of every other object found there
    tensors    address, shape, strides, dtype, device of every parameter, buffer and plain tensor attribute (same reach);
               host tensors also carry their version counter -- a 0-dim host tensor is baked into kernel arguments

The snapshot taken when a call starts is part of the capture key: a change that happened between two calls selects another
(or a new) capture, never a stale one.  What it cannot see -- closures, globals, attributes of foreign objects, in-place
changes of host-side containers deeper than three levels -- is what the periodic re-validation
(``-pn_graph_revalidate``) is for.
"""
import torch
import torch.nn as nn

_INTERNAL = frozenset(nn.Module().__dict__)            # training, _parameters, _buffers, _modules, the hook tables, ...
_SCALAR_TYPES = frozenset((int, float, bool, str, complex, type(None), torch.dtype, torch.device, torch.Size, bytes))
MAX_DEPTH = 3
MAX_ITEMS = 512                                        # per container: beyond it only the container's identity and length

# layout of a tensor record
T_MOD, T_PATH, T_PTR, T_SHAPE, T_STRIDE, T_DTYPE, T_DEV, T_VER, T_KIND = range(9)


def modules_of(funcs):
    """The modules a snapshot indexes, in its order."""
    out, seen = [], set()
    for f in funcs:
        if isinstance(f, nn.Module) and id(f) not in seen:
            seen.add(id(f))
            out += list(f.modules())
    return out


def _tensor_record(mi, path, v, kind):
    try:
        ptr = v.data_ptr()
    except Exception:                                  # no storage (meta / functional wrappers): identity only
        ptr = ("id", id(v))
    dev = v.device
    return (mi, path, ptr, tuple(v.shape), tuple(v.stride()) if v.layout == torch.strided else None, v.dtype,
            (dev.type, dev.index), v._version if dev.type == "cpu" else 0, kind)


def _leaf(mi, path, v, depth, scalars, tensors):
    tv = type(v)
    if tv in _SCALAR_TYPES:
        scalars.append((mi, path, v))
    elif isinstance(v, torch.Tensor):
        try:
            tensors.append(_tensor_record(mi, path, v, "a"))
        except Exception:                              # an exotic tensor subclass (nested, fake, ...): identity only
            scalars.append((mi, path, ("tensor", id(v))))
    elif tv in (list, tuple) or isinstance(v, (list, tuple)):
        scalars.append((mi, path, ("seq", tv.__name__, len(v))))
        if depth < MAX_DEPTH and len(v) <= MAX_ITEMS:
            base = path if isinstance(path, tuple) else (path,)
            for i, x in enumerate(v):
                _leaf(mi, base + (i,), x, depth + 1, scalars, tensors)
        else:
            scalars.append((mi, path, ("id", id(v))))
    elif isinstance(v, dict):
        scalars.append((mi, path, ("map", len(v))))
        if depth < MAX_DEPTH and len(v) <= MAX_ITEMS:
            base = path if isinstance(path, tuple) else (path,)
            for k, x in v.items():
                _leaf(mi, base + (k if type(k) in _SCALAR_TYPES else ("id", id(k)),), x, depth + 1, scalars, tensors)
        else:
            scalars.append((mi, path, ("id", id(v))))
    elif tv.__module__ == "numpy" and hasattr(v, "tobytes"):
        if getattr(v, "size", 1 << 30) <= 64:
            scalars.append((mi, path, ("np", str(getattr(v, "dtype", "")), tuple(getattr(v, "shape", ())), v.tobytes())))
        else:
            scalars.append((mi, path, ("np", id(v), tuple(v.shape), str(v.dtype))))
    elif tv.__name__ in ("Namespace", "SimpleNamespace") and depth < MAX_DEPTH:
        base = path if isinstance(path, tuple) else (path,)
        scalars.append((mi, path, ("ns", len(vars(v)))))
        for k, x in vars(v).items():
            _leaf(mi, base + (k,), x, depth + 1, scalars, tensors)
    else:
        scalars.append((mi, path, ("id", id(v))))      # a module kept outside _modules, a function, a foreign object


def snapshot(funcs):
    """(structure, scalars, tensors) of the modules of `funcs` -- see the module docstring.  ~1 us per attribute."""
    struct, scalars, tensors = [], [], []
    mi = 0
    for m in modules_of(funcs):
        struct.append((id(m), m.training))
        for k, v in m.__dict__.items():
            if k not in _INTERNAL:
                _leaf(mi, k, v, 0, scalars, tensors)
        for k, p in m._parameters.items():
            if p is None:
                scalars.append((mi, ("_parameters", k), None))
            else:
                tensors.append(_tensor_record(mi, k, p, "p") + (p.requires_grad,))
        for k, b in m._buffers.items():
            if b is None:
                scalars.append((mi, ("_buffers", k), None))
            else:
                tensors.append(_tensor_record(mi, k, b, "b"))
        mi += 1
    return tuple(struct), tuple(scalars), tuple(tensors)


def key_of(snap, counters, volatile):
    """The part of a capture key that stands for func's Python side.  `counters`: {(module index, name)} of the integer
    attributes func increments while it runs (their value is no configuration); `volatile`: {(module index, name)} of the
    tensor attributes that are fed to the captured sweep through a static copy (their address is no configuration)."""
    struct, scalars, tensors = snap
    if counters:
        scalars = tuple(s for s in scalars if (s[0], s[1]) not in counters)
    if volatile:
        tensors = tuple((t[:T_PTR] + ("fed",) + t[T_PTR + 1:]) if (t[T_MOD], t[T_PATH]) in volatile else t for t in tensors)
    return struct, scalars, tensors


def moved_tensors(prev, cur, device):
    """[(module index, name)] of the buffers and direct tensor attributes on `device` whose storage moved between two
    snapshots that describe the same structure: the ``func.x0 = x0.clone()`` of the reference's callers."""
    if prev is None or prev[0] != cur[0] or len(prev[2]) != len(cur[2]) or prev[2] == cur[2]:
        return []
    dev = (device.type, device.index)
    out = []
    for a, b in zip(prev[2], cur[2]):
        if a != b and a[:T_PTR] == b[:T_PTR] and a[T_PTR + 1:] == b[T_PTR + 1:] and a[T_KIND] in "ab" \
                and isinstance(a[T_PATH], str) and b[T_DEV] == dev:
            out.append((a[T_MOD], a[T_PATH]))
    return out


def holder_of(module, name):
    """The dictionary a buffer / plain tensor attribute of `module` lives in."""
    return module._buffers if name in module._buffers else module.__dict__


def counter_deltas(before, after):
    """What a sweep did to func's Python side, as [(module index, name, increment)] of integer attributes -- or None when
    it is not a set of call counters: something that is not an integer changed, an attribute appeared or vanished, a tensor
    attribute was (re)assigned, a train/eval flag flipped."""
    if before[0] != after[0] or len(before[1]) != len(after[1]) or before[2] != after[2]:
        return None
    out = []
    for (mi, k, v0), (mj, k2, v1) in zip(before[1], after[1]):
        if mi != mj or k != k2:
            return None
        if v0 is not v1 and v0 != v1:
            if type(v0) is not int or type(v1) is not int or not isinstance(k, str):
                return None
            out.append((mi, k, v1 - v0))
    return out


def describe_change(prev, cur, mods=None):
    """One line for warnings: the first difference between two snapshots."""
    def name(mi, path):
        cls = type(mods[mi]).__name__ if mods is not None and mi < len(mods) else "module %d" % mi
        return "%s.%s" % (cls, path if isinstance(path, str) else "".join("[%r]" % (p,) if i else str(p) for i, p in enumerate(path)))
    if prev[0] != cur[0]:
        return "a sub-module was added, removed, replaced or switched between train() and eval()"
    a, b = dict(((s[0], s[1]), s[2]) for s in prev[1]), dict(((s[0], s[1]), s[2]) for s in cur[1])
    for k in b:
        if k not in a or not (a[k] is b[k] or a[k] == b[k]):
            return "attribute %s changed (%r -> %r)" % (name(*k), a.get(k, "<absent>"), b[k])
    for k in a:
        if k not in b:
            return "attribute %s vanished" % name(*k)
    a, b = dict(((t[0], t[1]), t) for t in prev[2]), dict(((t[0], t[1]), t) for t in cur[2])
    for k in b:
        if k not in a:
            return "tensor attribute %s appeared" % name(*k)
        if a[k] != b[k]:
            return "tensor attribute %s was re-assigned or re-allocated" % name(*k)
    for k in a:
        if k not in b:
            return "tensor attribute %s vanished" % name(*k)
    return "no visible difference"
