"""hipGraphs of single evaluations of ``func`` for ADAPTIVE sweeps (``-pn_graph_capture`` with ``-ts_adapt_type basic``).

A fixed-step sweep is captured whole (pnode_amd/_sweepgraphs.py).  An adaptive one cannot be: the controller reads the
error norm on the host after every attempt (pn_ts_judge), and the reverse sweep's step sizes are new at every call.  What
does not change from step to step is what ONE evaluation of func launches: the reference calls ``func(t, y)`` once per stage
(pa.py:393-412 ``evalRHSFunction``) and once more, with a backward pass, per stage of a reversed step (pa.py:52-82, 341-363
``multTranspose``), and on an MI355X the host's dispatch of those few dozen PyTorch kernels -- not the device -- bounds an
adaptive solve at BASELINE's sizes (profiles/r06_stiff_reverse_host.txt: 91 % of the reverse sweep's host time is PyTorch's
own launches).  So each KIND of evaluation is captured once per stage index and replayed:

    F  slot i   K_i = func(t, Y_i), no tape                       (forward attempts; re-advancing from a checkpoint)
    A  slot i   the same, recorded by autograd                    (stage values recomputed for a reversed step, _stages_of)
    B  slot i   the backward pass over A_i's tape, cotangent w    (stage VJP with a tape)
    AB slot i   evaluation and backward pass in one               (stage VJP without one)

Everything that differs between two uses stays OUTSIDE the captured region and is launched per use, as before:

* the stage value: a unit reads either one of the solver's named work buffers (same address at every step: captured in
  place) or its own static copy (``pn_copy`` per use);
* the time: func receives a 0-dim float64 DEVICE tensor of the unit's own (autograd may save it: ``y * t``), filled before every
  replay.  The reference hands func a Python float (pa.py:405).  A func that needs one -- ``float(t)``, ``math.sin(t)``,
  ``if t < 0.5`` -- synchronises with the host, which a capture refuses: the solver then stays with eager launches (same
  results).  A func that ignores t or uses it in tensor arithmetic (``y * t``, ``torch.sin(t)``) is replayed; arithmetic an eager
  evaluation does with t on the HOST (``torch.sin(torch.as_tensor(t))``: a CPU sine) becomes device arithmetic, whose last bit may
  differ -- ``auto`` accepts that within its validation tolerance and says so in ``graph_status`` ("replays within ...");
* the step size: it only enters the solver's own kernels (pn_rk_stage, pn_adj_theta, pn_adj_accum, the scale of the parameter
  sensitivities), all of which are launched per use with the step's coefficients.  The grouped ``pn_linear_wgrad`` launch of a
  stage VJP (pnode_amd/_lineargrad.py) is deferred out of the captured backward pass for that reason: the hooks record which
  (cotangent, input) pairs the stage produces -- static addresses -- and the launch follows every replay with the stage's scale.

The launch mode is decided as for whole sweeps (``auto``): two calls run BOTH ways and must agree -- the second of them consists
of replays only, at other times and step sizes than the ones captured, so a scalar baked into a unit shows -- and replaying
must not be slower than launching.
"""
import weakref

import torch

from ._lib import PnError


class _Unit(object):
    __slots__ = ("graph", "y_static", "y_addr", "t", "out", "tape", "gy", "gp", "groups", "nfe_f", "nfe_b", "deltas", "uses")


class StageGraphs(object):
    """The captured evaluations of one solver under one capture key (func's Python-side state, shapes, modes)."""

    def __init__(self, ode):
        self._ode = weakref.ref(ode)
        self.units = {}
        self.by_tape = {}                 # id(tape) -> the A unit that owns it
        self._work_addrs, self._work_n = set(), -1
        self.captured = 0
        self.replayed = 0

    # ------------------------------------------------------------------ capture
    def _in_place(self, ode, y_flat):
        """The address of `y_flat` when it is one of the solver's named work buffers (the same at every step: a unit is captured
        ON it, and a unit per such buffer), else 0 (a trajectory slot, a view: the unit gets a static copy that is filled per use)."""
        work = ode._work
        if len(work) != self._work_n:          # (the named buffers are made on first use and never freed)
            self._work_addrs = {b.data_ptr() for b in work.values()}
            self._work_n = len(work)
        a = y_flat.data_ptr()
        return a if a in self._work_addrs else 0

    def _capture(self, ode, fn):
        """Run `fn` under stream capture; returns (graph, what fn returned, nfe increments, func's counter increments)."""
        ops = ode._ops
        g = torch.cuda.CUDAGraph()
        fp0 = ode._py_fingerprint()
        n0 = (ode.nfe_forward, ode.nfe_backward)
        pinned, ops._pinned_stream = ops._pinned_stream, None     # a launch of the solver's inside the region belongs to the capture
        sg, ode._sg = ode._sg, None
        ode._unit_capture = True
        if ode._lin is not None:
            ode._lin.unit_capture = True
        try:
            with torch.cuda.graph(g, capture_error_mode=ode.GRAPH_CAPTURE_MODE):
                res = fn()
        finally:
            ode._unit_capture = False
            if ode._lin is not None:
                ode._lin.unit_capture = False
            ode._sg = sg
            ops._pinned_stream = pinned
        fp1 = ode._py_fingerprint()
        d = [] if fp1 == fp0 else ode._counter_deltas(fp0, fp1)
        if d is None:
            raise PnError("func changes Python-side state during an evaluation in a way that is not a plain call counter: "
                          "replays would freeze it")
        self.captured += 1
        return g, res, (ode.nfe_forward - n0[0], ode.nfe_backward - n0[1]), d

    def _replayed(self, ode, u):
        ode.nfe_forward += u.nfe_f
        ode.nfe_backward += u.nfe_b
        if u.deltas:
            ode._bump(u.deltas)
        self.replayed += 1

    def _new_unit(self, ode):
        u = _Unit()
        # the time func sees: a device scalar of the unit's OWN (autograd may save it -- ``y * t`` -- for the backward unit that
        # is replayed after later stages have been given their times); filled through .data, as _restore writes func's buffers:
        # a write autograd can see would invalidate the captured tape
        u.t = torch.zeros((), dtype=torch.float64, device=ode.device)
        u.uses = 0
        return u

    def _feed(self, ode, u, y_flat):
        if u.y_static is not None:
            ode._ops.copy(u.y_static, y_flat)

    # ------------------------------------------------------------------ K = func(t, Y)
    def evaluate(self, ode, slot, t, y_flat, tape):
        kind = "A" if tape is not None else "F"
        yk = self._in_place(ode, y_flat)
        key = (kind, slot, yk)
        u = self.units.get(key)
        if u is None:
            u = self._new_unit(ode)
            u.t.data.fill_(t)
            u.y_static = None if yk else ode._ops.empty(ode._npad)
            u.y_addr = yk
            y_in = y_flat if u.y_static is None else u.y_static
            rec = [] if kind == "A" else None
            u.graph, u.out, (u.nfe_f, u.nfe_b), u.deltas = self._capture(ode, lambda: ode._call_func(u.t, y_in, rec))
            u.tape = rec[0] if rec else None
            if u.tape is not None:
                self.by_tape[id(u.tape)] = u
            u.gy = u.gp = u.groups = None
            self.units[key] = u
        else:
            u.t.data.fill_(t)
            self._replayed(ode, u)
        self._feed(ode, u, y_flat)
        u.graph.replay()
        u.uses += 1
        if tape is not None:
            tape.append(u.tape)
        if ode._fsal and slot == ode._s - 1:
            return u.out.clone()          # first-same-as-last: this derivative outlives the step (and a rejected attempt's retry)
        return u.out

    # ------------------------------------------------------------------ (J^T w, parameter cotangents)
    def vjp(self, ode, slot, t, y_flat, w_flat, tape, alpha, last):
        kind = "B" if tape is not None else "AB"
        yk = self._in_place(ode, y_flat) if kind == "AB" else 0
        key = (kind, slot, w_flat.data_ptr(), id(tape) if kind == "B" else yk)
        u = self.units.get(key)
        if ode._pend_g or ode._pend_bias:
            ode._flush_param_accum()       # what is queued may sit in the static outputs of the unit that is replayed next
        if kind == "B":
            a = self.by_tape.get(id(tape))
            if a is None or a.tape is not tape:
                raise PnError("pnode_amd: a stage VJP was handed a tape that is not its captured evaluation's")
        if u is None:
            lin = ode._lin
            u = self._new_unit(ode)
            u.t.data.fill_(t)
            u.y_static = ode._ops.empty(ode._npad) if (kind == "AB" and not yk) else None
            u.y_addr = yk
            y_in = y_flat if u.y_static is None else u.y_static
            if lin is not None:
                if lin.side_on:
                    raise PnError("-pn_linear_side_stream and per-evaluation graphs do not combine")
                lin.defer = []
            try:
                u.graph, (u.gy, gp), (u.nfe_f, u.nfe_b), u.deltas = self._capture(
                    ode, lambda: ode._vjp(u.t, y_in, w_flat, tape, alpha=alpha, last=last))
                groups = lin.defer if lin is not None else []
            finally:
                if lin is not None:
                    lin.defer = None
            u.gp = list(gp)
            # the deferred products: (cotangent, input, the layer's partial-sum record) per pair, one list per launch
            u.groups = []
            for grp in groups:
                items = []
                for g, x, _, pw, pb in grp:
                    st = next(s for s in lin.partials.values() if s[0] is pw)
                    items.append((g, x, st))
                u.groups.append(items)
            u.out = u.tape = None
            self.units[key] = u
        else:
            if kind == "AB":
                u.t.data.fill_(t)
            self._replayed(ode, u)
        if kind == "AB":
            self._feed(ode, u, y_flat)
        u.graph.replay()
        u.uses += 1
        for items in u.groups:
            for _, _, st in items:
                st[2] = True
            ode._ops.linear_wgrad_group([(g, x, alpha, st[0], st[1]) for g, x, st in items])
        return u.gy, list(u.gp)

    def min_uses(self):
        """Fewest uses of any captured unit: a validated sweep must have REPLAYED each of them at least once."""
        return min((u.uses for u in self.units.values()), default=0)
