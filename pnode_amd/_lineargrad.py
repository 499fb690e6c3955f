"""Parameter sensitivities of func's ``nn.Linear`` layers, accumulated by the engine instead of by autograd.

Row a-9 of the hot path (``RHSJacPShell.multTranspose`` + ``_flatten_convert_none_to_zeros``, pa.py:341-363, misc.py:9-14,
and the VecAXPY on mu inside TSAdjointStep_RK): per stage VJP the reference takes every parameter gradient from
``torch.autograd.grad``, flattens, copies and scales them.  For a Linear layer ``out = x W^T + b`` those gradients are
``dW = G^T x`` and ``db = colsum(G)`` with G the cotangent at the layer's output -- and at BASELINE's target configuration
(4096 x 512, 4 layers, 4 stages per time step) autograd's ``db`` alone is sixteen 14.7 us reductions per time step, 15 % of
the whole step and four times the solver's own kernels.  Here, for the explicit RK path:

* a forward hook on every eligible ``nn.Linear`` of func registers, while THIS solver evaluates func with autograd on, a
  tensor hook on the layer's output; when the stage VJP's backward reaches it the hook receives G and
    - adds ``alpha * G^T x`` straight into W's slice of mu with one accumulating GEMM (``torch.addmm(out=mu_W)``: the GEMM
      autograd would run, without the separate gradient tensor and the later pass of pn_param_accum over it), and
    - adds ``alpha * colsum(G)`` into b's slice with ``pn_colsum_accum`` (one pass over G, include/pnode_amd.h);
  or -- on the device, for rows >= 256 and features % 64 == 0 -- queues (G, x, alpha): when the stage VJP's backward pass is
  through, the pairs of ALL its layers go through ONE launch of a hand-written MFMA kernel (``pn_linear_wgrad_group``,
  csrc/pn_linear.hip): dW and db in a single pass over G and x, accumulated over the stages and time steps of the reverse sweep
  in per-layer partial buffers that ``finish`` adds to mu when the sweep ends.  fp32 states: on the bf16 matrix cores, every
  operand split exactly into three bf16 terms (six bf16 products per fp32 product; ``-pn_linear_wgrad_exact 1``: the fp32 matrix
  instruction); fp64 states: v_mfma_f64_16x16x4_f64.  (``-pn_linear_param_grads gemm`` keeps the library GEMM everywhere;
  ``-pn_linear_side_stream 1`` puts the launch on a second stream beside the next stage's backward pass.)
* the stage VJP asks autograd for dL/dy and the parameters of every OTHER module only, so autograd prunes dW / db.

Eligible: exactly ``nn.Linear`` (no subclass), weight (and bias, if any) trainable and owned by no other module.

The guarantee (the reference differentiates f with respect to EVERY parameter at EVERY stage, pa.py:66-74; the engine-side
path is never less right than that): what a static look at the modules cannot rule out -- a weight that is ALSO used outside
its layer's call (``F.linear(h, self.l2.weight)``), perhaps only at some stage times (``if t < 0.15: ...``) -- is ruled out
per EVALUATION, structurally: every grad-enabled evaluation of func the solver records is followed by a walk of ITS autograd
graph from ``out.grad_fn`` (``Evaluation.clean``).  The walk passes through each hooked layer call along the layer's INPUT
edge only; if the ``AccumulateGrad`` node of any handled weight or bias is still reachable, autograd has a contribution the
hooks would not see, and THAT evaluation is differentiated by autograd with respect to all parameters (its hooks muted).
Evaluations recorded inside a hipGraph capture are checked when they are captured and again at every re-validation.
The walk is ~10 us for a four-layer MLP.  On top of it the first engine-side VJP of a solver is computed BOTH ways and
compared (``self_check``: this guards the kernels, not the structure); ``-pn_linear_param_grads 0`` switches all of it off.
"""
import ctypes
import functools
import warnings
import weakref

import torch
import torch.nn as nn

from ._lib import PnError


class Evaluation(object):
    """One grad-enabled evaluation of func recorded by the solver: the layer calls that were hooked in it and the verdict of
    the structural check."""
    __slots__ = ("muted", "through", "unfused")

    def __init__(self):
        self.muted = False         # True: this evaluation is differentiated by autograd alone, its hooks do nothing
        self.unfused = False       # (per-evaluation graphs) a hooked layer of this evaluation is outside the fused kernel's shapes
        self.through = {}          # grad_fn of a hooked layer output -> grad_fn of that call's input (or None)

    def clean(self, out, handled_ids):
        """True when no handled parameter takes part in this evaluation outside its hooked layer calls.

        Walks the autograd graph of `out`.  At the node that produced a hooked layer's output the walk does not descend into
        the node (whose other edges lead to that layer's weight and bias) but continues from the node that produced the
        call's input.  Reaching the AccumulateGrad node of a handled parameter any other way = another use of it."""
        root = getattr(out, "grad_fn", None)
        if root is None:
            return True
        through = self.through
        seen = set()
        stack = [root]
        while stack:
            fn = stack.pop()
            if fn in seen:
                continue
            seen.add(fn)
            if fn in through:
                nxt = through[fn]
                if nxt is not None:
                    stack.append(nxt)
                continue
            var = getattr(fn, "variable", None)          # AccumulateGrad
            if var is not None:
                if id(var) in handled_ids:
                    return False
                continue
            for nxt, _ in fn.next_functions:
                if nxt is not None:
                    stack.append(nxt)
        return True


class LinearParamGrads(object):
    def __init__(self, ode):
        self._ode = weakref.ref(ode)
        self.handles = []
        self.slots = {}            # id(module) -> (weight offset, weight numel, bias offset or None, bias numel)
        self.shapes = {}           # id(module) -> (out features, in features)
        self.handled = ()          # indices (into the solver's parameter list) this object accumulates
        self.rest = ()             # ... and the ones autograd still differentiates
        self.recording = None      # the Evaluation in progress (the solver is evaluating func with autograd on), else None
        self.n_clean = 0           # evaluations whose Linear sensitivities the hooks take ...
        self.n_autograd = 0        # ... and evaluations left to autograd by the structural check
        self.alpha = None          # the scale of the stage VJP in progress; None: no VJP of ours is running
        self.target = None         # flat buffer the hooks accumulate into (mu, or a scratch buffer during the self-check)
        self.checked = False
        self.muted = False         # hooks fire and do nothing (the autograd half of the self-check)
        self.disabled = False      # the self-check failed: hooks that are still registered on older tapes do nothing
        self.why = None
        self.cot_storage = None    # storage address of the cotangent buffer of the stage VJP in progress
        self.fused = True          # use the fused MFMA kernel where the shape allows (-pn_linear_param_grads gemm switches it off)
        self.partials = {}         # id(module) -> [pw, pb, dirty]: partial sums of the fused kernel over a reverse sweep
        self.pending = []          # (G, x, alpha, pw, pb) of the stage VJP in progress, waiting for the grouped launch
        self.unit_capture = False  # a stage evaluation is being captured as a hipGraph of its own
        self.defer = None          # a list while ONE stage VJP is captured as a hipGraph of its own (pnode_amd/_stagegraphs.py): flush
                                   # hands the queued pairs over instead of launching -- the launch follows every replay, with that use's scale
        self._fused_ok = {}        # (rows, out, in, dtype) -> the fused kernel takes this shape
        self.side_on = False       # launch the grouped products on a second stream, beside the next stage's backward pass
        self.side = None           # ... that stream (made on first use)
        self.side_priority = True  # ... at the device's lowest priority
        self._side_handle = None
        self.inflight = []         # [(done event, the tensors the launch reads)] of launches the main stream has not waited for
        self.events = []           # a small ring of events, reused

    # ------------------------------------------------------------------ set-up
    def install(self, func, params, offsets):
        """Hook the eligible Linear layers of `func`; returns True when there is at least one."""
        self.remove()
        if not isinstance(func, nn.Module):
            return False
        index = {id(p): k for k, p in enumerate(params)}
        self.offsets, self.lens = list(offsets), [p.numel() for p in params]      # where each of `params` lives in mu
        owners = {}
        for m in func.modules():
            for p in m._parameters.values():
                if p is not None:
                    owners[id(p)] = owners.get(id(p), 0) + 1
        handled = []
        for m in func.modules():
            if type(m) is not nn.Linear:
                continue
            w, b = m.weight, m.bias
            if id(w) not in index or owners.get(id(w), 0) != 1:
                continue
            if b is not None and (id(b) not in index or owners.get(id(b), 0) != 1):
                continue
            kw = index[id(w)]
            kb = index[id(b)] if b is not None else None
            self.slots[id(m)] = (offsets[kw], w.numel(), None if kb is None else offsets[kb], 0 if b is None else b.numel())
            self.shapes[id(m)] = (int(w.shape[0]), int(w.shape[1]))
            handled.append(kw)
            if kb is not None:
                handled.append(kb)
            self.handles.append(m.register_forward_hook(self._forward_hook))
        self.handled = tuple(sorted(handled))
        hs = set(handled)
        self.rest = tuple(k for k in range(len(params)) if k not in hs)
        self.checked = False
        return bool(self.handled)

    def remove_hooks_only(self):
        for h in self.handles:
            try:
                h.remove()
            except Exception:
                pass
        self.handles = []

    def remove(self):
        self.remove_hooks_only()
        self.slots, self.handled, self.rest = {}, (), ()
        self.shapes, self.partials = {}, {}
        self._fused_ok = {}

    def __del__(self):
        self.remove()
        if self._side_handle is not None:
            lib, h = self._side_handle
            self._side_handle = None
            try:
                lib.pn_stream_destroy(h)
            except Exception:
                pass

    @property
    def active(self):
        return bool(self.handled) and not self.disabled

    # ------------------------------------------------------------------ hooks
    def begin(self):
        """The solver starts a grad-enabled evaluation of func: the layer calls made until `end` are hooked."""
        self.recording = Evaluation()

    def end(self, out, handled_params):
        """The evaluation is over.  Returns True when the hooks account for every use of the handled parameters in it
        (`handled_params`: the tensors func saw as those parameters -- the aliases, inside a capture); else the evaluation
        is muted and the caller differentiates it with respect to all parameters."""
        ev, self.recording = self.recording, None
        if ev is None or out is None:
            return True
        try:
            # (unfused: inside a per-evaluation graph a layer outside the fused kernel's shapes would take the library GEMM, whose
            # scale is a host scalar -- it would be baked into the captured backward pass: autograd differentiates such an evaluation)
            ok = not ev.unfused and ev.clean(out, {id(p) for p in handled_params})
        finally:
            # the map's keys are nodes of the graph whose hooks hold `ev`: kept, that is a reference cycle through C++ objects the
            # garbage collector cannot see -- the evaluation's whole graph and its saved activations would never be freed
            ev.through = None
        if ok:
            self.n_clean += 1
            return True
        ev.muted = True
        self.n_autograd += 1
        return False

    def abort(self):
        ev, self.recording = self.recording, None
        if ev is not None:
            ev.through = None

    def _forward_hook(self, module, inputs, output):
        ev = self.recording
        if ev is None or not isinstance(output, torch.Tensor) or not output.requires_grad or output.grad_fn is None:
            return None
        if id(module) not in self.slots or not inputs or not isinstance(inputs[0], torch.Tensor):
            return None
        x = inputs[0]
        ode0 = self._ode()
        if output.dtype != x.dtype or output.dtype != module.weight.dtype or (ode0 is not None and output.dtype != ode0.tensor_dtype):
            # func runs this layer under autocast (or in another precision than the state's, by hand): autograd forms dW from the
            # low-precision copies and rounds it to that precision; the hook would form it in the state's precision -- more accurate,
            # but NOT what differentiating func gives: left to autograd
            ev.unfused = True
            return None
        if self.unit_capture:
            out_f, in_f = module.weight.shape
            rows = x.numel() // max(in_f, 1)
            ode = self._ode()
            st = self.partials.get(id(module))
            if not (self.fused and st is not None and ode is not None and x.is_cuda and x.dtype == ode.tensor_dtype
                    and ode._ops.linear_wgrad_supported(rows, out_f, in_f)):
                ev.unfused = True
                return None
        # the hook keeps an alias of the input outside autograd's saved tensors: the version is checked by hand, as
        # autograd checks its own ("modified by an inplace operation")
        output.register_hook(functools.partial(self._grad_hook, module, x.detach(), x._version, ev))
        ev.through[output.grad_fn] = x.grad_fn
        return None

    def _grad_hook(self, module, x, version, ev, g):
        if self.muted or self.disabled or ev.muted:
            return None
        if x._version != version:
            raise RuntimeError("pnode_amd: the input of an nn.Linear layer of func was modified by an inplace operation after "
                               "the layer was evaluated (version %d, expected %d): its weight sensitivity cannot be formed; "
                               "-pn_linear_param_grads 0 leaves the layer to autograd, which raises the same way"
                               % (x._version, version))
        ode = self._ode()
        if self.alpha is None or ode is None or self.target is None:
            # A backward pass that is not one of this solver's stage VJPs reached the layer: func differentiates through its own
            # layers inside forward (FFJORD's divergence, ffjord-pnode/lib/layers/odefunc.py; a CNF's trace estimator).  The
            # stage VJP then also runs through that inner differentiation's graph, where these layers take part a second time
            # -- not something a hook on the forward output can account for.  Autograd does all of it from here on: the
            # evaluations recorded so far are differentiated with respect to every parameter (ODEPetsc._vjp).
            self.disabled = True
            self.why = "func differentiates through its nn.Linear layers inside its own forward"
            self.remove_hooks_only()
            return None
        ow, nw, ob, nb = self.slots[id(module)]
        out_f, in_f = module.weight.shape
        g2 = g.reshape(-1, out_f)
        x2 = x.reshape(-1, in_f)
        if g2.dtype != self.target.dtype:
            g2 = g2.to(self.target.dtype)
        if x2.dtype != self.target.dtype:
            x2 = x2.to(self.target.dtype)
        rows = g2.shape[0]
        ops = ode._ops
        key = (rows, out_f, in_f, g2.dtype)
        fused = self._fused_ok.get(key)
        if fused is None:                 # (asked once per shape: the hook runs in every stage VJP of an eager sweep)
            fused = self._fused_ok[key] = bool(self.fused and g2.is_cuda and hasattr(ops, "linear_wgrad_group")
                                               and ops.linear_wgrad_supported(rows, out_f, in_f))
        if fused:
            # the fused MFMA kernel (csrc/pn_linear.hip): dW and db in one pass over G and X, accumulated over the stages and
            # steps of the sweep in the layer's partial buffers; ODEPetsc._finish_linear_accum adds them to mu at the sweep's end
            st = self.partials.get(id(module))
            if st is None and not torch.cuda.is_current_stream_capturing():
                st = self.partials[id(module)] = list(ops.linear_wgrad_buffers(out_f, in_f, ob is not None)) + [False]
            if st is not None:
                g2c, x2c = g2.contiguous(), x2.contiguous()
                if g2c.data_ptr() % 16 == 0 and x2c.data_ptr() % 16 == 0:
                    # Queued until the stage VJP in progress is through (ODEPetsc._vjp calls flush): the pairs of all layers of the
                    # stage go through ONE grouped launch.  Within a stage VJP every G and x stays valid -- autograd temporaries
                    # held here by reference, the solver's cotangent buffer, the stage value.  (NOT across stages -- NOTES_r05.md:
                    # the last layer's cotangent IS the solver's buffer, rewritten for the next stage.)  A layer applied twice in
                    # one evaluation would have two pairs adding to the same partial tiles in one launch: the queue is flushed first.
                    if any(q[3] is st[0] for q in self.pending):
                        self.flush(ode)
                    self.pending.append((g2c, x2c, self.alpha, st[0], st[1]))
                    st[2] = True
                    return None
        if self.defer is not None:
            # (the library GEMM below takes the stage's scale as a host scalar: it would be baked into the captured backward pass)
            raise PnError("pnode_amd: an nn.Linear layer outside the fused kernel's shapes inside a per-evaluation graph")
        mw = self.target[ow: ow + nw].view(out_f, in_f)
        if g2.dtype == torch.float64 and rows % 8 == 0 and rows >= 4 * max(out_f, in_f):
            # the K-deep double-precision GEMM (K = rows) is the one shape hipBLASLt serves badly here: 129 us at 4096 x 512 x 512
            # against 40 us for the forward- and dX-shaped products of the same size.  Split K by hand -- one batched GEMM over
            # eight row chunks, the chunk sum folded into the accumulation: 46 us (tools/mb_dw_gemm.py, profiles/r05_microbench.txt).
            # Double precision only: in fp32 the library's K-deep kernel is as fast as the split (29 us)
            part = torch.bmm(g2.reshape(8, rows // 8, out_f).transpose(1, 2), x2.reshape(8, rows // 8, in_f))
            mw.add_(part.sum(0), alpha=self.alpha)
        else:
            torch.addmm(mw, g2.t(), x2, beta=1.0, alpha=self.alpha, out=mw)
        if ob is not None:
            # the bias sums are queued (ODEPetsc._colsum_accum): a cotangent that IS the solver's cotangent buffer -- the last
            # layer of func receives it unchanged -- is rewritten for the next stage before the queue is flushed: copy it
            if self.cot_storage is not None and g2.untyped_storage().data_ptr() == self.cot_storage:
                g2 = g2.clone()
            ode._colsum_accum(g2, self.target[ob: ob + nb], self.alpha)
        return None

    # ------------------------------------------------------------------ the fused kernel's partial sums
    def _make_side_stream(self, ode, dev):
        """The second stream, at the device's lowest priority where it has one: the products are background work, the chain
        of the next stage's backward pass is what the sweep waits for."""
        if self.side_priority:
            try:
                h = ctypes.c_void_p()
                with torch.cuda.device(dev):
                    if ode._lib.pn_stream_create(1, ctypes.byref(h)) == 0 and h.value:
                        self._side_handle = (ode._lib, h)
                        return torch.cuda.ExternalStream(h.value, device=dev)
            except Exception:
                pass
        return torch.cuda.Stream(device=dev)

    def _event(self):
        if len(self.events) < 8:
            self.events.append(torch.cuda.Event())
            return self.events[-1]
        ev = self.events.pop(0)
        self.events.append(ev)
        return ev

    def flush(self, ode, lam=None):
        """The queued pairs of the stage VJP that has just run: one grouped launch (pn_linear_wgrad_group).

        With `side_on` the launch goes to a second stream: nothing in the reverse sweep reads dW or db before the sweep ends, so
        the product of stage i runs BESIDE the backward pass of stage i-1 (its dX GEMMs run one workgroup per CU, its
        elementwise kernels are HBM-bound: the matrix pipes have room) instead of in front of it.  Protocol: the side stream
        waits for an event recorded here (everything the pairs read has been enqueued), the launch follows, a `done` event
        after it; the main stream waits for the done event of the launch BEFORE this one -- so exactly one launch is in flight
        behind the main stream -- and the tensors a launch reads are kept referenced here until the main stream has waited for
        it (an allocation the main stream makes later can then safely reuse their memory, in eager and in captured sweeps).
        What the in-flight launch may still read when this returns: autograd temporaries (referenced), the stage value (the
        trajectory's: read-only during the sweep; recomputed ones: ODEPetsc joins before it recomputes), and the solver's
        cotangent buffer of THIS stage -- the explicit RK sweep writes the next stage's cotangent to the OTHER of two buffers
        (pn_rk_adjoint_step, wbuf2).  `lam`: lambda, when this was the last stage VJP of a reversed step: lambda is rewritten
        next, a launch that reads it as a cotangent is waited for at once."""
        if self.defer is not None:
            if self.pending:
                self.defer.append(self.pending)
                self.pending = []
            return
        older, self.inflight = self.inflight, []
        try:
            if self.pending:
                items, self.pending = self.pending, []
                if not (self.side_on and items[0][0].is_cuda):
                    ode._ops.linear_wgrad_group(items)
                else:
                    dev = items[0][0].device
                    if self.side is None:
                        self.side = self._make_side_stream(ode, dev)
                    main = torch.cuda.current_stream(dev)
                    fork = self._event()
                    fork.record(main)
                    self.side.wait_event(fork)
                    ode._ops.linear_wgrad_group(items, stream=ctypes.c_void_p(self.side.cuda_stream))
                    done = self._event()
                    done.record(self.side)
                    self.inflight.append((done, items))
        finally:
            # the launches of EARLIER stage VJPs (whether or not this one queued anything: an evaluation the structural check left
            # to autograd queues nothing, the cotangent buffers turn all the same)
            for ev, items in older:
                torch.cuda.current_stream(items[0][0].device).wait_event(ev)
        if lam is not None:
            self._join_readers_of(lam)

    def _join_readers_of(self, lam):
        st = lam.untyped_storage().data_ptr()
        if any(g.untyped_storage().data_ptr() == st or x.untyped_storage().data_ptr() == st for _, items in self.inflight for g, x, _, _, _ in items):
            self.join()

    def join(self):
        """The main stream waits for every launch on the side stream; their operands are released."""
        if self.inflight:
            main = torch.cuda.current_stream(self.inflight[0][1][0][0].device)
            for ev, _ in self.inflight:
                main.wait_event(ev)
            self.inflight = []

    def finish(self, ode, target):
        """mu slices of `target` += the partial sums of the sweep (then zero)."""
        self.flush(ode)
        self.join()
        for mid, st in self.partials.items():
            if not st[2]:
                continue
            ow, nw, ob, nb = self.slots[mid]
            pw, pb = st[0], st[1]
            mu_w = target[ow: ow + nw]
            mu_b = target[ob: ob + nb] if (ob is not None and pb is not None) else None
            M = self.shapes[mid][0]
            N = self.shapes[mid][1]
            if mu_w.data_ptr() % 16 == 0 and mu_w.dtype == pw.dtype:
                ode._ops.linear_wgrad_finish(M, N, pw, pb, mu_w, mu_b)
            else:                                   # an unaligned slice of mu: the same sums, in the same order, through torch
                acc = pw.view(8, M * N)
                tot = acc[0].clone()
                for k in range(1, 8):
                    tot += acc[k]
                mu_w.add_(tot.to(mu_w.dtype))
                pw.zero_()
                if mu_b is not None:
                    accb = pb.view(-1, M)              # 8 K ranges x the tile columns that shared the sums
                    totb = accb[0].clone()
                    for k in range(1, accb.shape[0]):
                        totb += accb[k]
                    mu_b.add_(totb.to(mu_b.dtype))
                    pb.zero_()
            st[2] = False

    def reset(self):
        self.pending = []
        self.join()
        for st in self.partials.values():
            if st[2]:
                st[0].zero_()
                if st[1] is not None:
                    st[1].zero_()
                st[2] = False

    # ------------------------------------------------------------------ what _vjp calls
    def expand(self, grads_rest, n_params):
        """autograd's gradients of the parameters it still differentiates, placed in a list over ALL parameters."""
        full = [None] * n_params
        for k, g in zip(self.rest, grads_rest):
            full[k] = g
        return full

    def self_check(self, ode, out, y, wrt_all, cotangent):
        """First VJP of a solver: the parameter gradients of the handled layers by autograd and by the hooks, on the same
        tape.  Returns True when they agree (the tape is kept: retain_graph)."""
        self.checked = True
        handled_params = tuple(wrt_all[k] for k in self.handled)
        saved = (self.alpha, self.target)
        try:
            self.muted = True                                     # hooks fire, do nothing: autograd's own gradients
            ref = torch.autograd.grad(out, handled_params, cotangent, allow_unused=True, retain_graph=True)
            self.muted = False
            scratch = torch.zeros_like(ode.adj_p_tensor)
            self.alpha, self.target = 1.0, scratch
            torch.autograd.grad(out, (y,), cotangent, allow_unused=True, retain_graph=True)
            ode._flush_bias_accum()                               # (the bias sums are queued: into the scratch buffer, now)
            self.finish(ode, scratch)                             # (... and the fused kernel's partial sums)
            worst = 0.0
            # a parameter whose whole gradient is round-off of a sum that cancels (the bias of a layer in front of a train-mode
            # BatchNorm: the mean subtraction removes it) has no scale of its own: such differences are measured against the largest
            # gradient of the handled parameters instead (anything below 1e-5 of it is noise of either way of summing)
            gmax = max([float(r.abs().max()) for r in ref if r is not None] or [0.0])
            for k, r in zip(self.handled, ref):
                o, l = self.offsets[k], self.lens[k]
                got = scratch[o: o + l]
                want = torch.zeros_like(got) if r is None else r.reshape(-1).to(got.dtype)
                scale = float(want.abs().max())
                err = float((got - want).abs().max())
                if err <= (1e-5 if ode.tensor_dtype == torch.float32 else 3e-11) * gmax:
                    continue
                worst = max(worst, err / scale if scale > 0 else float(got.abs().max()))
            # the two ways run the same GEMM on the same operands: they differ by accumulation round-off (1e-6 / 1e-15 observed);
            # anything above this is a contribution autograd sees and the hooks do not
            tol = 3e-5 if ode.tensor_dtype == torch.float32 else 1e-10
            return worst <= tol, worst
        finally:
            self.muted = False
            self.alpha, self.target = saved
