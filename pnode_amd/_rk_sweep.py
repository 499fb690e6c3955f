"""The explicit Runge-Kutta sweeps behind ``ODEPetsc``: what PETSc's ``TSStep_RK`` / ``TSAdjointStep_RK`` / ``TSTrajectoryGet`` do
between two callbacks into Python, as a mixin.  ``ODEPetsc`` itself (pnode_amd/petsc_adjoint.py)
keeps the reference-shaped surface -- setupTS / odeint / petsc_adjointsolve / odeint_adjoint, pa.py:366-900 -- and calls in here
where the reference calls ``ts.solve`` / ``ts.adjointSolve``."""
import contextlib
import ctypes
import warnings

import torch
import torch.nn as nn  # noqa: F401

from . import _lib, options
from ._lib import PnError, check  # noqa: F401


class RKSweep(object):
    def _func_with_grad(self, t, y, which="EX"):
        """f(t, y) recorded by autograd; returns (output, parameter tensors to differentiate
        with respect to).  While a hipGraph is being captured the parameters are replaced by
        fresh detached aliases (same storage): the real parameters' AccumulateGrad nodes live
        on the stream of the enclosing autograd graph and a gradient edge to them would make
        autograd synchronise the capture stream with that stream."""
        fn, params, names = ((self.funcIM, self._paramsI, self._pnamesI) if which == "IM"
                             else (self.funcEX, self._paramsE, self._pnamesE))
        lin = self._lin if (which == "EX" and self._lin is not None and self._lin.active) else None
        capturing = self.device.type == "cuda" and params and torch.cuda.is_current_stream_capturing()
        seen = tuple(p.detach().requires_grad_(True) for p in params) if capturing else params
        if lin is not None:
            lin.begin()                # func's Linear layers hook their outputs: dW / db are accumulated by the engine
        out = None
        try:
            if capturing:
                out = torch.func.functional_call(fn, dict(zip(names, seen)), (t, y))
            else:
                out = fn(t, y)
        finally:
            if lin is not None and out is None:
                lin.abort()
        # the structural check of THIS evaluation (pnode_amd/_lineargrad.py): a handled weight or bias that func also used
        # outside its layer's call leaves the whole evaluation to autograd, as the reference does with every evaluation
        if lin is None or not lin.end(out, [seen[k] for k in lin.handled]):
            return out, seen
        return out, tuple(seen[k] for k in lin.rest)

    def _call_func(self, t, y_flat, tape=None, slot=None):
        """evalRHSFunction (pa.py:393-412): K = f(t, Y); no copy of the result.  With `tape`
        (a list) the evaluation is recorded by autograd and (input, output) is appended.  `slot`: the stage index, given by
        the callers whose evaluations an adaptive sweep replays from per-evaluation hipGraphs (pnode_amd/_stagegraphs.py)."""
        if slot is not None and self._sg is not None:
            return self._sg.evaluate(self, slot, t, y_flat, tape)
        y = self._shaped(y_flat)
        if tape is not None:
            with torch.enable_grad():
                y = y.detach().requires_grad_(True)
                k, wrt = self._func_with_grad(t, y)
            tape.append((y, k, wrt))
        else:
            k = self.funcEX(t, y)
        if k.dtype != self.tensor_dtype or k.device != self.device or k.numel() != self.n:
            raise ValueError("func must return a tensor with the state's shape, dtype and device")
        if not k.is_contiguous():
            k = k.contiguous()
        if k.untyped_storage().data_ptr() == y_flat.untyped_storage().data_ptr():
            k = k.clone()      # func returned (a view of) its input; the input buffer is recycled
        self.nfe_forward += 1
        return k.detach().reshape(-1)

    def _rk_step(self, t, h, u, K0, unew, stage_dest, want_err, tapes=None, t_first=None):
        """One explicit RK step attempt from the flat state `u` (TSStep_RK's body).

        `t_first`: time at which the first stage derivative is evaluated when it is not handed in
        (see `_first_stage_time`).

        stage_dest(i) -> flat buffer for stage value Y_i, 1 <= i < s (FSAL: Y_{s-1} is `unew`).
        Returns the stage derivatives K (K[s-1] is the FSAL derivative of the next step).
        `tapes` (list of s entries, filled here) receives the autograd tape of each stage.
        """
        ops, s, A, b = self._ops, self._s, self._A, self._b
        if self._native:
            return self._rk_step_native(t, h, u, K0, unew, stage_dest, want_err, tapes, t_first)
        plan = self._stage_plan(h)
        K = [None] * s
        for i in range(s):
            if i == 0:
                y = u
            else:
                y = unew if (self._fsal and i == s - 1) else stage_dest(i)
                idx, coef = plan[i]
                ops.rk_stage(y, u, [K[j] for j in idx], coef)
            if i == 0 and K0 is not None:
                K[0] = K0
            elif tapes is not None:
                rec = []
                K[i] = self._call_func(t + self._c[i] * h, y, rec, slot=i)
                tapes[i] = rec[0]
            else:
                K[i] = self._call_func(t_first if (i == 0 and t_first is not None) else t + self._c[i] * h, y, slot=i)
        if want_err:
            idx = [j for j in range(s) if self._e[j] != 0.0 or (not self._fsal and b[j] != 0.0)]
            ops.combine_wrms(None if self._fsal else unew, unew if self._fsal else u, [K[j] for j in idx],
                             [h * b[j] for j in idx], [h * self._e[j] for j in idx], self._atol, self._rtol)
        elif not self._fsal:
            idx, coef = plan[s]
            ops.rk_stage(unew, u, [K[j] for j in idx], coef)
        return K

    # ---- the C++ step loops (include/pnode_amd.h section 3a) and their two callbacks
    def _make_callbacks(self):
        import weakref
        ref = weakref.ref(self)

        def stage_cb(user, i, t):
            o = ref()
            try:
                tens, tapes, K = o._cbs
                if tapes is not None:
                    rec = []
                    k = o._call_func(t, tens[i], rec, slot=i)
                    tapes[i] = rec[0]
                else:
                    k = o._call_func(t, tens[i], slot=i)
                K[i] = k                                # keeps the derivative alive; the loop gets its address
                return k.data_ptr()
            except BaseException as exc:                # (an exception must not propagate through the C frame)
                o._cb_exc = exc
                return 0

        def vjp_cb(user, i, t, cot_in_w, scale):
            o = ref()
            try:
                Y, tapes, dlam, t0 = o._rcbs
                if i == 0 and t0 is not None:
                    t = t0                              # first-same-as-last: where the forward sweep evaluated this stage
                w = o.adj_u_flat if not cot_in_w else o._buf("w_a" if cot_in_w == 1 else "w_b")
                gy, gp = o._vjp(t, Y[i], w, tapes[i] if tapes else None, alpha=scale, last=(i == 0), slot=i)
                if tapes:
                    tapes[i] = None                     # release the stage's activations as soon as they are used
                if gy is not None and gy.data_ptr() == w.data_ptr():
                    gy = gy.clone()                     # f returned its cotangent unchanged (identity-like f)
                dlam[i] = gy
                if o.np > 0 and any(g is not None for g in gp):
                    if o._accum_mode == "stage":
                        o._ops.param_accum(o.adj_p_tensor, scale, gp, o._poff, o._plen)
                    else:
                        o._pend_a.append(scale)
                        o._pend_g.append(gp)
                return 0 if gy is None else gy.data_ptr()
            except BaseException as exc:
                o._cb_exc = exc
                return -1

        self._stage_cb_c = _lib.STAGE_CB(stage_cb)
        self._vjp_cb_c = _lib.VJP_CB(vjp_cb)
        self._ystage = (ctypes.c_void_p * _lib.PN_MAX_STAGES)()
        self._kout = (ctypes.c_void_p * _lib.PN_MAX_STAGES)()
        self._ytens = [None] * _lib.PN_MAX_STAGES
        self._cb_exc = None

    def _raise_from_loop(self, rc):
        exc, self._cb_exc = self._cb_exc, None
        if exc is not None:
            raise exc
        check(rc)

    def _rk_step_native(self, t, h, u, K0, unew, stage_dest, want_err, tapes, t_first):
        ops, s = self._ops, self._s
        if getattr(self, "_stage_cb_c", None) is None:
            self._make_callbacks()
        ys, tens = self._ystage, self._ytens
        tens[0] = u
        for i in range(1, s):
            y = unew if (self._fsal and i == s - 1) else stage_dest(i)
            tens[i] = y
            ys[i] = y.data_ptr()
        K = [None] * s
        K[0] = K0
        self._cbs = (tens, tapes, K)
        work, res = ops.wrms_buffers() if want_err else (None, None)
        rc = self._lib.pn_rk_attempt(ops.stream(), ops.code, self.n, self._ts, ops.vec_ops, t, h, u.data_ptr(), unew.data_ptr(), ys,
                                     None if K0 is None else K0.data_ptr(),
                                     1 if (K0 is None and t_first is not None) else 0, 0.0 if t_first is None else t_first,
                                     self._stage_cb_c, None, 1 if want_err else 0, work, res, self._kout)
        self._cbs = None
        if rc:
            self._raise_from_loop(rc)
        return K

    def _stage_plan(self, h):
        """Per stage i: (indices j of the non-zero a_ij, the coefficients h*a_ij as a C array); entry s: the same for
        the weights b.  Built once per step size (fixed-step sweeps use one; adaptive ones a few dozen)."""
        plan = self._plans.get(h)
        if plan is None:
            if len(self._plans) >= 256:
                self._plans.clear()
            mk = getattr(self._ops, "dbl", list)
            s, A, b = self._s, self._A, self._b
            plan = []
            for i in range(s):
                idx = [j for j in range(i) if A[i][j] != 0.0]
                plan.append((idx, mk([h * A[i][j] for j in idx])))
            idx = [j for j in range(s) if b[j] != 0.0]
            plan.append((idx, mk([h * b[j] for j in idx])))
            self._plans[h] = plan
        return plan

    def _first_stage_time(self, k):
        """Time argument of f for the first stage of step k when it is RE-computed from a checkpoint.
        In the original sweep of a first-same-as-last tableau that derivative was the previous step's
        last stage, evaluated at t_{k-1} + c_{s-1} h_{k-1}; that is not t_k to the last bit (5dp's
        c_{s-1} is the row sum 0.9999999999999998; matched output times are set exactly), and a
        time-dependent f would see it.  Same expression here, so that every checkpoint mode
        reproduces the store-all sweep bit for bit."""
        if self._fsal and k > 0 and not self._ref_defaults:
            tp, hp = self._step_info(k - 1)
            return tp + self._c[self._s - 1] * hp
        # (-pn_reference_defaults: PETSc's TSTrajectory restarts the stepper at a restored checkpoint, so the first stage is
        # re-evaluated -- and its Jacobian taken, TSAdjointStep_RK -- at t_k; for a time-dependent f under a first-same-as-last
        # tableau that is the forward sweep's derivative only up to the last bits of the time argument, as with the reference)
        return None

    def _stages_of(self, step):
        """Stage values Y_0..Y_{s_eff-1} of `step` as flat tensors: read from the store-all
        trajectory, or recomputed from the nearest kept state (TSTrajectoryGet)."""
        traj, ops = self._traj, self._ops
        s_eff = self._s_eff
        if self._tmode == _lib.PN_TRAJ_ALL:
            fs, fl, _ = traj.rev_plan(step)
            v = traj.view(fl)
            return [v[i] for i in range(s_eff)]
        fs, fl, stores = traj.rev_plan(step)
        keep = self._budget_stages
        if keep and fs == step and traj.stage_step.get(fl) == step:
            v = traj.view(fl)              # the checkpoint of this very step holds its stage values
            return [v[i] for i in range(s_eff)]
        slot_view = traj.view(fl)
        cur, cur_slot = slot_view[0], fl
        K_fsal = None
        k = fs
        pp = 0
        while k < step:                   # re-advance k -> k+1, keeping what the plan asks for
            tn, h = self._step_info(k)
            if (k + 1) in stores:
                nxt_slot = stores[k + 1]
                nxt_view = traj.claim(nxt_slot)
                nxt = nxt_view[0]
                traj.stage_step.pop(nxt_slot, None)
            else:
                pp ^= 1
                nxt_slot, nxt_view = -1, None
                nxt = self._buf("r_a" if pp else "r_b")
            if keep and cur_slot >= 0:
                dest = lambda i, c=slot_view: c[i]          # stage values of step k go behind its checkpoint
            else:
                dest = lambda i: self._buf("y_scratch")
            K = self._rk_step(tn, h, cur, K_fsal, nxt, dest, False,
                              t_first=self._first_stage_time(k) if K_fsal is None else None)
            if keep and cur_slot >= 0:
                traj.stage_step[cur_slot] = k
                traj.seal(cur_slot)                  # (disk tier) the checkpoint now carries its stage values
            if nxt_slot >= 0 and not keep:
                traj.seal(nxt_slot)                  # (disk tier) a new state-only checkpoint is complete
            K_fsal = K[self._s - 1] if self._fsal else None
            cur, cur_slot, slot_view = nxt, nxt_slot, nxt_view
            k += 1
        # stage values of `step` itself (its own derivatives K_0..K_{s_eff-2} are needed)
        tn, h = self._step_info(step)
        Y = [cur]
        K = [K_fsal]
        # The derivatives K_0..K_{s_eff-2} evaluated here are evaluations of f at exactly the points the stage VJPs of this
        # step differentiate f at: unless tapes are switched off (-pn_trajectory_retain_graph 0, -pn_reference_defaults) they
        # are recorded by autograd and the VJPs of those stages run their backward half only -- (s_eff - 1) evaluations of f
        # fewer per reversed step in every mode that recomputes stage values (solution-only, checkpoint budgets); same bits.
        rt = [None] * self._s if self._retain_graph != 0 else None
        self._rtapes = rt
        for i in range(1, s_eff):
            if K[i - 1] is None:
                t_eval = self._first_stage_time(step) if i == 1 else None
                tt = tn + self._c[i - 1] * h if t_eval is None else t_eval
                if rt is not None:
                    rec = []
                    K[i - 1] = self._call_func(tt, Y[i - 1], rec, slot=i - 1)
                    rt[i - 1] = rec[0]
                else:
                    K[i - 1] = self._call_func(tt, Y[i - 1], slot=i - 1)
            y = self._buf("ys%d" % i)
            idx = [j for j in range(i) if self._A[i][j] != 0.0]
            ops.rk_stage(y, cur, [K[j] for j in idx], [h * self._A[i][j] for j in idx])
            Y.append(y)
            K.append(None)
        if self._ref_defaults:
            # -pn_reference_defaults: PETSc's TSTrajectory re-runs the WHOLE step (TSStep) to get the stage values back,
            # i.e. it also evaluates the stage derivatives nothing in the reverse sweep reads.  Evaluated here too (and
            # dropped), so that a func that counts its calls sees s evaluations per recomputed step.
            if K[s_eff - 1] is None:
                K[s_eff - 1] = self._call_func(tn + self._c[s_eff - 1] * h, Y[s_eff - 1])
            if self._fsal:
                i = self._s - 1
                y = self._buf("y_scratch")
                idx = [j for j in range(i) if self._A[i][j] != 0.0]
                ops.rk_stage(y, cur, [K[j] for j in idx], [h * self._A[i][j] for j in idx])
                self._call_func(tn + self._c[i] * h, y)
        return Y

    def _vjp(self, t, y_flat, w_flat, tape=None, which="EX", alpha=None, last=False, slot=None):
        """RHSJacShell.multTranspose + RHSJacPShell.multTranspose (pa.py:52-82, 341-363): one
        forward of f with grad and one backward with the cotangent `w`; returns
        (J^T w as a flat tensor or None, list of parameter cotangents over ALL parameters of that f).  With a `tape`
        (input, output) recorded in the forward sweep only the backward runs.  `alpha`: the scale the caller will give the
        parameter cotangents when it adds them to mu -- the explicit RK path passes it so that the sensitivities of func's
        nn.Linear layers can be accumulated during the backward pass itself (pnode_amd/_lineargrad.py); those entries of
        the returned list are then None.  `last`: this is the last stage VJP of a reversed step (lambda is rewritten next)."""
        if slot is not None and self._sg is not None and which == "EX" and alpha is not None:
            return self._sg.vjp(self, slot, t, y_flat, w_flat, tape, alpha, last)
        lin = self._lin if (which == "EX" and self._lin is not None) else None
        all_params = self._paramsI if which == "IM" else self._paramsE
        # (a tape that belongs to a captured evaluation is differentiated again at every replay of its backward unit)
        keep = True if (self._unit_capture and tape is not None) else None
        if tape is not None:
            y, out, wrt = tape
        else:
            self.nfe_backward += 1
        with torch.enable_grad() if tape is None else contextlib.nullcontext():
            if tape is None:
                y = self._shaped(y_flat).detach().requires_grad_(True)
                out, wrt = self._func_with_grad(t, y, which)
            cot = self._shaped(w_flat).view(out.shape)
            hooked = lin is not None and len(wrt) != len(all_params)      # this evaluation left the Linear layers to the hooks
            if hooked and not lin.disabled and alpha is not None:
                capturing = self.device.type == "cuda" and torch.cuda.is_current_stream_capturing()
                if not lin.checked and not capturing:
                    ok, worst = lin.self_check(self, out, y, all_params, cot)
                    if not ok:
                        lin.disabled = True
                        lin.why = "its result differed from autograd's at the self-check (relative %.1e)" % worst
                        lin.remove_hooks_only()
                        warnings.warn("pnode_amd: the engine-side accumulation of the nn.Linear layers' parameter sensitivities is "
                                      "switched off for this solver: its result differs from autograd's (relative %.1e) -- a "
                                      "weight or bias of such a layer is also used somewhere else in func.  Results are autograd's; "
                                      "-pn_linear_param_grads 0 silences this." % worst, RuntimeWarning)
                if not lin.disabled:
                    # The hooks add to mu during this backward pass (or queue bias sums behind what is queued already); parameter
                    # cotangents autograd formed for EARLIER stages -- evaluations the structural check left to it -- may still
                    # wait in the batched queue: add them first, so that mu sees every stage's contribution in the order of
                    # the stages whatever -pn_param_accum says (the same bits in every mode, also for a func that mixes the two
                    # kinds of evaluation in one solve)
                    if self._pend_g and self._pend_mixed:
                        self._flush_param_accum()
                    lin.alpha, lin.target = float(alpha), self.adj_p_tensor
                    lin.cot_storage = w_flat.untyped_storage().data_ptr()
                    try:
                        grads = torch.autograd.grad(out, (y,) + wrt, cot, allow_unused=True, retain_graph=keep)
                    finally:
                        lin.alpha = None
                    grads = (grads[0],) + tuple(lin.expand(grads[1:], len(all_params)))
                    hooked = None
            if hooked:
                # evaluated with the hooks on, differentiated without them (the self-check failed, or a caller that adds the
                # parameter cotangents itself): autograd differentiates with respect to every parameter
                if self.device.type == "cuda" and torch.cuda.is_current_stream_capturing():
                    raise PnError("pnode_amd: a stage evaluation recorded for the engine-side Linear accumulation cannot be "
                                  "differentiated by autograd alone inside a hipGraph capture")
                lin.muted = True
                try:
                    grads = torch.autograd.grad(out, (y,) + tuple(all_params), cot, allow_unused=True, retain_graph=keep)
                finally:
                    lin.muted = False
            elif hooked is False:
                grads = torch.autograd.grad(out, (y,) + wrt, cot, allow_unused=True, retain_graph=keep)
                if lin is not None:
                    self._pend_mixed = True            # what the caller queues now holds cotangents of parameters the hooks also serve
                    if self._pend_bias:
                        self._flush_bias_accum()       # (the mirror case: bias sums queued by the hooks of earlier stages go first)
            if lin is not None:
                # the stage's queued (cotangent, input) pairs: one grouped launch of the fused kernel, beside the next stage; the
                # launches of earlier stages are waited for (every stage VJP, also one autograd did alone: the buffers turn)
                lin.flush(self, lam=self.adj_u_flat if last else None)
        gy = grads[0]
        if gy is not None:
            if gy.dtype != self.tensor_dtype:
                gy = gy.to(self.tensor_dtype)
            gy = gy.contiguous().reshape(-1)
        gp = []
        dt = self.tensor_dtype
        # Deferred accumulation (-pn_param_accum batch|step) reads these gradients launches later, after the
        # cotangent buffer (w_a, or lambda itself for a folded stage) has been rewritten in place.  Autograd hands
        # the cotangent, or ANY view of it, straight through for f = ... + p, cat([z[:2] + b1, ...]), stack((.. + p0, ..)):
        # every gradient that shares the cotangent's storage is copied, whatever its size.
        wst = None if self._accum_mode == "stage" else w_flat.untyped_storage().data_ptr()
        for g in grads[1:]:
            if g is not None:
                if g.dtype != dt or not g.is_contiguous():
                    g = g.to(dt).contiguous()
                if wst is not None and g.untyped_storage().data_ptr() == wst:
                    g = g.clone()
            gp.append(g)
        return gy, gp

    def _adjoint_steps(self, nsteps, forcing):
        """TSAdjointSolve over `nsteps` steps, newest first (TSAdjointStep_RK per step), then
        add `forcing` (dL/dy at the span point reached; pa.py:938) fused into the last update.

        Per step [t_n, t_n+H] with stage values Y_i, incoming lambda and mu:
            for i = s-1 .. 0:   w_i = H*(b_i*lambda + sum_{j>i} a_ji*dlam_j)
                                (dlam_i, dmu_i) = VJP of f at Y_i with cotangent w_i
            mu     <- mu + sum_i dmu_i      (stages added in the order s-1..0: one multi-tensor launch per
                                             stage, or per time step with -pn_param_accum step; same rounding)
            lambda <- lambda + sum_i dlam_i
        (the scale PETSc applies after MatMultTranspose is applied to the cotangent instead).
        A stage whose cotangent is a pure multiple of lambda -- the last non-trivial stage of
        every tableau -- is differentiated with lambda itself and the scalar is folded into
        the coefficients of everything that consumes its result: no kernel, no extra vector."""
        if self._theta is not None:
            return self._theta.adjoint_steps(nsteps, forcing)
        ops, s_eff, A, b = self._ops, self._s_eff, self._A, self._b
        lam = self.adj_u_flat
        if nsteps == 0 and forcing is not None:
            ops.adj_accum(lam, lam, [], [], forcing)
        # two cotangent buffers in turn while the weight-sensitivity products of a stage run beside the next stage on a second
        # stream (pnode_amd/_lineargrad.py): the product of stage i reads stage i's cotangent while stage i-1's is written
        two_w = self._lin is not None and self._lin.side_on
        for r in range(nsteps):
            step = self._rev_next
            tn, H = self._step_info(step)
            if self._lin is not None and self._tmode != _lib.PN_TRAJ_ALL:
                self._lin.join()             # a product still running may read stage values the recomputation below rewrites
            Y = self._stages_of(step)
            tapes = self._tapes.pop(step, None) if self._tapes else None
            if tapes is None and self._rtapes is not None:
                tapes = self._rtapes             # recorded while the stage values were recomputed (_stages_of)
            self._rtapes = None
            dlam = [None] * self._s          # raw VJP results
            if self._native:
                if getattr(self, "_vjp_cb_c", None) is None:
                    self._make_callbacks()
                self._rcbs = (Y, tapes, dlam, self._first_stage_time(step))
                fo = forcing if r == nsteps - 1 else None
                rc = self._lib.pn_rk_adjoint_step(ops.stream(), ops.code, self.n, self._ts, ops.vec_ops, tn, H, lam.data_ptr(),
                                                  self._buf("w_a").data_ptr(), self._buf("w_b").data_ptr() if two_w else None,
                                                  self._vjp_cb_c, None,
                                                  None if fo is None else fo.data_ptr())
                self._rcbs = None
                if rc:
                    self._raise_from_loop(rc)
                if self._pend_g and (self._accum_mode == "step" or self._sg is not None or len(self._pend_g) + s_eff > self._accum_cap):
                    self._flush_param_accum()          # (per-evaluation graphs: the cotangents sit in static outputs)
                elif self._pend_bias and self._accum_mode == "step":
                    self._flush_bias_accum()
                self._traj.rev_done(step)
                self._rev_next = step - 1
                continue
            scale = [1.0] * self._s          # true dlam_i = scale[i] * dlam[i]
            pend_a, pend_g = self._pend_a, self._pend_g      # parameter gradients waiting to be added to mu
            nw = 0
            for i in range(s_eff - 1, -1, -1):
                js = [j for j in range(i + 1, s_eff) if A[j][i] != 0.0 and dlam[j] is not None]
                if b[i] == 0.0 and not js:
                    continue                   # structurally zero cotangent
                if not js:
                    w, scale[i] = lam, H * b[i]
                else:
                    w = self._buf("w_b" if (two_w and nw % 2) else "w_a")
                    nw += 1
                    ops.adj_theta(w, lam if b[i] != 0.0 else None, H * b[i],
                                  [dlam[j] for j in js], [H * A[j][i] * scale[j] for j in js])
                # (stage 0 of a first-same-as-last tableau was evaluated at the previous step's last stage time, which is
                # t_n only to the last bit: the VJP differentiates f THERE, with and without a tape -- the exact discrete
                # adjoint, the same bits in every checkpoint mode for a time-dependent f; PETSc passes t_n)
                t0 = self._first_stage_time(step) if i == 0 else None
                gy, gp = self._vjp(tn + self._c[i] * H if t0 is None else t0, Y[i], w, tapes[i] if tapes else None, alpha=scale[i], last=(i == 0), slot=i)
                if tapes:
                    tapes[i] = None            # release the stage's activations as soon as they are used
                if gy is not None and gy.data_ptr() == w.data_ptr():
                    gy = gy.clone()            # f returned its cotangent unchanged (identity-like f)
                dlam[i] = gy
                if self.np > 0 and any(g is not None for g in gp):
                    if self._accum_mode == "stage":
                        ops.param_accum(self.adj_p_tensor, scale[i], gp, self._poff, self._plen)
                    else:
                        pend_a.append(scale[i])
                        pend_g.append(gp)
            if pend_g and (self._accum_mode == "step" or self._sg is not None or len(pend_g) + s_eff > self._accum_cap):
                self._flush_param_accum()      # mu += sum_j scale_j * dmu_j, oldest first: one launch
            elif self._pend_bias and self._accum_mode == "step":
                self._flush_bias_accum()
            idx = [i for i in range(s_eff) if dlam[i] is not None]
            ops.adj_accum(lam, lam, [dlam[i] for i in idx], [scale[i] for i in idx],
                          forcing if r == nsteps - 1 else None)
            self._traj.rev_done(step)
            self._rev_next = step - 1

    def _add_param_grads(self, alpha, gp, first=0, stable=True, cotangent=None):
        """mu[parameters first .. first+len(gp)) += alpha * gp for the implicit / IMEX steppers: one launch per call with
        -pn_param_accum stage, else queued for the batched launch of _flush_param_accum (same order, same rounding).
        `stable` False: the gradients sit in buffers that are rewritten before a deferred launch would read them (the
        outputs of a replayed graph): what is queued is added first, then these, at once.  `cotangent`: the buffer the
        gradients were computed FROM when they did not come through _vjp -- a gradient that is a view of it is copied."""
        if not any(g is not None for g in gp):
            return
        n_all = len(self._poff)
        full = first == 0 and len(gp) == n_all
        if self._accum_mode == "stage" or not stable:
            self._flush_param_accum()
            if full:
                off, ln = self._poff, self._plen
            elif first == 0:
                off, ln = self._poffI, self._plenI
            else:
                off, ln = self._poffE, self._plenE
            self._ops.param_accum(self.adj_p_tensor, alpha, list(gp), off, ln)
            return
        if cotangent is not None:
            st = cotangent.untyped_storage().data_ptr()
            gp = [g.clone() if (g is not None and g.untyped_storage().data_ptr() == st) else g for g in gp]
        self._pend_a.append(alpha)
        self._pend_g.append(list(gp) if full else [None] * first + list(gp) + [None] * (n_all - first - len(gp)))
        if len(self._pend_g) >= self._accum_cap:
            self._flush_param_accum()

    def _colsum_accum(self, g2, mu_slice, alpha):
        """mu_slice += alpha * column sums of g2 (rows x cols): the sensitivity of a bias.  Queued like the parameter
        cotangents of autograd (-pn_param_accum batch|step: the cotangent tensors stay alive, at most 1 GiB of them, and up to
        32 are summed by ONE pn_colsum_accum_multi pass; stage: at once) -- same bits whatever the grouping."""
        g2 = g2.contiguous()
        self._pend_bias.append((g2, mu_slice, float(alpha)))
        self._pend_bias_bytes += g2.numel() * g2.element_size()
        if self._accum_mode == "stage" or len(self._pend_bias) >= 32 or self._pend_bias_bytes >= (1 << 30):
            self._flush_bias_accum()

    def _flush_bias_accum(self):
        if self._pend_bias:
            fn = getattr(self._ops, "colsum_accum_multi", None)
            if fn is not None and self._pend_bias[0][0].device.type == "cuda":
                fn(self._pend_bias)
            else:                                        # the CPU test stand-in: same order, double sums
                for g2, mu_slice, alpha in self._pend_bias:
                    mu_slice.add_(g2.double().sum(0).to(mu_slice.dtype), alpha=alpha)
            self._pend_bias = []
            self._pend_bias_bytes = 0

    @property
    def linear_param_grads(self):
        """How the parameter sensitivities of func's nn.Linear layers are formed: "engine (N parameters)" or "autograd (why)"."""
        lin = self._lin
        if lin is None:
            return "autograd (no eligible nn.Linear layer, a theta stepper, or -pn_linear_param_grads 0)"
        if lin.disabled:
            return "autograd (%s)" % lin.why
        note = "; fused dW + db MFMA kernel on %d layers" % len(lin.partials) if lin.partials else ""
        if lin.n_autograd:
            # the structural check (LinearParamGrads.end): evaluations in which a handled parameter was also used outside its layer
            note += "; %d of %d recorded evaluations of func left to autograd (a handled weight or bias is also used outside its layer there, " \
                    "or a layer runs in another precision than its parameters -- autocast)" % (lin.n_autograd, lin.n_autograd + lin.n_clean)
        return "engine (%d of %d parameter tensors%s)" % (len(lin.handled), len(self._paramsE), note)

    def _setup_linear_grads(self):
        """(Re)install the engine-side accumulation of func's nn.Linear layers (pnode_amd/_lineargrad.py): explicit RK path
        only; -pn_linear_param_grads auto|gemm|0 (not a PETSc option)."""
        opt = str(options.get_all().get("pn_linear_param_grads", "auto"))
        gemm = opt == "gemm"             # the library GEMM + pn_colsum_accum_multi for every layer (no fused MFMA kernel)
        on = opt in ("auto", "gemm") or options.truthy(opt, False)
        # explicit RK (func), and ARKIMEX's explicitly treated func2: the only grad-enabled evaluations of that function are the
        # solver's own taped stage evaluations and stage VJPs.  Not the theta methods: their Newton-Krylov solves differentiate
        # func in ways of their own (double VJPs, captured linearisations)
        side = str(options.get_all().get("pn_linear_side_stream", "0"))
        # -pn_linear_side_stream 1 | same-priority (default 0): the products on a second stream beside the next stage's backward
        # pass -- the explicit RK sweep only (its cotangent buffers are doubled for it); ARKIMEX's stage vectors are rewritten on a
        # schedule of their own.  Measured at BASELINE's target configuration (profiles/r06_side_stream.txt): +1.5 % time-steps/s,
        # the same bits; the dX GEMMs of the next stage take 36 us beside the product against 19.4 alone -- the two share the
        # matrix pipes -- and every kernel's own duration stops being a statement about that kernel, so it is not the default.
        side_on = (side == "same-priority" or options.truthy(side, False)) and self._stepper_kind is None and self.device.type == "cuda"
        # -pn_linear_wgrad_exact 1: fp32 states on the fp32 matrix instruction (a k-ordered fmaf chain) instead of the default --
        # operands split exactly into three bf16 terms, six bf16 MFMA products per fp32 product, fp32 accumulation (csrc/pn_linear.hip)
        exact = options.truthy(options.get_all().get("pn_linear_wgrad_exact", 0), False)
        # -pn_linear_wgrad_tile64 1: the split-bf16 form always on 64 x 64 workgroup tiles (by default a launch that fills whole rounds
        # of the chip takes 128 x 128 ones; the same bits) -- for comparisons
        tile64 = options.truthy(options.get_all().get("pn_linear_wgrad_tile64", 0), False) or side_on
        # (the second stream implies the small tiles: a 128 x 128 workgroup holds a CU's whole LDS, nothing runs beside it -- measured
        # 869 against 958 time-steps/s; with 64 x 64 tiles the second stream is worth +1.5 %, profiles/r06_side_stream.txt)
        if hasattr(self._ops, "wgrad_flags"):
            self._ops.wgrad_flags = (_lib.PN_WGRAD_EXACT_FP32 if exact else 0) | (_lib.PN_WGRAD_TILE_64 if tile64 else 0)
        sig = (id(self.funcEX), on, self._stepper_kind in (None, "imex"), tuple(id(p) for p in self._paramsE), gemm, side_on, exact, tile64)
        if sig == self._lin_sig:
            return
        self._lin_sig = sig
        if self._lin is not None:
            self._lin.remove()
            self._lin = None
        if on and sig[2] and self._paramsE:
            from ._lineargrad import LinearParamGrads
            lin = LinearParamGrads(self)
            lin.fused = not gemm
            lin.side_on = side_on
            lin.side_priority = side != "same-priority"
            if lin.install(self.funcEX, self._paramsE, self._poffE if self._stepper_kind == "imex" else self._poff):
                self._lin = lin

    def _flush_param_accum(self):
        self._flush_bias_accum()
        self._pend_mixed = False
        if self._pend_g:
            self._ops.param_accum_multi(self.adj_p_tensor, self._pend_a, self._pend_g, self._poff, self._plen)
            del self._pend_a[:], self._pend_g[:]

    def _begin_adjoint(self, seed):
        if self._traj is None:
            raise RuntimeError("adjoint requested but no trajectory was saved "
                               "(setupTS(enable_adjoint=True) and a differentiable input are required)")
        if self.adj_u_tensor is None:
            self.adj_u_tensor = self._ops.empty(self._npad)
        if self.adj_p_tensor is None or self.adj_p_tensor.numel() != self.np:
            self.adj_p_tensor = self._ops.empty(max(self.np, 1))[: self.np]
        self.adj_u_flat = self.adj_u_tensor
        self._ops.copy(self.adj_u_flat, seed)
        self.adj_p_tensor.zero_()
        self._traj.begin_reverse()
        self._rev_next = self._nsteps - 1
        self._pend_a, self._pend_g = [], []
        self._pend_mixed = False             # the queue holds autograd's cotangents of parameters the Linear hooks also serve
        self._pend_bias, self._pend_bias_bytes = [], 0
        if self._lin is not None:
            self._lin.reset()              # (partial sums a sweep that raised may have left behind)
        # pending stage results are kept alive until they are added: bound them to 1 GiB
        esize = 4 if self.tensor_dtype == torch.float32 else 8
        self._accum_cap = max(1, min(self._accum_sources, (1 << 30) // max(self.np * esize, 1)))

    def _reverse_sweep(self, g, T):
        """The body of OdeintAdjointMethod.backward (pa.py:924-944) on the (T, n) cotangent."""
        with self._device_guard():
            if self._trace:
                torch.cuda.nvtx.range_push("pnode_amd.reverse_sweep")
                try:
                    return self._reverse_sweep_impl(g, T)
                finally:
                    torch.cuda.nvtx.range_pop()
            return self._reverse_sweep_impl(g, T)

    def _reverse_sweep_impl(self, g, T):
        self._begin_adjoint(g[T - 1])
        if T == 1:
            self._adjoint_steps(self._nsteps, None)
        for i in range(T - 1, 0, -1):
            self._adjoint_steps(self.cur_sol_steps[i], g[i - 1])
        self._flush_param_accum()
        self._finish_linear_accum()

    def _finish_linear_accum(self):
        """End of a reverse sweep: the partial sums of the fused Linear-sensitivity kernel go into mu (pn_linear_wgrad_finish)."""
        if self._lin is not None:
            self._lin.finish(self, self.adj_p_tensor)

    # ------------------------------------------------------------------ reverse (pa.py:871-890)
    def _step_info(self, k):
        if self._log_override is not None:
            return self._log_override[k]
        tt, hh = ctypes.c_double(), ctypes.c_double()
        check(self._lib.pn_ts_step_log(self._ts, k, ctypes.byref(tt), ctypes.byref(hh)))
        return tt.value, hh.value
