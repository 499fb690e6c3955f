"""hipGraph capture and replay of whole sweeps (``-pn_graph_capture``), kept apart from the solver it serves.

The reference launches everything every time (``ts.solve`` / ``ts.adjointSolve``, pa.py:829, 878).  On an MI355X a fixed-step
sweep at the sizes of BASELINE.json's configurations is bound by the host's dispatch of func and of autograd, not by the
device: after ``GRAPH_WARMUP_CALLS`` eager calls with the same signature the forward sweep and the reverse sweep are
captured as two hipGraphs and replayed.  ``ODEPetsc`` (pnode_amd/petsc_adjoint.py) inherits this class; everything here talks
to the solver through ``_odeint`` / ``_reverse_sweep`` and its host-side state.

What makes the default (``auto``) safe:

* the capture key holds everything a captured sweep bakes in: output times, step size(s), trajectory and tape modes, the
  autocast state, and a snapshot of func's Python side (pnode_amd/_funcguard.py) taken when the call starts -- a scalar
  attribute that was annealed, a flag that was toggled, a tensor attribute or buffer that was re-assigned between two calls
  selects another capture (or eager launches until one exists), never a stale one;
* tensor attributes that are re-assigned before every call (``func.x0 = x0.clone()``, examples-sinode/grand/src/
  base_classes.py:58-60) are recognised after two calls and fed to the captured sweep through a static copy, like ``y0``;
* plain call counters of func keep counting (their increments are learnt in the warm-up calls); any other change of
  Python-side state DURING a sweep keeps the solver eager;
* the call that captures a sweep also runs it eagerly, the first replay has to reproduce it, and replay must not be slower;
* every ``-pn_graph_revalidate`` N-th replayed call (default 100) is run eagerly as well and compared again: state no guard
  can see (closures, globals) is caught there, with a warning, and the solver goes back to eager launches.
"""
import contextlib
import gc
import time
import warnings

import torch

from . import _funcguard as fg


class _GraphEntry(object):
    """One captured (forward sweep, reverse sweep) pair of hipGraphs and the host-side state
    that belongs to it."""

    def __init__(self):
        self.calls = 0
        self.pool = None
        self.g_f = self.g_b = None
        self.static_y0 = self.static_gout = self.sol = None
        self.host = None
        self.pending_eager = None        # auto mode: host state of the eager forward sweep of the validating call
        self.revalidating = False        # ... which is a periodic re-validation of an existing pair, not its capture
        self.time_replay = False
        self.t_replay_f = self.t_replay_b = self.t_eager_f = None
        self.nfe_f = (0, 0)              # (nfe_forward, nfe_backward) one forward replay stands for
        self.nfe_b = (0, 0)              # ... one reverse replay
        self.deltas_f = self.deltas_b = None     # auto mode: increments of func's call counters per forward / reverse sweep
        self.orphans = 0                 # auto mode: validating forward sweeps that no backward followed
        self.eager_only = False
        self.replays = 0                 # replayed calls since the pair was validated last
        self.bitwise = False             # the first replays reproduced their eager twins bit for bit: re-validation demands the same
        self.replay_diff = 0.0
        self.static_in = None            # [(module index, name, static tensor)]: re-assigned tensor attributes fed by copy
        self.key_base = self.snap = None
        # adaptive sweeps (pnode_amd/_stagegraphs.py): per-evaluation graphs instead of a captured pair
        self.sg = None                   # StageGraphs
        self.state = "validate"          # validate (units are captured) -> validate2 (replays only, timed) -> graph | eager
        self.t_eager_b = None
        self.tries = 0
        self.grew = False                # the last validating call still captured evaluations of new kinds


class SweepGraphs(object):
    """Mixin of ODEPetsc: the launch mode of its sweeps."""

    # "thread_local": only the capturing thread is held to capture-safe API calls, so helper
    # threads of the process (RCCL watchdog, data loaders) cannot invalidate a capture
    GRAPH_CAPTURE_MODE = "thread_local"
    GRAPH_WARMUP_CALLS = 2
    GRAPH_CACHE_ENTRIES = 4
    GRAPH_MAX_EVICTIONS = 8        # auto: captured pairs dropped from the cache before the solver gives up on replay
    GRAPH_REVALIDATE_EVERY = 100   # default of -pn_graph_revalidate
    AUTO_THETA = True              # auto mode also covers the capturable IMEX / theta configuration (direct solves, ksponly)
    AUTO_MIN_GAIN = 1.02           # replay time must stay below this multiple of the eager sweeps' wall time (2 %: timing noise;
                                   # a solve the GPU bounds either way is replayed -- it frees the host)

    def _init_sweep_graphs(self):
        self._graphs = {}
        self._graph_mode = False
        self._graph_warned = False
        self._graph_status = "eager (setupTS not called)"
        self._auto_veto = None
        self._last_fp = None
        self._prev_fp = None
        self._counters = set()         # {(module index, attribute)}: func's call counters, learnt in the warm-up calls
        self._volatile = set()         # {(module index, attribute)}: tensor attributes re-assigned between calls
        self._revalidate_every = self.GRAPH_REVALIDATE_EVERY
        self._evicted_captured = 0
        self._log_override = None      # [(t_n, h_n)]: the step log the reverse sweep reads instead of the stepper's own

    def _reset_sweep_graphs(self, new_func=False):
        """Captured sweeps belong to the func, scheme, shapes and modes they were captured with."""
        self._graphs = {}
        if new_func:
            self._auto_veto = None     # a new func gets a new chance to be captured
            self._counters, self._volatile, self._prev_fp = set(), set(), None
            self._evicted_captured = 0

    @property
    def graphs_captured(self):
        """True once a (forward, reverse) hipGraph pair exists for some call signature."""
        return any((e.g_f is not None and e.g_b is not None) or (e.sg is not None and e.state == "graph") for e in self._graphs.values())

    @property
    def graph_status(self):
        """How the sweeps of this solver are launched, and why: "graph", "graph(auto)", or "eager (...)"."""
        return self._graph_status

    # ------------------------------------------------------------------ the capture key
    def _graph_entry(self, y0, t, need):
        """Cache entry for this call, or None when the call must run eagerly."""
        if not self._graph_mode or self.device.type != "cuda" or self._traj_disk:
            return None                              # (file I/O of the disk tier is host work inside the sweeps)
        if self._adaptive and (self._theta is not None or not self._native or self._sharded()):
            # adaptive sweeps: per-evaluation graphs (_stagegraphs.py), explicit RK only; not for a batch sharded over ranks -- the
            # ranks meet in the error norm's all-reduce at every attempt, and a rank that validates (two sweeps) beside one that
            # does not would leave them waiting for each other
            if self._theta is None and self._sharded():
                self._graph_status = "eager (adaptive steps over a batch that is sharded across ranks)"
            return None
        auto = self._graph_mode == 2
        if auto and self._auto_veto:
            return None
        if auto and self._theta is not None and not self.AUTO_THETA:
            return None                              # (switch: auto for the explicit RK sweeps only)
        if torch.cuda.is_current_stream_capturing():
            return None                              # the caller is capturing a graph of its own: be part of it
        if self._theta is not None and not (hasattr(self._theta, "capturable") and self._theta.capturable()):
            return None                              # Newton/GMRES iterations synchronise with the host
        import pnode_amd
        from . import petsc_adjoint as pa
        e = None
        if pnode_amd.GRAPH_REPLAY_SAFE and not self._lib.pn_prof_is_enabled():
            try:
                e = self._graph_lookup(y0, t, need)
            except Exception as exc:                  # func holds something the guard cannot describe: never guess
                why = "func's Python side could not be inspected (%s: %s)" % (type(exc).__name__, str(exc).split("\n")[0][:160])
                if auto:
                    self._veto_auto(why, warn=True)
                else:
                    self._graph_mode = False
                    self._graph_status = "eager (%s)" % why
                    warnings.warn("pnode_amd: -pn_graph_capture switched off for this solver: %s" % why, RuntimeWarning)
                return None
            if e.g_f is None and e.calls >= self.GRAPH_WARMUP_CALLS:
                from . import _graphcheck             # once per process and device, before the first capture (~1 s: not
                if not _graphcheck.replay_is_sound(self.device):      # spent on solvers that never get that far)
                    pnode_amd.GRAPH_REPLAY_SAFE = False
        if not pnode_amd.GRAPH_REPLAY_SAFE and auto:
            self._veto_auto("the HIP runtime was initialised before pnode_amd was imported (or DEBUG_CLR_GRAPH_PACKET_CAPTURE "
                            "is not 0, or the replay self-test failed): import pnode_amd (or pnode) before the first CUDA call",
                            warn=not pa._ENV_WARNED[0])
            pa._ENV_WARNED[0] = True
            return None
        if not pnode_amd.GRAPH_REPLAY_SAFE:
            if not self._graph_warned:
                self._graph_warned = True
                warnings.warn("pnode_amd: -pn_graph_capture ignored (eager launches instead): the HIP runtime was "
                              "initialised before pnode_amd was imported, or DEBUG_CLR_GRAPH_PACKET_CAPTURE is not 0, "
                              "or the replay self-test failed; hipGraph replays of PyTorch reductions are unreliable "
                              "on this ROCm in that state. Import pnode_amd (or pnode) before the first CUDA call, or "
                              "export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0.", RuntimeWarning)
            return None
        return e                                     # (None while per-dispatch events are on: they cannot be attached to graph nodes)

    def _funcs(self):
        return (self.funcEX, self.funcIM)

    def _py_fingerprint(self):
        """Snapshot of func's Python side (pnode_amd/_funcguard.py): structure and train/eval flags, scalar attributes,
        tensor attributes.  A captured sweep replays kernels, not Python."""
        return fg.snapshot(self._funcs())

    def _graph_lookup(self, y0, t, need):
        # what a captured sweep bakes in: the times, the step, the modes, the autocast state of the calling context -- and
        # func's Python-side configuration as it is NOW (module flags, scalar attributes, storage of every tensor it holds)
        snap = self._last_fp = self._py_fingerprint()
        moved = fg.moved_tensors(self._prev_fp, snap, self.device)
        self._prev_fp = snap
        if moved and not set(moved) <= self._volatile:
            self._volatile |= set(moved)             # re-assigned before every call: from now on fed through a static copy
            self._rekey_all(drop_captured=True)      # (a sweep captured before reads the old ADDRESS: it cannot be fed)
        base = (tuple(t.detach().cpu().to(torch.float64).tolist()), repr(self.step_size), bool(need),
                tuple(y0.shape), y0.dtype, self._traj_mode, self._max_cps, self._budget_stages, self._retain_graph,
                (torch.is_autocast_enabled(), torch.get_autocast_gpu_dtype()) if torch.is_autocast_enabled() else None)
        key = (base, fg.key_of(snap, self._counters, self._volatile))
        e = self._graphs.get(key)
        if e is None:
            if len(self._graphs) >= self.GRAPH_CACHE_ENTRIES:
                old = self._graphs.pop(next(iter(self._graphs)))
                if old.g_f is not None and self._graph_mode == 2:
                    # captured sweeps that are dropped before they pay: more configurations of func / output times alternate
                    # than the cache keeps -- every re-capture costs an eager twin
                    self._evicted_captured += 1
                    if self._evicted_captured >= self.GRAPH_MAX_EVICTIONS:
                        self._veto_auto("more than %d configurations (output times, step sizes, func's Python-side state) keep "
                                        "alternating: captured sweeps were dropped %d times before they paid"
                                        % (self.GRAPH_CACHE_ENTRIES, self._evicted_captured), warn=True)
            e = self._graphs[key] = _GraphEntry()
            e.key_base, e.snap = base, snap
        return e

    def _rekey_all(self, drop_captured=False):
        """The set of counters or of fed tensors changed: every entry gets the key its own snapshot has under the new sets
        (entries that are still warming up keep their call counts).  `drop_captured`: entries that hold captured sweeps are
        dropped instead -- their graphs read the address a now-fed tensor had at capture."""
        entries = list(self._graphs.values())
        self._graphs = {}
        for v in entries:
            if drop_captured and v.g_f is not None:
                continue
            k = (v.key_base, fg.key_of(v.snap, self._counters, self._volatile))
            w = self._graphs.get(k)
            if w is None or (v.g_f is not None, v.calls) > (w.g_f is not None, w.calls):
                self._graphs[k] = v

    def _counter_deltas(self, before, after):
        """[(module, name, increment)] of the integer attributes a sweep moved, or None (_funcguard.counter_deltas)."""
        d = fg.counter_deltas(before, after)
        if d is None:
            return None
        mods = fg.modules_of(self._funcs())
        return [(mods[mi], k, x, mi) for mi, k, x in d]

    @staticmethod
    def _bump(deltas, sign=1):
        for m, k, d, _ in deltas or ():
            setattr(m, k, getattr(m, k) + sign * d)

    def _learn_counters(self, e, deltas):
        new = {(mi, k) for _, k, _, mi in deltas} - self._counters
        if new:
            self._counters |= new
            self._rekey_all()                        # the keys must not hold the counters' values

    def _note_side_effects(self, e, which, before, veto=True):
        """Eager warm-up calls (`veto` False: the explicit -pn_graph_capture 1, which only learns which attributes are
        counters so that they stay out of the capture key, and does no bookkeeping at replay).  auto mode: a func that counts its calls (``self.nfe += 1``: the NFE of the reference's ODE
        blocks, examples-pnode/models/sqnxt_PETSc.py, spiral_unstable.py:326-347) is capturable as long as the counting
        is all it does on the Python side and every sweep counts the same: the increments are remembered and applied at
        every replay.  Anything else that changes func's Python side during a sweep -- a float, a flag, a tensor attribute
        that is created or re-assigned (ffjord-pnode/lib/layers/odefunc.py:341-364 samples ``self._e`` inside the first
        evaluation) -- keeps the solver eager."""
        after = self._py_fingerprint()
        d = [] if after == before else self._counter_deltas(before, after)
        prev = getattr(e, "deltas_" + which)
        if not veto:
            if d:
                self._learn_counters(e, d)
            return True
        if d is None or (prev is not None and [(id(m), k, x) for m, k, x, _ in prev] != [(id(m), k, x) for m, k, x, _ in d]):
            self._veto_auto("func changes Python-side state during a sweep in a way that is not a plain call counter "
                            "(%s): replays would freeze it" % (fg.describe_change(before, after, fg.modules_of(self._funcs()))
                                                               if d is None else "a counter that does not count the same every call"))
            return False
        setattr(e, "deltas_" + which, d)
        self._learn_counters(e, d)
        return True

    def _func_buffers(self):
        out, seen = [], set()
        for f in self._funcs():
            if isinstance(f, torch.nn.Module) and id(f) not in seen:
                seen.add(id(f))
                out += [b for b in f.buffers() if b.device == self.device]
        return out

    def _veto_auto(self, why, warn=False):
        """auto mode only: this solver stays with eager launches -- same results; ``graph_status`` says why, and a
        RuntimeWarning (once per solver object) when the reason is something the user may want to fix."""
        self._auto_veto = why
        self._graph_status = "eager (auto: %s)" % why
        self._graphs = {}
        if warn and not self._graph_warned:
            self._graph_warned = True
            warnings.warn("pnode_amd: the sweeps of this solver are launched eagerly instead of being replayed from hipGraphs "
                          "(-pn_graph_capture auto): %s.  Results are the same; -pn_graph_capture 0 silences this." % why,
                          RuntimeWarning, stacklevel=2)

    def _give_up_on_graphs(self, which, exc):
        """Capturing a sweep failed (func synchronises with the host, allocates with the wrong stream, ...):
        say so once and launch eagerly from now on -- same results."""
        self._graph_mode = False
        self._graphs = {}
        self._graph_status = "eager (capturing the %s sweep failed: %s)" % (which, type(exc).__name__)
        gc.collect()
        torch.cuda.synchronize(self.device)
        warnings.warn("pnode_amd: -pn_graph_capture switched off for this solver: capturing the %s sweep failed (%s: %s). "
                      "func must not synchronise with the host or depend on host-side data." % (which, type(exc).__name__, exc),
                      RuntimeWarning)

    def _host_state(self):
        return (self._nsteps, list(self.cur_sol_steps), self.cur_sol_index, self.sol_times, self._traj, self._tapes,
                getattr(self._theta, "traj", None), self._tmode)

    def _set_host_state(self, st):
        self._nsteps, self.cur_sol_steps, self.cur_sol_index, self.sol_times, self._traj, self._tapes, ttraj, self._tmode = st
        self.cur_sol_steps = list(self.cur_sol_steps)
        if self._theta is not None:
            self._theta.traj = ttraj

    # ------------------------------------------------------------------ tensor attributes fed through static copies
    def _make_static_inputs(self, e):
        mods = fg.modules_of(self._funcs())
        e.static_in = []
        for mi, name in sorted(self._volatile):
            if mi < len(mods):
                h = fg.holder_of(mods[mi], name)
                cur = h.get(name)
                if isinstance(cur, torch.Tensor) and cur.device == self.device:
                    e.static_in.append((mi, name, torch.empty_like(cur)))

    def _feed_static_inputs(self, e):
        if e.static_in:
            mods = fg.modules_of(self._funcs())
            for mi, name, st in e.static_in:
                # (.data: the retained stage tapes of the captured forward sweep may hold the static copy as a saved tensor --
                # ``out * self.mask`` -- and a write autograd can see would invalidate them, see _restore)
                st.data.copy_(fg.holder_of(mods[mi], name)[name].detach())

    @contextlib.contextmanager
    def _static_inputs_swapped_in(self, e):
        """While a sweep is captured func reads the static copies; the user's tensors are put back whatever happens."""
        if not e.static_in:
            yield
            return
        mods = fg.modules_of(self._funcs())
        saved = []
        try:
            for mi, name, st in e.static_in:
                h = fg.holder_of(mods[mi], name)
                saved.append((h, name, h[name]))
                h[name] = st
            yield
        finally:
            for h, name, v in saved:
                h[name] = v

    def _static_inputs_untouched(self, e):
        """A tensor attribute that func WRITES cannot be fed by copy (the user's tensor would never see the update)."""
        if e.static_in:
            mods = fg.modules_of(self._funcs())
            for mi, name, st in e.static_in:
                if not torch.equal(st, fg.holder_of(mods[mi], name)[name]):
                    return False
        return True

    # ------------------------------------------------------------------ capture / replay
    def _timed_replay(self, graph):
        torch.cuda.synchronize(self.device)
        t0 = time.perf_counter()
        graph.replay()
        torch.cuda.synchronize(self.device)
        return time.perf_counter() - t0

    def _graph_forward(self, e, y0, t, need):
        if self._theta is not None:
            self._theta.graph_prepare(y0)            # Jacobian + LU factors for the current parameters
        if e.g_f is None:
            t = t.detach().cpu()             # no device->host copy inside the captured region
            gc.collect()
            torch.cuda.synchronize(self.device)
            e.static_y0 = torch.empty_like(y0, memory_format=torch.contiguous_format)
            self._make_static_inputs(e)
            self._feed_static_inputs(e)
            e.pool = torch.cuda.graph_pool_handle()
            g = torch.cuda.CUDAGraph()
            nf, nb = self.nfe_forward, self.nfe_backward
            with self._static_inputs_swapped_in(e):
                with torch.cuda.graph(g, pool=e.pool, capture_error_mode=self.GRAPH_CAPTURE_MODE):
                    e.sol = self._odeint(e.static_y0, t, need)
            e.g_f = g
            e.host = self._host_state()
            e.nfe_f = (self.nfe_forward - nf, self.nfe_backward - nb)     # the Python of this call has counted already
        else:
            self.nfe_forward += e.nfe_f[0]       # a replay runs no Python: count what the captured sweep evaluates
            self.nfe_backward += e.nfe_f[1]
            self._bump(e.deltas_f)               # ... and func's own call counters (auto mode)
        self._set_host_state(e.host)
        e.static_y0.copy_(y0.detach())
        self._feed_static_inputs(e)
        if e.time_replay:
            e.t_replay_f = self._timed_replay(e.g_f)
        else:
            e.g_f.replay()
        return e.sol.clone()

    def _graph_backward(self, e, g, T):
        if e.g_b is None:
            gc.collect()
            torch.cuda.synchronize(self.device)
            e.static_gout = torch.zeros_like(g)
            gb = torch.cuda.CUDAGraph()
            nf, nb = self.nfe_forward, self.nfe_backward
            with self._static_inputs_swapped_in(e):
                with torch.cuda.graph(gb, pool=e.pool, capture_error_mode=self.GRAPH_CAPTURE_MODE):
                    self._reverse_sweep(e.static_gout, T)
            e.g_b = gb
            e.nfe_b = (self.nfe_forward - nf, self.nfe_backward - nb)
            self._graph_status = "graph(auto)" if self._graph_mode == 2 else "graph"
        else:
            self.nfe_forward += e.nfe_b[0]
            self.nfe_backward += e.nfe_b[1]
            self._bump(e.deltas_b)
        e.static_gout.copy_(g)
        self._feed_static_inputs(e)
        if e.time_replay:
            e.t_replay_b = self._timed_replay(e.g_b)
        else:
            e.g_b.replay()

    # -- auto mode: the call that captures a sweep also runs it eagerly, and the first replay has to reproduce the eager
    # result bit for bit (and must not be slower).  func's buffers (BatchNorm statistics) are put back in between, so that
    # the call leaves them updated once, as every other call does.  The same twin run re-validates an existing pair.
    @staticmethod
    def _restore(bufs, values):
        """Put func's buffers back WITHOUT touching autograd's version counters: the stage tapes of the eager sweep hold
        these tensors (BatchNorm's running statistics are inputs of its forward), and an in-place write autograd can see
        would invalidate them."""
        for b, v in zip(bufs, values):
            b.data.copy_(v)

    def _reproduces(self, got, want):
        """Does the first replay reproduce the eager sweep?  Bit for bit -- or, for a func whose kernels are not
        bit-reproducible from one launch to the next (MIOpen's weight gradients use atomics), to that noise: what the
        check guards against (a replayed reduction that drops partial sums, pnode_amd/__init__.py; func state that went
        stale) is wrong in its leading digits.  Returns (ok, relative difference)."""
        worst = 0.0
        for a, b in zip(got, want):
            if torch.equal(a, b):
                continue
            if not (torch.isfinite(a).all() and torch.isfinite(b).all()):
                return False, float("inf")
            d = float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-300))
            worst = max(worst, d)
        tol = 1e-4 if self.tensor_dtype == torch.float32 else 1e-9
        return worst <= tol, worst

    def _retimed(self, graph, first, limit, bufs, b0, b_after):
        """A replay that looks slower than the eager launches on ONE wall-clock sample is timed twice more (func's buffers
        put back before each), and the minimum counts: one noisy sample must not decide the launch mode of a whole run."""
        best, again = first, False
        for _ in range(2):
            if best <= limit:
                break
            self._restore(bufs, b0)
            best, again = min(best, self._timed_replay(graph)), True
        if again:
            self._restore(bufs, b_after)
        return best

    _STALE = ("func no longer computes what was captured -- Python-side state that the capture guard cannot see (a closure, a "
              "global, an attribute of a foreign object) changed between calls; up to %d earlier replayed calls may have used "
              "the stale value.  Keep such state in attributes of func's modules, or pass -pn_graph_capture 0")

    def _restore_counters(self, before):
        """Set func's call counters back to what `before` (a snapshot) recorded."""
        if self._counters:
            mods = fg.modules_of(self._funcs())
            for mi, k, v in before[1]:
                if (mi, k) in self._counters and mi < len(mods) and type(v) is int:
                    setattr(mods[mi], k, v)

    def _auto_capture_forward(self, e, y0, t, need, revalidate=False):
        """Returns (answer, entry or None)."""
        bufs = self._func_buffers()
        b0 = [b.clone() for b in bufs]
        fp0 = self._last_fp
        torch.cuda.synchronize(self.device)
        t0 = time.perf_counter()
        ans_e = self._odeint(y0, t, need)
        torch.cuda.synchronize(self.device)
        e.t_eager_f = time.perf_counter() - t0
        host_e = self._host_state()
        counts = (self.nfe_forward, self.nfe_backward)
        b1 = [b.clone() for b in bufs]
        self._restore(bufs, b0)
        why, broken = None, False
        try:
            e.time_replay = True
            ans_g = self._graph_forward(e, y0, t, need)
            self._bump(e.deltas_f, -1)               # func's counters moved twice: the eager sweep, and the capturing pass (or,
                                                     # when the forward graph exists already, the increment a replay applies)
            ok, diff = self._reproduces((ans_g,), (ans_e,))
            if revalidate and e.bitwise and diff > 0.0:
                ok = False                           # the pair reproduced its eager twin bit for bit when it was captured:
            else:                                    # any difference now is a change, not launch-to-launch noise
                e.replay_diff = diff
            if not ok and revalidate:
                why = self._STALE % e.replays + " (forward sweep: relative difference %.1e)" % diff
            elif not ok:
                why = "the first replay of the forward sweep does not reproduce the eager sweep (relative difference %.1e)" % diff
            elif not self._static_inputs_untouched(e):
                why = "func writes to a tensor attribute that is re-assigned between calls"
            elif not revalidate:
                limit = self.AUTO_MIN_GAIN * e.t_eager_f
                if e.t_replay_f > limit:
                    e.t_replay_f = self._retimed(e.g_f, e.t_replay_f, limit, bufs, b0, [b.clone() for b in bufs])
                if e.t_replay_f > limit:
                    why = "replaying the forward sweep is not faster than launching it (%.3g ms vs %.3g ms)" % (1e3 * e.t_replay_f, 1e3 * e.t_eager_f)
        except Exception as exc:                     # func cannot be captured (host synchronisation, ...)
            why, broken = "capturing the forward sweep failed (%s: %s)" % (type(exc).__name__, str(exc).split("\n")[0][:200]), True
            gc.collect()
            torch.cuda.synchronize(self.device)
        if why is not None:
            self._veto_auto(why, warn="not faster" not in why)
            if broken:
                # the aborted capture restarted the stepper's state machine (its step log is what the reverse sweep reads):
                # run the sweep again, eagerly, from the buffers and the call counters this call started with
                host_e = None
                self._restore_counters(fp0)
                ans_e = self._odeint(y0, t, need)
            else:
                self._restore(bufs, b1)
                self._set_host_state(host_e)
            self.nfe_forward, self.nfe_backward = counts
            return ans_e, None
        self.nfe_forward, self.nfe_backward = counts
        if need and (e.g_b is None or revalidate):
            e.pending_eager = host_e                 # the eager trajectory: the reverse sweep is validated against it
            e.revalidating = revalidate
        else:
            e.time_replay = False
            e.replays = 0
            if not revalidate:
                e.bitwise = e.replay_diff == 0.0
            self._graph_status = "graph(auto)"
        return ans_g, e

    def _auto_capture_backward(self, e, g, T):
        """Eager reverse sweep on the eager trajectory, then capture (or, re-validating, replay) on the graph's; the results
        (adj_u_flat, adj_p_tensor) are the eager sweep's bits either way."""
        host_g, host_e = e.host, e.pending_eager
        revalidate, e.revalidating = e.revalidating, False
        e.pending_eager = None
        bufs = self._func_buffers()
        b0 = [b.clone() for b in bufs]
        self._set_host_state(host_e)
        torch.cuda.synchronize(self.device)
        fp0 = self._py_fingerprint() if e.deltas_b is None else None
        t0 = time.perf_counter()
        self._reverse_sweep(g, T)
        torch.cuda.synchronize(self.device)
        t_eager = time.perf_counter() - t0
        why = None
        if fp0 is not None:                         # (no warm-up call had a backward: learn the counters' increments here)
            fp1 = self._py_fingerprint()
            d = [] if fp1 == fp0 else self._counter_deltas(fp0, fp1)
            if d is None:                           # ... and what is not a counter vetoes, as _note_side_effects does
                why = ("func changes Python-side state during a sweep in a way that is not a plain call counter (%s): replays "
                       "would freeze it" % fg.describe_change(fp0, fp1, fg.modules_of(self._funcs())))
            else:
                e.deltas_b = d
                self._learn_counters(e, d)
        n = self.n                                  # (the buffer is padded to 64 elements; the padding is never written)
        adj_u, adj_p = self.adj_u_flat[:n].clone(), self.adj_p_tensor.clone()
        counts = (self.nfe_forward, self.nfe_backward)
        if why is not None:
            e.time_replay = False
            self._veto_auto(why)
            return
        b1 = [b.clone() for b in bufs]
        self._restore(bufs, b0)
        host_e = None
        self._set_host_state(host_g)
        try:
            self._graph_backward(e, g, T)
            self._bump(e.deltas_b, -1)
            ok, diff = self._reproduces((self.adj_u_flat[:n], self.adj_p_tensor), (adj_u, adj_p))
            if revalidate and e.bitwise and diff > 0.0:
                ok = False
            else:
                e.replay_diff = max(getattr(e, "replay_diff", 0.0), diff)
            if not ok and revalidate:
                why = self._STALE % e.replays + " (reverse sweep: relative difference %.1e)" % diff
            elif not ok:
                why = "the first replay of the reverse sweep does not reproduce the eager sweep (relative difference %.1e)" % diff
            elif not self._static_inputs_untouched(e):
                why = "func writes to a tensor attribute that is re-assigned between calls"
            elif not revalidate:
                limit = self.AUTO_MIN_GAIN * (e.t_eager_f + t_eager) - e.t_replay_f
                if e.t_replay_b > limit:
                    e.t_replay_b = self._retimed(e.g_b, e.t_replay_b, limit, bufs, b0, [b.clone() for b in bufs])
                if e.t_replay_b > limit:
                    why = ("replaying the sweeps is not faster than launching them (%.3g ms vs %.3g ms)"
                           % (1e3 * (e.t_replay_f + e.t_replay_b), 1e3 * (e.t_eager_f + t_eager)))
            if why is None and e.replay_diff > 0.0:
                self._graph_status = ("graph(auto; func is not bit-reproducible: first replays within %.0e of the eager sweeps)"
                                      % e.replay_diff)
        except Exception as exc:
            why = "capturing the reverse sweep failed (%s: %s)" % (type(exc).__name__, str(exc).split("\n")[0][:200])
            gc.collect()
            torch.cuda.synchronize(self.device)
        e.time_replay = False
        e.replays = 0
        if why is None and not revalidate:
            e.bitwise = e.replay_diff == 0.0
        self.nfe_forward, self.nfe_backward = counts
        if why is not None:
            self._veto_auto(why, warn="not faster" not in why)
            self.adj_u_flat[:n].copy_(adj_u)
            self.adj_p_tensor.copy_(adj_p)
            self._restore(bufs, b1)

    # ------------------------------------------------------------------ adaptive sweeps: per-evaluation graphs
    def _with_units(self, e, fn):
        """Run a sweep with func's evaluations replayed from `e`'s per-evaluation graphs."""
        self._sg = e.sg
        try:
            return fn()
        finally:
            self._sg = None

    def _step_log(self):
        return [self._step_info(k) for k in range(self._nsteps)]

    _UNITS = "graph(%sper-evaluation hipGraphs: adaptive steps)"
    _QUIET = "outside the fused kernel's shapes"

    def _adaptive_forward(self, e, y0, t, need):
        auto = self._graph_mode == 2
        if e.calls < self.GRAPH_WARMUP_CALLS:
            ans = self._odeint(y0, t, need)
            # (learns which attributes of func are call counters, so that they stay out of the capture key; what an evaluation
            # does to func's Python side is checked per captured evaluation, StageGraphs._capture)
            self._note_side_effects(e, "f", self._last_fp, veto=False)
            e.calls += 1
            return ans, None, e
        if self._volatile or e.state == "eager" or (self._lin is not None and self._lin.side_on):
            # (tensor attributes re-assigned before every call: whole sweeps only; -pn_linear_side_stream: its launches are ordered
            # by events between two streams, which a unit that stands for ONE evaluation cannot carry)
            return self._odeint(y0, t, need), None, None
        orphan = e.pending_eager is not None
        e.pending_eager = None
        if e.sg is None:
            from ._stagegraphs import StageGraphs
            e.sg = StageGraphs(self)
        if auto and e.state in ("validate", "validate2"):
            e.orphans += 1 if orphan else 0
            if e.orphans >= 2:                      # solves with gradients enabled that are never differentiated
                e.state, e.sg = "eager", None
                return self._odeint(y0, t, need), None, None
            return self._adaptive_validate_forward(e, y0, t, need)
        if auto and self._revalidate_every > 0:
            e.replays += 1
            if e.replays >= self._revalidate_every:
                e.state, e.revalidating, e.tries = "validate2", True, 0
                return self._adaptive_validate_forward(e, y0, t, need)
        try:
            ans = self._with_units(e, lambda: self._odeint(y0, t, need))
        except Exception as exc:                    # an evaluation of a kind that was not captured before, and cannot be
            if auto:
                self._veto_auto("capturing an evaluation of func failed (%s: %s)" % (type(exc).__name__, str(exc).split("\n")[0][:200]), warn=True)
            else:
                self._give_up_on_graphs("forward", exc)
            gc.collect()
            torch.cuda.synchronize(self.device)
            return self._odeint(y0, t, need), None, None
        if not auto:
            e.state = "graph"
            self._graph_status = self._UNITS % ""
        e.host = self._host_state()
        return ans, e, None

    def _adaptive_validate_forward(self, e, y0, t, need):
        """The call runs both ways; the eager sweep's answer is what the caller gets unless the replayed one reproduces it."""
        bufs = self._func_buffers()
        b0 = [b.clone() for b in bufs]
        torch.cuda.synchronize(self.device)
        t0 = time.perf_counter()
        ans_e = self._odeint(y0, t, need)
        torch.cuda.synchronize(self.device)
        e.t_eager_f = time.perf_counter() - t0
        host_e, log_e = self._host_state(), self._step_log()
        counts = (self.nfe_forward, self.nfe_backward)
        fp1 = self._py_fingerprint()
        b1 = [b.clone() for b in bufs]
        self._restore(bufs, b0)
        why = None
        captured0 = e.sg.captured
        try:
            torch.cuda.synchronize(self.device)
            t0 = time.perf_counter()
            ans_g = self._with_units(e, lambda: self._odeint(y0, t, need))
            torch.cuda.synchronize(self.device)
            e.t_replay_f = time.perf_counter() - t0
            ok, diff = self._reproduces((ans_g,), (ans_e,))
            if e.revalidating and e.bitwise and diff > 0.0:
                ok = False
            else:
                e.replay_diff = max(e.replay_diff, diff)
            if not ok and e.revalidating:
                why = self._STALE % e.replays + " (forward sweep: relative difference %.1e)" % diff
            elif not ok:
                why = "the replayed evaluations of the forward sweep do not reproduce the eager sweep (relative difference %.1e)" % diff
        except Exception as exc:
            why = "capturing an evaluation of func failed (%s: %s)" % (type(exc).__name__, str(exc).split("\n")[0][:200])
            gc.collect()
            torch.cuda.synchronize(self.device)
        self._restore_counters(fp1)
        self.nfe_forward, self.nfe_backward = counts
        if why is not None:
            self._veto_auto(why, warn=self._QUIET not in why)
            # the replayed sweep rewrote the stepper's step log (what the reverse sweep reads): once more, eagerly
            self._restore(bufs, b0)
            self._restore_counters(self._last_fp)
            ans_e = self._odeint(y0, t, need)
            self.nfe_forward, self.nfe_backward = counts
            return ans_e, None, None
        e.host = self._host_state()
        e.grew = e.sg.captured > captured0
        if need:
            e.pending_eager = (host_e, log_e)
        else:
            self._adaptive_advance(e, e.t_replay_f, e.t_eager_f)
        return ans_g, e, None

    def _adaptive_advance(self, e, t_replay, t_eager):
        """A call whose replayed sweeps reproduced their eager twins: validate -> validate2 -> graph."""
        if e.state == "validate":
            e.state, e.tries = "validate2", 0
            return
        if e.grew:                                   # evaluations of a new kind were still being captured: time the next call
            e.tries += 1
            if e.tries >= 4:
                self._veto_auto("func keeps launching evaluations of new kinds")
            return
        if not e.revalidating and t_replay > self.AUTO_MIN_GAIN * t_eager:
            self._veto_auto("replaying func's evaluations is not faster than launching them (%.3g ms vs %.3g ms)" % (1e3 * t_replay, 1e3 * t_eager))
            return
        if not e.revalidating:
            e.bitwise = e.replay_diff == 0.0
        e.state, e.revalidating, e.replays = "graph", False, 0
        self._graph_status = self._UNITS % "auto; "
        if e.replay_diff > 0.0:
            self._graph_status = (self._UNITS % "auto; ")[:-1] + "; replays within %.0e of the eager sweeps)" % e.replay_diff

    def _adaptive_backward(self, e, g, T):
        if e.pending_eager is None:
            try:
                self._with_units(e, lambda: self._reverse_sweep(g, T))
            except Exception as exc:
                if self._graph_mode == 2:
                    self._veto_auto("capturing a stage VJP failed (%s: %s)" % (type(exc).__name__, str(exc).split("\n")[0][:200]), warn=True)
                else:
                    self._give_up_on_graphs("reverse", exc)
                gc.collect()
                torch.cuda.synchronize(self.device)
                self._reverse_sweep(g, T)            # (from the seed: nothing of the aborted sweep is kept)
            return
        host_g = e.host
        host_e, log_e = e.pending_eager
        e.pending_eager = None
        bufs = self._func_buffers()
        b0 = [b.clone() for b in bufs]
        self._set_host_state(host_e)
        self._log_override = log_e                   # the eager forward sweep's steps (the stepper's own log is the replayed sweep's)
        torch.cuda.synchronize(self.device)
        t0 = time.perf_counter()
        try:
            self._reverse_sweep(g, T)
        finally:
            self._log_override = None
        torch.cuda.synchronize(self.device)
        e.t_eager_b = time.perf_counter() - t0
        n = self.n
        adj_u, adj_p = self.adj_u_flat[:n].clone(), self.adj_p_tensor.clone()
        counts = (self.nfe_forward, self.nfe_backward)
        fp1 = self._py_fingerprint()
        b1 = [b.clone() for b in bufs]
        self._restore(bufs, b0)
        self._set_host_state(host_g)
        why = None
        captured0 = e.sg.captured
        try:
            torch.cuda.synchronize(self.device)
            t0 = time.perf_counter()
            self._with_units(e, lambda: self._reverse_sweep(g, T))
            torch.cuda.synchronize(self.device)
            e.t_replay_b = time.perf_counter() - t0
            ok, diff = self._reproduces((self.adj_u_flat[:n], self.adj_p_tensor), (adj_u, adj_p))
            if e.revalidating and e.bitwise and diff > 0.0:
                ok = False
            else:
                e.replay_diff = max(e.replay_diff, diff)
            if not ok and e.revalidating:
                why = self._STALE % e.replays + " (reverse sweep: relative difference %.1e)" % diff
            elif not ok:
                why = "the replayed stage VJPs of the reverse sweep do not reproduce the eager sweep (relative difference %.1e)" % diff
        except Exception as exc:
            why = "capturing a stage VJP failed (%s: %s)" % (type(exc).__name__, str(exc).split("\n")[0][:200])
            gc.collect()
            torch.cuda.synchronize(self.device)
        self._restore_counters(fp1)
        self.nfe_forward, self.nfe_backward = counts
        if why is not None:
            self._veto_auto(why, warn=self._QUIET not in why)      # (a limit of this path, nothing the user should fix)
            self.adj_u_flat[:n].copy_(adj_u)
            self.adj_p_tensor.copy_(adj_p)
            self._restore(bufs, b1)
            return
        e.grew = e.grew or e.sg.captured > captured0
        self._adaptive_advance(e, e.t_replay_f + e.t_replay_b, e.t_eager_f + e.t_eager_b)

    # ------------------------------------------------------------------ what OdeintAdjointMethod calls
    def _sweep_forward(self, y0, t, need):
        """The forward sweep of one ``odeint_adjoint`` call in whatever launch mode applies.  Returns (states, the graph entry
        the reverse sweep of this call belongs to or None, the entry whose warm-up this call is or None)."""
        e = self._graph_entry(y0, t, need)
        auto = self._graph_mode == 2
        if e is not None and self._adaptive:
            return self._adaptive_forward(e, y0, t, need)
        if e is None or e.calls < self.GRAPH_WARMUP_CALLS:
            ans = self._odeint(y0, t, need)
            warm = None
            if e is not None:
                if self._note_side_effects(e, "f", self._last_fp, veto=auto):
                    warm = e                       # the reverse sweep of this call is watched the same way
                e.calls += 1
            return ans, None, warm
        orphan = e.pending_eager is not None       # the last validating call had no backward
        e.pending_eager, e.revalidating, e.time_replay = None, False, False
        if auto and e.eager_only:
            return self._odeint(y0, t, need), None, None
        if auto and (e.g_f is None or (need and e.g_b is None)):
            # A caller that solves with gradients enabled and never differentiates would pay for two sweeps per call
            # for ever: after two such calls this call signature stays with eager launches.
            e.orphans += 1 if orphan else 0
            if e.orphans >= 2:
                e.eager_only, e.g_f, e.host, e.sol = True, None, None, None
                return self._odeint(y0, t, need), None, None
            ans, e = self._auto_capture_forward(e, y0, t, need)
            return ans, e, None
        if auto and self._revalidate_every > 0:
            e.replays += 1
            if e.replays >= self._revalidate_every:
                ans, e = self._auto_capture_forward(e, y0, t, need, revalidate=True)
                return ans, e, None
        try:
            return self._graph_forward(e, y0, t, need), e, None
        except Exception as exc:
            if e.g_f is not None:
                raise                              # a replay failed: nothing to fall back from
            self._give_up_on_graphs("forward", exc)
            return self._odeint(y0, t, need), None, None

    def _sweep_backward(self, e, warm, g, T):
        """The reverse sweep that belongs to `_sweep_forward`'s (entry, warm entry)."""
        if e is not None and e.sg is not None:
            self._adaptive_backward(e, g, T)
        elif e is not None and e.pending_eager is not None:
            self._auto_capture_backward(e, g, T)
        elif e is not None:
            try:
                self._graph_backward(e, g, T)
            except Exception as exc:
                if e.g_b is not None:
                    raise
                self._give_up_on_graphs("reverse", exc)
                self._reverse_sweep(g, T)               # the replayed forward sweep left its trajectory in place
        else:
            before = self._py_fingerprint() if warm is not None else None
            self._reverse_sweep(g, T)
            if warm is not None and self._graph_mode and not self._auto_veto:
                self._note_side_effects(warm, "b", before, veto=self._graph_mode == 2 and not self._adaptive)
