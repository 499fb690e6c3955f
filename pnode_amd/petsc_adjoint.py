"""Drop-in for ``pnode.petsc_adjoint`` on the explicit-RK path, MI355X-native.

Same surface as the reference (``/root/reference/pnode/petsc_adjoint.py``, "pa.py"):
``ODEPetsc.setupTS / odeint / odeint_adjoint`` and ``OdeintAdjointMethod`` with the same
arguments, argument meaning and error behaviour -- but no PETSc/petsc4py underneath.  The
PETSc TS / TSAdapt / TSAdjoint / TSTrajectory / Vec machinery the reference drives
(pa.py:370, 637-656, 766-775, 812-829, 875-878) is replaced by ``libpnode_amd.so``
(``include/pnode_amd.h``): hand-written gfx950 kernels for all state-vector arithmetic and a
C++ host engine for the stepper state machine and the checkpoint schedule.  What stays in
Python is what is Python in the reference too: the callback shells around the user's
``nn.Module`` (pa.py:393-412 ``evalRHSFunction``, 52-82 ``RHSJacShell.multTranspose``,
341-363 ``RHSJacPShell.multTranspose``) and the autograd entry (pa.py:903-947).

The product path has no CPU fallback: states must live on a HIP device and the shared
library must be present, otherwise an exception is raised.
"""
import contextlib
import ctypes
import sys
import warnings

import torch
import torch.nn as nn

from . import _lib, options
from ._lib import PnError, check  # noqa: F401
from .misc import flat_parameters
from ._rk_sweep import RKSweep
from ._sweepgraphs import SweepGraphs
from ._trajectory import _DiskTrajectory, _Trajectory, _TwoLevelTrajectory  # noqa: F401
from ._vecops import HipVecOps, _KrylovBuffers  # noqa: F401

__all__ = ["ODEPetsc", "OdeintAdjointMethod", "PnError", "PnUnpinnedWarning"]


class PnUnpinnedWarning(RuntimeWarning):
    """Issued once per process and piece when a piece of PETSc's behaviour that is restated here WITHOUT anything
    PETSc-produced to check it against ("parity unpinned") decides the numbers of a solve."""


_UNPINNED_WARNED = set()
_ENV_WARNED = [False]           # the "HIP runtime initialised before import" note of auto graph capture: once per process


def _warn_unpinned(key, what):
    if key in _UNPINNED_WARNED:
        return
    _UNPINNED_WARNED.add(key)
    warnings.warn("pnode_amd: %s -- restated from PETSc's documentation and the literature, not checked against anything "
                  "PETSc produced (no PETSc build and no PETSc-made log exists in the reference): the numbers claim PETSc's "
                  "semantics, not bit-parity with it.  See DESIGN.md section 3 (parity unpinned).  Said once per process."
                  % what, PnUnpinnedWarning, stacklevel=3)



def _mem_now(device):
    """(bytes allocated, bytes reserved) by PyTorch's caching allocator on `device`.  torch.cuda.memory_allocated()
    flattens the whole statistics dictionary in Python (~90 us per call, measured in the eager sweep's profile); the
    nested dictionary underneath costs a tenth of that."""
    try:
        st = torch._C._cuda_memoryStats(device.index if device.index is not None else torch.cuda.current_device())
        return st["allocated_bytes"]["all"]["current"], st["reserved_bytes"]["all"]["current"]
    except Exception:
        return torch.cuda.memory_allocated(device), torch.cuda.memory_reserved(device)


class ODEPetsc(RKSweep, SweepGraphs):
    """Explicit-RK neural-ODE solver with discrete adjoint (drop-in for pa.py:366-900).  How its sweeps are launched
    (eagerly, or replayed from hipGraphs) lives in the SweepGraphs mixin, pnode_amd/_sweepgraphs.py."""

    def __init__(self, backend=None):
        self._lib = _lib.load()
        self._backend_cls = backend if backend is not None else HipVecOps
        self._ts = ctypes.c_void_p(self._lib.pn_ts_create())
        self.n = 0
        self.tensor_size = None
        self.tensor_dtype = None
        self.device = None
        self.adj_u = []
        self.adj_p = []
        self.mass = None
        self.funcIM = None
        self.funcEX = None
        self.flat_params = None
        self.npIM = self.npEX = self.np = None
        self.imex = None
        self.use_dlpack = True
        self.linear_solver = None
        self.matrixfree_jacobian = True
        self.step_size = 0.01
        self.enable_adjoint = True
        self._traj = None
        self._tmode = None
        self._ops = None
        self._nsteps = 0
        self._tapes = None
        self._init_sweep_graphs()
        self._lin = None               # engine-side parameter sensitivities of func's nn.Linear layers (_lineargrad.py)
        self._pend_mixed = False
        self._sg = None                # the per-evaluation hipGraphs the sweep in progress replays (pnode_amd/_stagegraphs.py)
        self._unit_capture = False
        self._lin_sig = None
        self._theta = None
        self._theta_method = None
        self._imex_built = False
        self._paramsI = self._paramsE = self._pnamesI = self._pnamesE = ()
        self._options_sig = None
        self._view = False
        self._rtapes = None
        self._trace = False
        self._pg_enabled = False
        self._pg = None
        self._pg_average = True
        self._pg_global_norm = True
        self.nfe_forward = 0      # f evaluations in forward sweeps (NFE-F of the reference's examples)
        self.nfe_backward = 0     # f evaluations (each followed by a VJP) in reverse sweeps (NFE-B)

    def __del__(self):
        try:
            if self._lin is not None:
                self._lin.remove()             # the forward hooks on func's Linear layers go with the solver
        except Exception:
            pass
        try:
            self._lib.pn_ts_destroy(self._ts)
        except Exception:
            pass

    # ------------------------------------------------------------------ introspection
    @property
    def num_steps(self):
        """Accepted time steps of the last forward solve."""
        return self._nsteps

    @property
    def num_rejections(self):
        """Rejected step attempts of the last forward solve (adaptive schemes)."""
        return int(self._lib.pn_ts_rejections(self._ts))

    def step_log(self):
        """[(t_n, h_n)] of the accepted steps of the last forward solve."""
        return [self._step_info(k) for k in range(self._nsteps)]

    # ------------------------------------------------------------------ multi-GPU
    def setProcessGroup(self, group=None, average=True, global_error_norm=True, enabled=True):
        """Shard the batch of trajectories over the ranks of a ``torch.distributed`` group
        (one process per GPU; backend "nccl" is RCCL over xGMI).  Every rank integrates its own
        contiguous batch shard with replicated parameters; the only data-path exchange is ONE
        all-reduce of the flat parameter-gradient buffer per backward.  ``average`` divides by
        the world size (loss = mean over the global batch).  With an adaptive method
        ``global_error_norm`` additionally all-reduces one scalar (sum of squares) per step
        attempt so that every rank takes the reference's step sequence, whose WRMS norm spans
        the whole flattened batch; turning it off gives per-shard controllers (not parity).
        The reference has no counterpart (it is single-process, pa.py:367 COMM_SELF)."""
        self._pg_enabled = enabled
        self._pg = group
        self._pg_average = average
        self._pg_global_norm = global_error_norm

    def _sharded(self):
        """True when a process group is attached: the collectives are then issued whatever the group's
        size (a one-rank group reduces to itself -- that is how the RCCL path is exercised on one GPU)."""
        import torch.distributed as dist
        return bool(self._pg_enabled and dist.is_available() and dist.is_initialized())

    def _world(self):
        import torch.distributed as dist
        return dist.get_world_size(self._pg) if self._sharded() else 1

    def _global_enorm(self, enorm):
        """sqrt(sum_r n_r*enorm_r^2 / sum_r n_r): the WRMS norm over the global batch."""
        import torch.distributed as dist
        if not self._sharded() or not self._pg_global_norm:
            return enorm
        v = torch.tensor([enorm * enorm * self.n, float(self.n)], dtype=torch.float64, device=self.device)
        dist.all_reduce(v, op=dist.ReduceOp.SUM, group=self._pg)
        s, n = v.tolist()
        return (s / n) ** 0.5

    def _allreduce_adj_p(self):
        import torch.distributed as dist
        if not self._sharded() or self.np == 0:
            return
        w = self._world()
        dist.all_reduce(self.adj_p_tensor, op=dist.ReduceOp.SUM, group=self._pg)
        if self._pg_average and w > 1:
            self.adj_p_tensor.mul_(1.0 / w)

    # ------------------------------------------------------------------ setup (pa.py:534-775)
    def setupTS(self, u_tensor, func, step_size=0.01, enable_adjoint=True, implicit_form=False,
                use_dlpack=True, method="dopri5", mass=None, imex_form=False, func2=None,
                batch_size=1, linear_solver="petsc", fixed_jacobian=False, matrixfree_jacobian=True,
                fixed_jacobian_across_solves=None):
        """Set up the solver before it is used.  Arguments as in pa.py:551-584.

        ``u_tensor`` only donates shape, dtype and device.  ``step_size`` is a float or a list
        (per-step sizes).  ``use_dlpack``, ``batch_size``, ``linear_solver``,
        ``fixed_jacobian`` and ``matrixfree_jacobian`` are accepted and have no effect on the
        explicit path (there is one zero-copy mode and no linear solve); on the IMEX path
        ``linear_solver="torch"`` selects the direct stage solve and ``fixed_jacobian=True`` keeps its
        factors across solves when funcIM has no trainable parameter.  As in the reference,
        ``method`` is applied only when shape, dtype or device change (pa.py:627-656), and
        command-line options (``pnode_amd.init(argv)``) override it (pa.py:775).
        """
        if imex_form and func2 is None:
            raise ValueError("func2 must be provided to enable imex_form=True")
        if fixed_jacobian_across_solves is not None:
            # examples-sinode/KS/KS.py:494,516 still pass the keyword's earlier name
            fixed_jacobian = bool(fixed_jacobian_across_solves)
        from .theta import THETA_METHODS
        theta_method = implicit_form and not imex_form and method in THETA_METHODS
        imex_method = bool(imex_form) and method == "imex"
        if (implicit_form or imex_form or method in ("beuler", "cn", "imex")) and not (theta_method or imex_method):
            raise NotImplementedError(
                "pnode_amd implements the explicit-RK path (euler/midpoint/rk2/bosh3/rk4/dopri5), the implicit "
                "theta methods (implicit_form=True with method 'beuler' or 'cn') and IMEX (imex_form=True with "
                "method 'imex'); other combinations are not built (DESIGN.md section 8)")
        self.imex = imex_form
        self.linear_solver = linear_solver
        self.fixed_jacobian = fixed_jacobian
        self.matrixfree_jacobian = matrixfree_jacobian
        tensor_dtype = u_tensor.dtype
        tensor_size = u_tensor.size()
        device = u_tensor.device
        n = u_tensor.numel()
        if self.funcIM is not func or (imex_form and self.funcEX is not func2) or bool(imex_form) != bool(self._imex_built):
            # pa.py:600-621 -- IMEX: func is treated implicitly, func2 explicitly, flat parameters
            # = [func's, func2's]; otherwise func2 is ignored
            self._imex_built = bool(imex_form)
            self.funcIM = func
            self.funcEX = func2 if imex_form else func
            def trainable(f):
                if not isinstance(f, nn.Module):
                    return (), ()
                return (tuple(p for p in f.parameters() if p.requires_grad),
                        tuple(n for n, p in f.named_parameters() if p.requires_grad))
            self._paramsI, self._pnamesI = trainable(self.funcIM)
            self._paramsE, self._pnamesE = trainable(self.funcEX)
            self._params = self._paramsI + self._paramsE if imex_form else self._paramsE
            # The reference routes dL/dtheta through a cat of parameter views (pa.py:618-620).
            # Here the parameters themselves are inputs of the autograd Function (its `*args`),
            # and flat_params is a detached copy kept for its size/order only: a live cat graph
            # pins the parameters' AccumulateGrad nodes to the stream it was built on, which
            # breaks hipGraph capture of the reverse sweep.
            with torch.no_grad():
                self.flat_params = flat_parameters(self._params)
            self.np = self.flat_params.numel()
            self.npIM = sum(p.numel() for p in self._paramsI) if imex_form else self.np
            self.npEX = self.np - self.npIM if imex_form else self.np
            self._poff, off = [], 0
            for p in self._params:
                self._poff.append(off)
                off += p.numel()
            self._plen = [p.numel() for p in self._params]
            nI = len(self._paramsI) if imex_form else 0
            self._poffI, self._plenI = self._poff[:nI], self._plen[:nI]
            self._poffE, self._plenE = self._poff[nI:], self._plen[nI:]
            self.adj_p_tensor = None
            self._reset_sweep_graphs(new_func=True)
        if self.mass is not mass:
            self.mass = mass
        if tensor_size != self.tensor_size or tensor_dtype != self.tensor_dtype or device != self.device:
            self._ops = self._backend_cls(device, tensor_dtype, n)
            self.tensor_size = tensor_size
            self.tensor_dtype = tensor_dtype
            self.device = device
            self.use_dlpack = use_dlpack
            self.n = n
            self._npad = (n + 63) // 64 * 64
            check(self._lib.pn_ts_set_rk_type(self._ts, self._lib.pn_method_to_rk_type(str(method).encode())))
            self._theta_method = method if theta_method else ("imex" if imex_method else None)   # on rebuild only
            self.adj_u_tensor = None
            self.adj_p_tensor = None
            self._traj = None
            self._work = {}
            self._reset_sweep_graphs()
        self.step_size = step_size
        self.enable_adjoint = enable_adjoint
        if not enable_adjoint:
            self._traj = None          # ts.removeTrajectory() (pa.py:773-774)
        # ts.setFromOptions() (pa.py:775).  Callers such as train-Cifar10.py:121-139 call setupTS on
        # every forward: when neither the tableau choice nor the options database changed since
        # the last call this is a dictionary comparison, not a re-parse.
        sig = (options.get_all(), self._theta_method, id(self._ops))
        if sig != self._options_sig:
            self._set_from_options()
            self._theta = None
            self._reset_sweep_graphs()  # captured sweeps belong to the scheme and modes they were captured with
            # -ts_type on the command line overrides the `method` keyword, as ts.setFromOptions() does
            # (README.md:89: "-ts_type cn will choose the Crank-Nicolson methods")
            ts_type = str(options.get_all().get("ts_type", ""))
            stepper = self._theta_method
            if ts_type in ("beuler", "cn", "theta") and stepper != "imex":
                stepper = ts_type
            elif ts_type == "rk" and stepper != "imex":
                stepper = None
            elif ts_type == "arkimex" and stepper != "imex":
                raise PnError("-ts_type arkimex needs the IMEX set-up (setupTS(..., imex_form=True, method='imex', func2=...))")
            elif ts_type in ("beuler", "cn", "theta", "rk") and stepper == "imex":
                raise PnError("-ts_type %s cannot override an IMEX set-up (two functions were given)" % ts_type)
            self._stepper_kind = stepper
            adapt_wanted = str(options.get_all().get("ts_adapt_type", "basic")) != "none"
            if stepper == "imex":
                from .arkimex import ArkimexStepper
                self._theta = ArkimexStepper(self, options.get_all())
                has_embed = self._theta.embedded() is not None
                check(self._lib.pn_ts_set_scheme(self._ts, self._theta.tab["adapt_order"], 1 if has_embed else 0))
                if adapt_wanted and not has_embed:
                    warnings.warn("pnode_amd: ARKIMEX type %s has no embedded weights here and takes the fixed steps of "
                                  "step_size; PETSc adapts unless -ts_adapt_type none is given (every IMEX run of the "
                                  "reference gives it).  Pass -ts_adapt_type none to state that explicitly, or use type "
                                  "3, 4, 5 or 1bee." % self._theta.name, RuntimeWarning)
                self._adaptive = bool(self._lib.pn_ts_is_adaptive(self._ts))
                if self._theta.name in ("l2", "2c", "2d", "2e"):
                    _warn_unpinned("arkimex_table_" + self._theta.name,
                                   "ARKIMEX type %s: the identification of this table with PETSc's type of that name "
                                   "(l2) / the free entries of its explicit part (2c, 2d, 2e) are" % self._theta.name)
                if self._adaptive:
                    _warn_unpinned("arkimex_adapt", "adaptive ARKIMEX steps (TSAdapt basic on the embedded pair; embedded "
                                   "weights, controller constants and step rejection are")
            elif stepper:
                from .theta import ThetaStepper
                self._theta = ThetaStepper(self, stepper, options.get_all())
                # TSCreate_Theta sets the TS's default adapt type to NONE: beuler / cn take the fixed steps of
                # step_size unless `-ts_adapt_type basic` is given explicitly (the reference's spiral_unstable.py
                # and ode_demo_petsc.py give no adapt option and run cn / beuler with fixed steps).  With it, PETSc
                # estimates the local truncation error from the last three solutions and the controller uses
                # order 2 (see ThetaStepper.error_norm).
                explicit_basic = str(options.get_all().get("ts_adapt_type", "")) == "basic"
                check(self._lib.pn_ts_set_scheme(self._ts, 2, 1 if explicit_basic else 0))
                self._adaptive = bool(self._lib.pn_ts_is_adaptive(self._ts))
                if self._adaptive:
                    _warn_unpinned("theta_adapt", "adaptive beuler / cn steps (-ts_adapt_type basic: the three-solution local "
                                   "truncation error estimate of TSEvaluateWLTE_Theta and its controller order are")
            else:
                check(self._lib.pn_ts_set_scheme(self._ts, 0, 0))          # the RK tableau drives the controller
            self._options_sig = sig
        self._setup_linear_grads()

    def _set_from_options(self):
        """ts.setFromOptions() (pa.py:775) for the option subset of this path."""
        db = options.get_all()
        self._monitor = "ts_monitor" in db
        self._view = "ts_view" in db
        if "log_view" in db and self.device.type == "cuda":
            from . import logview
            logview.enable()
        # not a PETSc option: bracket the sweeps with roctx ranges (visible to rocprofv3 --marker-trace)
        self._trace = self.device.type == "cuda" and options.truthy(db.get("pn_trace"), False) if "pn_trace" in db else False
        # not a PETSc option: ONE switch that takes back every default of this package that a user of the reference
        # could observe (all of them leave the gradients bit-identical):
        #   * stage tapes are not retained (-pn_trajectory_retain_graph 0): func is re-evaluated inside every stage VJP
        #     of the reverse sweep, as RHSJacShell.multTranspose does (pa.py:66-74), so funcs that count their calls
        #     (NFE-B of the reference's drivers, examples-pnode/spiral_unstable.py:326-347) read what they read there;
        #   * PETSc's trajectory default stays solution-only whatever fits in HBM (no automatic stage keeping), and a
        #     reversed step is recomputed WHOLE, last stage derivative included, as TSTrajectory's TSStep does;
        #   * the steps of an output interval are counted with the reference's +-1e-5 / 1e-3 window (-pn_span_count
        #     reference, pa.py:527-532);
        #   * the Newton-Krylov solves launch func eagerly (-pn_krylov_graph 0): its side effects happen at every call.
        # Each of these can still be set on its own; an explicit option wins over the switch.
        self._ref_defaults = options.truthy(db.get("pn_reference_defaults"), False) if "pn_reference_defaults" in db else False
        self._solution_only = options.truthy(db.get("ts_trajectory_solution_only"), True)
        # option not given: PETSc's default (states only, stages recomputed) unless everything fits easily, see _pick_traj_mode
        self._solution_only_auto = "ts_trajectory_solution_only" not in db and not self._ref_defaults
        ram = int(float(db["ts_trajectory_max_cps_ram"])) if db.get("ts_trajectory_max_cps_ram", "") != "" else 0
        disk = int(float(db["ts_trajectory_max_cps_disk"])) if db.get("ts_trajectory_max_cps_disk", "") != "" else 0
        if ram < 0 or disk < 0:
            raise PnError("-ts_trajectory_max_cps_ram / -ts_trajectory_max_cps_disk must not be negative")
        # PETSc's two levels: at most `ram` checkpoints in memory (HBM here) and `disk` more in files; one scheduler places the
        # ram + disk of them (_TwoLevelTrajectory).  Only -ts_trajectory_max_cps_disk: the bounded set lives in files.
        self._max_cps = ram + disk
        self._max_cps_ram, self._max_cps_disk = ram, disk
        # -ts_trajectory_type: "memory" = HBM (the default here; PETSc's default is "basic" = one file per checkpoint,
        # which is what "basic" selects here too); PETSc's other types are not built and are refused, not ignored
        ttype = str(db.get("ts_trajectory_type", "memory"))
        if ttype not in ("memory", "basic"):
            raise PnError("-ts_trajectory_type %s is not implemented by pnode_amd (memory: checkpoints in HBM; basic: one file "
                          "per checkpoint under -ts_trajectory_dirname)" % ttype)
        self._traj_disk = ttype == "basic" or disk > 0
        self._traj_all_on_disk = ttype == "basic"          # -ts_trajectory_type basic: every checkpoint is a file, whatever the budgets
        self._traj_dirname = str(db.get("ts_trajectory_dirname", "SA-data"))
        self._traj_keep = options.truthy(db.get("ts_trajectory_keep_files"), False) if "ts_trajectory_keep_files" in db else False
        # not a PETSc option.  With store-all checkpoints (-ts_trajectory_solution_only 0) the forward sweep can
        # also keep every stage's autograd tape, so that the reverse sweep runs only the backward half of each
        # stage VJP instead of re-evaluating f first (pa.py:66-68 re-evaluates).  Same bits either way.
        #   auto (default): tapes are kept while they fit in half of the HBM that is free when the sweep starts
        #   1: always   0: never (the reference's recompute)
        rg = str(db.get("pn_trajectory_retain_graph", "0" if self._ref_defaults else "auto"))
        self._retain_graph = 2 if rg == "auto" else (1 if options.truthy(rg, False) else 0)
        # not a PETSc option: how the steps of an output interval are counted (see _span_post_step)
        self._span_count_reference = str(db.get("pn_span_count", "reference" if self._ref_defaults else "exact")) == "reference"
        self._accum_mode = str(db.get("pn_param_accum", "batch"))
        if self._accum_mode not in ("batch", "step", "stage"):
            raise PnError("-pn_param_accum must be batch, step or stage")
        self._accum_sources = max(1, min(32, int(float(db.get("pn_param_accum_sources", 32)))))
        # not a PETSc option: who walks the tableau.  native (default): one C++ entry point per step attempt / reversed step
        # (pn_rk_attempt, pn_rk_adjoint_step -- PETSc's C loops behind ts.solve / ts.adjointSolve, pa.py:829, 878) that calls
        # back only for func and its VJP; python: the stage loop of rounds 1-3, one ctypes call per launch.  Same launches,
        # same coefficients, same bits.
        sl = str(db.get("pn_step_loop", "native"))
        if sl not in ("native", "python"):
            raise PnError("-pn_step_loop must be native or python")
        self._native = sl == "native" and bool(getattr(self._ops, "native_steps", False))
        # not a PETSc option: after GRAPH_WARMUP_CALLS eager calls with the same shapes/times,
        # capture the whole forward sweep and the whole reverse sweep as two hipGraphs and
        # replay them (fixed-step only; func must be capturable: no host-side data dependence)
        #   auto (the default on a HIP device; off under -pn_reference_defaults): explicit fixed-step RK sweeps only, and only
        #        when it is safe and pays -- see _graph_entry / _auto_capture_forward: plain call counters of func keep counting
        #        (their increments are learnt in the warm-up calls), any other Python-side change during a sweep keeps the solver
        #        eager, the first replay of each sweep is compared with the eager sweep of the same call, and replay must not
        #        be slower than the eager launches it replaces
        #   1: always (every capturable stepper; a failure to capture warns and falls back)      0: never
        gco = str(db.get("pn_graph_capture", "0" if self._ref_defaults else "auto"))
        self._graph_mode = 2 if gco == "auto" else (1 if options.truthy(gco, False) else 0)
        self._graph_status = "eager (-pn_graph_capture 0)" if self._graph_mode == 0 else "eager (warming up)"
        self._auto_veto = None
        # not a PETSc option: in auto mode every N-th replayed call of a captured pair is ALSO run eagerly and compared, as
        # the call that captured it was (0: never).  Catches state of func that the capture guard cannot see.
        self._revalidate_every = int(float(db.get("pn_graph_revalidate", self.GRAPH_REVALIDATE_EVERY)))
        if self._revalidate_every < 0:
            raise PnError("-pn_graph_revalidate must not be negative")
        for key, val in db.items():
            if key.startswith("ts_trajectory") or key in ("ts_monitor", "ts_view") or key.startswith("pn_"):
                continue
            if key == "ts_type" and str(val) in ("beuler", "cn", "theta", "arkimex"):
                continue
            if key.startswith("ts_arkimex") or key.startswith("ts_theta"):
                continue
            if key.startswith("ts_"):
                try:
                    check(self._lib.pn_ts_set_option(self._ts, key.encode(), str(val).encode()))
                except PnError as exc:
                    raise PnError("PETSc option -%s %s is not implemented by pnode_amd (%s). Implemented: -ts_type, -ts_rk_type, "
                                  "-ts_adapt_type none|basic, -ts_rtol, -ts_atol, -ts_max_steps, -ts_max_reject, -ts_adapt_safety, "
                                  "-ts_adapt_reject_safety, -ts_adapt_clip, -ts_adapt_dt_min, -ts_adapt_dt_max, -ts_arkimex_type, "
                                  "-ts_trajectory_*, -ts_monitor, -ts_view, -snes_*, -ksp_*, -log_view; an option that could change "
                                  "the numbers is refused rather than ignored" % (key, val, exc)) from None
        tab = _lib.Tableau()
        check(self._lib.pn_ts_get_tableau(self._ts, ctypes.byref(tab)))
        s = tab.s
        self._s = s
        self._fsal = bool(tab.fsal)
        self._s_eff = s - 1 if self._fsal else s      # stages whose adjoint is not structurally zero
        self._A = [[tab.A[i][j] for j in range(s)] for i in range(s)]
        self._b = [tab.b[j] for j in range(s)]
        self._c = [tab.c[j] for j in range(s)]
        self._e = [tab.bembed[j] - tab.b[j] for j in range(s)]
        self._plans = {}
        self._adaptive = bool(self._lib.pn_ts_is_adaptive(self._ts))
        a, r = ctypes.c_double(), ctypes.c_double()
        check(self._lib.pn_ts_get_tolerances(self._ts, ctypes.byref(a), ctypes.byref(r)))
        self._atol, self._rtol = a.value, r.value
        # budgeted checkpoints with -ts_trajectory_solution_only 0: a checkpoint holds the stage values
        # of its step as well (PETSc's checkpoints do), so reversing a checkpointed step recomputes nothing
        self._budget_stages = self._max_cps > 0 and not self._solution_only
        if self._max_cps > 0:
            self._traj_mode = _lib.PN_TRAJ_BUDGET
        elif self._solution_only:
            self._traj_mode = _lib.PN_TRAJ_SOLUTION
        else:
            self._traj_mode = _lib.PN_TRAJ_ALL

    def _new_trajectory(self, vecs, mode):
        """TSTrajectory of the coming forward sweep: HBM slabs, or files for -ts_trajectory_type basic (every placement:
        all steps, or the bounded set of -ts_trajectory_max_cps_ram)."""
        if self._max_cps_disk > 0 and self._max_cps_ram > 0 and mode == _lib.PN_TRAJ_BUDGET and not self._traj_all_on_disk:
            return _TwoLevelTrajectory(self._lib, self._ops, self.n, vecs, mode, self._max_cps, self._traj_dirname, self._traj_keep,
                                       self._max_cps_ram)
        if self._traj_disk:
            return _DiskTrajectory(self._lib, self._ops, self.n, vecs, mode, self._max_cps, self._traj_dirname, self._traj_keep)
        return _Trajectory(self._lib, self._ops, self.n, vecs, mode, self._max_cps)

    # ------------------------------------------------------------------ helpers
    def _flat(self, t):
        return t.reshape(-1)

    def _buf(self, name):
        b = self._work.get(name)
        if b is None:
            b = self._ops.empty(self._npad)
            self._work[name] = b
        return b

    def _shaped(self, flat):
        return flat[: self.n].view(self.tensor_size)


    # ------------------------------------------------------------------ forward (pa.py:777-869)
    def odeint(self, u0, t):
        """Solve du/dt = func(t, u), u(t[0]) = u0; returns the states at the times `t`
        (first dimension), or, when `t` has one element, integrates [0, t[0]] (pa.py:818-820)."""
        return self._odeint(u0, t, self.enable_adjoint)

    def _device_guard(self):
        """The device entry points launch on the calling thread's current HIP device: make it the
        solver's device for the duration of a sweep (a no-op context for the CPU test stand-in)."""
        if self.device is not None and self.device.type == "cuda":
            return self._sweep_context()
        return contextlib.nullcontext()

    @contextlib.contextmanager
    def _sweep_context(self):
        with torch.cuda.device(self.device):
            ops = self._ops
            prev = getattr(ops, "_pinned_stream", None)
            if ops is not None and hasattr(ops, "_pinned_stream"):
                ops._pinned_stream = ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
            try:
                yield
            finally:
                if ops is not None and hasattr(ops, "_pinned_stream"):
                    ops._pinned_stream = prev

    def _odeint(self, u0, t, save):
        with self._device_guard():
            if not self._trace:
                return self._odeint_impl(u0, t, save)
            torch.cuda.nvtx.range_push("pnode_amd.forward_sweep")      # roctx range on ROCm
            try:
                return self._odeint_impl(u0, t, save)
            finally:
                torch.cuda.nvtx.range_pop()

    def _odeint_impl(self, u0, t, save):
        if self._ops is None:
            raise RuntimeError("setupTS must be called before odeint")
        if u0.size() != self.tensor_size or u0.dtype != self.tensor_dtype or u0.device != self.device:
            raise ValueError("u0 does not match the tensor given to setupTS (shape, dtype, device)")
        if self._theta is not None:
            return self._theta.odeint(u0, t, save)
        lib, ops, ts = self._lib, self._ops, self._ts
        self.sol_times = t.detach().cpu().to(dtype=torch.float64)
        T = int(t.shape[0])
        times = self.sol_times.tolist()
        dt0 = float(self.step_size[0] if isinstance(self.step_size, list) else self.step_size)
        check(lib.pn_ts_begin(ts, 0.0, dt0, T, (ctypes.c_double * T)(*times)))
        self._span_begin(T)
        solution = ops.empty((T,) + tuple(self.tensor_size))
        sol_flat = solution.view(T, -1)
        u0f = u0.detach().contiguous().reshape(-1)

        # where the state at the start of step k lives
        self._tmode = self._pick_traj_mode(self._s_eff) if save else self._traj_mode
        if save:
            vecs = self._s_eff if (self._tmode == _lib.PN_TRAJ_ALL or self._budget_stages) else 1
            self._traj = self._new_trajectory(vecs, self._tmode)
            traj = self._traj
            if self._tmode == _lib.PN_TRAJ_BUDGET and not self._adaptive and not isinstance(self.step_size, list):
                total = lib.pn_ts_count_fixed_steps(ts)         # fixed step: the sweep length is known
                if total > 0:
                    check(lib.pn_traj_set_total(traj.handle, total))
        else:
            traj = self._traj = None
        store_stages = save and self._tmode == _lib.PN_TRAJ_ALL
        # (per-evaluation graphs, pnode_amd/_stagegraphs.py: a captured evaluation's tape is overwritten by its next replay)
        keep_tape = store_stages and self._retain_graph != 0 and self._sg is None
        tape_budget = None
        if keep_tape and self._retain_graph == 2:
            tape_budget = self._tape_budget()
            keep_tape = tape_budget is not None and tape_budget > 0
        self._tapes = {} if keep_tape else None
        tape_fsal = None
        pingpong = [self._buf("u_a"), self._buf("u_b")]
        pp = 0

        cur_slot = -1                # trajectory slot `cur` lives in, -1 when it is a ping-pong buffer

        def state_home(step):
            nonlocal pp, home_slot
            if traj is not None:
                slot = traj.fwd_slot(step)
                if slot >= 0:
                    home_slot = slot
                    traj.stage_step.pop(slot, None)  # a recycled slot no longer holds the old step's stages
                    return traj.claim(slot)         # (vecs, npad)
            home_slot = -1
            pp ^= 1
            return pingpong[pp].view(1, -1)

        home_slot = -1
        cur = state_home(0)
        cur_slot = home_slot
        ops.copy(cur[0], u0f)
        if T > 1:
            ops.copy(sol_flat[0], u0f)
        K_fsal = None
        tt, hh = ctypes.c_double(), ctypes.c_double()
        acc, hit, done = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(0)
        finished = not (times[-1] > (0.0 if T == 1 else times[0]))
        if self._monitor:
            print("%d TS dt %g time %g" % (0, dt0, 0.0 if T == 1 else times[0]))
        while not finished:
            step = lib.pn_ts_steps(ts)
            nxt = state_home(step + 1)
            nxt_slot = home_slot
            K0 = K_fsal
            tape0 = tape_fsal
            while True:
                check(lib.pn_ts_attempt(ts, ctypes.byref(tt), ctypes.byref(hh)))
                tn, h = tt.value, hh.value
                if store_stages or (self._budget_stages and cur_slot >= 0):
                    dest = lambda i, c=cur: c[i]
                else:
                    dest = lambda i: self._buf("y_scratch")
                tapes = [tape0] + [None] * (self._s - 1) if keep_tape else None
                K = self._rk_step(tn, h, cur[0], K0, nxt[0], dest, self._adaptive, tapes)
                if keep_tape:
                    tape0 = tapes[0]
                enorm = self._global_enorm(ops.read_enorm()) if self._adaptive else -1.0
                check(lib.pn_ts_judge(ts, enorm, ctypes.byref(acc), ctypes.byref(hit), ctypes.byref(done)))
                if acc.value:
                    break
                K0 = K[0]            # f(t_n, u_n) does not depend on h
            K_fsal = K[self._s - 1] if self._fsal else None
            if keep_tape:
                self._tapes[step] = tapes[: self._s_eff]
                tape_fsal = tapes[self._s - 1] if self._fsal else None
                if tape_budget is not None and tape_budget != float("inf"):
                    if step == 0:                 # one measurement: what a step's tapes (and its slot) take
                        per_step = max(_mem_now(self.device)[0] - self._tape_mem0, 1)
                        tape_steps = int(tape_budget // per_step) - 1
                    if step + 1 >= tape_steps:
                        keep_tape, tape_fsal = False, None       # later steps re-evaluate f in the reverse sweep
                        self._tape_all_fit = False
            if self._budget_stages and cur_slot >= 0 and save:
                traj.stage_step[cur_slot] = step
            if traj is not None and cur_slot >= 0:
                traj.seal(cur_slot)          # the step's checkpoint is complete (a no-op on the HBM tier)
            cur = nxt
            cur_slot = nxt_slot
            stepno = step + 1
            tnew = lib.pn_ts_time(ts)
            self._span_post_step(T, times, hit.value, done.value, stepno, tnew, cur[0], sol_flat)
            if self._monitor:
                print("%d TS dt %g time %g" % (stepno, h, tnew))
            finished = bool(done.value)
        self._nsteps = lib.pn_ts_steps(ts)
        if self._view:
            self._ts_view()
        if T == 1:
            ops.copy(sol_flat[0], cur[0])
        else:
            self._span_end(T)
        return solution

    # ------------------------------------------------------------------ time span (pa.py:518-532, 822-868)
    def _span_begin(self, T):
        self.cur_sol_steps = [0] * T      # steps taken from the previous output time to this one
        self.cur_sol_index = 1
        self._span_hits = 1               # output times whose solution has been kept (t[0] is u0)
        self._span_delta = 1e-5 if self.tensor_dtype == torch.double else 1e-3

    def _span_post_step(self, T, times, hit, done, stepno, tnew, cur, sol_flat):
        """What happens after an accepted step of a multi-output solve.

        * The output itself: the reference reads PETSc's ``getTimeSpanSolutions()`` (pa.py:845), i.e.
          the state of exactly the step that landed on t[i].  Here: ``pn_ts_judge`` reports that step
          (`hit` = i) and the state is copied out then.
        * ``tspanPostStep`` (pa.py:518-532): a ``step_size`` list sets the next step; the steps of
          each output interval are counted for the reverse sweep.  The reference advances its
          interval counter when ``|t - t[i]| < 1e-5`` (fp64) / ``1e-3`` (fp32), which is one step
          early whenever the step is shorter than that window: its backward pass then injects
          dL/dy(t[i]) one step off and never reverses the sweep's first step.  The default here
          counts with the exact hit (the discrete adjoint of what the forward sweep computed);
          ``-pn_span_count reference`` counts as the reference does (identical whenever every step
          is longer than the window)."""
        if T <= 1:
            return
        if hit >= 0:
            self._ops.copy(sol_flat[hit], cur)
            self._span_hits += 1
        if self.cur_sol_index < T:
            if isinstance(self.step_size, list) and stepno < len(self.step_size) and not done:
                check(self._lib.pn_ts_override_next_dt(self._ts, float(self.step_size[stepno])))
            self.cur_sol_steps[self.cur_sol_index] += 1
            if self._span_count_reference:
                if abs(tnew - times[self.cur_sol_index]) < self._span_delta:
                    self.cur_sol_index += 1
            elif hit >= 0:
                self.cur_sol_index = hit + 1

    def _span_end(self, T):
        if self.cur_sol_index != T or self._span_hits != T:
            raise Exception("TSSolve fails to step on all the specified points")

    def _pick_traj_mode(self, vecs_all):
        """Trajectory mode of the solve that was just begun (pn_ts_begin done).  When
        -ts_trajectory_solution_only is not given PETSc keeps the states only and recomputes a step's
        stages when it is reversed.  Every mode replays the same arithmetic -- gradients are identical bit
        for bit -- so on a 288 GB part the stage values are kept as well whenever the step count is known
        (fixed step) and the whole trajectory fits in a quarter of the HBM that is free right now: the
        reverse sweep then recomputes nothing.  Give the option (0 or 1) to decide yourself."""
        mode = self._traj_mode
        if (mode != _lib.PN_TRAJ_SOLUTION or not self._solution_only_auto or self.device.type != "cuda"
                or isinstance(self.step_size, list)):
            return mode
        if torch.cuda.is_current_stream_capturing():        # no driver query while capturing: as the last eager solve
            return getattr(self, "_tmode_auto", mode)
        total = self._lib.pn_ts_count_fixed_steps(self._ts)
        self._tmode_auto = mode
        if total > 0:
            esize = 4 if self.tensor_dtype == torch.float32 else 8
            need = (total + 1) * vecs_all * self._npad * esize
            free, _ = torch.cuda.mem_get_info(self.device)
            allocated, reserved = _mem_now(self.device)
            if need <= 0.25 * (free + max(reserved - allocated, 0)):
                self._tmode_auto = _lib.PN_TRAJ_ALL
        return self._tmode_auto

    def _tape_budget(self):
        """Bytes the retained tapes of this sweep may take in `auto` mode: half of the HBM that is free now
        (driver-free + cached-but-unused blocks of PyTorch's allocator); None on the CPU test stand-in.
        While a hipGraph is being captured no driver query is made: the sweep keeps what the eager
        warm-up call before it kept."""
        if self.device.type != "cuda":
            return None
        if torch.cuda.is_current_stream_capturing():
            return float("inf") if getattr(self, "_tape_all_fit", False) else None
        self._tape_mem0, reserved = _mem_now(self.device)
        free, _ = torch.cuda.mem_get_info(self.device)
        self._tape_all_fit = True
        return 0.5 * (free + max(reserved - self._tape_mem0, 0))


    def _ts_view(self):
        """-ts_view: the solver's settings and counters after a solve (PETSc prints its TS object there)."""
        modes = {_lib.PN_TRAJ_ALL: "every step, with stage values", _lib.PN_TRAJ_SOLUTION: "every step, solution only",
                 _lib.PN_TRAJ_BUDGET: "at most %d checkpoints (%s)" % (self._max_cps, "with stage values" if self._budget_stages else "solution only")}
        tab = _lib.Tableau()
        check(self._lib.pn_ts_get_tableau(self._ts, ctypes.byref(tab)))
        print("TS Object (pnode_amd): type rk, order %d, %d stages%s%s"
              % (tab.order, tab.s, ", first same as last" if tab.fsal else "", ", embedded error estimate" if tab.has_embed else ""))
        print("  adapt: %s%s" % ("basic, atol %g rtol %g" % (self._atol, self._rtol) if self._adaptive else "none (fixed steps)",
                                 "; final time matched exactly (MATCHSTEP)"))
        print("  state: %s %s on %s;  trainable parameters: %d" % (tuple(self.tensor_size), str(self.tensor_dtype).replace("torch.", ""), self.device, self.np))
        print("  launches: %s;  step loop: %s" % (self._graph_status, "C++ (pn_rk_attempt / pn_rk_adjoint_step)" if self._native else "Python"))
        print("  total number of time steps=%d, rejected=%d;  trajectory: %s"
              % (self._nsteps, self._lib.pn_ts_rejections(self._ts), modes[self._tmode] if self._traj is not None else "not saved"))


    def petsc_adjointsolve(self, t, i=1):
        """Reverse one output interval (pa.py:871-890): all steps when `t` has one element,
        else the ``cur_sol_steps[i]`` steps that led to output time i."""
        if t.shape[0] == 1:
            self._adjoint_steps(self._nsteps, None)
        else:
            self._adjoint_steps(self.cur_sol_steps[i], None)
        self._flush_param_accum()
        self._finish_linear_accum()
        return self._shaped(self.adj_u_flat), self.adj_p_tensor


    # ------------------------------------------------------------------ autograd entry (pa.py:892-900)
    def odeint_adjoint(self, y0, t):
        if not isinstance(self.funcIM, nn.Module):
            raise ValueError("func is required to be an instance of nn.Module.")
        # inside Function.forward grad mode is always off, so note here whether a backward can follow
        self._grad_mode = torch.is_grad_enabled()
        return OdeintAdjointMethod.apply(y0, t, self.flat_params, self, *self._params)


class OdeintAdjointMethod(torch.autograd.Function):
    """pa.py:903-947.  Forward: solve under no_grad.  Backward: seed lambda with dL/dy(t_T),
    reverse the output intervals newest first, adding dL/dy(t_{i-1}) after each one."""

    @staticmethod
    def forward(ctx, y0, t, flat_params, ode, *args):
        ctx.ode = ode
        need = ode.enable_adjoint and ode._grad_mode and (ctx.needs_input_grad[0] or any(ctx.needs_input_grad[4:]))
        with torch.no_grad():
            ans, e, warm = ode._sweep_forward(y0, t, need)
        ctx.graph_entry = e
        ctx.warm_entry = warm
        if "pnode_amd.logview" in sys.modules:
            sys.modules["pnode_amd.logview"].note_forward(ode)
        ctx.save_for_backward(t, flat_params, ans)
        return ans

    @staticmethod
    def backward(ctx, *grad_output):
        t, flat_params, ans = ctx.saved_tensors
        ode = ctx.ode
        T = ans.shape[0]
        g = grad_output[0]
        if g.dtype != ode.tensor_dtype:
            g = g.to(ode.tensor_dtype)
        g = g.contiguous().view(T, -1)
        with torch.no_grad():
            ode._sweep_backward(ctx.graph_entry, getattr(ctx, "warm_entry", None), g, T)
            ode._allreduce_adj_p()
            if "pnode_amd.logview" in sys.modules:
                sys.modules["pnode_amd.logview"].note_backward(ode)
            adj_u = ode._shaped(ode.adj_u_flat).detach().clone()
            adj_p = ode.adj_p_tensor.detach().clone()
            gparams = tuple(adj_p[o:o + l].view_as(p).to(p.dtype) for p, o, l in zip(ode._params, ode._poff, ode._plen))
        return (adj_u, None, None, None) + gparams
