"""Implicit one-step (theta) methods ``beuler`` and ``cn`` of the reference's implicit path
(``setupTS(..., implicit_form=True, method="beuler"|"cn")``; reference
``pnode/petsc_adjoint.py`` ("pa.py") 651-654 TS type BE/CN, 414-441 ``evalIFunction``
F = M udot - f, 98-197 ``IJacShell.mult/multTranspose`` = shift*M - df/du matrix-free) and
their discrete adjoint (PETSc ``TSAdjointStep_Theta`` behind pa.py:875-878).

PETSc's pieces restated on the C ABI:
  TSStep_Theta          stage equation  shift*M (X - u_n) - f(t_s, X) = b, shift = 1/(theta h)
                        (cn = theta 1/2 in endpoint form, b = ((1-theta)/theta) f(t_n,u_n);
                         beuler = theta 1 in stage form, b = 0)
  SNES newtonls|ksponly Newton iteration with PETSc's default tolerances (rtol 1e-8, atol 1e-50,
                        stol 1e-8, 50 iterations), full steps with halving when the residual grows
  KSP gmres             restarted GMRES(30), rtol 1e-5, classical Gram-Schmidt with one
                        re-orthogonalisation when most of ||w|| cancels.  Device-resident (pn_krylov_*):
                        products, Hessenberg column, Givens rotations, residual estimate and a stop flag
                        live in HBM, the host enqueues iterations in chunks and synchronises once per chunk
                        (-pn_krylov host: round 2's loop, one synchronisation per iteration, pn_gmres_*)
  operator              shift*M v - J v, J v by the double-VJP identity on one graph of f per Newton
                        iterate (the reference rebuilds the graph per product).  On the HIP device the
                        linearisation of f at a stage time and the product are captured ONCE as hipGraphs
                        with static input buffers and replayed for every Newton iterate / Krylov iteration
                        (-pn_krylov_graph auto|0|1; see _OpGraph)
  adjoint               (shift*M - J)^T nu = shift*rhs by the same GMRES on the transposed
                        operator (J^T v is one backward through the cached graph), then
                        lambda/mu updates as in oracle/theta_oracle.py
Vector arithmetic is on the device entry points; func, its VJP/JVP and the optional mass
matrix product (``torch.matmul(mass, .)``, as pa.py:429-431) are PyTorch.
"""
import contextlib
import ctypes
import gc
import warnings

import torch

from . import _lib
from ._lib import check

THETA_METHODS = {"beuler": (1.0, False), "cn": (0.5, True)}

# What the timed comparison of `-pn_krylov_graph auto` chose (double-VJP graph, forward-mode graph, eager launches) per
# (class of func, state size, dtype, which function): every solver of a process that runs the same dynamics at the same
# size takes the same form -- the three are equal to round-off only, and two solvers that must agree bit for bit (the
# checkpoint modes of one problem, say) must not be split by a near-tie of their timings.
_GRAPH_CHOICE = {}


class _OpGraph(object):
    """f linearised at one stage time, as replayable hipGraphs over static buffers (HIP device only).

    The matrix-free Newton-Krylov solves are launch-bound at the sizes the reference runs them (a stage system of
    64 x 1024 unknowns: every operator product is a dozen small kernels behind ~1 ms of eager PyTorch dispatch).
    So the two things that are repeated are captured once and replayed:
      A  the linearisation: out = f(t, x) recorded by autograd, x a static buffer -- and, for products with J, the
         graph of g = J^T dummy (create_graph) whose derivative with respect to dummy is v -> J v.  Replaying A
         after copying a new Newton iterate into x refreshes every saved activation in place, so ONE capture serves
         every iterate of every step that evaluates f at this stage time; out doubles as the residual's f(t, x).
      B  the product: w = shift*M vin - J vin (or the transposed operator), vin/w static (the Krylov buffers); one
         small graph per shift, sharing A.
      C  (transposed entries) the parameter cotangents (df/dp)^T vin through the same linearisation.
    f's time argument is a Python float (as PETSc passes it, pa.py:405): it is baked into the capture, so entries
    are keyed by the stage time (a fixed-step training loop sees the same stage times in every solve;
    -pn_krylov_autonomous 1 declares f time-independent and uses one entry).  Python-side effects of f (call counters)
    happen at capture only; what f does in device kernels -- the in-place update of BatchNorm's running statistics in
    train mode -- is part of graph A and re-runs at every linearise(); -pn_krylov_graph 0 gives the eager path."""

    def __init__(self, st, t, transpose, fwd=False):
        self.st, self.t, self.transpose = st, t, transpose
        # fwd: the product is one forward-mode pass of func (torch.func.jvp: primal and tangent together) instead of a
        # backward through the graph of J^T dummy.  Twice the GEMMs of the double-VJP form for a dense net, but the only
        # good form for a func whose double backward is a poor algorithm (PyTorch's fp64 convolution fallback on ROCm).
        self.fwd = bool(fwd) and not transpose
        self.B = {}
        self.gC = None
        self.gp = None
        self.x = st._buf("lin_x")
        # Every body runs once eagerly before it is captured: the libraries underneath (hipBLASLt heuristics and
        # workspaces, MIOpen's find step) initialise on first use with calls that are illegal inside a capture.
        self._linearisation()
        gA = torch.cuda.CUDAGraph()
        with st._capturing(gA):
            self.xx, self.out, self.wrt, self.dummy, self.g, self.fx = self._linearisation()
        self.gA = gA

    def _linearisation(self):
        st, o = self.st, self.st.ode
        if self.fwd:
            fn = o.funcIM if st.which == "IM" else o.funcEX
            with torch.no_grad():
                xx = o._shaped(self.x)
                out = fn(self.t, xx)
                if out.shape != xx.shape or out.dtype != xx.dtype:
                    raise ValueError("func must return a tensor with the state's shape and dtype")
            return xx, out, (), None, None, out.contiguous().reshape(-1)
        with torch.enable_grad():
            xx = o._shaped(self.x).detach().requires_grad_(True)
            out, wrt = o._func_with_grad(self.t, xx, st.which)
            if out.shape != xx.shape or out.dtype != xx.dtype:
                raise ValueError("func must return a tensor with the state's shape and dtype")
            dummy = g = None
            if not self.transpose:
                dummy = torch.zeros_like(out, requires_grad=True)
                g = torch.autograd.grad(out, xx, dummy, create_graph=True, allow_unused=True)[0]
        return xx, out, wrt, dummy, g, out.detach().contiguous().reshape(-1)

    def linearise(self, X):
        """Refresh the linearisation at the flat state X; returns f(t, X) (a static flat tensor)."""
        self.st.ode._ops.copy(self.x, X)
        self.gA.replay()
        return self.fx

    def _product_body(self, shift, kr):
        st, o = self.st, self.st.ode
        v = o._shaped(kr.vin)
        if self.fwd:
            fn = o.funcIM if st.which == "IM" else o.funcEX
            with torch.no_grad():
                jv = torch.func.jvp(lambda y: fn(self.t, y), (self.xx,), (v,))[1]
        elif self.transpose:
            jv = torch.autograd.grad(self.out, self.xx, v.view(self.out.shape), retain_graph=True, allow_unused=True)[0]
        elif self.g is not None:
            jv = torch.autograd.grad(self.g, self.dummy, v.view(self.g.shape), retain_graph=True, allow_unused=True)[0]
        else:
            jv = None
        st._apply(lambda _v: None if jv is None else jv.contiguous().reshape(-1), shift, kr.vin, kr.w, self.transpose)

    def product(self, shift, kr, with_step=False):
        """A callable that enqueues  kr.w <- shift*M kr.vin - J kr.vin  (transposed entry: the transposed operator) and, with
        `with_step`, the GMRES iteration that consumes it (pn_krylov_step with k = -1: the kernels take the iteration index
        from the state block, so ONE captured graph serves every iteration of every solve with this shift -- one graph launch
        per Krylov iteration is all the host does)."""
        key = (round(shift, 14), kr.vin.data_ptr(), kr.w.data_ptr(), bool(with_step))
        g = self.B.get(key)
        if g is None:
            ops = self.st.ode._ops
            self._product_body(shift, kr)                  # eager once (see __init__); kr.w is scratch at this point
            g = torch.cuda.CUDAGraph()
            with self.st._capturing(g):
                self._product_body(shift, kr)
                if with_step:
                    ops.krylov_step(kr, -1)
            if len(self.B) >= 8:
                self.B.pop(next(iter(self.B)))
            self.B[key] = g
        return g.replay

    def eager_product(self):
        """v -> J v (J^T v) through the current linearisation without a product graph (-pn_krylov host)."""
        o = self.st.ode

        def jprod(v):
            vv = o._shaped(v)
            if self.fwd:
                fn = o.funcIM if self.st.which == "IM" else o.funcEX
                with torch.no_grad():
                    return torch.func.jvp(lambda y: fn(self.t, y), (self.xx,), (vv,))[1].contiguous().reshape(-1)
            if self.transpose:
                r = torch.autograd.grad(self.out, self.xx, vv.view(self.out.shape), retain_graph=True, allow_unused=True)[0]
            elif self.g is not None:
                r = torch.autograd.grad(self.g, self.dummy, vv.view(self.g.shape), retain_graph=True, allow_unused=True)[0]
            else:
                r = None
            return None if r is None else r.contiguous().reshape(-1)
        return jprod

    def param_cotangents(self, nu, kr):
        """(df/dp)^T nu through the current linearisation: list aligned with the implicit function's parameters."""
        st, o = self.st, self.st.ode
        if not self.wrt:
            return []
        o._ops.copy(kr.vin, nu)
        if self.gC is None:
            def body():
                gp = torch.autograd.grad(self.out, self.wrt, o._shaped(kr.vin).view(self.out.shape), retain_graph=True,
                                         allow_unused=True)
                return [None if q is None else q.to(o.tensor_dtype).contiguous() for q in gp]
            body()                                          # eager once (see __init__)
            g = torch.cuda.CUDAGraph()
            with st._capturing(g):
                self.gp = body()
            self.gC = g
        self.gC.replay()
        return self.gp


class ThetaStepper(object):
    def __init__(self, ode, method, db):
        self.ode = ode
        self.lib = ode._lib
        if method == "theta":          # -ts_type theta -ts_theta_theta <x> [-ts_theta_endpoint]; PETSc's defaults 0.5 / off
            from . import options as _options
            self.theta = float(db.get("ts_theta_theta", 0.5))
            self.endpoint = _options.truthy(db.get("ts_theta_endpoint"), False) if "ts_theta_endpoint" in db else False
            if not 0.0 < self.theta <= 1.0:
                raise _lib.PnError("-ts_theta_theta must be in (0, 1]")
        else:
            self.theta, self.endpoint = THETA_METHODS[method]
        self.method = method
        f = lambda k, d: float(db.get(k, d))
        self.snes_rtol, self.snes_atol, self.snes_stol = f("snes_rtol", 1e-8), f("snes_atol", 1e-50), f("snes_stol", 1e-8)
        self.snes_max_it = int(f("snes_max_it", 50))
        self.ksponly = str(db.get("snes_type", "newtonls")) == "ksponly"
        self.ksp_rtol, self.ksp_atol = f("ksp_rtol", 1e-5), f("ksp_atol", 1e-50)
        self.ksp_max_it = int(f("ksp_max_it", 10000))
        self.restart = int(f("ksp_gmres_restart", 30))
        self.which = "EX"        # which of the solver's functions is the implicit one (IMEX: "IM")
        self._fwd_mode = None    # None: try forward-mode JVPs; True/False once known
        if str(db.get("pn_jvp", "")) == "double_vjp":
            self._fwd_mode = False
        self.gmres = ctypes.c_void_p(self.lib.pn_gmres_create(self.restart))
        self.V = []
        # not PETSc options.  -pn_krylov device|host: where GMRES keeps its decisions (device: pn_krylov_*, one host
        # synchronisation per chunk of iterations; host: one per iteration).  -pn_krylov_graph auto|0|1: replay the
        # linearisation of f and the operator product from hipGraphs (auto: on a HIP device, until capturing fails or
        # the stage times stop repeating).  -pn_krylov_autonomous 1 declares that f ignores its time argument: one
        # captured linearisation then serves every stage time.
        from . import options as _opt
        self._krylov_mode = str(db.get("pn_krylov", "device"))
        if self._krylov_mode not in ("device", "host"):
            raise _lib.PnError("-pn_krylov must be device or host")
        kg = str(db.get("pn_krylov_graph", "0" if _opt.truthy(db.get("pn_reference_defaults"), False) and "pn_reference_defaults" in db else "auto"))
        self._graph_mode = 2 if kg == "auto" else (1 if _opt.truthy(kg, False) else 0)
        self._autonomous = _opt.truthy(db.get("pn_krylov_autonomous"), False) if "pn_krylov_autonomous" in db else False
        self._kr = None                # _KrylovBuffers, made on first use
        self._its_guess = {}           # transpose flag -> iterations the last solve of that kind took
        self._op_graphs = {}           # key -> _OpGraph (insertion-ordered: oldest first)
        self._op_pool = None
        self._op_stats = [0, 0]        # look-ups, captures
        self._validate = False         # set at the start of every solve: check the first replay against eager func
        self._stage_times_this_solve = 0   # implicit stage times of the solve in progress (fixed step: known at its start)
        self._entry_bytes = None       # device memory one captured linearisation holds (measured at the first capture)
        self._entry_cap = self.GRAPH_CACHE_ENTRIES
        self._calibrated = False       # auto mode: graphs timed against eager launches once
        # form of the captured product J v: "" (default) = by what func is (double-VJP identity; one forward-mode pass when func
        # holds a convolution in double precision -- see _op_graph); "jvp" / "dvjp" force one
        self._graph_form = str(db.get("pn_krylov_graph_form", ""))
        if self._graph_form not in ("", "jvp", "dvjp"):
            raise _lib.PnError("-pn_krylov_graph_form must be jvp or dvjp")
        self._graph_fwd = self._graph_form == "jvp"
        self._form_decided = False     # the form of J v (captured and eager alike) is fixed at the first capture
        self._calibration = None
        self._graphs_dropped = None    # why the graphs were given up, if they were
        self.host_syncs = 0            # stream synchronisations made by the Krylov solves (diagnostic)
        self.second_passes = 0         # Gram-Schmidt re-orthogonalisation passes (diagnostic)
        self._its_log = [] if "pn_krylov_log" in db else None      # (transposed?, iterations) per linear solve (diagnostic)
        self.newton_its = self.linear_its = 0
        self.traj = None
        # linear_solver="torch" (torch_linearsolve.py): LU of shift*M - J with J = d f/du of ONE sample, frozen for the solve
        self.direct = ode.linear_solver == "torch"
        if not self.direct:
            from .petsc_adjoint import _warn_unpinned
            _warn_unpinned("krylov_matrix_free", "the matrix-free Newton-Krylov stage solves (linear_solver=\"petsc\": GMRES with "
                           "classical Gram-Schmidt refined when more than 3/4 of the norm cancels -- KSPGMRES's default never "
                           "refines --, its convergence test on the recurrence's residual estimate, PETSc's default tolerances "
                           "and the full-step Newton iteration around it; DESIGN.md section 3, differences 17-18) are")
        self._lu = {}
        self._J = None
        # not a PETSc option.  -pn_affine_vjp auto|0 (IMEX, direct solves): setupTS(fixed_jacobian=True) with a parameter-free
        # implicit part declares d funcIM/du constant, and its one-sample Jacobian is kept across solves anyway.  If funcIM then
        # passes an affinity check against that matrix (first eager solve; _check_affine), the reverse sweep forms J^T w with
        # ONE dense product per stage instead of differentiating funcIM, and evaluates funcIM itself as Y J^T + funcIM(t, 0-row) --
        # for BASELINE config 5's Conv1d in double precision that replaces ~130 tiny kernels per call of PyTorch's per-sample
        # fallback on ROCm by one product and a one-row call.  Same numbers to round-off;
        # funcIM is called less often (off under -pn_reference_defaults, like every default that changes call counts).
        from . import options as _o2
        ref = _o2.truthy(db.get("pn_reference_defaults"), False) if "pn_reference_defaults" in db else False
        av = str(db.get("pn_affine_vjp", "0" if ref else "auto"))
        self._affine_mode = av == "auto" or _o2.truthy(av, False)
        self._affine = None           # None: not decided yet (decided by the first eager solve)
        self._zero_row = None
        self._tape_rec = None         # per-stage autograd tapes of the step in progress (steppers with an explicit part)
        # hipGraph replay (-pn_graph_capture): factors live in persistent tensors refreshed in place
        self._static_lu = {}          # key -> (LU, pivots, info)
        self._seen_shifts = {}        # key -> shift, recorded by eager solves
        self._J_time = None           # stage time the Jacobian is taken at (first implicit stage of the solve)

    def __del__(self):
        try:
            self.lib.pn_gmres_destroy(self.gmres)
        except Exception:
            pass

    # ---------------------------------------------------------------- small helpers
    def _buf(self, name):
        return self.ode._buf("th_" + name)

    def _dots(self, x, ys):
        """[<x, y_j>] over the GLOBAL batch: with a process group the local products are summed over the
        ranks (one small all-reduce next to the host synchronisation the products need anyway), so Newton
        and GMRES take the decisions -- and build the Krylov space -- of the unsharded solve on every rank."""
        o = self.ode
        vals = o._ops.dots(x, ys)
        if o._world() > 1 and o._pg_global_norm:
            import torch.distributed as dist
            v = torch.tensor(vals, dtype=torch.float64, device=o.device)
            dist.all_reduce(v, op=dist.ReduceOp.SUM, group=o._pg)
            vals = v.tolist()
        return vals

    def _norm(self, x):
        return max(self._dots(x, [x])[0], 0.0) ** 0.5

    def _mass(self, v_flat, transpose=False):
        """M v (or M^T v) as a flat tensor; identity when no mass matrix was given."""
        m = self.ode.mass
        if m is None:
            return v_flat
        o = self.ode
        # pa.py:426-431: torch.matmul(mass, udot) -- udot carries the state's N-D shape in the DLPack
        # mode (mass acts on the second-to-last dimension; a vector state: plain M v) and is flat in the
        # copy mode (mass is numel x numel).  Extension: a (d x d) matrix, d = last state dimension,
        # acting on every row, when neither reference form fits.
        if m.dim() != 2 or m.shape[0] != m.shape[1]:
            raise ValueError("mass must be a square matrix")
        mm = m.T if transpose else m
        size = tuple(o.tensor_size)
        if m.shape[0] == o.n:
            out = torch.mv(mm, v_flat[: o.n])
        elif len(size) >= 2 and m.shape[0] == size[-2]:
            out = torch.matmul(mm, o._shaped(v_flat))
        elif m.shape[0] == size[-1]:
            out = torch.matmul(o._shaped(v_flat), mm.T)
        else:
            raise ValueError("mass is %dx%d but the state has shape %s (%d elements)"
                             % (m.shape[0], m.shape[1], size, o.n))
        r = self._buf("mv_t" if transpose else "mv")
        o._ops.copy(r, out.contiguous().reshape(-1))
        return r

    def _f(self, t, x_flat, which=None):
        o = self.ode
        im = (which or self.which) == "IM"
        fn = o.funcIM if im else o.funcEX
        with torch.no_grad():
            if im and self._affine:
                # affine and row-wise with the kept Jacobian (_check_affine): f(t, Y) = Y J^T + c(t), c(t) = f(t, 0) of ONE row
                n1 = self._J.shape[0]
                if self._zero_row is None:
                    self._zero_row = torch.zeros((1,) + tuple(o.tensor_size[1:]), dtype=o.tensor_dtype, device=o.device)
                k = torch.addmm(fn(t, self._zero_row).reshape(1, n1), x_flat[: o.n].view(-1, n1), self._J.t())
            else:
                k = fn(t, o._shaped(x_flat))
        o.nfe_forward += 1
        return k.contiguous().reshape(-1)

    # ---------------------------------------------------------------- replayed linearisations (_OpGraph)
    @contextlib.contextmanager
    def _capturing(self, graph):
        """Capture into `graph` (shared memory pool); the device entry points launch on the capture stream meanwhile."""
        o, ops = self.ode, self.ode._ops
        if self._op_pool is None:
            self._op_pool = torch.cuda.graph_pool_handle()
        prev = ops._pinned_stream
        with torch.cuda.graph(graph, pool=self._op_pool, capture_error_mode=o.GRAPH_CAPTURE_MODE):
            ops._pinned_stream = ctypes.c_void_p(torch.cuda.current_stream(o.device).cuda_stream)
            try:
                yield
            finally:
                ops._pinned_stream = prev

    GRAPH_CACHE_ENTRIES = 256          # captured linearisations kept (least recently captured dropped first) ...
    GRAPH_CACHE_FRACTION = 0.125       # ... and the share of the device's memory they may hold (measured at the first capture)

    def _graphs_allowed(self):
        o = self.ode
        if self._graph_mode == 0 or o.device.type != "cuda" or not self._device_krylov():
            return False
        if torch.cuda.is_current_stream_capturing() or self.lib.pn_prof_is_enabled():
            return False
        import pnode_amd
        if pnode_amd.GRAPH_REPLAY_SAFE:
            from . import _graphcheck                 # once per process and device
            if not _graphcheck.replay_is_sound(o.device):
                pnode_amd.GRAPH_REPLAY_SAFE = False
        if not pnode_amd.GRAPH_REPLAY_SAFE:
            if self._graph_mode == 1:
                warnings.warn("pnode_amd: -pn_krylov_graph ignored: hipGraph replays are not reliable in this process (see "
                              "-pn_graph_capture)", RuntimeWarning)
            self._graph_mode = 0
            return False
        return True

    def _op_graph(self, t, transpose, X=None):
        """The replayable linearisation of f at stage time t, or None (eager path).  `X`: a state to put into the static
        input before anything is run on it (warm-up, capture and calibration then see ordinary numbers)."""
        if not self._graphs_allowed():
            return None
        o = self.ode
        if self._graph_mode == 2 and not self._autonomous:
            if o._adaptive or isinstance(o.step_size, list):
                return None      # auto: stage times of adaptive / listed steps do not repeat from solve to solve
            if 2 * self._stage_times_this_solve > self._entry_cap:
                return None      # auto: more distinct stage times in ONE solve than the cache holds (every look-up a capture)
        params = o._paramsI if self.which == "IM" else o._paramsE
        fn = o.funcIM if self.which == "IM" else o.funcEX
        # a capture bakes in: the stage time, where the parameters live, the mass matrix, and the train/eval mode of every
        # submodule (BatchNorm, dropout)
        key = (self.which, bool(transpose), None if self._autonomous else float(t),
               tuple(p.data_ptr() for p in params), None if o.mass is None else o.mass.data_ptr(),
               tuple(m.training for m in fn.modules()) if hasattr(fn, "modules") else None)
        self._op_stats[0] += 1
        e = self._op_graphs.get(key)
        if e is not None:
            if self._validate and X is not None:
                # Once per solve: what a capture cannot see is a change of func that is not a change of a parameter's or
                # buffer's CONTENTS -- a Python attribute (a coefficient kept as a float), a swapped submodule.  So the first
                # replay of every solve is checked against one eager evaluation of func on the same state.
                self._validate = False
                fx = e.linearise(X)
                fe = self._f(t, X)
                o.nfe_forward -= 1                       # a check, not an evaluation of the solve (NFE as without graphs)
                tol = 1e-5 if o.tensor_dtype == torch.float32 else 1e-10
                if not torch.allclose(fe, fx, rtol=tol, atol=tol * float(fe.abs().max()) + 1e-300):
                    self._drop_graphs("func no longer computes what was captured (a Python attribute or a submodule changed "
                                      "since the capture?); pass -pn_krylov_graph 0 if func is changed between solves")
                    return None
            return e
        # auto mode: stage times that never repeat (adaptive steps, a time grid longer than the cache) would make every
        # look-up a capture -- give the graphs up then
        looks, caps = self._op_stats
        if self._graph_mode == 2 and caps >= 2 * self._entry_cap and caps * 2 > looks:
            self._drop_graphs("the stage times do not repeat (%d captures in %d look-ups); pass -pn_krylov_autonomous 1 "
                              "if func ignores its time argument" % (caps, looks))
            return None
        try:
            gc.collect()
            if X is not None:
                o._ops.copy(self._buf("lin_x"), X)
            if not transpose and not self._form_decided:
                # The arithmetic FORM of J v is not a matter of timing (two runs, or two ranks, could otherwise
                # differ in their last bits).  It follows from what func IS: the double-VJP identity -- half the GEMMs of a
                # forward-mode pass for a dense net under replay -- unless func contains a convolution in double precision,
                # whose double backward is PyTorch-ROCm's per-sample fallback (three times slower than its forward-mode pass,
                # DESIGN 5.4): then one forward-mode pass of func, where func supports it.  -pn_krylov_graph_form forces one.
                # Timing (below) decides only whether that arithmetic is replayed or launched -- the eager operator of this
                # stepper takes the same form from here on, so the bits do not depend on the verdict either.
                self._form_decided = True
                if self._graph_form == "":
                    conv64 = o.tensor_dtype == torch.float64 and hasattr(fn, "modules") and any(
                        isinstance(m, torch.nn.modules.conv._ConvNd) for m in fn.modules())
                    if conv64 and self._fwd_mode is None:
                        self._probe_fwd_mode(t, X)
                    self._graph_fwd = bool(conv64 and self._fwd_mode)
                if self._fwd_mode is not False or not self._graph_fwd:
                    self._fwd_mode = self._graph_fwd
            before = torch.cuda.memory_reserved(o.device) if self._entry_bytes is None else 0
            e = _OpGraph(self, t, transpose, self._graph_fwd)
            if self._entry_bytes is None:            # what one entry holds (activations of func and of its double backward)
                self._entry_bytes = max(torch.cuda.memory_reserved(o.device) - before, 1)
                total = torch.cuda.get_device_properties(o.device).total_memory
                self._entry_cap = max(4, min(self.GRAPH_CACHE_ENTRIES, int(self.GRAPH_CACHE_FRACTION * total / self._entry_bytes)))
        except Exception as exc:
            self._drop_graphs("capturing func failed (%s: %s); func must not synchronise with the host"
                              % (type(exc).__name__, exc))
            return None
        self._op_stats[1] += 1
        ckey = (type(fn).__module__, type(fn).__qualname__, o.n, str(o.tensor_dtype), self.which)
        if self._graph_mode == 2 and not self._calibrated and ckey in _GRAPH_CHOICE:
            self._calibrated = True                        # decided earlier in this process for the same dynamics and size
            choice = self._agree_on(_GRAPH_CHOICE[ckey])
            self._calibration = ("cached", choice)
            if choice == "eager":
                self._drop_graphs("an earlier timed comparison in this process chose eager launches for this func and size", warn=False)
                return None
        if self._graph_mode == 2 and not self._calibrated:
            # auto: keep the graphs only if they pay.  Once per stepper: one linearisation + eight products, replayed,
            # against the SAME arithmetic through eager launches, both timed to completion on the device.  (A func whose
            # kernels are many and tiny replays several times faster; PyTorch's fp64 convolution fallback on ROCm, which
            # loops over the batch, does not.)  Every rank of a sharded solve takes rank 0's verdict.
            self._calibrated = True
            try:
                tg, te = self._time_graph_against_eager(e, t, transpose)
            except Exception as exc:
                self._drop_graphs("timing the captured func failed (%s: %s)" % (type(exc).__name__, exc))
                return None
            self._calibration = (tg, te, None)
            choice = self._agree_on("eager" if tg > 0.9 * te else "graph")
            _GRAPH_CHOICE[ckey] = choice
            if choice == "eager":
                self._drop_graphs("replaying the captured func is not faster than launching it (%.0f us against %.0f us for a "
                                  "linearisation and eight products)" % (1e6 * tg, 1e6 * te), warn=False)
                return None
        while len(self._op_graphs) >= self._entry_cap:
            self._op_graphs.pop(next(iter(self._op_graphs)))
        self._op_graphs[key] = e
        return e

    def _probe_fwd_mode(self, t, X=None):
        """Does func support forward-mode AD (torch.func.jvp)?  One evaluation, once per stepper."""
        o = self.ode
        fn = o.funcIM if self.which == "IM" else o.funcEX
        x = o._shaped(X if X is not None else torch.zeros(o._npad, dtype=o.tensor_dtype, device=o.device)).detach()
        try:
            with torch.no_grad():
                torch.func.jvp(lambda y: fn(t, y), (x,), (x,))
            self._fwd_mode = True
        except Exception:
            self._fwd_mode = False

    def _agree_on(self, choice):
        """Rank 0's choice for every rank of a sharded solve (the ranks time independently)."""
        o = self.ode
        if not o._sharded():
            return choice
        import torch.distributed as dist
        flag = torch.tensor([1 if choice == "eager" else 0], dtype=torch.int32, device=o.device)
        src = dist.get_global_rank(o._pg, 0) if o._pg is not None else 0
        dist.broadcast(flag, src=src, group=o._pg)
        return "eager" if int(flag.item()) else "graph"

    def _time_graph_against_eager(self, e, t, transpose):
        import time
        o, ops = self.ode, self.ode._ops
        if self._kr is None:
            self._kr = ops.krylov_new(self.restart)
        kr = self._kr
        ops.copy(kr.vin, e.x)                               # finite numbers in the operator's input
        shift = 1.0
        prod = e.product(shift, kr)

        def graph_path():
            e.gA.replay()
            for _ in range(8):
                prod()

        def eager_path():
            jp, _ = self._linearise(t, e.x, transpose)
            for _ in range(8):
                self._apply(jp, shift, kr.vin, kr.w, transpose)

        out = []
        for fn in (graph_path, eager_path):
            fn()
            best = None
            for _ in range(2):
                torch.cuda.synchronize(o.device)
                t0 = time.perf_counter()
                fn()
                torch.cuda.synchronize(o.device)
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            out.append(best)
        return out[0], out[1]

    def _drop_graphs(self, why, warn=True):
        self._graph_mode = 0
        self._op_graphs = {}
        self._graphs_dropped = why
        gc.collect()
        torch.cuda.synchronize(self.ode.device)
        if warn:
            warnings.warn("pnode_amd: the Newton-Krylov solves launch func eagerly from now on: " + why, RuntimeWarning)

    # ---------------------------------------------------------------- Jacobian products
    def _linearise(self, t, x_flat, transpose):
        """One graph of f at x; returns a function v -> J v (or J^T v) on flat tensors."""
        o = self.ode
        with torch.enable_grad():
            xx = o._shaped(x_flat).detach().requires_grad_(True)
            out, wrt = o._func_with_grad(t, xx, self.which)
            if transpose:
                def jt(v):
                    g = torch.autograd.grad(out, xx, o._shaped(v).view(out.shape), retain_graph=True, allow_unused=True)[0]
                    return None if g is None else g.contiguous().reshape(-1)
                return jt, (out, xx, wrt)
            dummy = torch.zeros_like(out, requires_grad=True)
            g = torch.autograd.grad(out, xx, dummy, create_graph=True, allow_unused=True)[0]

        def jv_double_vjp(v):
            if g is None:
                return None
            r = torch.autograd.grad(g, dummy, o._shaped(v).view(g.shape), retain_graph=True, allow_unused=True)[0]
            return None if r is None else r.contiguous().reshape(-1)

        fn = o.funcIM if self.which == "IM" else o.funcEX
        x_const = o._shaped(x_flat).detach()

        def jv(v):
            # forward-mode AD is one dual-number pass of f (about a third of the double-VJP cost);
            # dynamics with operators that lack a forward rule fall back to the double-VJP identity
            if self._fwd_mode is not False:
                try:
                    with torch.no_grad():
                        r = torch.func.jvp(lambda y: fn(t, y), (x_const,), (o._shaped(v).detach(),))[1]
                    self._fwd_mode = True
                    return r.contiguous().reshape(-1)
                except Exception:
                    if self._fwd_mode is True:
                        raise
                    self._fwd_mode = False
            return jv_double_vjp(v)
        return jv, (out, xx, wrt)

    def _apply(self, jprod, shift, v, out, transpose):
        """out = shift*M v - J v   (or the transposed operator)."""
        ops = self.ode._ops
        mv = self._mass(v, transpose)
        jv = jprod(v)
        if jv is None:
            ops.lincomb(out, [mv], [shift])
        else:
            ops.lincomb(out, [mv, jv], [shift, -1.0])

    def _lincomb_terms(self, out, terms):
        """out = sum c*x over [(x, c)] of any length (out may be the first x)."""
        ops = self.ode._ops
        first = terms[:8]
        ops.lincomb(out, [x for x, _ in first], [c for _, c in first])
        k = 8
        while k < len(terms):
            chunk = terms[k:k + 7]
            ops.lincomb(out, [out] + [x for x, _ in chunk], [1.0] + [c for _, c in chunk])
            k += 7

    # ---------------------------------------------------------------- GMRES
    def _device_krylov(self):
        ops = self.ode._ops
        return (self._krylov_mode == "device" and hasattr(ops, "krylov_begin")
                and self.restart <= getattr(ops, "MAX_KRYLOV_RESTART", 0))

    def _reduce_fn(self):
        """Sum of a small device tensor over the ranks (stream-ordered), or None for a one-rank solve."""
        o = self.ode
        # issued whatever the group's size (a one-rank group reduces to itself): that is how the RCCL path -- the
        # all-reduce of the product block between the deferred parts of pn_krylov_step -- is exercised on one GPU
        if o._sharded() and o._pg_global_norm:
            import torch.distributed as dist
            return lambda v: dist.all_reduce(v, op=dist.ReduceOp.SUM, group=o._pg)
        return None

    def _gmres(self, jprod, shift, rhs, x, transpose, graph=None, tag=0):
        """Solve A x = rhs from x = 0 (A = shift*M - J or its transpose); returns #iterations.  `jprod`: v -> J v (or
        J^T v) on flat tensors; `graph`: an _OpGraph whose linearisation is current, in place of jprod; `tag`: which
        linear solve of its kind this is (the Newton iteration index): solves with the same tag need about the same
        number of iterations from step to step, which sizes the first chunk of the device-resident loop."""
        if self._device_krylov():
            if self._kr is None:
                self._kr = self.ode._ops.krylov_new(self.restart)
            kr = self._kr
            fused = False
            if graph is not None:
                fused = self._reduce_fn() is None          # (a sharded solve all-reduces the products between the parts of a step)
                op = graph.product(shift, kr, with_step=fused)
            else:
                op = lambda: self._apply(jprod, shift, kr.vin, kr.w, transpose)
            return self._gmres_device(op, rhs, x, transpose, tag, fused)
        if graph is not None:
            jprod = graph.eager_product()
        return self._gmres_host(jprod, shift, rhs, x, transpose)

    def _gmres_device(self, op, rhs, x, transpose, tag=0, fused=False):
        """GMRES with its state on the device.  `op()` enqueues  kr.w <- A kr.vin  (`fused`: and the iteration's kernels).  The host enqueues whole chunks of
        iterations (operator + pn_krylov_step) and looks at the device's stop flag once per chunk; the first chunk is
        as long as the previous solve of this kind was (stage systems of consecutive steps need the same number of
        iterations give or take one), later ones two iterations.  Launches past convergence are no-ops on the device
        (the operator applications among them are wasted work, nothing else)."""
        ops, kr, m = self.ode._ops, self._kr, self.restart
        reduce = self._reduce_fn()
        ops.lincomb(x, [rhs], [0.0])
        ops.krylov_begin(kr, rhs, self.ksp_rtol, self.ksp_atol, self.ksp_max_it, True, reduce)
        key = (bool(transpose), tag)
        chunk = max(1, min(self._its_guess.get(key, self._its_guess.get((bool(transpose), 0), 4)), m))
        k = 0
        while True:
            for _ in range(chunk):
                if k >= m:
                    break
                op()                                        # fused: the replayed graph contains the iteration's kernels too
                if not fused:
                    ops.krylov_step(kr, k, reduce)
                k += 1
            ops.krylov_close(kr, x)                         # acts only when the cycle has ended
            stop, kdone, total, res = ops.krylov_status(kr)
            self.host_syncs += 1
            if stop:
                break
            chunk = 2
            if kdone >= m:                                  # restart: x is updated, continue from the true residual
                r = self._buf("r")
                ops.copy(kr.vin, x)
                op()                                        # (fused: the step inside returns at its entry check -- the cycle is closed)
                ops.lincomb(r, [rhs, kr.w], [1.0, -1.0])
                ops.krylov_begin(kr, r, self.ksp_rtol, self.ksp_atol, self.ksp_max_it, False, reduce)
                k = 0
        if stop == 5:
            raise _lib.PnError("KSP diverged (GMRES breakdown after %d iterations): the %sstage system shift*M - J "
                               "is singular on its Krylov space -- for a DAE, the algebraic part of dfunc/du has lost rank"
                               % (total, "transposed " if transpose else ""))
        if stop == 4:
            raise _lib.PnError("KSP diverged: not-a-number in the %sstage system (GMRES iteration %d)"
                               % ("transposed " if transpose else "", total))
        self.linear_its += total
        self.second_passes += getattr(kr, "second_passes", 0)
        if 0 < total <= m:
            self._its_guess[key] = total
        if self._its_log is not None:
            self._its_log.append((1 if transpose else 0, total))
        return total

    def _gmres_host(self, jprod, shift, rhs, x, transpose):
        """Round 2's loop (-pn_krylov host, and the CPU test stand-in without the device entry points): the small dense
        part on the host (pn_gmres_*), one stream synchronisation per iteration."""
        ops, lib, m = self.ode._ops, self.lib, self.restart
        while len(self.V) < m + 1:
            self.V.append(ops.empty(self.ode._npad))
        V = self.V
        w, r = self._buf("w"), self._buf("r")
        bnorm = self._norm(rhs)
        tol = max(self.ksp_rtol * bnorm, self.ksp_atol)
        ops.lincomb(x, [rhs], [0.0])
        if bnorm == 0.0 or bnorm <= self.ksp_atol:
            return 0
        total = 0
        first = True
        while total < self.ksp_max_it:
            if first:
                ops.copy(r, rhs)
                beta = bnorm
                first = False
            else:
                self._apply(jprod, shift, x, w, transpose)
                ops.lincomb(r, [rhs, w], [1.0, -1.0])
                beta = self._norm(r)
                if beta <= tol:
                    break
            ops.lincomb(V[0], [r], [1.0 / beta])
            check(lib.pn_gmres_begin(self.gmres, beta))
            res = ctypes.c_double(beta)
            k = -1
            for k in range(m):
                self._apply(jprod, shift, V[k], w, transpose)
                # classical Gram-Schmidt with ONE host synchronisation: all <w,V_j> and <w,w> come
                # from the same multi-dot; ||w - sum h_j V_j|| follows from Pythagoras.  When that
                # loses digits (strong cancellation) the step is repeated the two-pass way.
                d = self._dots(w, V[: k + 1] + [w])
                ww, d = d[-1], d[:-1]
                rest = ww - sum(c * c for c in d)
                if rest > 0.25 * ww and rest > 0.0:
                    hk1 = rest ** 0.5
                    h = d + [hk1]
                    terms = [(w, 1.0 / hk1)] + [(V[j], -d[j] / hk1) for j in range(k + 1)]
                    self._lincomb_terms(V[k + 1], terms)
                else:
                    # two passes (CGS2), two synchronisations: the second multi-dot returns the
                    # correction coefficients and ||w||^2 together; after the first pass the
                    # corrections are tiny, so Pythagoras is safe
                    self._lincomb_terms(w, [(w, 1.0)] + [(V[j], -d[j]) for j in range(k + 1)])
                    d2 = self._dots(w, V[: k + 1] + [w])
                    ww2, d2 = d2[-1], d2[:-1]
                    rest = max(ww2 - sum(c * c for c in d2), 0.0)
                    hk1 = rest ** 0.5
                    h = [a + b for a, b in zip(d, d2)] + [hk1]
                    if hk1 > 0.0:
                        self._lincomb_terms(V[k + 1], [(w, 1.0 / hk1)] + [(V[j], -d2[j] / hk1) for j in range(k + 1)])
                check(lib.pn_gmres_column(self.gmres, k, (ctypes.c_double * (k + 2))(*h), ctypes.byref(res)))
                total += 1
                if res.value <= tol or hk1 == 0.0 or total >= self.ksp_max_it:
                    break
            y = (ctypes.c_double * (k + 1))()
            if lib.pn_gmres_solve(self.gmres, k, y):
                # PETSc: KSP_DIVERGED_BREAKDOWN -> SNES_DIVERGED_LINEAR_SOLVE -> TS_DIVERGED_NONLINEAR_SOLVE
                raise _lib.PnError("KSP diverged (GMRES breakdown after %d iterations): the %sstage system shift*M - J "
                                   "(shift = %g) is singular on its Krylov space -- for a DAE, the algebraic part of "
                                   "dfunc/du has lost rank (%s)" % (k + 1, "transposed " if transpose else "", shift,
                                                                  lib.pn_last_error().decode()))
            ys = list(y)
            self._lincomb_terms(x, [(x, 1.0)] + [(V[j], ys[j]) for j in range(k + 1)])
            if res.value <= tol:
                break
        self.linear_its += total
        return total

    # ---------------------------------------------------------------- direct stage solves (linear_solver="torch")
    def _direct_factor(self, t, u_flat, shift):
        """LU of shift*M - J, J = d f/du of the first batch row for the implicitly treated f (pa.py:474-508),
        cached per shift for the duration of one odeint (pa.py:792-799 resets the factor at every odeint)."""
        o = self.ode
        key = round(shift, 12)
        if o.device.type == "cuda" and torch.cuda.is_current_stream_capturing():
            # a captured sweep only reads the factors; graph_prepare() refreshes them before every replay
            if key not in self._static_lu:
                raise _lib.PnError("graph capture: shift %r was not seen in the eager warm-up solves" % shift)
            return self._static_lu[key][:2]
        if self._J is None:
            self._J = self._jacobian(t, o._shaped(u_flat))
            self._J_time = t
            if self._affine is None:
                self._affine = self._check_affine(t, u_flat)
        if key not in self._lu:
            if len(self._lu) >= 64:              # adaptive steps: one factorisation per distinct shift; keep the newest
                self._lu.pop(next(iter(self._lu)))        # (an evicted one is recomputed from the same frozen J: same bits)
            self._lu[key] = torch.linalg.lu_factor(self._shifted(shift))
            self._seen_shifts[key] = shift
        return self._lu[key]

    def _check_affine(self, t, u_flat):
        """Is the implicitly treated f affine in u with the kept one-sample Jacobian, f(t, y) = y J^T + c(t) row by row?
        Asked only where the user has declared the Jacobian constant (fixed_jacobian=True, no trainable parameter in f: the
        factors are kept across solves) and the solves are direct; answered by evaluating f at two points, at two times,
        on up to four rows.  One host synchronisation, once per solver."""
        o = self.ode
        if not (self._affine_mode and self.which == "IM" and self.direct and self._reuse_factors() and o.mass is None):
            return False
        fn = o.funcIM
        n1 = self._J.shape[0]
        allrows = o._shaped(u_flat).detach().reshape(-1, n1)
        B = allrows.shape[0]
        rows = allrows[sorted({0, min(1, B - 1), B // 2, B - 1})]          # spread over the batch: f must act on every row alike
        shape = (rows.shape[0],) + tuple(o.tensor_size[1:])
        tol = 1e-10 if o.tensor_dtype == torch.float64 else 1e-4
        try:
            with torch.no_grad():
                worst = scale = 0.0
                for tt in (t, t + 0.37):
                    f0 = fn(tt, torch.zeros(shape, dtype=rows.dtype, device=rows.device)).reshape(-1, n1)
                    for y in (rows, 1.0 - 2.0 * rows):
                        lin = y @ self._J.T
                        d = fn(tt, y.reshape(shape)).reshape(-1, n1) - f0 - lin
                        worst = max(worst, float(d.abs().max()))
                        scale = max(scale, float(lin.abs().max()))
            return bool(worst <= tol * max(scale, 1e-300))
        except Exception:
            return False

    def _jt_rows(self, w_flat):
        """J^T applied to every row of the cotangent: (B, n1) @ J -- the VJP of an affine f."""
        o = self.ode
        n1 = self._J.shape[0]
        return torch.mm(w_flat[: o.n].view(-1, n1), self._J).reshape(-1)

    def _jacobian(self, t, u):
        o = self.ode
        fn = o.funcIM if self.which == "IM" else o.funcEX
        with torch.no_grad():
            jac = torch.func.jacrev(lambda y: fn(t, y))(u[0:1].detach().clone())
        n1 = u[0:1].numel()
        return jac.reshape(n1, n1)

    def _shifted(self, shift):
        n1 = self._J.shape[0]
        m = self.ode.mass
        M = torch.eye(n1, dtype=self._J.dtype, device=self._J.device) if m is None else m.to(self._J.dtype)
        return shift * M - self._J

    def _use_direct(self):
        if not self.direct:
            return False
        if not self._direct_ok():
            if not getattr(self, "_direct_warned", False):
                self._direct_warned = True
                import warnings
                warnings.warn("pnode_amd: linear_solver='torch' needs a mass matrix that acts on one sample (d x d, d = last "
                              "state dimension); solving matrix-free instead", RuntimeWarning)
            return False
        return True

    def _direct_ok(self):
        """The direct solve works on one sample's Jacobian: a mass matrix must act per sample too."""
        m = self.ode.mass
        if m is None:
            return True
        n1 = self.ode.n // max(int(self.ode.tensor_size[0]), 1) if len(self.ode.tensor_size) > 1 else self.ode.n
        return m.dim() == 2 and m.shape[0] == m.shape[1] == n1 and m.shape[0] == self.ode.tensor_size[-1]

    def _reuse_factors(self):
        """setupTS(fixed_jacobian=True) declares d f/du constant across solves (pa.py:582).  The
        reference recomputes it at every odeint anyway (pa.py:792-799); here the factors are kept
        when, in addition, the implicitly treated f has no trainable parameter -- then nothing the optimiser does can
        change them and the results are the same."""
        return bool(self.ode.fixed_jacobian) and (self.ode.npIM if self.which == "IM" else self.ode.np) == 0

    # ---------------------------------------------------------------- hipGraph support
    def capturable(self):
        """True when a whole sweep has no host synchronisation: one direct solve per implicit stage
        (-snes_type ksponly + linear_solver="torch"); Newton/GMRES iterations read norms on the host."""
        return self.direct and self.ksponly and self._direct_ok()

    def graph_prepare(self, u0):
        """Before a captured sweep runs: d f/du at the solve's first state with the CURRENT
        parameters and its LU factors for every shift, written into the tensors the graph reads
        (what an eager solve does at its first implicit stage, pa.py:474-508, 792-799)."""
        o = self.ode
        if self._reuse_factors() and len(self._static_lu) == len(self._seen_shifts) and self._static_lu:
            return
        self._J = self._jacobian(self._J_time, u0.detach().reshape(o.tensor_size))
        for key, shift in self._seen_shifts.items():
            A = self._shifted(shift)
            if key not in self._static_lu:
                LU, piv, info = torch.linalg.lu_factor_ex(A, check_errors=False)
                self._static_lu[key] = (LU, piv, info)
            else:
                torch.linalg.lu_factor_ex(A, check_errors=False, out=self._static_lu[key])

    def _direct_solver(self, t, u_flat, shift, transpose):
        o, ops = self.ode, self.ode._ops
        LU, piv = self._direct_factor(t, u_flat, shift)
        n1 = LU.shape[0]

        def solve(rhs, out):
            R = rhs[: o.n].view(-1, n1)
            # rows x with x (shift I - J)^T = r  <=>  (shift I - J) x^T = r^T ; transposed system: x (shift I - J) = r
            X = torch.linalg.lu_solve(LU, piv, R, left=False, adjoint=not transpose)
            ops.copy(out, X.contiguous().reshape(-1))
        return solve

    # ---------------------------------------------------------------- Newton
    def _newton(self, ts, shift, Z, b, X, linear_solve=None):
        """Solve  shift*M (X - Z) - f(ts, X) - b = 0  for X (updated in place; its entry value is
        the initial guess).  SNES newtonls restated: full steps, halved while the residual norm
        grows; `-snes_type ksponly` = exactly one linear solve.  `linear_solve(rhs, out)`, when
        given, replaces GMRES (the reference's linear_solver="torch" direct solve)."""
        o, ops = self.ode, self.ode._ops
        G, dX, d = self._buf("G"), self._buf("dX"), self._buf("d")
        # Krylov path on the HIP device: f and its linearisation at X come from ONE replayed graph -- the evaluation
        # the residual needs is also the linearisation the next linear solve needs
        ent = self._op_graph(ts, False, X) if linear_solve is None else None

        def residual():
            if ent is not None:
                fx = ent.linearise(X)
                o.nfe_forward += 1
            else:
                fx = self._f(ts, X)
            if o.mass is None:
                xs, cs = [X, Z, fx], [shift, -shift, -1.0]
            else:
                ops.lincomb(d, [X, Z], [shift, -shift])
                xs, cs = [self._mass(d), fx], [1.0, -1.0]
            if b is not None:
                xs, cs = xs + [b], cs + [-1.0]
            ops.lincomb(G, xs, cs)
            # SNESKSPONLY takes one full step whatever the residual is: no norm, no host synchronisation
            return 1.0 if self.ksponly else self._norm(G)

        fnorm = fnorm0 = residual()
        for it in range(self.snes_max_it):
            if not self.ksponly and (fnorm <= self.snes_atol or fnorm <= self.snes_rtol * fnorm0) and it > 0:
                break
            if fnorm == 0.0:
                break
            ops.lincomb(G, [G], [-1.0])                     # right-hand side -G
            if linear_solve is not None:
                linear_solve(G, dX)
            elif ent is not None:
                self._gmres(None, shift, G, dX, False, graph=ent, tag=it)      # the last residual() linearised f at this X
            else:
                jv, _ = self._linearise(ts, X, False)
                self._gmres(jv, shift, G, dX, False, tag=it)
            self.newton_its += 1
            if self.ksponly:                                # SNESKSPONLY: one solve, full step, no re-evaluation
                ops.lincomb(X, [X, dX], [1.0, 1.0])
                break
            lam = 1.0
            ops.copy(d, X)                                  # keep the iterate for the step control
            while True:
                ops.lincomb(X, [d, dX], [1.0, lam])
                fnew = residual()
                if fnew <= fnorm or lam < 1e-3 or not (fnew == fnew):
                    break
                lam *= 0.5
            if not (fnew == fnew) or fnew == float("inf"):
                raise _lib.PnError("SNES diverged: function norm is NaN/Inf (implicit stage at t=%g)" % ts)
            fnorm = fnew
            xnorm = self._norm(X)
            dnorm = lam * self._norm(dX)
            if dnorm <= self.snes_stol * xnorm:
                break
        else:
            raise _lib.PnError("SNES did not converge in %d iterations (implicit stage at t=%g)" % (self.snes_max_it, ts))

    # ---------------------------------------------------------------- one step
    def _step(self, tn, h, u, unew):
        """Solve the stage equation starting from X = u_n; writes u_{n+1}; returns X (flat)."""
        ops = self.ode._ops
        theta, shift = self.theta, 1.0 / (self.theta * h)
        ts = tn + h if self.endpoint else tn + theta * h
        X = unew if self.endpoint else self._buf("X")
        ops.copy(X, u)
        b = None
        if self.endpoint:
            b = self._buf("b")
            ops.lincomb(b, [self._f(tn, u)], [(1.0 - theta) / theta])
        lin = self._direct_solver(ts, u, shift, False) if self._use_direct() else None
        self._newton(ts, shift, u, b, X, lin)
        if not self.endpoint:
            ops.lincomb(unew, [u, X], [1.0 - 1.0 / theta, 1.0 / theta])
        return X

    def _implicit_stages(self):
        """Implicit stage solves per step (distinct stage times a step contributes to the capture cache)."""
        return 1

    # ---------------------------------------------------------------- stage storage
    def nstage(self):
        """Stage vectors a reversed step needs besides the state at its start (theta: the stage solution X)."""
        return 1

    def _do_step(self, tn, h, u, unew, stage_dest):
        """One step u -> unew; the stage values go to stage_dest(i).  Returns them."""
        X = self._step(tn, h, u, unew)
        dst = stage_dest(0)
        self.ode._ops.copy(dst, X)
        return [dst]

    # ---------------------------------------------------------------- step-size control (TSAdapt basic)
    def lte_available(self):
        """The theta methods have no embedded pair; PETSc estimates their local truncation error from the last three
        solutions instead (TSEvaluateWLTE_Theta)."""
        return type(self) is ThetaStepper

    def error_begin(self):
        self._prev = None                 # (state at the start of the previous accepted step, its size)

    def error_norm(self, h, u, unew):
        """Returns False when no estimate exists yet (first step: accepted with its size unchanged).  Otherwise the
        WRMS norm of the estimate lands in the pinned scalar: with a = 1 + h_prev/h,
            LTE ~ X/a - X0/(a-1) + Xprev/(a(a-1)),   X = unew, X0 = u, Xprev = the state one step earlier
        (a scaled second backward difference on the non-uniform grid; the controller takes order 2 for it).  Restated from
        memory of PETSc's theta.c: PARITY UNPINNED."""
        if self._prev is None:
            return False
        o, ops = self.ode, self.ode._ops
        xprev, h_prev = self._prev
        a = 1.0 + h_prev / h
        E = self._buf("E")
        ops.lincomb(E, [unew, u, xprev], [1.0 / a, -1.0 / (a - 1.0), 1.0 / (a * (a - 1.0))])
        ops.combine_wrms(None, unew, [E], [0.0], [1.0], o._atol, o._rtol)
        return True

    def error_accept(self, h, u):
        """The step from `u` with size `h` was accepted: it becomes the previous step of the next estimate."""
        keep = self._buf("prev")
        self.ode._ops.copy(keep, u)
        self._prev = (keep, h)

    # ---------------------------------------------------------------- forward sweep
    def odeint(self, u0, t, save):
        """TSSolve for the one-step implicit / IMEX steppers.  What the reverse sweep needs is kept by the
        same trajectory store and scheduler as on the explicit path (TSTrajectory applies to every TS type,
        pa.py:771-775): -ts_trajectory_solution_only 0 keeps the state and the stage values of every step,
        the default keeps the states and re-solves a step's stages when it is reversed, and
        -ts_trajectory_max_cps_ram N keeps at most N checkpoints and re-advances between them.  All modes
        replay the same arithmetic, so gradients are identical bit for bit."""
        o, ops, lib, ts = self.ode, self.ode._ops, self.lib, self.ode._ts
        o.sol_times = t.detach().cpu().to(dtype=torch.float64)
        T = int(t.shape[0])
        times = o.sol_times.tolist()
        dt0 = float(o.step_size[0] if isinstance(o.step_size, list) else o.step_size)
        check(lib.pn_ts_begin(ts, 0.0, dt0, T, (ctypes.c_double * T)(*times)))
        o._span_begin(T)
        solution = ops.empty((T,) + tuple(o.tensor_size))
        sol_flat = solution.view(T, -1)
        self.newton_its = self.linear_its = self.host_syncs = self.second_passes = 0
        self._validate = True
        nfix = lib.pn_ts_count_fixed_steps(ts)                  # -1: adaptive
        self._stage_times_this_solve = max(int(nfix), 0) * self._implicit_stages()
        if self._its_log is not None:
            del self._its_log[:]
        if not self._reuse_factors() and not (o.device.type == "cuda" and torch.cuda.is_current_stream_capturing()):
            # pa.py:792-799: refactor at every odeint.  NOT while a sweep is being captured: a captured sweep reads the static
            # factors of graph_prepare() only, and the eager cache belongs to the eager solve around it -- the call that
            # validates a capture runs the reverse sweep eagerly AFTER the forward sweep was captured, and a cache emptied
            # here made it refactor with the LAST step's shift, which differs from the first step's in its last bits
            # (matched steps): factors, and every transposed solve, one ulp off the replay's (round 4's "tried and reverted").
            self._lu, self._J = {}, None
        nst = self.nstage()
        o._tmode = o._pick_traj_mode(1 + nst) if save else o._traj_mode
        with_stages = o._tmode == _lib.PN_TRAJ_ALL or o._budget_stages
        if save:
            traj = o._traj = o._new_trajectory(1 + nst if with_stages else 1, o._tmode)
            if o._tmode == _lib.PN_TRAJ_BUDGET and not isinstance(o.step_size, list):
                total = lib.pn_ts_count_fixed_steps(ts)
                if total > 0:
                    check(lib.pn_traj_set_total(traj.handle, total))
        else:
            traj = o._traj = None
        self.traj = traj
        # stage autograd tapes of the explicitly treated part (steppers that have one: ARKIMEX), kept for the reverse sweep
        # under the rules of the explicit RK path (-pn_trajectory_retain_graph auto|1|0; store-all trajectories only: the
        # tape's input IS the stage value in its trajectory slot).  The reverse sweep then runs only the backward half of
        # funcEX's VJPs (the reference re-evaluates, pa.py:66-68).  Same bits.
        keep_tape = bool(save and o._tmode == _lib.PN_TRAJ_ALL and o._retain_graph != 0 and getattr(self, "tapes_ex", False))
        tape_budget = None
        if keep_tape and o._retain_graph == 2:
            tape_budget = o._tape_budget()
            keep_tape = tape_budget is not None and tape_budget > 0
        o._tapes = {} if keep_tape else None
        tape_steps = 1 << 62
        pingpong = [self._buf("u_a"), self._buf("u_b")]
        state = {"pp": 0, "slot": -1}

        def state_home(step):
            if traj is not None:
                slot = traj.fwd_slot(step)
                if slot >= 0:
                    state["slot"] = slot
                    traj.stage_step.pop(slot, None)
                    return traj.claim(slot)
            state["slot"] = -1
            state["pp"] ^= 1
            return pingpong[state["pp"]].view(1, -1)

        cur = state_home(0)
        cur_slot = state["slot"]
        ops.copy(cur[0], u0.detach().contiguous().reshape(-1))
        if T > 1:
            ops.copy(sol_flat[0], cur[0])
        tt, hh = ctypes.c_double(), ctypes.c_double()
        acc, hit, done = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(0)
        finished = not (times[-1] > (0.0 if T == 1 else times[0]))
        adaptive = bool(o._adaptive)
        if adaptive:
            self.error_begin()
        while not finished:
            step = lib.pn_ts_steps(ts)
            nxt = state_home(step + 1)
            nxt_slot = state["slot"]
            keep = with_stages and cur_slot >= 0
            dest = (lambda i, c=cur: c[1 + i]) if keep else (lambda i: self._buf("ys%d" % i))
            while True:                                # attempts: rejected ones rewrite the same buffers
                check(lib.pn_ts_attempt(ts, ctypes.byref(tt), ctypes.byref(hh)))
                tn, h = tt.value, hh.value
                self._tape_rec = [None] * nst if (keep_tape and keep) else None
                self._do_step(tn, h, cur[0], nxt[0], dest)
                enorm = -1.0
                if adaptive and self.error_norm(h, cur[0], nxt[0]):
                    enorm = o._global_enorm(ops.read_enorm())
                check(lib.pn_ts_judge(ts, enorm, ctypes.byref(acc), ctypes.byref(hit), ctypes.byref(done)))
                if acc.value:
                    break
            if adaptive:
                self.error_accept(h, cur[0])
            if self._tape_rec is not None:
                o._tapes[step] = self._tape_rec              # (of the accepted attempt: its buffers are not rewritten)
                self._tape_rec = None
                if tape_budget is not None and tape_budget != float("inf"):
                    if step == 0:                             # one measurement: what a step's tapes (and its slot) take
                        from .petsc_adjoint import _mem_now
                        per_step = max(_mem_now(o.device)[0] - o._tape_mem0, 1)
                        tape_steps = int(tape_budget // per_step) - 1
                    if step + 1 >= tape_steps:
                        keep_tape = False                     # later steps re-evaluate funcEX in the reverse sweep
                        o._tape_all_fit = False
            if keep and o._budget_stages:
                traj.stage_step[cur_slot] = step
            if traj is not None and cur_slot >= 0:
                traj.seal(cur_slot)
            cur, cur_slot = nxt, nxt_slot
            stepno = lib.pn_ts_steps(ts)
            tnew = lib.pn_ts_time(ts)
            o._span_post_step(T, times, hit.value, done.value, stepno, tnew, cur[0], sol_flat)
            if o._monitor:
                print("%d TS dt %g time %g" % (stepno, h, tnew))
            finished = bool(done.value)
        o._nsteps = lib.pn_ts_steps(ts)
        if T == 1:
            ops.copy(sol_flat[0], cur[0])
        else:
            o._span_end(T)
        return solution

    def _stages_of(self, step):
        """(state at the start of `step`, its stage values): read from the trajectory, or re-solved from the
        nearest kept state (TSTrajectoryGet), storing on the way what the checkpoint plan asks for."""
        o, traj, nst = self.ode, self.ode._traj, self.nstage()
        fs, fl, stores = traj.rev_plan(step)
        view = traj.view(fl)
        if fs == step and (o._tmode == _lib.PN_TRAJ_ALL or (o._budget_stages and traj.stage_step.get(fl) == step)):
            return view[0], [view[1 + i] for i in range(nst)]
        cur, cur_slot, cur_view = view[0], fl, view
        k, pp = fs, 0
        while k < step:                   # re-advance k -> k+1
            tn, h = o._step_info(k)
            if (k + 1) in stores:
                nxt_slot = stores[k + 1]
                nxt_view = traj.claim(nxt_slot)
                nxt = nxt_view[0]
                traj.stage_step.pop(nxt_slot, None)
            else:
                pp ^= 1
                nxt_slot, nxt_view = -1, None
                nxt = self._buf("r_a" if pp else "r_b")
            keep = o._budget_stages and cur_slot >= 0
            dest = (lambda i, c=cur_view: c[1 + i]) if keep else (lambda i: self._buf("ys%d" % i))
            self._do_step(tn, h, cur, nxt, dest)
            if keep:
                traj.stage_step[cur_slot] = k
                traj.seal(cur_slot)                  # (disk tier) the checkpoint now carries its stage values
            if nxt_slot >= 0 and not o._budget_stages:
                traj.seal(nxt_slot)                  # (disk tier) a new state-only checkpoint is complete
            cur, cur_slot, cur_view = nxt, nxt_slot, nxt_view
            k += 1
        tn, h = o._step_info(step)
        stages = self._do_step(tn, h, cur, self._buf("r_c"), lambda i: self._buf("ys%d" % i))
        return cur, stages

    # ---------------------------------------------------------------- reverse sweep
    def adjoint_steps(self, nsteps, forcing):
        """TSAdjointStep_Theta over `nsteps` steps, newest first; then lambda += forcing."""
        o, ops = self.ode, self.ode._ops
        lam, theta = o.adj_u_flat, self.theta
        nu, rhs = self._buf("nu"), self._buf("rhs")
        for r in range(nsteps):
            step = o._rev_next
            tn, h = o._step_info(step)
            u, (X,) = self._stages_of(step)
            shift = 1.0 / (theta * h)
            ts = tn + h if self.endpoint else tn + theta * h
            ops.lincomb(rhs, [lam], [shift if self.endpoint else shift / theta])
            direct = self._use_direct()
            ent = None if direct else self._op_graph(ts, True, X)
            stable = True
            if ent is not None:
                # replayed linearisation: J^T products and the parameter cotangents through the same captured graph
                ent.linearise(X)
                self._gmres(None, shift, rhs, nu, True, graph=ent)
                gp = ent.param_cotangents(nu, self._kr) if o.np > 0 else []
                stable = False                           # static outputs of the replayed graph
            else:
                jt, (out, xx, wrt) = self._linearise(ts, X, True)
                if direct:
                    self._direct_solver(ts, u, shift, True)(rhs, nu)       # frozen one-sample Jacobian, as the reference
                else:
                    self._gmres(jt, shift, rhs, nu, True)
                # parameter part at the stage point: (df/dp)_X^T nu through the same graph
                gp = []
                if o.np > 0:
                    gp = torch.autograd.grad(out, wrt, o._shaped(nu).view(out.shape), allow_unused=True)
                    gp = [None if g is None else g.to(o.tensor_dtype).contiguous() for g in gp]
            if o.np > 0:
                o._add_param_grads(theta * h, gp, stable=stable, cotangent=nu)
            o.nfe_backward += 1
            mtnu = self._mass(nu, transpose=True)
            if self.endpoint:
                gy, gp = o._vjp(tn, u, nu)
                xs, cs = [mtnu], [1.0]
                if gy is not None:
                    xs, cs = xs + [gy], cs + [(1.0 - theta) * h]
                ops.lincomb(lam, xs, cs)
                if o.np > 0:
                    o._add_param_grads((1.0 - theta) * h, gp)
            else:
                ops.lincomb(lam, [lam, mtnu], [1.0 - 1.0 / theta, 1.0])
            if o._pend_g and (o._accum_mode == "step" or len(o._pend_g) + 2 > o._accum_cap):
                o._flush_param_accum()
            o._traj.rev_done(step)
            o._rev_next = step - 1
        if forcing is not None:
            ops.lincomb(lam, [lam, forcing], [1.0, 1.0])
