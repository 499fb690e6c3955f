#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path on MI355X.

Workload (the configuration BASELINE.json's metric/north_star is
quoted on): batch 4096 x state_dim 512 fp32, func = 3x[Linear(512,512)+Tanh]+Linear(512,512)
(W ~ N(0,0.02), b = 0), rk4 fixed step h = 0.01, 100 time steps, adjoint on, stages stored.
One bench "step" = one forward sweep + one reverse (discrete adjoint) sweep over the 100 time
steps; value = time-steps/s (fwd+adjoint) summed over ranks (each rank integrates its own
batch shard of 4096 trajectories: weak scaling, one all-reduce of the parameter gradient per
backward over RCCL).  The timed region runs the product's defaults for this option set: the two
sweeps replayed from hipGraphs (--mode eager for plain stream launches) and, HBM permitting, the
stage autograd tapes of the forward sweep kept for the reverse sweep (-pn_trajectory_retain_graph auto;
`variants.recompute` is the same run with the reference's per-stage re-evaluation of func, pa.py:66-74).
--config c4 runs BASELINE config 4's shard instead (conv block, 128 x 64 x 32 x 32 per GPU, rk4, t=[1]); --config c2 the
batched spiral (4096 x 2, rk4 x 100: launch-bound), --config c3b BASELINE config 3 as written (dopri5 adaptive, one scalar
all-reduce per step attempt when sharded), --config c5 the Burgers IMEX shard (64 x 1024 fp64 per GPU, ARKIMEX 3 + ksponly +
batched LU; --strong: 512 global).  --dtype f64 runs c3a / c2 / c3b in the reference's CI precision.

  python bench.py [--gpus N --steps K --warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

Without WORLD_SIZE in the environment `--gpus N` (N > 1) starts the N ranks itself (fresh child processes
under torch.distributed.run; this process never touches a GPU) and fails if the box has fewer GPUs.

Rank 0 prints ONE JSON line.  `roofline` covers the solver kernels (the pn_* entry points:
stage AXPYs, adjoint cotangents, lambda update AND the parameter-sensitivity accumulation -- SURVEY
8(d): 32*N*w + s*3*np*w algorithmic bytes per time step), timed live with HIP events bound to each
dispatch (hipExtLaunchKernelGGL start/stop events); func's GEMMs are PyTorch/hipBLASLt and
are not part of it.  `cpu_baseline` is the oracle (restated PETSc path) on the host cores.
"""
import argparse
import ctypes
import json
import os

# see pnode_amd/__init__.py: hipGraph replays of PyTorch reductions need this on ROCm 7.2
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0           # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
MFMA_F32_PEAK_TFLOPS = 157.3    # dense fp32-input MFMA (v_mfma_f32_32x32x2_f32: 64 FLOP/clk/SIMD), same guide
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA (the guide: ~2.5 PFLOP/s; AMD's headline figure includes 2:1 sparsity)
MFMA_F64_PEAK_TFLOPS = 78.6     # fp64 matrix (v_mfma_f64_16x16x4_f64): AMD's MI355X data sheet (the guide does not list an fp64 MFMA rate)
ALG_VECTORS_PER_STEP = 32       # rk4 fwd (15) + adjoint (17) vector moves per step


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", choices=["c3a", "c4", "c2", "c3b", "c5"], default="c3a",
                    help="c3a (default, the headline): MLP 4096 x 512 rk4 x 100; c4: conv block 128 x 64 x 32 x 32 per GPU, "
                         "rk4, t = [1.0], --nt steps (BASELINE config 4, the one that names 8 GPUs); c2: batched spiral 4096 x 2, "
                         "rk4 x 100 (launch-bound); c3b: MLP 4096 x 512, dopri5 adaptive, T = 1, max_cps 50 (config 3 as written: "
                         "one scalar all-reduce per step attempt when sharded); c5: Burgers IMEX 64 x 1024 fp64 per GPU "
                         "(config 5: ARKIMEX 3, ksponly, batched LU; --strong --batch 512 is the reference's global batch)")
    ap.add_argument("--dtype", choices=["f32", "f64"], default=None,
                    help="state precision (default f32; c5 is f64 as the reference runs it; f64 is the reference's CI precision)")
    ap.add_argument("--batch", type=int, default=None, help="trajectories per GPU (default 4096 for c3a/c2/c3b, 128 for c4, 64 for c5)")
    ap.add_argument("--dim", type=int, default=512)
    ap.add_argument("--hw", type=int, default=32, help="c4: spatial size of the feature maps (BASELINE config 4: 32 x 32)")
    ap.add_argument("--dump", default=None,
                    help="after the timed region: one more solve whose states (gathered over the ranks, in rank order) and "
                         "all-reduced parameter gradient rank 0 saves to this file (torch.save) -- how the tests check that the "
                         "shards of a sharded run concatenate to the one-rank answer")
    ap.add_argument("--nt", type=int, default=None, help="time steps per solve (default 100 for c3a, 4 for c4)")
    ap.add_argument("--dt", type=float, default=None, help="step size (default 0.01 for c3a, 1/nt for c4)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-variants", action="store_true", help="skip the extra (non-headline) measurements")
    ap.add_argument("--no-pmc", action="store_true",
                    help="do not run the two rocprofv3 --pmc children (FETCH_SIZE, WRITE_SIZE) that measure roofline.traffic")
    ap.add_argument("--no-rocprof", action="store_true",
                    help="do not run the rocprofv3 --kernel-trace child that times the solver kernels of the replayed "
                         "(graph-mode) region; `roofline` then rests on the HIP-event pass alone")
    ap.add_argument("--no-roofline-pass", action="store_true",
                    help="skip the eager HIP-event pass (profiling runs that must contain the timed region's launches only; "
                         "`roofline` is then null)")
    ap.add_argument("--mode", choices=["graph", "eager"], default="graph",
                    help="launch mode of the timed region: hipGraph replay (default) or plain stream launches")
    ap.add_argument("--no-ceiling", action="store_true", help="do not measure roofline.copy_ceiling (the streaming microbenchmark)")
    ap.add_argument("--ceiling-only", action="store_true",
                    help="run only the streaming microbenchmark behind roofline.copy_ceiling and print its layout (what the "
                         "rocprofv3 child of that measurement runs)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--tunableop", action="store_true",
                    help="let PyTorch's TunableOp pick func's GEMM kernels (tuned in the untimed setup solves; +2-3 %% "
                         "at C3a, profiles/README.md); off by default: the headline uses PyTorch's stock heuristics")
    ap.add_argument("--stiff", action="store_true",
                    help="with --config c3b: the same shapes on dynamics that make the controller work (problems.SwitchedMLPFunc: "
                         "W ~ N(0, 0.08), right-hand side gated by tanh(20 (sin(15 pi t) + 1/2)), T = 4): > 100 accepted steps, "
                         "rejections at every reversal, a checkpoint budget of 50 that binds.  BASELINE config 3 as written takes 3 steps")
    ap.add_argument("--stiff-T", type=float, default=None, dest="stiff_T",
                    help="final time of the --stiff workload (default 4.0; the counter children use a shorter horizon)")
    ap.add_argument("--strong", action="store_true",
                    help="strong scaling: --batch is the GLOBAL batch, split evenly over the ranks (default: weak, "
                         "--batch trajectories per GPU)")
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=VALUE",
                    help="an engine option for every solver of the run (A/B experiments; e.g. --opt pn_linear_side_stream=0); "
                         "the headline is the run WITHOUT any")
    a = ap.parse_args()
    a.opt = dict(kv.split("=", 1) for kv in a.opt)
    if a.config in ("c3a", "c3b"):
        a.batch = a.batch or 4096
        a.nt = a.nt or 100
        a.dt = a.dt or 0.01
    elif a.config == "c2":
        a.batch = a.batch or 4096
        a.nt = a.nt or 100
        a.dt = a.dt or 0.025
    elif a.config == "c5":
        a.batch = a.batch or 64
        a.nt = a.nt or 10
        a.dt = a.dt or 0.01
    else:
        a.batch = a.batch or 128
        a.nt = a.nt or 4
        a.dt = a.dt or 1.0 / a.nt
    a.dtype = a.dtype or ("f64" if a.config == "c5" else "f32")
    # (c3b, adaptive steps: the controller reads the error norm on the host at every attempt, so no sweep is captured whole; in
    # "graph" mode the solver replays func's single evaluations from per-evaluation hipGraphs, pnode_amd/_stagegraphs.py)
    return a


class Problem(object):
    """One BASELINE configuration: dynamics, state shape, scheme, options and what a solve is."""


def make_problem(args, torch, dtype=None):
    """The selected BASELINE config on the CPU: Problem with .func (.func2), .shape, .method, .setup (setupTS keywords),
    .opts (options database), .t (output times), .adaptive, .workload (description)."""
    from problems import BurgersEX, BurgersIM, ConvBlockFunc, MLPFunc, SpiralFunc, SwitchedMLPFunc
    dt = dtype or (torch.float64 if getattr(args, "dtype", "f32") == "f64" else torch.float32)
    p = Problem()
    p.dtype, p.func2, p.setup, p.adaptive = dt, None, {}, False
    p.opts = {"ts_adapt_type": "none", "ts_trajectory_type": "memory", "ts_trajectory_solution_only": "0"}
    p.method, p.t = "rk4", torch.tensor([args.dt * args.nt])
    if args.config == "c3a":
        p.func, p.shape = MLPFunc(args.dim, dt), (args.batch, args.dim)
        p.workload = ("C3a: MLP dynamics 3x%d tanh, batch %d x state_dim %d per GPU, rk4 fixed h=%g, %d time steps, adjoint on, "
                      "stages stored in HBM" % (args.dim, args.batch, args.dim, args.dt, args.nt))
    elif args.config == "c4":
        hw = getattr(args, "hw", 32)
        p.func, p.shape = ConvBlockFunc(64, dt), (args.batch, 64, hw, hw)
        p.workload = ("C4 shard: conv block (5 x conv+BN(eval)+ReLU, 9744 parameters) on %d x 64 x %d x %d per GPU, rk4 fixed "
                      "h=%g, t=[1.0], %d time steps, adjoint on, stages stored in HBM, setupTS before every forward"
                      % (args.batch, hw, hw, args.dt, args.nt))
    elif args.config == "c2":
        p.func, p.shape = SpiralFunc(dt), (args.batch, 2)
        p.workload = ("C2: batched spiral (Linear(2,50)-Tanh-Linear(50,2) on y^3), batch %d x 2 per GPU, rk4 fixed h=%g, %d time "
                      "steps, adjoint on, stages stored in HBM (launch-bound: N = %d elements)" % (args.batch, args.dt, args.nt, 2 * args.batch))
    elif args.config == "c3b":
        p.func, p.shape = MLPFunc(args.dim, dt), (args.batch, args.dim)
        p.method, p.adaptive = "dopri5", True
        p.opts = {"ts_trajectory_type": "memory", "ts_trajectory_max_cps_ram": "50"}
        p.t = torch.tensor([1.0])
        p.workload = ("C3b: MLP dynamics 3x%d tanh, batch %d x state_dim %d per GPU, dopri5 adaptive (rtol = atol = 1e-4, h0 = %g, "
                      "T = 1), adjoint on, -ts_trajectory_max_cps_ram 50" % (args.dim, args.batch, args.dim, args.dt))
        if getattr(args, "stiff", False):
            p.func = SwitchedMLPFunc(args.dim, dt)
            p.t = torch.tensor([getattr(args, "stiff_T", None) or SwitchedMLPFunc.T_END])
            p.workload = ("C3b --stiff: config 3's shapes on dynamics that adapt -- MLP 3x%d tanh with W ~ N(0, 0.08), right-hand side "
                          "gated by tanh(20 (sin(15 pi t) + 1/2)), batch %d x state_dim %d per GPU, dopri5 adaptive (rtol = atol = 1e-4, "
                          "h0 = %g, T = %g), adjoint on, -ts_trajectory_max_cps_ram 50 (binding)"
                          % (args.dim, args.batch, args.dim, args.dt, float(p.t[0])))
    else:
        n5 = 1024
        p.func, p.func2, p.shape = BurgersIM(n5, dtype=dt), BurgersEX(n5, dt), (args.batch, n5)
        p.method = "imex"
        p.setup = dict(implicit_form=True, imex_form=True, batch_size=args.batch, linear_solver="torch", matrixfree_jacobian=False,
                       fixed_jacobian=True)
        p.opts = {"ts_adapt_type": "none", "ts_trajectory_type": "memory", "ts_trajectory_solution_only": "0",
                  "ts_arkimex_type": "3", "snes_type": "ksponly"}
        p.workload = ("C5 shard: SINODE Burgers, IMEX split (circular Laplacian implicit, 5-layer ReLU MLP explicit), %d x 1024 %s "
                      "per GPU, ARKIMEX type 3, -snes_type ksponly, linear_solver=torch (batched LU of the one-sample Jacobian), "
                      "h=%g, %d time steps, adjoint on" % (args.batch, "fp64" if dt == torch.float64 else "fp32", args.dt, args.nt))
    return p


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N fresh ranks (the parent never initialises a GPU;
    no exec of a process that did) and hand their exit code back."""
    import socket
    import subprocess
    import torch
    have = torch.cuda.device_count()             # does not initialise the HIP runtime on this image
    if have < args.gpus and os.environ.get("PN_BENCH_BACKEND", "nccl") == "nccl":
        sys.stderr.write("bench.py: --gpus %d needs %d GPUs, this box has %d\n" % (args.gpus, args.gpus, have))
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd).returncode


def cpu_baseline(args, budget_s):
    """Oracle (C restatement of the PETSc op sequence + per-stage Python callbacks) on the
    host: same state size, same func, same scheme, fewer time steps (bounded sample)."""
    import torch
    from oracle.ts_oracle import ODEPetscOracle

    if args.config == "c5":
        return cpu_baseline_c5(args, budget_s)
    torch.manual_seed(0)
    pb = make_problem(args, torch)
    f, shape = pb.func, pb.shape
    y0 = torch.randn(*shape, dtype=pb.dtype)
    adaptive = pb.adaptive

    def solve(nt):
        opts = {"ts_trajectory_solution_only": 0} if adaptive else {"ts_adapt_type": "none", "ts_trajectory_solution_only": 0}
        ode = ODEPetscOracle(opts)
        ode.setupTS(y0, f, step_size=args.dt, method=pb.method)
        f.zero_grad()
        y = y0.clone().requires_grad_(True)
        t0 = time.perf_counter()
        out = ode.odeint_adjoint(y, torch.tensor([args.dt * nt], dtype=torch.float64))
        out.abs().mean().backward()
        el = time.perf_counter() - t0
        return el, len(ode.step_log()[1])

    solve(1)                                   # warm-up (allocations, thread pools)
    # func's GEMMs run on torch's intra-op pool: calibrate the pool size (more threads than
    # the 4096x512 GEMMs can feed only adds synchronisation cost), then keep the fastest
    ncpu = os.cpu_count() or 1
    best = None
    for th in sorted({min(ncpu, c) for c in (8, 16, 32, 64, ncpu)}):
        torch.set_num_threads(th)
        solve(1)
        dt1 = solve(1)[0]
        if best is None or dt1 < best[1]:
            best = (th, dt1)
    threads = best[0]
    torch.set_num_threads(threads)
    el2, n2 = solve(2)
    per = el2 / max(n2, 1)
    nt = int(max(2, min(args.nt, budget_s / max(per, 1e-6))))
    dt, nsteps = solve(nt)
    return {"value": nsteps / dt, "unit": "time-steps/s", "cores": threads, "kind": "port",
            "sample": "%d %s time steps (horizon %d x h of %d) fwd+adjoint on a %s %s state, stages stored; "
                      "vector ops single-threaded C (VecSeq-like), func on %d torch threads "
                      "(fastest of 8..%d on this host)"
                      % (nsteps, pb.method, nt, args.nt, "x".join(str(d) for d in shape), getattr(args, "dtype", "f32"), threads, ncpu)}


def cpu_baseline_c5(args, budget_s):
    """Config 5 on the host: the oracle's restatement of the reference's direct IMEX path (oracle/arkimex_oracle.py
    `odeint_adjoint_arkimex_direct`: one-sample Jacobian, LU once per odeint, lu_solve on the (B, n) right-hand sides,
    transposed solve in the adjoint -- /root/reference/pnode/torch_linearsolve.py:15-35, pa.py:474-508), same state, same
    funcs, same scheme, a bounded number of time steps."""
    import torch
    from oracle.arkimex_oracle import odeint_adjoint_arkimex_direct
    torch.manual_seed(0)
    pb = make_problem(args, torch)
    fI, fE = pb.func, pb.func2
    torch.manual_seed(1234)
    y0 = torch.rand(*pb.shape, dtype=pb.dtype)
    name = pb.opts["ts_arkimex_type"]

    def solve(nt):
        fE.zero_grad()
        y = y0.clone().requires_grad_(True)
        t0 = time.perf_counter()
        out = odeint_adjoint_arkimex_direct(fI, fE, y, torch.tensor([args.dt * nt], dtype=torch.float64), args.dt, name, ksponly=True)
        out.abs().mean().backward()
        return time.perf_counter() - t0

    ncpu = os.cpu_count() or 1
    solve(1)
    best = None
    for th in sorted({min(ncpu, c) for c in (8, 16, 32, 64, ncpu)}):
        torch.set_num_threads(th)
        solve(1)
        d1 = solve(2) / 2
        if best is None or d1 < best[1]:
            best = (th, d1)
    threads, per = best
    torch.set_num_threads(threads)
    nt = int(max(2, min(50 * args.nt, budget_s / max(per, 1e-6))))
    el = solve(nt)
    return {"value": nt / el, "unit": "time-steps/s", "cores": threads, "kind": "port",
            "sample": "%d ARKIMEX-%s time steps of h = %g fwd+adjoint on a %s %s state with the reference's direct stage solve "
                      "(Jacobian of one sample by jacrev, one LU per odeint, lu_solve on the (B, n) right-hand sides, transposed "
                      "solve in the adjoint; -snes_type ksponly), torch on %d threads (fastest of 8..%d on this host)"
                      % (nt, name, args.dt, "x".join(str(d) for d in pb.shape), getattr(args, "dtype", "f64"), threads, ncpu)}


def under_profiler():
    return any(k.startswith(("ROCPROF", "ROCP_TOOL", "ROCPROFILER")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "")


def rocprof_child(args):
    """Kernel durations of the TIMED mode by the profiler itself: a child `rocprofv3 --kernel-trace -- python3
    bench.py ...` of the same workload and launch mode (graph mode: 2 eager set-up solves + the capturing call;
    then 1 warm-up and 3 timed solves), summarised over its last 3 solves.  Started before this process initialises the GPU; any failure
    returns None and the bench line falls back to the HIP-event numbers."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    # this process is itself being profiled (rocprofv3 -- python3 bench.py ...): no profiler inside a profiler
    if under_profiler():
        return None
    if args.config == "c3b" and args.mode == "graph":
        return None        # (per-evaluation graphs add whole-state copies of their own: the solves of a trace cannot be told apart below)
    d = tempfile.mkdtemp(prefix="pn_rocprof_", dir="/tmp")
    try:
        k_timed, k_warm = 3, 1
        # graph mode (-pn_graph_capture auto, the default): 2 eager calls, then the capturing call, which runs the sweeps
        # eagerly AND replays what it captured (the first-replay check): 4 solves' worth of launches before the warm-up
        k_setup = 4 if args.mode == "graph" else 0
        cmd = [exe, "--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__),
               "--config", args.config, "--mode", args.mode, "--steps", str(k_timed), "--warmup", str(k_warm),
               "--batch", str(args.batch), "--dim", str(getattr(args, "dim", 512)), "--nt", str(args.nt), "--dt", repr(args.dt),
               "--dtype", getattr(args, "dtype", "f32"),
               "--no-cpu-baseline", "--no-variants", "--no-roofline-pass", "--no-rocprof", "--no-pmc"] + (["--stiff"] if getattr(args, "stiff", False) else [])
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
        env.update(DEBUG_CLR_GRAPH_PACKET_CAPTURE="0", TMPDIR="/tmp")
        r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=900)
        if r.returncode != 0:
            sys.stderr.write("bench: rocprofv3 child failed (rc %d): %s\n" % (r.returncode, r.stderr[-500:]))
            return None
        rows = []
        for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                rows.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]), row["Kernel_Name"]))
        rows.sort()
        copies = ("pn_lincomb_kernel<float, 1,", "pn_lincomb_kernel<double, 1,")
        if args.config != "c5":
            # every solve of these configs (one output time) makes exactly three whole-state copies -- u0 into its slot, the
            # answer out, the cotangent in -- and the first of them is its first solver launch: the last 3*k_timed copies
            # delimit the timed solves whatever the launch mode of the child turned out to be
            cidx = [i for i, x in enumerate(rows) if any(c in x[2] for c in copies)]
            if len(cidx) < 3 * k_timed or len(cidx) % 3:
                sys.stderr.write("bench: rocprofv3 child: %d whole-state copies do not make whole solves\n" % len(cidx))
                return None
            total = len(cidx) // 3
            rows = rows[cidx[len(cidx) - 3 * k_timed]:]
        else:
            idx = [i for i, x in enumerate(rows) if "pn_lincomb_kernel" in x[2]]
            total = k_setup + k_warm + k_timed
            if not idx or len(idx) % total:
                sys.stderr.write("bench: rocprofv3 child: %d pn_lincomb launches do not divide into %d solves\n" % (len(idx), total))
                return None
            rows = rows[idx[len(idx) - (len(idx) // total) * k_timed]:]
        per = {}
        for s0, e0, name in rows:
            if "pn_" in name:
                name = name[name.index("pn_"):].split("(")[0]
                per.setdefault(name, []).append((e0 - s0) / 1e3)
        vec_us = sum(sum(v) for k, v in per.items() if k.startswith("pn_lincomb_kernel") and not k.startswith(copies))
        par_us = sum(sum(v) for k, v in per.items() if k.startswith(("pn_param_accum", "pn_colsum")))
        wgrad_us = sum(sum(v) for k, v in per.items() if "pn_linear_wgrad_kernel" in k)
        wrms_us = sum(sum(v) for k, v in per.items() if k.startswith("pn_combine_wrms"))
        all_kernels_us = sum(e0 - s0 for s0, e0, _ in rows) / 1e3
        return {"time_steps": args.nt * k_timed, "solves": k_timed, "vec_us": vec_us, "par_us": par_us, "wrms_us": wrms_us, "wgrad_us": wgrad_us,
                "all_kernels_us": all_kernels_us,
                "wall_us": (rows[-1][1] - rows[0][0]) / 1e3,
                "per_kernel": {k: {"launches": len(v), "avg_us": sum(v) / len(v)} for k, v in sorted(per.items())},
                "command": "rocprofv3 --kernel-trace --output-format csv -- python3 bench.py " + " ".join(cmd[cmd.index(os.path.abspath(__file__)) + 1:]),
                "region": "the last %d of %d solves of the child (its timed region; launch mode %s)" % (k_timed, total, args.mode)}
    except Exception as exc:
        sys.stderr.write("bench: rocprofv3 child not usable (%r)\n" % (exc,))
        return None
    finally:
        shutil.rmtree(d, ignore_errors=True)


def ceiling_blocks(torch, dev, dtype, n, shape, launch, meter=None):
    """The streaming microbenchmark behind roofline.copy_ceiling ("also report measured copy-kernel ceiling"):
    y = u + c*K -- two vectors in, one out, the shape of the dominant launch of an rk4 step -- through the product's own
    pn_rk_stage entry point (pn_lincomb_kernel<T, 2>), in four settings:

      large_stream            256 MiB per vector: what this chip streams through this kernel when launch cost is amortised
      state_size_cold         vectors of the state's size, rotating over > 1 GiB of distinct buffers (nothing is cache-resident)
      state_size_behind_gemm  as in the sweep: K is the output of a GEMM of func's shape launched just before (2-D states only)
      state_size_hot          the same three buffers every launch (resident in the 256 MB Infinity Cache)

    `launch(y, u, K)` enqueues one launch.  `meter`, when given, is the HIP-event instrument: meter.start() after the warm-up
    launches of a block, meter.stop() -> microseconds after its measured launches.  Returns
    [(name, discarded warm-up launches, measured launches, bytes moved per launch, measured us or None)]."""
    w = 8 if dtype == torch.float64 else 4
    out = []

    def block(name, triples, warm, reps, before=None):
        us = None
        for k in range(warm + reps):
            if k == warm and meter is not None:
                meter.start()
            y, u, K = triples[k % len(triples)]
            if before is not None:
                K = before(k)
            launch(y, u, K)
        if meter is not None:
            us = meter.stop()
        out.append((name, warm, reps, 3 * triples[0][0].numel() * w, us))

    big = (256 << 20) // w
    bufs = [torch.empty(big, dtype=dtype, device=dev).normal_() for _ in range(3)]
    block("large_stream", [tuple(bufs)], 2, 10)
    del bufs
    torch.cuda.empty_cache()
    R = max(4, min(256, -(-(1200 << 20) // (3 * n * w))))
    pool = torch.empty(R, 3, n, dtype=dtype, device=dev).normal_()
    block("state_size_cold", [(pool[r, 0], pool[r, 1], pool[r, 2]) for r in range(R)], R, R)
    if len(shape) == 2 and shape[1] >= 64:
        W = torch.randn(shape[1], shape[1], dtype=dtype, device=dev) * 0.02
        Ks = [torch.empty(shape, dtype=dtype, device=dev) for _ in range(2)]
        Y = pool[0, 0].view(shape)

        def gemm(k):
            torch.mm(Y, W, out=Ks[k % 2])
            return Ks[k % 2].view(-1)
        block("state_size_behind_gemm", [(pool[r, 0], pool[r, 1], pool[r, 2]) for r in range(1, R)], 4, 24, before=gemm)
    block("state_size_hot", [(pool[0, 0], pool[0, 1], pool[0, 2])], 4, 24)
    torch.cuda.synchronize()
    return out


def ceiling_run(args, torch, dev, events):
    """Run the microbenchmark on this process's device: `events` True = HIP-event durations (pn_prof), False = launches only
    (the rocprofv3 child: the profiler's dispatch timestamps are the instrument)."""
    from pnode_amd import _lib, petsc_adjoint
    pb = make_problem(args, torch)
    n = 1
    for d in pb.shape:
        n *= d
    ops = petsc_adjoint.HipVecOps(dev, pb.dtype, n)
    big = None
    lib = _lib.load()

    def launch(y, u, K):
        nonlocal big
        if y.numel() != n:                       # (the large-stream block: an entry-point object of that length)
            if big is None:
                big = petsc_adjoint.HipVecOps(dev, pb.dtype, y.numel())
            big.rk_stage(y, u, [K], [0.5])
        else:
            ops.rk_stage(y, u, [K], [0.5])
    L = (ctypes.c_int64 * len(_lib.KERNEL_IDS))()
    us = (ctypes.c_double * len(_lib.KERNEL_IDS))()
    by = (ctypes.c_double * len(_lib.KERNEL_IDS))()

    class Meter(object):
        @staticmethod
        def start():
            torch.cuda.synchronize()
            lib.pn_prof_enable(1)

        @staticmethod
        def stop():
            torch.cuda.synchronize()
            _lib.check(lib.pn_prof_collect(len(L), L, us, by))
            lib.pn_prof_enable(0)
            return us[0]
    blocks = ceiling_blocks(torch, dev, pb.dtype, n, pb.shape, launch, Meter if events else None)
    # an independent instrument for the chip's ceiling: the runtime's own device-to-device copy of 256 MiB (read + write)
    src = torch.empty((256 << 20) // 4, dtype=torch.float32, device=dev).normal_()
    dst = torch.empty_like(src)
    for _ in range(2):
        dst.copy_(src)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        dst.copy_(src)
    e1.record()
    torch.cuda.synchronize()
    d2d = 10 * 2 * src.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9
    return blocks, d2d


def ceiling_summary(blocks, us_of, instrument):
    """{block: GB/s} from [(name, warm, reps, bytes per launch, us)]; `us_of(name, index range)` overrides the block's own us."""
    res = {}
    pos = 0
    for blk in blocks:
        name, warm, reps, nbytes = blk[:4]
        us = blk[4] if len(blk) > 4 else None
        if us_of is not None:
            us = us_of(pos + warm, pos + warm + reps)
        pos += warm + reps
        if us:
            res[name] = {"GBps": reps * nbytes / (us * 1e-6) / 1e9, "avg_us": us / reps, "bytes_per_launch": nbytes, "launches": reps}
    res["instrument"] = instrument
    return res


def ceiling_child(args):
    """The microbenchmark under `rocprofv3 --kernel-trace` in a child (started before this process touches the GPU): the
    profiler's kernel durations of the pn_lincomb_kernel<T, 2> launches, split into the blocks by the layout the child prints."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe) or under_profiler():
        return None
    d = tempfile.mkdtemp(prefix="pn_ceiling_", dir="/tmp")
    try:
        cmd = [exe, "--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__),
               "--ceiling-only", "--config", args.config, "--batch", str(args.batch), "--dim", str(getattr(args, "dim", 512)),
               "--dtype", getattr(args, "dtype", "f32")]
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
        env.update(DEBUG_CLR_GRAPH_PACKET_CAPTURE="0", TMPDIR="/tmp")
        r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=600)
        if r.returncode != 0:
            sys.stderr.write("bench: rocprofv3 ceiling child failed (rc %d): %s\n" % (r.returncode, r.stderr[-400:]))
            return None
        layout = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        rows = []
        for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                if "pn_lincomb_kernel<" in row["Kernel_Name"]:
                    rows.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"])))
        rows.sort()
        blocks = [tuple(b) for b in layout["blocks"]]
        if len(rows) != sum(b[1] + b[2] for b in blocks):
            sys.stderr.write("bench: rocprofv3 ceiling child: %d launches traced, %d expected\n" % (len(rows), sum(b[1] + b[2] for b in blocks)))
            return None
        res = ceiling_summary(blocks, lambda a, b: sum(e - s0 for s0, e in rows[a:b]) / 1e3,
                              "kernel durations of a child `rocprofv3 --kernel-trace -- python3 bench.py --ceiling-only ...`")
        return res
    except Exception as exc:
        sys.stderr.write("bench: rocprofv3 ceiling child not usable (%r)\n" % (exc,))
        return None
    finally:
        shutil.rmtree(d, ignore_errors=True)


def moved_bytes_per_launch(name, n, w, n_par, sources_per_launch):
    """HBM bytes one launch of a solver kernel moves (every input read once, the output written once), from its name:
    pn_lincomb_kernel<T, K, ...> reads K vectors and writes one; the batched parameter accumulation reads its gradient sets
    and reads + writes mu.  None for kernels whose operand count is not in the name (pn_combine_wrms_kernel)."""
    import re
    m = re.match(r"pn_lincomb_kernel<(?:float|double), (\d+)", name)
    if m:
        return (int(m.group(1)) + 1) * n * w
    if name.startswith("pn_param_accum_multi_kernel") and sources_per_launch:
        return (sources_per_launch + 2) * n_par * w
    return None


def pmc_traffic_from_profiles():
    """Fallback when the PMC children cannot run: the newest committed summary under profiles/ (a constant of the
    repository, NOT a measurement of this run) and its file name."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    if not files:
        return None
    try:
        return {"hbm_bytes_per_launch": json.load(open(files[-1]))["rk4_time_step"]["hbm_bytes_per_launch_avg"],
                "file": os.path.relpath(files[-1], ROOT)}
    except Exception:
        return None


def pmc_child(args, counter):
    """One counter pass of the workload in a child: `rocprofv3 --kernel-trace --pmc <counter> -- python3 bench.py ...` (eager
    launches, one solve of at most 16 time steps, nothing else in the process).  FETCH_SIZE and WRITE_SIZE do not fit into
    one pass (MI355X_MICROARCH.md, rocprofv3 PMC slots), hence one child per counter.  Returns {kernel: [values in KiB]} for
    the pn_* kernels, or None."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe) or under_profiler():
        return None
    d = tempfile.mkdtemp(prefix="pn_pmc_", dir="/tmp")
    try:
        # counter collection serialises every dispatch: configs whose func is thousands of tiny launches per time step (C2's
        # skinny GEMMs, C5's per-sample fp64 convolution fallback) get two time steps
        nt = min(args.nt, 2 if args.config in ("c2", "c5") else 16)
        cmd = [exe, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", d, "--", sys.executable,
               os.path.abspath(__file__), "--config", args.config, "--mode", "eager", "--steps", "1", "--warmup", "0",
               "--batch", str(args.batch), "--dim", str(getattr(args, "dim", 512)), "--nt", str(nt), "--dt", repr(args.dt),
               "--dtype", getattr(args, "dtype", "f32"),
               "--no-cpu-baseline", "--no-variants", "--no-roofline-pass", "--no-rocprof", "--no-pmc"] + \
              (["--stiff", "--stiff-T", "0.3"] if getattr(args, "stiff", False) else [])    # (two reversals: ~20 steps, ~25 rejections)
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
        env.update(DEBUG_CLR_GRAPH_PACKET_CAPTURE="0", TMPDIR="/tmp")
        r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=300)
        if r.returncode != 0:
            sys.stderr.write("bench: rocprofv3 --pmc %s child failed (rc %d): %s\n" % (counter, r.returncode, r.stderr[-400:]))
            return None
        per = {}
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                if row["Counter_Name"] != counter or "pn_" not in row["Kernel_Name"]:
                    continue
                name = row["Kernel_Name"]
                per.setdefault(name[name.index("pn_"):].split("(")[0], []).append(float(row["Counter_Value"]))
        try:                                    # accepted time steps of the child's solve (an adaptive solve decides them itself)
            nt = int(json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])["config"]["time_steps"])
        except Exception:
            pass
        return {"per_kernel": per, "time_steps": nt, "command": " ".join(cmd[:cmd.index("--") + 1]) + " python3 bench.py " +
                " ".join(cmd[cmd.index(os.path.abspath(__file__)) + 1:])} if per else None
    except Exception as exc:
        sys.stderr.write("bench: rocprofv3 --pmc child not usable (%r)\n" % (exc,))
        return None
    finally:
        shutil.rmtree(d, ignore_errors=True)


def pmc_traffic(args):
    """HBM bytes the solver kernels really moved, measured in this run: two counter children (FETCH_SIZE, WRITE_SIZE), units
    and the gfx950 correction as MI355X_MICROARCH.md prescribes (both counters are KiB; FETCH_SIZE reports half of the bytes
    of a wide coalesced streaming read and is doubled; WRITE_SIZE is exact).  Copies of whole states (pn_lincomb_kernel with
    one input: u0 into its slot, outputs) are left out, as they are left out of `achieved`."""
    fetch = pmc_child(args, "FETCH_SIZE")
    write = pmc_child(args, "WRITE_SIZE") if fetch else None
    if not fetch or not write:
        return None
    copies = ("pn_lincomb_kernel<float, 1,", "pn_lincomb_kernel<double, 1,")
    kernels, total, launches = {}, 0.0, 0
    for name in sorted(set(fetch["per_kernel"]) | set(write["per_kernel"])):
        fv, wv = fetch["per_kernel"].get(name, []), write["per_kernel"].get(name, [])
        n = max(len(fv), len(wv))
        rb = 2.0 * 1024.0 * sum(fv) / max(len(fv), 1)
        wb = 1024.0 * sum(wv) / max(len(wv), 1)
        kernels[name] = {"launches": n, "read_bytes": rb, "write_bytes": wb}
        if not name.startswith(copies) and not name.startswith("pn_linear_"):     # (the MFMA-bound kernel: roofline.linear_wgrad)
            total += n * (rb + wb)
            launches += n
    if not launches:
        return None
    return {"hbm_bytes_per_launch": total / launches, "hbm_bytes_per_time_step": total / fetch["time_steps"],
            "launches": launches, "time_steps": fetch["time_steps"], "per_kernel": kernels,
            "commands": [fetch["command"], write["command"]],
            "units": "counters are KiB; FETCH_SIZE doubled (gfx950 reports half of a wide coalesced read stream), WRITE_SIZE exact; "
                     "averages per launch over the HBM-bound solver kernels of one eager solve (state copies and the MFMA-bound "
                     "pn_linear_wgrad kernels excluded from the average; all of them are listed in per_kernel)"}


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    import copy
    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but the launcher started %d ranks" % (args.gpus, world))
    # ---- profiler children, started before this process touches the GPU: kernel durations of the timed mode
    # (--kernel-trace), HBM bytes of the solver kernels (--pmc FETCH_SIZE / WRITE_SIZE, one child each) and, for the
    # headline config, the same solve in the reference's CI precision (fp64)
    prof = traffic = prof64 = ceil_prof = None
    solo = world == 1 and not args.no_roofline_pass
    if args.ceiling_only:
        ge.build_library()
        torch.cuda.set_device(local)
        blocks, _ = ceiling_run(args, torch, torch.device("cuda", local), events=False)
        print(json.dumps({"blocks": [b[:4] for b in blocks]}), flush=True)
        return
    if solo and not args.no_rocprof:
        prof = rocprof_child(args)
        if not args.no_ceiling:
            ceil_prof = ceiling_child(args)
        if args.config == "c3a" and args.dtype == "f32" and not args.no_variants:
            a64 = copy.copy(args)
            a64.dtype = "f64"
            prof64 = rocprof_child(a64)
    if solo and not args.no_pmc:
        traffic = pmc_traffic(args)
    # PN_BENCH_BACKEND=gloo is a test hook: it lets the multi-rank flow be exercised on a box
    # with fewer GPUs than ranks (ranks then share devices); the real runs use RCCL ("nccl")
    backend = os.environ.get("PN_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local = local % max(torch.cuda.device_count(), 1)
    elif torch.cuda.device_count() < world:
        sys.exit("bench.py: %d ranks need %d GPUs, this box has %d" % (world, world, torch.cuda.device_count()))
    # the library ships prebuilt in-tree; it is rebuilt (by whichever rank gets the lock) only when the
    # hash of its sources differs from the one stored beside it -- before this process touches the GPU
    ge.build_library()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    from pnode_amd import _lib, options, petsc_adjoint

    lib = _lib.load()
    if args.tunableop:
        torch.cuda.tunable.enable(True)
        torch.cuda.tunable.set_filename(os.path.join("/tmp", "pnode_amd_tunableop_rank%d.csv" % rank))   # results file: scratch
    if args.strong:
        if args.batch % world:
            raise SystemExit("--strong: --batch must be divisible by the number of ranks")
        args.batch //= world

    def build(a):
        """Problem of argument set `a` on the device: parameters equal on every rank, a different batch shard per rank."""
        torch.manual_seed(0)
        pb = make_problem(a, torch)
        pb.func = pb.func.to(dev)
        if pb.func2 is not None:
            pb.func2 = pb.func2.to(dev)
        draw = torch.rand if a.config == "c5" else torch.randn
        if a.strong:
            # strong scaling: ONE global batch (drawn on the host, the same on every rank), rank r integrates rows
            # [r*b, (r+1)*b) -- the shards of an N-rank run concatenate to the one-rank run's states
            torch.manual_seed(1234)
            pb.y0 = draw(pb.shape[0] * world, *pb.shape[1:], dtype=pb.dtype)[rank * pb.shape[0]:(rank + 1) * pb.shape[0]].to(dev)
        else:
            torch.manual_seed(1234 + rank)
            pb.y0 = draw(*pb.shape, device=dev, dtype=pb.dtype)
        pb.params = [q for m in (pb.func, pb.func2) if m is not None for q in m.parameters() if q.requires_grad]
        pb.every_forward = a.config == "c4"   # the reference's ODE block calls setupTS before every forward
        pb.step = a.dt                        # (train-Cifar10.py:121-139)
        return pb

    pb = build(args)

    def make_ode(extra, q=None):
        q = q or pb
        options.clear()
        opts = dict(dict(q.opts, **args.opt), **extra)
        for k, v in opts.items():
            options.set_option(k, v)
        o = petsc_adjoint.ODEPetsc()
        kw = dict(q.setup, **({"func2": q.func2} if q.func2 is not None else {}))
        o.setupTS(q.y0, q.func, step_size=q.step, method=q.method, enable_adjoint=True, **kw)
        options.clear()
        o._bench_opts, o._bench_kw, o._bench_pb = opts, kw, q
        return o

    def one_solve(o):
        q = o._bench_pb
        for par in q.params:
            par.grad = None
        y = q.y0.detach().requires_grad_(True)
        if q.every_forward:
            for k, v in o._bench_opts.items():
                options.set_option(k, v)
            o.setupTS(y, q.func, step_size=q.step, method=q.method, enable_adjoint=True, **o._bench_kw)
            options.clear()
        out = o.odeint_adjoint(y, q.t)
        loss = out.abs().mean()
        loss.backward()
        return loss

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    rep_ms = []

    def timed(o, k):
        """Wall time of exactly k solves between two barrier + synchronize brackets (max over ranks) -- the contract's clock.
        Every solve is also bracketed by HIP events on the stream the sweeps are launched on (torch's current stream):
        rep_ms holds this rank's per-solve durations of the LAST call (median of the reps, HIP-event timed)."""
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(k + 1)]
        sync()
        t0 = time.perf_counter()
        for i in range(k):
            ev[i].record()
            one_solve(o)
        ev[k].record()
        sync()
        tv = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        rep_ms[:] = [ev[i].elapsed_time(ev[i + 1]) for i in range(k)]
        if world > 1:
            dist.all_reduce(tv, op=dist.ReduceOp.MAX)
        return tv.item()

    def graph_ode(extra=None, q=None):
        # no launch option at all: -pn_graph_capture defaults to `auto` (two eager calls, then a call that runs the sweeps
        # eagerly AND captures them, checks the first replays bit for bit against the eager results and that replay is faster)
        extra = dict(extra or {})
        o = make_ode(extra, q)
        # (adaptive: two validating calls -- the second consists of replays only and is timed against its eager twin)
        for _ in range(petsc_adjoint.ODEPetsc.GRAPH_WARMUP_CALLS + (2 if (q or pb).adaptive else 1)):
            one_solve(o)
        torch.cuda.synchronize()
        if not o.graphs_captured and not (q or pb).adaptive:
            raise RuntimeError("the solver stayed with eager launches: %s" % o.graph_status)
        return o                                      # (adaptive: whatever the solver settled on -- graph_status says which and why)

    # ---- headline solver.  mode "graph": the whole forward sweep and the whole reverse sweep
    # are replayed from two hipGraphs (same kernels, same order, bit-identical results; two
    # eager calls + one capturing call happen here, untimed).  Captured BEFORE the process
    # group exists, so no RCCL helper thread is alive while a capture is in progress.
    mode = args.mode
    ode = None
    if mode == "graph":
        try:
            ode = graph_ode()
        except Exception as exc:                      # fall back to eager launches, say so
            sys.stderr.write("bench: hipGraph capture failed (%r); falling back to eager launches\n" % (exc,))
            mode = "eager(graph-capture-failed)"
            ode = None
    if ode is None:
        ode = make_ode({"pn_graph_capture": "0"})
    else:
        mode = ode.graph_status                       # "graph(auto)": what a default-constructed ODEPetsc does
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        ode.setProcessGroup(None, average=True)
        dist.barrier()

    for _ in range(args.warmup):
        one_solve(ode)
    elapsed = timed(ode, args.steps)
    if pb.adaptive and not mode.startswith("eager("):
        mode = ode.graph_status                       # (an adaptive solve over a sharded batch stays with eager launches)
    head_rep_ms = sorted(rep_ms)
    nsteps = ode.num_steps
    if not pb.adaptive:
        assert nsteps == args.nt, (nsteps, args.nt)
    tapes_kept = ode._tapes is not None
    # every rank must have taken the same steps (fixed step: by construction; adaptive: the global error norm's all-reduce)
    steps_per_rank = [nsteps]
    if world > 1:
        sv = torch.tensor([nsteps, ode.num_rejections if pb.adaptive else 0], dtype=torch.int64, device=dev)
        got = [torch.zeros_like(sv) for _ in range(world)]
        dist.all_gather(got, sv)
        steps_per_rank = [int(g[0]) for g in got]
        if len(set(tuple(g.tolist()) for g in got)) != 1:
            raise SystemExit("bench.py: the ranks took different steps: %r" % [g.tolist() for g in got])
    if args.dump:
        for par in pb.params:
            par.grad = None
        y = pb.y0.detach().requires_grad_(True)
        out = ode.odeint_adjoint(y, pb.t)
        (out.abs().sum() / (out.numel() * world)).backward()       # = the global batch's mean |y(T)|, every rank's share of it
        gflat = torch.cat([q.grad.reshape(-1) for q in pb.params]) * (world if world > 1 else 1)   # (setProcessGroup averages)
        outs, gys = [out.detach()], [y.grad.detach()]
        if world > 1:
            outs = [torch.empty_like(out) for _ in range(world)]
            gys = [torch.empty_like(y.grad) for _ in range(world)]
            dist.all_gather(outs, out.detach().contiguous())
            dist.all_gather(gys, y.grad.detach().contiguous())
        if rank == 0:
            torch.save({"out": torch.cat([o.cpu() for o in outs], dim=1), "dy0": torch.cat([g.cpu() for g in gys], dim=0),
                        "dtheta": gflat.cpu(), "steps_per_rank": steps_per_rank, "world": world}, args.dump)

    # ---- roofline pass: the same solve with eager launches, every solver-kernel dispatch
    # bracketed by HIP start/stop events (events cannot be attached to graph nodes).  Same
    # kernels on the same data as the timed region above.
    kr = max(1, min(args.steps, 5))
    L = (ctypes.c_int64 * len(_lib.KERNEL_IDS))()
    us = (ctypes.c_double * len(_lib.KERNEL_IDS))()
    by = (ctypes.c_double * len(_lib.KERNEL_IDS))()
    elapsed_e = None
    if not args.no_roofline_pass:
        ode_e = ode if mode.startswith("eager") else make_ode({"pn_graph_capture": "0"})
        if world > 1 and ode_e is not ode:
            ode_e.setProcessGroup(None, average=True)
        one_solve(ode_e)
        sync()
        lib.pn_prof_enable(1)
        elapsed_e = timed(ode_e, kr)
        _lib.check(lib.pn_prof_collect(len(L), L, us, by))
        lib.pn_prof_enable(0)

    # ---- the measured streaming ceiling of this chip for launches of this shape, HIP-event instrument
    ceil_ev = ceil_d2d = None
    if world == 1 and not args.no_ceiling and not args.no_roofline_pass:
        try:
            torch.cuda.empty_cache()
            blocks, ceil_d2d = ceiling_run(args, torch, dev, events=True)
            ceil_ev = ceiling_summary(blocks, None, "HIP start/stop events bound to each dispatch (each duration holds one marker-to-"
                                                    "dispatch hand-over, 0.6-0.8 us, that the profiler's timestamps do not)")
        except Exception as exc:
            sys.stderr.write("bench: copy-ceiling microbenchmark failed (%r)\n" % (exc,))
        torch.cuda.empty_cache()

    # ---- the collectives of the path, timed alone: the all-reduce of dL/dtheta after every backward and, for
    # an adaptive scheme, the two-double all-reduce (+ read-back) every step attempt makes so that all ranks take the same step
    allreduce_us = enorm_allreduce_us = None
    if world > 1:
        flat = torch.zeros(max(sum(q.numel() for q in pb.params), 1), device=dev, dtype=pb.dtype)
        for _ in range(3):
            dist.all_reduce(flat)
        sync()
        t0 = time.perf_counter()
        for _ in range(20):
            dist.all_reduce(flat)
        torch.cuda.synchronize()
        allreduce_us = 1e6 * (time.perf_counter() - t0) / 20
        if pb.adaptive:
            for _ in range(3):
                ode._global_enorm(0.5)
            sync()
            t0 = time.perf_counter()
            for _ in range(20):
                ode._global_enorm(0.5)
            enorm_allreduce_us = 1e6 * (time.perf_counter() - t0) / 20

    # ---- extra, NOT the headline (single GPU only)
    variants = None
    if world == 1 and not args.no_variants:
        variants = {}
        if elapsed_e:
            variants["eager"] = {"value": nsteps * kr / elapsed_e, "unit": "time-steps/s",
                                 "note": "plain stream launches, events on (the roofline pass)"}
        extra_runs = []
        if args.config in ("c3a", "c4", "c2") and not mode.startswith("eager"):
            extra_runs = [
                ("recompute", {"pn_trajectory_retain_graph": "0"},
                 "graph replay; f re-evaluated inside every stage VJP of the reverse sweep, as the reference does (pa.py:66-74)"),
                ("eager+recompute", {"pn_graph_capture": "0", "pn_trajectory_retain_graph": "0"},
                 "plain stream launches and the reference's per-stage re-evaluation of f"),
                ("autograd-param-grads", {"pn_linear_param_grads": "0"},
                 "graph replay; the parameter sensitivities of func's nn.Linear layers taken from autograd and added by "
                 "pn_param_accum_multi (rounds 1-4) instead of being formed by the engine during the backward pass (round 5)"),
                ("library-gemm-param-grads", {"pn_linear_param_grads": "gemm"},
                 "graph replay; engine-side Linear sensitivities with the BLAS library's GEMM (torch.addmm into mu) + "
                 "pn_colsum_accum_multi for the bias sums instead of the fused pn_linear_wgrad_kernel"),
                ("fp32-mfma-param-grads", {"pn_linear_wgrad_exact": "1"},
                 "graph replay; the fused kernel on the fp32 matrix instruction (v_mfma_f32_32x32x2_f32, a k-ordered fmaf chain) instead of "
                 "the default split-bf16 form"),
                ("tile-64-param-grads", {"pn_linear_wgrad_tile64": "1"},
                 "graph replay; the fused split-bf16 kernel on 64 x 64 workgroup tiles (eight waves) instead of the 128 x 128 ones "
                 "(sixteen waves) a stage's four layers take by default; the same bits"),
                ("side-stream", {"pn_linear_side_stream": "1"},
                 "graph replay; the grouped pn_linear_wgrad launch of a stage VJP on a second, lowest-priority stream beside the next "
                 "stage's backward pass (two cotangent buffers in turn; same bits).  The matrix pipes are shared: the dX GEMMs beside it "
                 "take 36 us instead of 19.4 (profiles/r06_side_stream.txt; measured with the 64 x 64 tiles: +1.5 %; the 128 x 128 tiles "
                 "hold a whole CU's LDS and the second stream costs time), hence opt-in"),
                ("solution-only", {"ts_trajectory_solution_only": "1"},
                 "PETSc's default trajectory contents (-ts_trajectory_solution_only 1: states only); the stage values of a reversed step "
                 "are recomputed, with autograd's tape (DESIGN section 3, difference 20)"),
            ]
        for name, extra, note in extra_runs:
            try:
                ov = make_ode(extra)
                for _ in range(3):
                    one_solve(ov)
                tv = timed(ov, args.steps)
                variants[name] = {"value": args.nt * args.steps / tv, "unit": "time-steps/s", "note": note}
            except Exception as exc:
                variants[name] = {"error": repr(exc)}
            ov = None
            torch.cuda.empty_cache()
        if args.config == "c3a" and args.dtype == "f32":
            # the reference's CI precision (tests/test_pnode.py:127-130: PETSc built in double): the same solve in fp64
            try:
                a64 = copy.copy(args)
                a64.dtype = "f64"
                pb64 = build(a64)
                try:
                    o64 = graph_ode(q=pb64) if not mode.startswith("eager") else make_ode({"pn_graph_capture": "0"}, pb64)
                except RuntimeError:                   # auto mode kept the eager launches (a solve the GPU bounds): time those
                    o64 = make_ode({}, pb64)
                for _ in range(2):
                    one_solve(o64)
                k64 = max(2, min(args.steps, 5))
                tv = timed(o64, k64)
                v64 = {"value": args.nt * k64 / tv, "unit": "time-steps/s", "dtype": "f64", "steps": k64, "launch_mode": o64.graph_status,
                       "note": "the headline solve in double precision (the reference's CI precision), default launch mode"}
                if prof64 and prof64["vec_us"] > 0:
                    n64, w64 = pb64.y0.numel(), 8
                    npar64 = sum(q.numel() for q in pb64.params)
                    pts = args.nt * prof64["solves"]
                    alg = (ALG_VECTORS_PER_STEP * n64 * w64 + 12 * npar64 * w64) * pts
                    ach = alg / ((prof64["vec_us"] + prof64["par_us"]) * 1e-6) / 1e9
                    avec = ALG_VECTORS_PER_STEP * n64 * w64 * pts / (prof64["vec_us"] * 1e-6) / 1e9
                    v64["roofline"] = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                                       "vector_only": {"achieved": avec, "frac": avec / HBM_PEAK_GBS,
                                                       "us_per_time_step": prof64["vec_us"] / pts},
                                       "solver_kernel_us_per_time_step": (prof64["vec_us"] + prof64["par_us"]) / pts,
                                       "algorithmic_bytes_per_time_step": alg / pts, "per_kernel": prof64["per_kernel"],
                                       "command": prof64["command"], "region": prof64["region"]}
                variants["f64"] = v64
                o64 = pb64 = None
            except Exception as exc:
                variants["f64"] = {"error": repr(exc)}
            torch.cuda.empty_cache()

    if rank == 0:
        n = pb.y0.numel()
        w = 8 if pb.dtype == torch.float64 else 4
        mfma_peak = MFMA_F64_PEAK_TFLOPS if pb.dtype == torch.float64 else MFMA_F32_PEAK_TFLOPS
        n_par = sum(q.numel() for q in pb.params)
        contract = args.config in ("c3a", "c4", "c2")        # rk4, fixed steps, stages stored: SURVEY 8(d)'s closed form
        lin_engine = str(getattr(ode, "linear_param_grads", "")).startswith("engine")
        vec = (0, 2, 3)                       # pn_rk_stage, pn_adj_theta, pn_adj_accum
        solver_ids = tuple(i for i, nm in enumerate(_lib.KERNEL_IDS) if nm not in ("pn_copy", "pn_linear_wgrad"))
        i_wgrad = _lib.KERNEL_IDS.index("pn_linear_wgrad")
        nts = max(nsteps * kr, 1)
        per_kernel = {}
        for i, name in enumerate(_lib.KERNEL_IDS):
            if L[i] and i == i_wgrad:           # the MFMA-bound one: its "bytes" are FLOPs (include/pnode_amd.h)
                per_kernel[name] = {"launches": int(L[i]), "avg_us": us[i] / L[i], "bound": "mfma",
                                    "TFLOPs": by[i] / (us[i] * 1e-6) / 1e12,
                                    "frac_of_mfma_peak": by[i] / (us[i] * 1e-6) / 1e12 / mfma_peak}
            elif L[i]:
                per_kernel[name] = {"launches": int(L[i]), "avg_us": us[i] / L[i],
                                    "GBps_moved": by[i] / (us[i] * 1e-6) / 1e9}
        wgrad_flops = by[i_wgrad] / L[i_wgrad] if L[i_wgrad] else 0.0        # per launch, average over the layers
        if contract:
            v_usec = sum(us[i] for i in vec)
            v_launch = sum(L[i] for i in vec)
            alg_vec = ALG_VECTORS_PER_STEP * n * w
            if lin_engine:
                # round 5: the sensitivities of func's nn.Linear layers are formed by the engine during the stage VJP's backward
                # pass -- dW by the GEMM that forms it, accumulating straight into mu (no separate pass: nothing to count), db by
                # pn_colsum_accum_multi, which reads every layer-output cotangent once.  The algorithmic bytes of that kernel are
                # what its entry point accounts for (each cotangent read once + mu read and written); SURVEY's s*3*np*w credit
                # for a separate accumulation pass no longer applies
                alg_par = by[4] / nts
                if L[i_wgrad] and alg_par == 0.0:
                    alg_note = ("32*N*w (rk4 forward 15 + adjoint 17 state vectors); the parameter sensitivities have no HBM-bound pass "
                                "of their own any more: dW and db of every nn.Linear layer are formed and accumulated by "
                                "pn_linear_wgrad_kernel, an MFMA-bound product priced on its own roofline (roofline.linear_wgrad)")
                else:
                    alg_note = ("32*N*w (rk4 forward 15 + adjoint 17 state vectors) + the bias-sensitivity pass of the engine-side "
                                "Linear accumulation (pn_colsum_accum_multi: every layer-output cotangent of a time step read once: "
                                "%.1f MB; the weight sensitivities are accumulated by the GEMM that forms them, no pass of their own)"
                                % (alg_par / 1e6))
            else:
                # the engine (not autograd's AccumulateGrad) accumulates mu, so the parameter-sensitivity kernel
                # belongs to the path: s stages x (read g, read mu, write mu) x np x w algorithmic bytes per time step
                alg_par = 12 * n_par * w
                alg_note = ("SURVEY 8(d): 32*N*w (rk4 forward 15 + adjoint 17 state vectors) + s*3*np*w (the engine, not autograd, "
                            "accumulates the parameter sensitivities)")
            alg_step = alg_vec + alg_par
            all_usec, all_launch = v_usec + us[4], v_launch + L[4]
        else:
            # adaptive / IMEX sweeps have no closed form in SURVEY 8(d): the algorithmic bytes are what the entry points
            # account for themselves (every distinct input vector read once + every output written once, per launch),
            # summed over the solver launches of a solve; copies of whole states are left out
            all_usec = sum(us[i] for i in solver_ids)
            all_launch = sum(L[i] for i in solver_ids)
            alg_step = sum(by[i] for i in solver_ids) / nts
            v_usec, v_launch, alg_vec, alg_par = all_usec, all_launch, alg_step, 0.0
            alg_note = ("sum over the solver launches of a solve of (distinct input vectors read once + outputs written once), as "
                        "the entry points account for it (pn_prof_collect), divided by the accepted time steps")
        achieved = alg_step * nts / (all_usec * 1e-6) / 1e9 if all_usec > 0 else 0.0
        v_achieved = alg_vec * nts / (v_usec * 1e-6) / 1e9 if v_usec > 0 else 0.0
        ev = {"achieved": achieved, "frac": achieved / HBM_PEAK_GBS, "us_per_time_step": all_usec / nts,
              "launches_per_time_step": all_launch / nts,
              "vector_only": {"achieved": v_achieved, "frac": v_achieved / HBM_PEAK_GBS, "us_per_time_step": v_usec / nts,
                              "avg_launch_us": v_usec / max(v_launch, 1), "launches_per_time_step": v_launch / nts},
              "per_kernel": per_kernel,
              "method": "separate eager pass of the same solve (%d solves); every pn_* dispatch is launched with hipExtLaunchKernelGGL "
                        "start/stop events.  The start event is a marker packet in front of the dispatch, so these durations "
                        "contain one marker-to-dispatch hand-over each (0.6-0.8 us; profiles/README.md) that the profiler's "
                        "dispatch timestamps do not" % kr}
        rp = None
        if prof and prof["vec_us"] > 0:
            pts = nsteps * prof["solves"]
            p_all = prof["vec_us"] + prof["par_us"] + (0.0 if contract else prof["wrms_us"])
            rp_all = alg_step * pts / (p_all * 1e-6) / 1e9
            rp_vec = alg_vec * pts / ((prof["vec_us"] if contract else p_all) * 1e-6) / 1e9
            rp = {"achieved": rp_all, "frac": rp_all / HBM_PEAK_GBS, "us_per_time_step": p_all / pts,
                  "vector_only": {"achieved": rp_vec, "frac": rp_vec / HBM_PEAK_GBS,
                                  "us_per_time_step": (prof["vec_us"] if contract else p_all) / pts},
                  "all_kernels_us_per_time_step": prof["all_kernels_us"] / pts, "wall_us_per_time_step": prof["wall_us"] / pts,
                  "per_kernel": prof["per_kernel"], "command": prof["command"], "region": prof["region"]}
        head = rp or ev
        fallback = None if traffic else pmc_traffic_from_profiles()

        # ---- every kernel on ITS OWN moved bytes (what the inclusive figure cannot show), and the figures next to `frac`
        def kernel_table(per, solves, usec_total=None):
            """per: {kernel name: {launches, avg_us}} of `solves` solves -> the same with bytes moved per launch, GB/s and the
            fraction of the HBM peak; (moved bytes per time step, their kernels' us per time step)."""
            tab, moved, usec = {}, 0.0, 0.0
            srcs = ode._s_eff * nsteps * solves if contract else 0
            for name, v in per.items():
                row = dict(v)
                pl = v["launches"]
                b = moved_bytes_per_launch(name, n, w, n_par, srcs / pl if (pl and name.startswith("pn_param_accum_multi")) else 0)
                if b is None and name.startswith("pn_colsum_partial") and lin_engine and pl:
                    b = alg_par * nsteps * solves / pl             # every cotangent of the launch's sources read once
                if "pn_linear_wgrad_kernel" in name and wgrad_flops:      # (profiler pass: FLOPs per launch from the event pass)
                    row.update(bound="mfma", TFLOPs=wgrad_flops / (v["avg_us"] * 1e-6) / 1e12,
                               frac_of_mfma_peak=wgrad_flops / (v["avg_us"] * 1e-6) / 1e12 / mfma_peak)
                if b is None and "GBps_moved" in v:                # the entry points' own accounting (HIP-event pass)
                    b = v["GBps_moved"] * 1e9 * v["avg_us"] * 1e-6
                if b is not None:
                    row["bytes_moved_per_launch"] = b
                    row["GBps_moved"] = b / (v["avg_us"] * 1e-6) / 1e9
                    row["frac"] = row["GBps_moved"] / HBM_PEAK_GBS
                    if not name.startswith(("pn_lincomb_kernel<float, 1,", "pn_lincomb_kernel<double, 1,", "pn_copy")):
                        moved += b * pl
                        usec += v["avg_us"] * pl
                tab[name] = row
            return tab, moved / max(nsteps * solves, 1), usec / max(nsteps * solves, 1)

        if rp:
            rp["per_kernel"], moved_step, moved_us = kernel_table(rp["per_kernel"], prof["solves"])
        else:
            moved_step, moved_us = kernel_table({k: v for k, v in per_kernel.items()}, kr)[1:]
        ev["per_kernel"] = kernel_table(ev["per_kernel"], kr)[0]
        frac_moved = moved_step / (moved_us * 1e-6) / 1e9 / HBM_PEAK_GBS if moved_us > 0 else None
        lin = {k: v for k, v in head["per_kernel"].items() if k.startswith(("pn_lincomb_kernel", "pn_rk_stage", "pn_adj_", "pn_colsum_partial"))
               and not k.startswith(("pn_lincomb_kernel<float, 1,", "pn_lincomb_kernel<double, 1,")) and "frac" in v}
        dominant = max(lin.items(), key=lambda kv: kv[1]["launches"] * kv[1]["avg_us"])[0] if lin else None
        credits = []
        if contract and not lin_engine:
            # SURVEY 8(d)'s inclusive formula credits s*3*np*w bytes per time step to the parameter accumulation; the batched
            # kernel moves a third of that (it reads mu once per 32 gradient sets, not once per stage): a "fraction" above 1
            # says the credit is accounting, not bytes the kernel moved
            par_us = (prof["par_us"] / (nsteps * prof["solves"])) if rp else us[4] / nts
            if par_us > 0:
                cf = alg_par / (par_us * 1e-6) / 1e9 / HBM_PEAK_GBS
                credits.append({"kernel": "pn_param_accum_multi_kernel", "credited_bytes_per_time_step": alg_par,
                                "us_per_time_step": par_us, "credit_frac": cf, "over_1": cf > 1.0,
                                "note": "credited bytes / kernel time / peak; > 1 means the credit exceeds what the kernel can have moved "
                                        "(see per_kernel[...].frac for the bytes it does move)"})
        ceiling = None
        if ceil_prof or ceil_ev:
            ceiling = dict(ceil_prof or ceil_ev)
            ceiling["unit"] = "GB/s"
            ceiling["hip_events"] = ceil_ev if ceil_prof else None
            ceiling["memcpy_d2d_256MiB_GBps"] = ceil_d2d
            ceiling["kernel"] = "pn_lincomb_kernel<T, 2> through pn_rk_stage: y = u + c*K, two vectors in, one out (3*N*w bytes per launch)"
        def of_ceiling(key):
            c = (ceiling or {}).get(key)
            return c["GBps"] if c else None
        dom = head["per_kernel"].get(dominant) if dominant else None
        # the one MFMA-bound kernel of the path (round 5): dW + db of func's nn.Linear layers, on the fp32 MFMA roofline
        linear_wgrad = None
        wg = [(k, v) for k, v in head["per_kernel"].items() if "pn_linear_wgrad" in k and "finish" not in k and "TFLOPs" in v]
        if wg:
            k, v = wg[0]
            linear_wgrad = {"bound": "mfma", "achieved": v["TFLOPs"], "peak": mfma_peak, "unit": "TFLOP/s",
                            "frac": v["frac_of_mfma_peak"], "kernel": k, "avg_us": v["avg_us"],
                            "launches_per_time_step": v["launches"] / max(nsteps * (prof["solves"] if rp else kr), 1),
                            "flops_per_launch": wgrad_flops,
                            "hbm_bytes_per_launch": (lambda t: (t["read_bytes"] + t["write_bytes"]) if t else None)(
                                next((v for kk, v in ((traffic or {}).get("per_kernel") or {}).items() if kk.startswith("pn_linear_wgrad_kernel")), None)),
                            "arithmetic": ("fp32 operands split exactly into three bf16 terms each; six bf16 x bf16 products per fp32 product "
                                           "(each exact in fp32; the three dropped ones are below 2^-23 |g x|) on v_mfma_f32_32x32x16_bf16 with "
                                           "fp32 accumulation -- error against float64 below an fp32 fmaf chain's (tools/mb_wgrad_bf16x3.hip)"
                                           if "f32x3" in k else
                                           ("v_mfma_f64_16x16x4_f64" if pb.dtype == torch.float64 else "v_mfma_f32_32x32x2_f32: a k-ordered fp32 fmaf chain")),
                            "matrix_pipe": ({"instruction": "v_mfma_f32_32x32x16_bf16", "flops_per_launch": 6.0 * wgrad_flops,
                                             "achieved": 6.0 * v["TFLOPs"], "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                                             "frac": 6.0 * v["TFLOPs"] / MFMA_BF16_PEAK_TFLOPS,
                                             "note": "what the bf16 matrix pipe executes: six times the algorithmic FLOPs, against the dense "
                                                     "bf16 peak.  `frac` above prices the ALGORITHMIC fp32 FLOPs against the fp32 matrix peak "
                                                     "(the pipe this product would otherwise run on); the kernel is no longer bound by either "
                                                     "pipe but by what feeds it -- fragment reads from LDS, the stores of the split parts, the "
                                                     "split's VALU work: the 128 x 128 workgroup tile (kernel ...x3w: sixteen waves, wave tiles "
                                                     "64 x 32) needs 0.75 KB of fragments per MFMA where the 64 x 64 one needs 1, and half the "
                                                     "split work (variants.tile-64-param-grads: the 64 x 64 form, same bits)"}
                                            if "f32x3" in k else None),
                            "pairs_per_launch": round(wgrad_flops / (2.0 * args.batch * args.dim * args.dim), 3) if args.config in ("c3a", "c3b") else None,
                            "us_per_pair": (v["avg_us"] / (wgrad_flops / (2.0 * args.batch * args.dim * args.dim))) if args.config in ("c3a", "c3b") and wgrad_flops else None,
                            "note": "grouped launches (pn_linear_wgrad_group): the (cotangent, input) pairs of all Linear layers of one stage "
                                    "VJP in ONE launch; achieved = 2 * rows * out * in ALGORITHMIC FLOPs per PAIR / time, against the dense "
                                    "matrix peak of the state's dtype (fp32 157.3; fp64 78.6, data sheet); see `arithmetic`; row a-9 of the hot "
                                    "path for func's nn.Linear layers: sum over stages of alpha * (G^T X, column sums of G) into the layer's "
                                    "partial buffers, added to mu once per reverse sweep (pnode_amd/csrc/pn_linear.hip); hbm_bytes_per_launch: "
                                    "the PMC children's FETCH_SIZE (doubled) + WRITE_SIZE for this kernel, G + X + the partial tiles read "
                                    "and the partial tiles written.  The BLAS library's "
                                    "kernel for the same product: variants.library-gemm-param-grads"}
        roofline = {"bound": "hbm", "achieved": head["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": head["frac"],
                    "frac_note": ("SURVEY 8(d)'s figure: 32*N*w algorithmic bytes per time step / (the time of the HBM-bound pn_* kernels: "
                                  "the state-vector launches and the finishing passes) / peak.  The MFMA-bound pn_linear_wgrad_kernel "
                                  "(dW + db of func's nn.Linear layers) is NOT in this time: it is priced on its own roofline, "
                                  "roofline.linear_wgrad; SURVEY's s*3*np*w credit for a parameter-accumulation pass is not taken (no "
                                  "such pass exists on this path)" if (contract and wg) else
                                  "SURVEY 8(d)'s inclusive figure: (32*N*w + s*3*np*w algorithmic bytes) / (all pn_* kernel time) / peak")
                                 if contract else "algorithmic bytes of the solver launches / their kernel time / peak",
                    "frac_state_vectors": head["vector_only"]["frac"],
                    "frac_moved": frac_moved,
                    "frac_traffic": (traffic["hbm_bytes_per_time_step"] / (head["us_per_time_step"] * 1e-6) / 1e9 / HBM_PEAK_GBS)
                                    if traffic and head["us_per_time_step"] > 0 else None,
                    "fracs_note": "frac_state_vectors: the state-vector kernels on SURVEY's 32*N*w; frac_moved: every solver kernel on the "
                                  "bytes it moves (inputs read once + output written once, from its operand count); frac_traffic: "
                                  "on the HBM bytes the PMC counters measured (roofline.traffic_measured)",
                    "dominant_kernel": None if dom is None else {
                        "name": dominant, "avg_us": dom["avg_us"], "GBps_moved": dom["GBps_moved"], "frac": dom["frac"],
                        "frac_of_copy_ceiling": dom["GBps_moved"] / of_ceiling("large_stream") if of_ceiling("large_stream") else None,
                        "frac_of_state_size_ceiling": dom["GBps_moved"] / (of_ceiling("state_size_behind_gemm") or of_ceiling("state_size_cold"))
                        if (of_ceiling("state_size_behind_gemm") or of_ceiling("state_size_cold")) else None},
                    "copy_ceiling": ceiling,
                    "frac_of_copy_ceiling": (frac_moved * HBM_PEAK_GBS / of_ceiling("large_stream"))
                                            if frac_moved and of_ceiling("large_stream") else None,
                    "credits": credits,
                    "linear_wgrad": linear_wgrad,
                    "by_time": ("of the solver's own kernels pn_linear_wgrad takes the most time per time step (MFMA-bound: roofline."
                                "linear_wgrad, %.0f us against %.0f us of the HBM-bound state-vector kernels this object prices by "
                                "SURVEY 8(d)'s 32*N*w); func's own kernels -- the BLAS library's GEMMs and PyTorch's elementwise "
                                "kernels -- are the rest of the step" % (linear_wgrad["avg_us"] * linear_wgrad["launches_per_time_step"],
                                                                         head["us_per_time_step"])) if linear_wgrad else None,
                    "traffic": traffic["hbm_bytes_per_launch"] if traffic else None,
                    "traffic_note": ("HBM bytes per solver-kernel launch MEASURED in this run by two rocprofv3 --pmc children "
                                     "(roofline.traffic_measured)") if traffic else
                                    ("not measured in this run (no counter children: --no-pmc, several ranks, or the profiler is "
                                     "unavailable)" + ("; roofline.traffic_from_profiles is a constant committed under profiles/, NOT a "
                                                       "measurement of this run" if fallback else "")),
                    "traffic_measured": traffic,
                    "traffic_from_profiles": fallback if args.config == "c3a" and args.dtype == "f32" else None,
                    "kernel": "all pn_* kernels of a time step: pn_lincomb_kernel (pn_rk_stage + pn_adj_theta + pn_adj_accum) "
                              "and " + ("pn_colsum_partial_kernel + pn_colsum_finish_kernel (bias sensitivities of the layers the fused "
                                        "kernel does not take; none at this configuration when roofline.linear_wgrad is present)" if lin_engine else
                                        "pn_param_accum_multi_kernel") + ("" if contract else ", pn_combine_wrms_kernel"),
                    "algorithmic_bytes_per_time_step": alg_step,
                    "algorithmic_bytes_note": alg_note,
                    "solver_kernel_us_per_time_step": head["us_per_time_step"],
                    "vector_only": dict(head["vector_only"], algorithmic_bytes_per_time_step=alg_vec,
                                        note="the state-vector kernels alone (32*N*w per time step), round 1's headline figure"
                                        if contract else "same as the whole (no separate parameter term for this config)"),
                    "measured_in": ("the timed mode itself (launch mode %s): kernel durations from a child `rocprofv3 --kernel-trace` run of "
                                    "this workload, last 3 solves (roofline.rocprofv3); the HIP-event figures of the eager pass are in "
                                    "roofline.hip_events" % args.mode) if rp else
                                   ("separate eager pass, HIP start/stop events bound to each dispatch (roofline.hip_events)"
                                    if not mode.startswith("eager") else "the timed region (eager launches), HIP events"),
                    "rocprofv3": rp, "hip_events": ev}
        if args.config == "c2":
            roofline["note"] = ("launch-bound configuration (N = %d elements per state vector): read us_per_time_step, not the "
                                "bandwidth fraction" % n)
        if under_profiler():
            roofline["measured_in"] += ("; NOTE: this process runs under a profiler -- HIP-event durations read inflated there (the tool's "
                                        "interception sits between the event markers): take the kernel durations from the profiler's own output")
        out = {
            "metric": "time-steps/sec (fwd+adjoint)",
            "value": world * nsteps * args.steps / elapsed,
            "unit": "time-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "median": {"value": world * nsteps / (1e-3 * head_rep_ms[len(head_rep_ms) // 2]) if head_rep_ms else None,
                       "ms_per_step": head_rep_ms[len(head_rep_ms) // 2] if head_rep_ms else None,
                       "min_ms": head_rep_ms[0] if head_rep_ms else None, "max_ms": head_rep_ms[-1] if head_rep_ms else None,
                       "reps": len(head_rep_ms),
                       "method": "SURVEY 8(d)(i): every solve (forward sweep + backward) bracketed by HIP events recorded on the stream "
                                 "the sweeps run on (rank 0); median of the --steps reps after --warmup warm-ups.  `value` above is the "
                                 "contract's clock: all reps between two barrier + synchronize brackets, max over ranks"},
            "higher_is_better": True,
            "scaling": "strong" if args.strong else "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {"workload": pb.workload,
                       "batch_per_gpu": args.batch, "state_elements_per_gpu": n, "time_steps": nsteps,
                       "time_steps_per_rank": steps_per_rank,
                       "rejected_attempts": ode.num_rejections if pb.adaptive else 0,
                       "launch_mode": mode, "graph_revalidate_every": getattr(ode, "_revalidate_every", None) if not mode.startswith("eager") else None,
                       "linear_param_grads": getattr(ode, "linear_param_grads", None),
                       "extra_options": args.opt or None,
                       "stage_tapes_retained": bool(tapes_kept), "tunableop": bool(args.tunableop),
                       "parallelism": "batch-sharded x%d, one RCCL all-reduce of dL/dtheta per backward" % world +
                                      (" + one 2-double all-reduce per step attempt (global error norm)" if pb.adaptive else ""),
                       "allreduce_us": allreduce_us, "enorm_allreduce_us": enorm_allreduce_us},
            "roofline": roofline,
            "variants": variants,
        }
        if args.no_roofline_pass:
            out["roofline"] = None
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, args.cpu_seconds)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
