"""CPU-side checks of bench.py's host logic (no GPU): the self-launcher's refusal and the parsing of the profiler child's
kernel trace (exercised with a stand-in `rocprofv3` that writes a synthetic trace)."""
import os
import stat
import subprocess
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gpus_flag_without_launcher_refuses_when_the_box_has_fewer_gpus():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "PN_BENCH_BACKEND")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0 and "needs 64 GPUs" in r.stderr and "{" not in r.stdout


def test_launcher_rank_count_must_match_gpus_flag():
    env = dict({k: v for k, v in os.environ.items() if k != "PN_BENCH_BACKEND"}, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0 and "launcher started 2 ranks" in r.stderr


def test_profiler_child_trace_is_summarised_over_the_timed_solves_only(tmp_path, monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    nt = 2
    fake = tmp_path / "rocprofv3"
    # per solve: 3 state copies (NIN=1), 6*nt three-vector and 2*nt six-vector launches, one accumulation launch and some of
    # func's kernels; graph mode (-pn_graph_capture auto) = 4 set-up solves' worth of launches (two eager calls, then the call that
    # runs the sweeps eagerly AND replays its capture) + 1 warm-up + 3 timed.  Set-up solves get absurd durations: they must not
    # show up in the summary.
    fake.write_text('''#!%s
import os, sys
d = sys.argv[sys.argv.index("-d") + 1]
os.makedirs(os.path.join(d, "host"), exist_ok=True)
rows = ["Kind,Agent_Id,Queue_Id,Kernel_Name,Start_Timestamp,End_Timestamp"]
t = 1000
def k(name, dur):
    global t
    rows.append('KERNEL_DISPATCH,1,1,"%%s",%%d,%%d' %% (name, t, t + dur)); t += dur + 100
broken = "eager" in sys.argv
for solve in range(8):
    slow = 50000 if solve < 4 else 0
    if not (broken and solve == 6):
        k("void (anonymous namespace)::pn_lincomb_kernel<float, 1, 4, 2, false, 256, 0, 1>(x)", 4000 + slow)
    for step in range(%d):
        for _ in range(3):
            k("Cijk_gemm", 20000)
            k("void (anonymous namespace)::pn_lincomb_kernel<float, 2, 4, 2, false, 256, 0, 1>(x)", 5000 + slow)
        k("void (anonymous namespace)::pn_lincomb_kernel<float, 5, 4, 2, false, 256, 0, 1>(x)", 8000 + slow)
    k("void (anonymous namespace)::pn_lincomb_kernel<float, 1, 4, 2, false, 256, 0, 1>(x)", 4000 + slow)
    k("void (anonymous namespace)::pn_lincomb_kernel<float, 1, 4, 2, false, 256, 0, 1>(x)", 4000 + slow)
    for step in range(%d):
        for _ in range(3):
            k("Cijk_gemm_bwd", 60000)
            k("void (anonymous namespace)::pn_lincomb_kernel<float, 2, 4, 2, false, 256, 0, 1>(x)", 5000 + slow)
        k("void (anonymous namespace)::pn_lincomb_kernel<float, 5, 4, 2, false, 256, 0, 1>(x)", 8000 + slow)
    k("void (anonymous namespace)::pn_param_accum_multi_kernel<float, 4>(x)", 30000 + slow)
open(os.path.join(d, "host", "123_kernel_trace.csv"), "w").write("\\n".join(rows) + "\\n")
''' % (sys.executable, nt, nt))
    fake.chmod(fake.stat().st_mode | stat.S_IEXEC)
    monkeypatch.setenv("PATH", str(tmp_path) + os.pathsep + os.environ["PATH"])
    args = types.SimpleNamespace(config="c3a", mode="graph", batch=64, dim=16, nt=nt, dt=0.01)
    prof = bench.rocprof_child(args)
    assert prof is not None and prof["time_steps"] == 3 * nt
    assert prof["vec_us"] == 3 * nt * (6 * 5.0 + 2 * 8.0)            # only the last three solves, copies excluded
    assert prof["par_us"] == 3 * 30.0
    pk = prof["per_kernel"]
    assert pk["pn_lincomb_kernel<float, 2, 4, 2, false, 256, 0, 1>"] == {"launches": 3 * 6 * nt, "avg_us": 5.0}
    assert pk["pn_param_accum_multi_kernel<float, 4>"]["launches"] == 3
    # a trace whose whole-state copies do not make whole solves is refused, not mis-summarised
    args.mode = "eager"                                              # (the fake then drops one copy)
    assert bench.rocprof_child(args) is None


def test_moved_bytes_and_ceiling_summary_helpers():
    """The two pieces of arithmetic behind roofline.per_kernel[*].frac and roofline.copy_ceiling: bytes a
    launch moves from its kernel name (K inputs + 1 output for pn_lincomb_kernel<T, K>; gradient sets + mu read + mu written
    for the batched accumulation) and GB/s per block of the streaming microbenchmark from a flat list of launch durations."""
    import bench
    n, w, npar = 2097152, 4, 1050624
    assert bench.moved_bytes_per_launch("pn_lincomb_kernel<float, 2, 4, 2, false, 256, 0, 1>", n, w, npar, 0) == 3 * n * w
    assert bench.moved_bytes_per_launch("pn_lincomb_kernel<double, 5, 2, 1, false, 256, 0, 1>", n, 8, npar, 0) == 6 * n * 8
    assert bench.moved_bytes_per_launch("pn_lincomb_kernel<float, 1, 4, 2, false, 256, 0, 1>", n, w, npar, 0) == 2 * n * w
    # 400 stage results of a 100-step rk4 solve in 13 launches: (400 / 13 + 2) * np * w per launch = 17.9 MB per time step
    b = bench.moved_bytes_per_launch("pn_param_accum_multi_kernel<float, 4, 1, true>", n, w, npar, 400 / 13)
    assert abs(b * 13 / 100 - (4 + 2 * 0.13) * npar * w) < 1
    assert bench.moved_bytes_per_launch("pn_combine_wrms_kernel<float, 6, 4, 4, false, 1, 0>", n, w, npar, 0) is None
    blocks = [("large_stream", 2, 10, 3 * (256 << 20)), ("state_size_cold", 4, 4, 3 * n * w)]
    durations = [999.0] * 2 + [134.0] * 10 + [50.0] * 4 + [6.0] * 4          # us, in launch order (warm-up launches first)
    res = bench.ceiling_summary(blocks, lambda a, b2: sum(durations[a:b2]), "test")
    assert abs(res["large_stream"]["GBps"] - 3 * (256 << 20) / 134e-6 / 1e9) < 1e-6 and res["large_stream"]["launches"] == 10
    assert abs(res["state_size_cold"]["avg_us"] - 6.0) < 1e-12 and res["instrument"] == "test"
    # the HIP-event flavour carries its own durations
    res = bench.ceiling_summary([("state_size_hot", 4, 24, 3 * n * w, 24 * 4.5)], None, "events")
    assert abs(res["state_size_hot"]["GBps"] - 3 * n * w / 4.5e-6 / 1e9) < 1e-6


def test_measured_traffic_keeps_the_mfma_bound_kernel_out_of_the_hbm_average(monkeypatch):
    """roofline.traffic is the HBM bytes per launch of the HBM-bound solver kernels: state copies and the MFMA-bound
    pn_linear_wgrad kernels are listed per kernel but stay out of the average (they are priced in roofline.linear_wgrad);
    FETCH_SIZE is doubled, both counters are KiB."""
    sys.path.insert(0, ROOT)
    import bench
    fetch = {"per_kernel": {"pn_lincomb_kernel<float, 2, 4, 2, false, 256, 0, 1>": [8192.0] * 6,            # KiB, half of the bytes read
                            "pn_lincomb_kernel<float, 1, 4, 2, false, 256, 0, 1>": [4096.0] * 3,
                            "pn_linear_wgrad_kernel": [12288.0] * 16, "pn_linear_wgrad_finish_kernel": [4608.0] * 4},
             "time_steps": 2, "command": "fetch"}
    write = {"per_kernel": {"pn_lincomb_kernel<float, 2, 4, 2, false, 256, 0, 1>": [8192.0] * 6,
                            "pn_lincomb_kernel<float, 1, 4, 2, false, 256, 0, 1>": [8192.0] * 3,
                            "pn_linear_wgrad_kernel": [8448.0] * 16, "pn_linear_wgrad_finish_kernel": [9216.0] * 4},
             "time_steps": 2, "command": "write"}
    monkeypatch.setattr(bench, "pmc_child", lambda args, counter: fetch if counter == "FETCH_SIZE" else write)
    t = bench.pmc_traffic(types.SimpleNamespace())
    per_launch = 2 * 8192 * 1024 + 8192 * 1024
    assert t["launches"] == 6 and t["hbm_bytes_per_launch"] == per_launch and t["hbm_bytes_per_time_step"] == 6 * per_launch / 2
    wg = t["per_kernel"]["pn_linear_wgrad_kernel"]
    assert wg["launches"] == 16 and wg["read_bytes"] == 2 * 12288 * 1024 and wg["write_bytes"] == 8448 * 1024
    assert "pn_linear_wgrad" in t["units"]
