R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/suite_r06
mkdir -p $O; cd $R
python -m pytest tests -q -m gpu -x --durations=15 > $O/gpu_suite.txt 2>&1; tail -4 $O/gpu_suite.txt
timeout 600 python tools/profile_stiff_reverse.py > $O/stiff_reverse_host.txt 2>&1
timeout 900 python bench.py --config c3b --no-cpu-baseline --no-variants > $O/bench_c3b.json 2> $O/bench_c3b.err
timeout 900 python bench.py --config c3b --stiff --no-cpu-baseline --no-variants > $O/bench_c3b_stiff.json 2> $O/bench_c3b_stiff.err
python - <<PY
import json
for f in ("bench_c3b", "bench_c3b_stiff"):
    try:
        d = json.loads(open("$O/%s.json" % f).read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d["config"].get("time_steps"), d["roofline"].get("frac"), d.get("gpu_busy") or d["roofline"].get("gpu_busy"))
    except Exception as e:
        print(f, "failed", e)
PY
