# last collection of round 6: the -m gpu suite, the adaptive bench lines, the other configs
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/final3_r06
mkdir -p $O; cd $R
python -m pytest tests -q -m gpu --durations=15 > $O/gpu_suite.txt 2>&1; tail -2 $O/gpu_suite.txt
timeout 600 python bench.py --config c3b --no-variants > $O/r06_bench_c3b.json 2> $O/c3b.err
timeout 700 python bench.py --config c3b --stiff --no-variants > $O/r06_bench_c3b_stiff.json 2> $O/c3b_stiff.err
for c in c2 c4 c5; do timeout 900 python bench.py --config $c --steps 5 --warmup 2 > $O/r06_bench_$c.json 2> $O/$c.err; done
timeout 900 python bench.py --dtype f64 --no-variants --no-cpu-baseline > $O/r06_bench_f64.json 2> $O/f64.err
python - <<PY
import json
for f in ("c3b", "c3b_stiff", "c2", "c4", "c5", "f64"):
    try:
        d = json.loads(open("$O/r06_bench_%s.json" % f).read().strip().splitlines()[-1])
        print(f, round(d["value"], 1), d["config"].get("launch_mode"), round(d["roofline"]["frac"], 3))
    except Exception as e:
        print(f, "FAILED", e)
PY
