cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2700 python -m pytest tests -m gpu -q 2>&1 | tail -12 > gpurun_out/t10.log; tail -5 gpurun_out/t10.log
for i in 1 2 3; do timeout 600 python bench.py --steps 10 --warmup 3 --no-variants --no-cpu-baseline >> gpurun_out/bench_repeat_r02.jsonl 2>/dev/null; done
