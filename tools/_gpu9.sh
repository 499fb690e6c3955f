cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2700 python -m pytest tests -m gpu -q 2>&1 | tail -8 > gpurun_out/t9.log; tail -4 gpurun_out/t9.log
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_v4.json 2> gpurun_out/bench_v4.err; tail -2 gpurun_out/bench_v4.err
timeout 900 python bench.py --config c4 --steps 20 --warmup 5 > gpurun_out/bench_c4_v4.json 2> gpurun_out/bench_c4_v4.err; tail -2 gpurun_out/bench_c4_v4.err
