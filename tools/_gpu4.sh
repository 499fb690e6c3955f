cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python bench.py --steps 10 --warmup 3 > gpurun_out/bench_v2.json 2> gpurun_out/bench_v2.err
timeout 900 python -m pytest tests/test_gpu_distributed.py tests/test_gpu_configs.py -m gpu -q 2>&1 | tail -15 > gpurun_out/t4.log
tail -5 gpurun_out/bench_v2.err; tail -6 gpurun_out/t4.log
