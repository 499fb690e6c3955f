set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_configs.py tests/test_gpu_distributed.py tests/test_gpu_kernels.py -m gpu -q -x 2>&1 | tail -40 > gpurun_out/t1.log
timeout 600 python bench.py --steps 5 --warmup 2 > gpurun_out/bench_v0.json 2> gpurun_out/bench_v0.err
timeout 600 python tools/ab_r02.py "" "block=512" "block=1024" "block=512,vpt=1" "block=1024,vpt=1" "vpt=1" > gpurun_out/ab_r02.txt 2>&1
tail -5 gpurun_out/t1.log; cat gpurun_out/bench_v0.json | head -c 1500; tail -30 gpurun_out/ab_r02.txt
