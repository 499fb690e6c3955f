set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_configs.py tests/test_gpu_kernels.py "tests/test_gpu_parity.py::test_retain_graph_mode_bitwise_identical_on_gpu" -m gpu -q 2>&1 | tail -30 > gpurun_out/t3.log
timeout 600 python tools/ab_r02.py "" > gpurun_out/ab_r02_c.txt 2>&1
tail -8 gpurun_out/t3.log; cat gpurun_out/ab_r02_c.txt
