cd ${GRAFT_REPO_ROOT:-/root/repo}
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
mkdir -p gpurun_out/r03_n
timeout 900 python -m pytest tests/test_gpu_krylov.py tests/test_gpu_abi_client.py -x -q -m gpu > gpurun_out/r03_n/tests.log 2>&1; echo "rc $?" >> gpurun_out/r03_n/tests.log
timeout 600 python tools/bench_c5_theta.py --only-default > gpurun_out/r03_n/c5_default.txt 2>&1
tail -n 5 gpurun_out/r03_n/tests.log; grep "C5 shard" gpurun_out/r03_n/c5_default.txt | cut -c1-280
