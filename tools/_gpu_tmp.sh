cd ${GRAFT_REPO_ROOT:-/root/repo}
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
mkdir -p gpurun_out/r03_q
timeout 900 python -m pytest tests/test_gpu_krylov.py tests/test_gpu_parity.py -x -q -m gpu -k "krylov or gmres or replayed or forward_mode or auto_mode or theta or imex or captured or changed" > gpurun_out/r03_q/tests.log 2>&1; echo "rc $?" >> gpurun_out/r03_q/tests.log
timeout 900 python tools/bench_c5_imex_krylov.py > gpurun_out/r03_q/imex.txt 2>&1
tail -n 4 gpurun_out/r03_q/tests.log; grep "C5 shard" gpurun_out/r03_q/imex.txt | cut -c1-250
