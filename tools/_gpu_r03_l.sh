R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_l
mkdir -p $O
cd $R
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
timeout 900 python -m pytest tests/test_gpu_krylov.py tests/test_gpu_abi_client.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc $?" >> $O/tests.log
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_distributed.py -x -q -m gpu -k "theta or imex or c5 or dae or implicit or rccl" > $O/tests2.log 2>&1; echo "tests2 rc $?" >> $O/tests2.log
timeout 1500 python tools/bench_c5_imex_krylov.py > $O/r03_c5_imex_krylov.txt 2>&1
timeout 1200 python tools/bench_c5_theta.py --only-default > $O/c5_default.txt 2>&1
timeout 900 python tools/bench_c5_theta.py --only-default --tunableop > $O/c5_default_tunableop.txt 2>&1
tail -n 4 $O/tests.log $O/tests2.log
grep "C5 shard" $O/r03_c5_imex_krylov.txt $O/c5_default.txt $O/c5_default_tunableop.txt | cut -c1-300
