R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_k
mkdir -p $O
cd $R
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
timeout 600 python -m pytest tests/test_gpu_abi_client.py -x -q -m gpu -s > $O/abi_client.log 2>&1; echo "abi rc $?" >> $O/abi_client.log
timeout 1500 python tools/bench_c5_imex_krylov.py > $O/r03_c5_imex_krylov.txt 2>&1
tail -n 6 $O/abi_client.log
grep "C5 shard" $O/r03_c5_imex_krylov.txt | cut -c1-300
