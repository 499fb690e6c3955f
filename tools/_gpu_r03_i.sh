R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_i
mkdir -p $O
cd $R
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
timeout 900 python -m pytest tests/test_gpu_krylov.py tests/test_gpu_distributed.py -x -q -m gpu -k "krylov or forward_mode or auto_mode or rccl or two_rank or replayed" > $O/tests.log 2>&1; echo "tests rc $?" >> $O/tests.log
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_configs.py -x -q -m gpu -k "wrms or c3b" > $O/tests2.log 2>&1; echo "tests2 rc $?" >> $O/tests2.log
timeout 600 python tools/leak_check.py > $O/r03_leak_check.txt 2>&1
timeout 900 python tools/fuzz_modes.py 120 3 > $O/r03_fuzz_modes.txt 2>&1
timeout 600 python tools/fuzz_imex.py > $O/r03_fuzz_imex.txt 2>&1
timeout 1200 python tools/bench_c5_theta.py --only-default > $O/c5_default_stencil.txt 2>&1
timeout 600 python tools/prof_krylov.py default > $O/prof_krylov_conv1d_default.txt 2>&1
tail -4 $O/tests.log $O/tests2.log
cat $O/r03_leak_check.txt
tail -3 $O/r03_fuzz_modes.txt $O/r03_fuzz_imex.txt
grep "C5 shard" $O/c5_default_stencil.txt
head -8 $O/prof_krylov_conv1d_default.txt
