R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_j
mkdir -p $O
cd $R
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
timeout 900 python -m pytest tests/test_gpu_distributed.py tests/test_gpu_kernels.py tests/test_gpu_krylov.py -x -q -m gpu -k "rccl or wrms or krylov or gmres or replayed or forward_mode or auto_mode" > $O/tests.log 2>&1; echo "tests rc $?" >> $O/tests.log
timeout 1500 python tools/bench_c5_theta.py > $O/r03_c5_theta.txt 2>&1
timeout 900 python tools/bench_c5_theta.py --only-default --tunableop > $O/r03_c5_theta_tunableop.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_kr
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_kr -- python3 $R/tools/prof_krylov.py default stencil --trace-only > $O/kr_trace.log 2>&1
cp $(find /tmp/p_kr -name "*kernel_stats.csv" | head -1) $O/r03_krylov_stencil_kernel_stats.csv
python3 $R/tools/krylov_trace_summary.py /tmp/p_kr > $O/r03_krylov_stencil_trace_summary.txt 2>&1
cd $R
tail -n 4 $O/tests.log
grep "C5 shard" $O/r03_c5_theta.txt $O/r03_c5_theta_tunableop.txt | cut -c1-330
head -14 $O/r03_krylov_stencil_trace_summary.txt
