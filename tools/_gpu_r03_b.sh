# Round 3, GPU call B: combine kernel with sharded arrival counters; device-resident GMRES + replayed linearisations.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_b
mkdir -p $O
cd $R
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
timeout 900 python -m pytest tests/test_gpu_krylov.py -x -q -m gpu > $O/krylov_tests.log 2>&1; echo "krylov rc $?" >> $O/krylov_tests.log
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "wrms or dots" > $O/kernels.log 2>&1; echo "kernels rc $?" >> $O/kernels.log
timeout 900 python tools/bench_c5_theta.py > $O/r03_c5_theta.txt 2>&1
cd /tmp && export TMPDIR=/tmp
for v in 1 2; do
  rm -rf /tmp/p_c3b_$v
  PN_TUNE="wvpt=$v" timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_c3b_$v -- python3 $R/tools/prof_c3b.py --solves 2 > $O/c3b_wvpt$v.log 2>&1
  python3 $R/tools/trace_stats.py /tmp/p_c3b_$v $O/r03_c3b_wvpt${v}_trace_stats.csv --label "PN_TUNE=wvpt=$v tools/prof_c3b.py --solves 2" > /dev/null
done
rm -rf /tmp/p_c5
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_c5 -- python3 $R/tools/bench_c5_theta.py --only-default > $O/c5_prof.log 2>&1
cp $(find /tmp/p_c5 -name "*kernel_stats.csv" | head -1) $O/r03_c5_theta_kernel_stats.csv
cd $R
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_distributed.py -x -q -m gpu -k "theta or imex or c5 or dae or implicit or c3b or adaptive" > $O/parity_subset.log 2>&1; echo "subset rc $?" >> $O/parity_subset.log
tail -n 5 $O/krylov_tests.log $O/kernels.log $O/parity_subset.log
cat $O/r03_c5_theta.txt | grep "C5 shard"
grep -h "combine" $O/*trace_stats.csv
