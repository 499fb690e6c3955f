R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_d
mkdir -p $O
cd $R
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
timeout 900 python -m pytest tests/test_gpu_krylov.py -x -q -m gpu > $O/krylov_tests.log 2>&1; echo "krylov rc $?" >> $O/krylov_tests.log
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_abi_client.py -x -q -m gpu -k "wrms or dots or abi" > $O/kernels.log 2>&1; echo "kernels rc $?" >> $O/kernels.log
for w in default nograph; do timeout 300 python tools/prof_krylov.py $w > $O/prof_krylov_$w.txt 2>&1; done
timeout 900 python tools/bench_c5_theta.py > $O/r03_c5_theta.txt 2>&1
cd /tmp && export TMPDIR=/tmp
for v in "wfin=0" "wfin=1" "wfin=0,wvpt=1"; do
  rm -rf /tmp/p_c3b_x
  PN_TUNE="$v" timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_c3b_x -- python3 $R/tools/prof_c3b.py --solves 2 > $O/c3b_$v.log 2>&1
  python3 $R/tools/trace_stats.py /tmp/p_c3b_x "$O/r03_c3b_${v}_trace_stats.csv" --label "PN_TUNE=$v tools/prof_c3b.py --solves 2" > /dev/null
done
cd $R
timeout 300 python tools/profile_host_noop.py > $O/profile_host_noop.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q -m gpu -k "c3b or adaptive or dopri5 or reference_defaults or c1_literal or ten_thousand" > $O/parity_subset.log 2>&1; echo "subset rc $?" >> $O/parity_subset.log
tail -n 6 $O/krylov_tests.log $O/kernels.log $O/parity_subset.log
grep "C5 shard" $O/r03_c5_theta.txt
grep -h "combine" $O/*trace_stats.csv
head -12 $O/prof_krylov_default.txt
head -4 $O/profile_host_noop.txt
