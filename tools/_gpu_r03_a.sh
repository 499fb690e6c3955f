# Round 3, GPU call A: kernel tests of the single-launch reductions, the whole -m gpu suite, C3b profile (combine kernel).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_a
mkdir -p $O
cd $R
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "wrms or dots" > $O/kernels.log 2>&1; echo "kernels rc $?" >> $O/kernels.log
timeout 1500 python -m pytest tests -x -q -m gpu > $O/gpu_suite.log 2>&1; echo "suite rc $?" >> $O/gpu_suite.log
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_c3b
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_c3b -- python3 $R/tools/prof_c3b.py --solves 3 > $O/c3b.log 2>&1
cp $(find /tmp/p_c3b -name "*kernel_stats.csv" | head -1) $O/r03_c3b_kernel_stats.csv
python3 $R/tools/trace_stats.py /tmp/p_c3b $O/r03_c3b_trace_stats.csv --label "tools/prof_c3b.py --solves 3 (C3b: dopri5 adaptive, 4096 x 512 fp32, max_cps 50), whole run" > /dev/null
for v in 1 2; do
  rm -rf /tmp/p_c3b_$v
  PN_TUNE="wvpt=$v" timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_c3b_$v -- python3 $R/tools/prof_c3b.py --solves 2 > $O/c3b_wvpt$v.log 2>&1
  python3 $R/tools/trace_stats.py /tmp/p_c3b_$v $O/r03_c3b_wvpt${v}_trace_stats.csv --label "PN_TUNE=wvpt=$v tools/prof_c3b.py --solves 2" > /dev/null
done
tail -5 $O/kernels.log $O/gpu_suite.log $O/c3b.log
grep -h "combine" $O/*trace_stats.csv
