R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_e
mkdir -p $O
cd $R
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
timeout 600 python -m pytest tests/test_gpu_krylov.py -x -q -m gpu -k "two_rank" > $O/krylov_two_rank.log 2>&1; echo "two_rank rc $?" >> $O/krylov_two_rank.log
timeout 900 python -m pytest tests/test_gpu_krylov.py -x -q -m gpu -k "not two_rank" > $O/krylov_tests.log 2>&1; echo "krylov rc $?" >> $O/krylov_tests.log
timeout 1200 python tools/bench_c5_theta.py > $O/r03_c5_theta.txt 2>&1
for w in default; do timeout 300 python tools/prof_krylov.py $w > $O/prof_krylov_$w.txt 2>&1; done
cd /tmp && export TMPDIR=/tmp
for v in "wvpt=2" "wvpt=4"; do
  rm -rf /tmp/p_c3b_x
  PN_TUNE="$v" timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_c3b_x -- python3 $R/tools/prof_c3b.py --solves 2 > $O/c3b_$v.log 2>&1
  python3 $R/tools/trace_stats.py /tmp/p_c3b_x "$O/r03_c3b_${v}_trace_stats.csv" --label "PN_TUNE=$v tools/prof_c3b.py --solves 2" > /dev/null
done
cd $R
timeout 1500 python -m pytest tests/test_gpu_distributed.py -x -q -m gpu -k "bench" > $O/bench_tests.log 2>&1; echo "bench tests rc $?" >> $O/bench_tests.log
timeout 600 python bench.py --steps 5 --warmup 2 > $O/r03_bench.json 2> $O/r03_bench.err; echo "bench rc $?" >> $O/r03_bench.err
tail -n 8 $O/krylov_two_rank.log $O/krylov_tests.log $O/bench_tests.log
grep "C5 shard" $O/r03_c5_theta.txt
grep -h "combine" $O/*trace_stats.csv
head -8 $O/prof_krylov_default.txt
tail -3 $O/r03_bench.err; head -c 1500 $O/r03_bench.json
