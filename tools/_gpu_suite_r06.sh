R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/suite_r06
mkdir -p $O; cd $R
python -m pytest tests -q -m gpu --durations=15 > $O/gpu_suite.txt 2>&1; tail -4 $O/gpu_suite.txt
timeout 900 python tools/fuzz_modes.py ${FUZZ_CASES:-60} 6 > $O/fuzz_modes.txt 2>&1; tail -2 $O/fuzz_modes.txt
