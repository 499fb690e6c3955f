cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
./tools/mb_events > gpurun_out/mb_events.txt 2>&1; cat gpurun_out/mb_events.txt
timeout 2700 python -m pytest tests -m gpu -q 2>&1 | tail -12 > gpurun_out/t7.log; tail -5 gpurun_out/t7.log
