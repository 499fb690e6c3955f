set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -40 > gpurun_out/t2.log
timeout 600 python tools/mb_floor.py > gpurun_out/mb_floor.txt 2>&1
timeout 600 python tools/ab_r02.py "" > gpurun_out/ab_r02_b.txt 2>&1
tail -8 gpurun_out/t2.log; cat gpurun_out/mb_floor.txt; cat gpurun_out/ab_r02_b.txt
