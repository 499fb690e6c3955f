cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2700 python -m pytest tests -m gpu -q 2>&1 | tail -8 > gpurun_out/t11.log; tail -4 gpurun_out/t11.log
timeout 600 python tools/bench_budget.py > gpurun_out/bench_budget_r02.txt 2>&1; tail -20 gpurun_out/bench_budget_r02.txt
