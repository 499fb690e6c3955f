cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2700 python -m pytest tests -m gpu -q 2>&1 | tail -25 > gpurun_out/t6.log; tail -8 gpurun_out/t6.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
