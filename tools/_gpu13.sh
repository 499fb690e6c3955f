cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for i in 1 2; do timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -3 >> gpurun_out/t13.log; done
cat gpurun_out/t13.log
