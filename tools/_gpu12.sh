cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2700 python -m pytest tests -m gpu -q 2>&1 | tail -6 > gpurun_out/t12.log; tail -3 gpurun_out/t12.log
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_v5.json 2> gpurun_out/bench_v5.err; tail -2 gpurun_out/bench_v5.err
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
