# robustness runs of the final tree (after the per-evaluation graphs and the 128 x 128 tile form)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/robust_r06
mkdir -p $O; cd $R
timeout 900 python tools/fuzz_guard.py 100 5 24 > $O/fuzz_guard.txt 2>&1
timeout 1500 python tools/fuzz_modes.py 300 12 > $O/fuzz_modes.txt 2>&1
timeout 900 python tools/fuzz_imex.py > $O/fuzz_imex.txt 2>&1
ITERS=100 timeout 600 python tools/soak_graph.py > $O/soak_graph.txt 2>&1
timeout 900 python tools/leak_check.py > $O/leak_check.txt 2>&1
timeout 600 python tools/accuracy_c3a.py > $O/accuracy.txt 2>&1
for f in fuzz_guard fuzz_modes fuzz_imex soak_graph leak_check accuracy; do echo "== $f"; grep -v "Warning\|warnings.warn\|^  \|amdgpu.ids" $O/$f.txt | tail -4; done
