cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
FIXED=1 timeout 900 python tools/bench_c5.py 3 l2 4 5 > gpurun_out/c5_r02_fixed.txt 2>&1
FIXED=0 timeout 600 python tools/bench_c5.py 3 l2 > gpurun_out/c5_r02.txt 2>&1
timeout 600 python tools/bench_c5_theta.py > gpurun_out/c5_theta_r02.txt 2>&1
grep "time-steps/s" gpurun_out/c5_r02_fixed.txt gpurun_out/c5_r02.txt | head -30; tail -12 gpurun_out/c5_theta_r02.txt
