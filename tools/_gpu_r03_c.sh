R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_c
mkdir -p $O
cd $R
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
for w in default nograph host; do timeout 300 python tools/prof_krylov.py $w > $O/prof_krylov_$w.txt 2>&1; done
timeout 300 python tools/profile_host_noop.py > $O/profile_host_noop.txt 2>&1
head -30 $O/prof_krylov_default.txt
