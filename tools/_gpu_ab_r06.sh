# interleaved A/B of bench.py option sets on one box: each line of stdin = the extra bench.py arguments of one arm
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/ab_r06
mkdir -p $O; cd $R
mapfile -t ARMS
for round in 1 2 ${ROUNDS:-}; do
  for i in "${!ARMS[@]}"; do
    timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-variants --no-roofline-pass --no-rocprof --no-pmc --no-ceiling ${ARMS[$i]} 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('arm %s round %s: %8.2f %s  (%s)' % ('$i', '$round', d['value'], d['unit'], '${ARMS[$i]}'))"
  done
done | tee -a $O/ab.txt
