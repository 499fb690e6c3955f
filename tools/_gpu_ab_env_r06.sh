# interleaved A/B of environment settings: each line of stdin = "ENV=... ENV2=..." (may be empty) for one arm
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/ab_r06
mkdir -p $O; cd $R
mapfile -t ARMS
for round in 1 2 ${ROUNDS:-}; do
  for i in "${!ARMS[@]}"; do
    env ${ARMS[$i]} timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-variants --no-pmc --no-ceiling 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
lw = d['roofline'].get('linear_wgrad') or {}
print('arm %s round %s: %8.2f %s  hbm frac %.3f  wgrad us/pair %s frac %s (%s)' % ('$i', '$round', d['value'], d['unit'], d['roofline']['frac'], lw.get('us_per_pair'), lw.get('frac'), '${ARMS[$i]}'))"
  done
done | tee -a $O/ab_env.txt
