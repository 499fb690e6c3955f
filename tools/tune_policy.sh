#!/bin/bash
# Runs bench.py once per load/store cache policy of the streaming kernels (PN_TUNE) on the GPU
# box and prints the solver-kernel roofline of each.  Usage: tools/tune_policy.sh "cfg1" "cfg2" ...
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
i=0
for cfg in "$@"; do
  i=$((i+1))
  PN_TUNE="$cfg" python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/tune_policy_$i.json
  python - "$cfg" gpurun_out/tune_policy_$i.json <<'PY'
import json, sys
cfg, f = sys.argv[1], sys.argv[2]
d = json.load(open(f))
r = d["roofline"]
pk = r["per_kernel"]
print("%-28s value %7.1f  achieved %7.1f GB/s frac %.3f  us/step %.2f  | " % (cfg, d["value"], r["achieved"], r["frac"], r["solver_kernel_us_per_time_step"])
      + "  ".join("%s %.2fus" % (k.replace("pn_", ""), v["avg_us"]) for k, v in pk.items()))
PY
done
