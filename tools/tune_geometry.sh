#!/bin/bash
# Runs bench.py once per launch geometry of the streaming kernels (PN_TUNE) on the GPU box and
# prints the solver-kernel roofline of each.  Usage: tools/tune_geometry.sh [extra bench args]
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
# block=512|1024 are real instantiations since round 2 (vpt 1|2, default cache policy); vpt=4 exists for block=256 only
for geo in "vpt=1,block=256" "vpt=2,block=256" "vpt=4,block=256" "vpt=1,block=512" "vpt=2,block=512" "vpt=1,block=1024" "vpt=2,block=1024"; do
  PN_TUNE="$geo" python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-rocprof --no-variants "$@" 2>/dev/null | tail -1 > gpurun_out/tune_$geo.json
  python - "$geo" <<'PY'
import json, sys
geo = sys.argv[1]
d = json.load(open("gpurun_out/tune_%s.json" % geo))
r = d["roofline"]
pk = r["hip_events"]["per_kernel"]
print("%-18s value %7.1f  achieved %7.1f GB/s frac %.3f  us/step %.2f  | " % (geo, d["value"], r["achieved"], r["frac"], r["solver_kernel_us_per_time_step"])
      + "  ".join("%s %.2fus" % (k.replace("pn_", ""), v["avg_us"]) for k, v in pk.items()))
PY
done
