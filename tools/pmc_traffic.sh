#!/bin/bash
# Two rocprofv3 counter passes over a short eager bench run (counters only with --kernel-trace, no other trace
# domain), then tools/pmc_traffic.py.  Writes gpurun_out/pmc_traffic.json.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
# rocprofv3 initialises the HIP runtime before python starts: the graph-replay switch must already be in the environment
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
rm -rf /tmp/pmc_f /tmp/pmc_w
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_f -- python3 $R/bench.py --mode eager --steps 1 --warmup 0 --nt 10 --no-cpu-baseline --no-variants > /tmp/pmc_f.log 2>&1
echo "FETCH pass rc=$?"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_w -- python3 $R/bench.py --mode eager --steps 1 --warmup 0 --nt 10 --no-cpu-baseline --no-variants > /tmp/pmc_w.log 2>&1
echo "WRITE pass rc=$?"
mkdir -p $R/gpurun_out
python3 $R/tools/pmc_traffic.py /tmp/pmc_f /tmp/pmc_w $R/gpurun_out/pmc_traffic.json "${1:-}"
