# kernel trace of adaptive solves replayed from per-evaluation hipGraphs; EAGER=1 for the eager twin
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/units_trace${EAGER:+_eager}
mkdir -p $O
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_units
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_units -- python3 $R/tools/try_units_prof.py > $O/run.log 2>&1
cp $(find /tmp/p_units -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
grep -v "^W20\|^E20" $O/run.log | tail -4
python3 - <<PY
import csv, glob
f = glob.glob("/tmp/p_units/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last three solves: everything after the last long idle gap pattern is hard to find; take the last 3/8 of the launches
n = len(rows)
sel = rows[int(n * 5 / 8):]
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in sel)
wall = int(sel[-1]["End_Timestamp"]) - int(sel[0]["Start_Timestamp"])
print("launches %d (selected %d): kernel time %.1f ms of %.1f ms wall = %.1f %% busy" % (n, len(sel), busy / 1e6, wall / 1e6, 100.0 * busy / wall))
PY
head -12 $O/kernel_stats.csv | cut -c1-150
