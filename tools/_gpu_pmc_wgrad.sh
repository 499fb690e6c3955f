# SQ counters of pn_linear_wgrad_kernel (one pass, 8 SQ slots) under the C++ loop of tools/mb_wgrad_abi (single-pair and grouped launches)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmc_wgrad
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export LD_LIBRARY_PATH=$R/pnode_amd/lib
rm -rf /tmp/p_wg
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d /tmp/p_wg/a -- $R/tools/mb_wgrad_abi > $O/run.log 2>&1
export MB_EXACT=1          # (read by the program: the grouped launches on the fp32 matrix instruction)
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d /tmp/p_wg/b -- $R/tools/mb_wgrad_abi >> $O/run.log 2>&1
unset MB_EXACT
export MB_TILE64=1         # (the grouped launches of the split-bf16 form on 64 x 64 tiles)
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d /tmp/p_wg/c -- $R/tools/mb_wgrad_abi >> $O/run.log 2>&1
unset MB_TILE64
python3 - <<'PY' > $O/r06_pmc_wgrad.txt 2>&1
import csv, glob, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/p_wg/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        m = re.search(r"pn_\w+", row["Kernel_Name"])
        name = m.group(0) if m else row["Kernel_Name"].split("(")[0]
        acc[(name, int(row.get("Grid_Size", 0) or 0))][row["Counter_Name"]].append(float(row["Counter_Value"]))
print("rocprofv3 --kernel-trace --pmc <8 SQ counters> -- tools/mb_wgrad_abi, as it is (grouped launches: the split-bf16 form on 128 x 128 tiles), with MB_EXACT=1 (grouped launches on the fp32 matrix instruction) and with MB_TILE64=1 (the split-bf16 form on 64 x 64 tiles)   (averages per dispatch; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are quad-cycles)")
for (k, grid), d in sorted(acc.items(), key=lambda kv: ("pn_linear" not in kv[0][0], kv[0][1])):
    print(k, "grid", grid, "threads; dispatches", max(len(v) for v in d.values()))
    for c, v in sorted(d.items()):
        print("   %-28s %16.1f" % (c, sum(v) / len(v)))
    if "pn_linear_wgrad_kernel" not in k:
        continue
    g = lambda c: (sum(d[c]) / len(d[c]) if c in d and d[c] else float("nan")) or float("nan")
    wide = k.endswith("x3w")              # 128 x 128 tiles: 128 workgroups of 1024 threads per pair, one per CU
    pairs = max(grid // (128 * 1024 if wide else 512 * 512), 1)   # (64 x 64 tiles: 512 workgroups of 512 threads per 4096 x 512 x 512 pair)
    waves = (128 * 16 if wide else 512 * 8) * pairs
    busy = g("SQ_VALU_MFMA_BUSY_CYCLES") / 1024
    life = 4 * g("SQ_WAVE_CYCLES") / waves
    x3 = k.endswith("x3") or wide
    # waves per SIMD at a time: the fp32-instruction form keeps to 80 VGPRs (three workgroups per CU), the split-bf16 form has two
    conc = min(4 if x3 else 6, waves // 1024)
    resid = waves / 1024.0 * life / conc
    print("   pairs per launch                 %d" % pairs)
    print("   MFMA busy cycles per SIMD        %.0f  (%s)" % (busy, "6 products x 32 cycles x 16 slabs x 4 waves x pairs = %d (the same MFMAs on either tile)" % (6 * 32 * 16 * 4 * pairs) if x3 else "128 MFMAs x 64 cycles x 4 waves x pairs = %d" % (128 * 64 * 4 * pairs)))
    print("   wave lifetime, cycles            %.0f  -> a SIMD is occupied %.0f cycles (%d waves, %d at a time); MFMA pipe busy %.3f of that" % (life, resid, waves // 1024, conc, busy / resid))
    print("   wave cycles: waiting %.3f, issue-stalled %.3f, issuing %.3f" % (g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"), g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES"), g("SQ_ACTIVE_INST_ANY") / g("SQ_WAVE_CYCLES")))
    print("   LDS bank-conflict cycles / LDS active cycles   %.3f" % (g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE")))
PY
cat $O/r06_pmc_wgrad.txt; tail -12 $O/run.log
