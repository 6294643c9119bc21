#!/bin/bash
# K4 with and without the hash dedup (SLR_OPT_MF_MATCH_ALGO 8 / 9; a `make FORMS=all` build), counters of the grouped launch the bench
# times (8 frames per launch, verged rig).  Run on the GPU box from the repo root: bash profiles/exp/r06/k4_pmc.sh
set -u
REPO=$PWD
OUT=$REPO/gpurun_out/r06_k4pmc
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
for A in 8 9; do
  i=0
  for PMC in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
             "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
             "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    D=/tmp/k4pmc_${A}_$i
    rm -rf "$D"
    rocprofv3 --kernel-trace --pmc $PMC -f csv -d "$D" -o pmc -- python "$REPO/bench.py" --pmc-child 1 --match-algo $A > "$OUT/algo${A}_pass$i.log" 2>&1; ls -R "$D" | head -20 >> "$OUT/algo${A}_pass$i.log"
    find "$D" -name "*counter_collection.csv" -exec cat {} + 2>/dev/null | grep -E "Counter_Name|mf_match_lean_kernel" > "$OUT/algo${A}_pass$i.csv"
    rm -rf "$D"
  done
done
python - <<PY
import csv, glob, collections
for A in (8, 9):
    acc = collections.defaultdict(list)
    for f in sorted(glob.glob("$OUT/algo%d_pass*.csv" % A)):
        for row in csv.DictReader(open(f)):
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    print("SLR_OPT_MF_MATCH_ALGO", A, "(8 = hash dedup, the shipped kernel; 9 = no hash)  -- mean per 8-frame launch")
    for k in sorted(acc):
        print("    %-24s %.6g  (%d dispatches)" % (k, sum(acc[k]) / len(acc[k]), len(acc[k])))
    px = 8 * 4096 * 3000.0
    g = lambda k: sum(acc[k]) / len(acc[k]) if acc[k] else float("nan")
    print("    VALU lane-instructions per pixel %.1f, SALU wave-instructions x 64 per pixel %.1f, LDS instructions x 64 per pixel %.1f" % (
        g("SQ_INSTS_VALU") * 64 / px, g("SQ_INSTS_SALU") * 64 / px, g("SQ_INSTS_LDS") * 64 / px))
    print("    LDS bank conflict cycles / LDS active cycles = %.3f" % (g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE")))
PY
