#!/bin/bash
# Collects the rocprofv3 evidence for one round.  Run ON THE GPU BOX through gpurun from the repo root:
#   gpurun --timeout 1200 -- 'bash profiles/run_profile.sh r01'
# Kernel timing (--kernel-trace --stats) and counters (--pmc, one pass each, never mixed with API tracing) are
# separate runs, as MI355X_MICROARCH.md prescribes (FETCH_SIZE and WRITE_SIZE do not fit one pass).
# Raw output goes to gpurun_out/prof_<tag>/ (scratch); the summaries are copied to gpurun_out/profiles_<tag>/ and
# from there committed under profiles/.
set -u
TAG=${1:-r01}
WHAT=${2:-mf}
OUT=$PWD/gpurun_out/prof_$TAG
SUM=$PWD/gpurun_out/profiles_$TAG
mkdir -p "$OUT" "$SUM"
export TMPDIR=/tmp SLR_WHAT=$WHAT
REPO=$PWD
cd /tmp

# 1. kernel trace + stats of the bench command itself (the numbers bench.py's roofline must agree with; --self-check 0: the self check and
#    the first-call probe launch the same kernels one frame at a time, which would be averaged into the 8-frame launches' mean)
rocprofv3 --kernel-trace --stats -f csv -d "$OUT/bench" -o bench -- python "$REPO/bench.py" --steps 10 --warmup 2 --cpu-baseline 0 --host-io 0 --traffic off --map-sweep 0 --self-check 0 \
    > "$OUT/bench_stdout.log" 2>&1
cp "$OUT"/bench/*kernel_stats.csv "$SUM/${TAG}_bench_kernel_stats.csv" 2>/dev/null
grep '^{"metric"' "$OUT/bench_stdout.log" | tail -1 > "$SUM/${TAG}_bench_line.json"

# 2. kernel trace + stats of the per-kernel driver
SLR_WHAT=mf,gray,ge,ray rocprofv3 --kernel-trace --stats -f csv -d "$OUT/drv" -o drv -- python "$REPO/profiles/prof_driver.py" > "$OUT/drv_stdout.log" 2>&1
cp "$OUT"/drv/*kernel_stats.csv "$SUM/${TAG}_driver_kernel_stats.csv" 2>/dev/null
for M in ge gray hybrid mfn; do
    rocprofv3 --kernel-trace --stats -f csv -d "$OUT/bench_$M" -o bench -- python "$REPO/bench.py" --mode $M --steps 5 --warmup 1 --cpu-baseline 0 --host-io 0 --traffic off \
        > "$OUT/bench_${M}_stdout.log" 2>&1
    cp "$OUT"/bench_$M/*kernel_stats.csv "$SUM/${TAG}_bench_${M}_kernel_stats.csv" 2>/dev/null
    grep '^{"metric"' "$OUT/bench_${M}_stdout.log" | tail -1 > "$SUM/${TAG}_bench_${M}_line.json"
done

# 3. PMC passes (each its own run), one driver group at a time so that a kernel that serves several plane counts (the Gray decode:
#    26 planes in GRAY_EPI, 44+ in GRAY_ONLY) is summarised per group: the summary keys on <group>:<kernel>
for GROUP in mf gray ge ray; do
i=0
for PMC in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    SLR_WHAT=$GROUP rocprofv3 --kernel-trace --pmc $PMC -f csv -d "$OUT/pmc_${GROUP}_$i" -o pmc -- python "$REPO/profiles/prof_driver.py" > "$OUT/pmc_${GROUP}_${i}_stdout.log" 2>&1
    cp "$OUT"/pmc_${GROUP}_$i/*counter_collection.csv "$OUT/pmc_${GROUP}_${i}_counters.csv" 2>/dev/null
    rm -rf "$OUT/pmc_${GROUP}_$i"
done
done
python "$REPO/profiles/summarize_pmc.py" "$OUT" "$SUM/${TAG}_pmc_summary.csv" "$SUM/pmc_traffic.json" > "$SUM/${TAG}_pmc_summary.txt" 2>&1
ls -la "$SUM"
# the raw rocprofv3 output is tens of MB per pass and gpurun_out/ is capped at 64 MiB: keep logs + summaries only
mkdir -p "$SUM/logs"; cp "$OUT"/*_stdout.log "$SUM/logs/" 2>/dev/null
rm -rf "$OUT"
