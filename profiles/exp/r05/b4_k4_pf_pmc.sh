#!/bin/bash
# round 5, batch 4: does K4's LDS-DMA prefetch reach the L2?  HBM reads (FETCH_SIZE x 2 on gfx950) and L2 hits / misses of the grouped K4 with
# SLR_K4_PF = 0 / 48
mkdir -p gpurun_out/r05e; O=$PWD/gpurun_out/r05e/k4_pf_pmc.txt; : > $O
export TMPDIR=/tmp; R=$PWD; cd /tmp
for PF in 0 48; do for PMC in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES"; do
  rm -rf /tmp/pm; SLR_K4_PF=$PF K4_GROUP=8 rocprofv3 --kernel-trace --pmc $PMC -f csv -d /tmp/pm -o pmc -- python $R/profiles/exp/r04/k4_group_pmc.py > /tmp/pm.log 2>&1
  python - $PF <<'P' | tee -a $O
import csv,glob,sys,collections
pf=sys.argv[1]; acc=collections.defaultdict(list)
for fn in glob.glob('/tmp/pm/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(fn)):
        if 'mf_match_lean_kernel' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
print("pf %s (per 8-frame launch):" % pf, "  ".join("%s %.6g" % (k, sum(v)/len(v)) for k,v in sorted(acc.items())))
P
done; done
