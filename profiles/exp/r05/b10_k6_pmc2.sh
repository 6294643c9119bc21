#!/bin/bash
# round 5, batch 10: K6 counters through profiles/prof_driver.py (SLR_WHAT=ray), round 4's kernel against the packed two-pair form
mkdir -p gpurun_out/r05j; O=$PWD/gpurun_out/r05j/k6_pmc2.txt; : > $O
P=$PWD/structure-light-reconstructor_amd/libslr_hip.so
cp $P /tmp/keep.so
export TMPDIR=/tmp SLR_WHAT=ray; R=$PWD; cd /tmp
for V in base packed; do
  if [ $V = base ]; then cp $R/profiles/exp/ab/so/var_k6base.so $P; else cp /tmp/keep.so $P; fi
  for PMC in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
    rm -rf /tmp/pm; timeout 200 rocprofv3 --kernel-trace --pmc $PMC -f csv -d /tmp/pm -o pmc -- python $R/profiles/prof_driver.py > /tmp/pm.log 2>&1
    python - $V <<'PY' | tee -a $O
import csv,glob,sys,collections
acc=collections.defaultdict(list)
for fn in glob.glob('/tmp/pm/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(fn)):
        if 'ray_triangulate_small' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
print(sys.argv[1], "  ".join("%s %.5g" % (k, sum(v)/len(v)) for k,v in sorted(acc.items())))
PY
  done
done
cp /tmp/keep.so $P
