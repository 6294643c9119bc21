#!/bin/bash
# round 5, batch 9: VALU instruction counts / busy cycles of K6, round 4's kernel against the packed two-pair form (one PMC pass each;
# a second pass with SQ_ACTIVE_INST_* / SQ_WAIT_* counters did not finish within 15 minutes on bench.py --mode gray and is not repeated)
mkdir -p gpurun_out/r05j; O=$PWD/gpurun_out/r05j/k6_pmc.txt
P=$PWD/structure-light-reconstructor_amd/libslr_hip.so
cp $P /tmp/keep.so
export TMPDIR=/tmp; R=$PWD; cd /tmp
for V in ${1:-base packed}; do
  if [ $V = base ]; then cp $R/profiles/exp/ab/so/var_k6base.so $P; else cp /tmp/keep.so $P; fi
  for PMC in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
    rm -rf /tmp/pm; timeout 240 rocprofv3 --kernel-trace --pmc $PMC -f csv -d /tmp/pm -o pmc -- python $R/bench.py --mode gray --steps 2 --warmup 1 --cpu-baseline 0 --host-io 0 --traffic off --batch-streams 1 > /tmp/pm.log 2>&1
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
