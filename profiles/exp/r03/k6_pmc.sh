#!/bin/bash
# PMC view of K6 (ray_triangulate_kernel) inside bench.py --mode gray.  Run on the GPU box from the repo root.
OUT=$PWD/gpurun_out/r03/k6_pmc; mkdir -p $OUT; REPO=$PWD; export TMPDIR=/tmp; cd /tmp
i=0
for PMC in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $PMC -f csv -d $OUT/p$i -o pmc -- python $REPO/bench.py --mode gray --steps 3 --warmup 1 --cpu-baseline 0 --host-io 0 --traffic off > $OUT/p${i}.log 2>&1
  python - $OUT/p$i <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "ray_" not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k in acc:
    print(k, {c: round(v / n[(k, c)]) for c, v in acc[k].items()}, "vgpr/lds:", )
PY
  rm -rf $OUT/p$i
done
