#!/bin/bash
# round 5, batch 11: instruction-cache counters of the hot kernels (the shipped fused MF decode is 66 KB of code, K6 40 KB; two CUs share a 64 KB I-cache)
mkdir -p gpurun_out/r05k; O=$PWD/gpurun_out/r05k/icache.txt; : > $O
export TMPDIR=/tmp; R=$PWD; cd /tmp
rocprofv3 --list-avail 2>/dev/null | grep -i -E "icache|ifetch|SQC_INST|SQ_INST_LEVEL|SQC_TC_INST" | head -40 >> $O
for G in mf ray; do
for PMC in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU"; do
  rm -rf /tmp/pm; SLR_WHAT=$G timeout 200 rocprofv3 --kernel-trace --pmc $PMC -f csv -d /tmp/pm -o pmc -- python $R/profiles/prof_driver.py > /tmp/pm.log 2>&1 || tail -3 /tmp/pm.log >> $O
  python - $G <<'PY' | tee -a $O
import csv,glob,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob('/tmp/pm/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(fn)):
        k=r['Kernel_Name']
        if any(x in k for x in ('mf_rect_decode_dma','mf_match_lean','ray_triangulate_small','mf_decode_kernel')): acc[k[:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in acc.items(): print(sys.argv[1], k, "  ".join("%s %.5g" % (c, sum(x)/len(x)) for c,x in sorted(v.items())))
PY
done; done
