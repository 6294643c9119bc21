#!/bin/bash
# PMC of K4 inside the grouped batch (8 frames per launch)
export TMPDIR=/tmp; R=$PWD; cd /tmp
for PMC in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD"; do
  rm -rf /tmp/pm; K4_GROUP=8 rocprofv3 --kernel-trace --pmc $PMC -f csv -d /tmp/pm -o pmc -- python $R/profiles/exp/r04/k4_group_pmc.py > /tmp/pm.log 2>&1
  python - <<'P'
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob('/tmp/pm/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(fn)):
        k=r['Kernel_Name']
        if 'mf_match_lean' in k or 'mf_rect_decode_dma' in k:
            acc[k.split('(')[0][-44:]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in acc.items():
    print(k, {c: round(sum(x)/len(x)/8,1) for c,x in v.items()}, "(per frame) n=%d"%len(next(iter(v.values()))))
P
done
