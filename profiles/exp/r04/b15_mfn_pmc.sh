#!/bin/bash
# PMC of the config-5 LDS-DMA decode (shipped shape): VALU / LDS busy, bank conflicts, waits
export TMPDIR=/tmp; R=$PWD; cd /tmp
for PMC in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf /tmp/pm; MFN_FORMS=0 MFN_REPS=1 rocprofv3 --kernel-trace --pmc $PMC -f csv -d /tmp/pm -o pmc -- python $R/profiles/exp/r04/mfn_time.py > /tmp/pm.log 2>&1
  python - <<'P'
import csv,glob,collections
f=glob.glob('/tmp/pm/**/*counter_collection.csv',recursive=True)
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for fn in f:
    for r in csv.DictReader(open(fn)):
        k=r['Kernel_Name']
        if 'mfn_rect_dma' in k or 'mfn_decode_kernel' in k:
            acc[k.split('(')[0][-40:]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in acc.items():
    print(k, {c: round(sum(x)/len(x),1) for c,x in v.items()}, "n=%d"%len(next(iter(v.values()))))
P
done
