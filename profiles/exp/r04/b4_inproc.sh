#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out/r04_b4.txt
S=profiles/exp/ab/so
python profiles/exp/r04/ab_inproc.py --variants base,abl1=$S/var_abl1.so,abl2=$S/var_abl2.so,abl3=$S/var_abl3.so,mode0=$S/var_mode0.so,mode10=$S/var_mode10.so \
   --maps near-identity,verged,verged:0.3:-0.2 --flags 0,32 --reps 3 > $O 2>&1
python profiles/exp/r04/ab_inproc.py --variants base --maps near-identity,verged --shapes 3,0,1 --depths 2,1 --reps 2 >> $O 2>&1
python profiles/exp/r04/ab_inproc.py --variants base --maps near-identity,verged --k4 1 --reps 2 >> $O 2>&1
export TMPDIR=/tmp; R=$PWD; cd /tmp
for PMC in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum"; do
  rm -rf /tmp/pp; timeout 400 rocprofv3 --kernel-trace --pmc $PMC -f csv -d /tmp/pp -o p -- python $R/profiles/exp/r04/ab_inproc.py --variants base --maps near-identity,verged --reps 1 > /tmp/pp.log 2>&1
  python - <<PY >> $R/$O
import csv,glob,collections
f=glob.glob('/tmp/pp/**/*counter_collection.csv',recursive=True)
rows=collections.defaultdict(list)
if f:
    for r in csv.DictReader(open(f[0])):
        if 'mf_rect_decode_dma' in r['Kernel_Name']: rows[r['Counter_Name']].append((int(r['Dispatch_Id']), float(r['Counter_Value'])))
for k,v in sorted(rows.items()):
    v.sort(); n=len(v)//2
    a=[x for _,x in v[:n]]; b=[x for _,x in v[n:]]
    print("PMC %-28s near-identity %.6g   verged %.6g   (%d dispatches each)" % (k, sum(a)/max(1,len(a)), sum(b)/max(1,len(b)), n))
if not rows: print("PMC: nothing for $PMC", open('/tmp/pp.log').read()[-400:])
PY
done
