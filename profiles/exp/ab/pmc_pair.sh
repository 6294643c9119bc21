export TMPDIR=/tmp; R=$PWD; cd /tmp
for PMC in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY"; do
  rm -rf /tmp/pp; rocprofv3 --kernel-trace --pmc $PMC -f csv -d /tmp/pp -o p -- python $R/bench.py --steps 2 --warmup 1 --cpu-baseline 0 --host-io 0 --traffic off > /tmp/pp.log 2>&1
  python - <<PY
import csv,glob,collections
f=glob.glob('/tmp/pp/**/*counter_collection.csv',recursive=True)
acc=collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    if 'mf_rect_decode_dma' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
print("  ".join("%s %.4g" % (k, sum(v)/len(v)) for k,v in sorted(acc.items())))
PY
done
