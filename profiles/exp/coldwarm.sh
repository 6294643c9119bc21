#!/bin/bash
# PMC comparison warm vs cold for the fused kernel; prints mean counter values for dispatches 2..7 (warm) and 10..15 (cold)
REPO=$PWD; export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/cw; mkdir -p /tmp/cw
i=0
for PMC in "$@"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $PMC -f csv -d /tmp/cw/p$i -o p$i -- python $REPO/profiles/exp/coldwarm_driver.py > /tmp/cw/log$i 2>&1
  python - <<PY
import csv,glob,collections
f=glob.glob('/tmp/cw/p$i/**/*counter_collection.csv',recursive=True)
rows=[r for r in csv.DictReader(open(f[0])) if 'mf_rect_decode_' in r['Kernel_Name']]
by=collections.defaultdict(dict)
for r in rows: by[int(r['Dispatch_Id'])][r['Counter_Name']]=float(r['Counter_Value'])
ids=sorted(by)
warm=ids[2:8]; cold=ids[10:16]
for c in sorted(by[ids[0]]):
    w=sum(by[d][c] for d in warm)/len(warm); k=sum(by[d][c] for d in cold)/len(cold)
    print("%-32s warm %16.0f   cold %16.0f   ratio %.3f"%(c,w,k,k/w if w else 0))
PY
done
