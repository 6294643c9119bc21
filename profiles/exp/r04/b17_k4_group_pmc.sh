#!/bin/bash
# HBM traffic of K4 per frame, frame by frame (group 1) and in groups of 8 frames per launch: FETCH_SIZE x 2 (gfx950) + WRITE_SIZE, KiB
export TMPDIR=/tmp; R=$PWD; cd /tmp
for G in 1 8; do for PMC in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pm; K4_GROUP=$G rocprofv3 --kernel-trace --pmc $PMC -f csv -d /tmp/pm -o pmc -- python $R/profiles/exp/r04/k4_group_pmc.py > /tmp/pm.log 2>&1
  python - $G $PMC <<'P'
import csv,glob,sys
G=int(sys.argv[1]); pmc=sys.argv[2]; v=[]
for fn in glob.glob('/tmp/pm/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(fn)):
        if 'mf_match_lean_kernel' in r['Kernel_Name'] and r['Counter_Name']==pmc: v.append(float(r['Counter_Value']))
m=sum(v)/len(v)*1024*(2 if pmc=='FETCH_SIZE' else 1)/G
print("group %d %s: %d dispatches, %.1f MB per frame" % (G, pmc, len(v), m/1e6))
P
done; done
