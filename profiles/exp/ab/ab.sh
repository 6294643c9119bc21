#!/bin/bash
# A/B two builds of libslr_hip.so on ONE box (box-to-box variation is ~5 %): ab.sh <old.so> <new.so> [microbench args]
P=structure-light-reconstructor_amd/libslr_hip.so
cp $P /tmp/keep.so
for r in 1 2 3; do
  for v in old new; do
    if [ $v = old ]; then cp "$1" $P; else cp "$2" $P; fi
    echo "== $v run $r"
    python profiles/microbench.py 4096 3000 10 2>&1 | grep -E "vec=4|mf_rectify_decode auto|binned" 
    python bench.py --steps 20 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['north_star_kernel']['avg_launch_us'])"
  done
done
cp /tmp/keep.so $P
