#!/bin/bash
# time several builds of libslr_hip.so on ONE box: multi.sh a.so b.so ...   (microbench lines of the decode kernels)
P=structure-light-reconstructor_amd/libslr_hip.so
cp $P /tmp/keep.so
for r in 1 2; do
  for f in "$@"; do
    cp "$f" $P
    echo "== $(basename $f) run $r"
    python profiles/microbench.py 4096 3000 10 2>&1 | grep -E "mf_rectify_decode auto"
  done
done
cp /tmp/keep.so $P
