#!/bin/bash
# times K4 of the variant builds on ONE box, alternating: k4_run.sh reps name[:stop] ...   (repo root, through gpurun)
P=structure-light-reconstructor_amd/libslr_hip.so
cp $P /tmp/keep.so
reps=$1; shift
for r in $(seq $reps); do
for v in "$@"; do
  n=${v%%:*}; stop=""; [[ "$v" == *:* ]] && stop=${v##*:}
  if [ "$n" = base ]; then cp /tmp/keep.so $P; else cp profiles/exp/ab/so/k4_$n.so $P; fi
  echo "$v : $(K4_STOP=$stop python profiles/exp/ab/k4_time.py 2>&1 | tail -1)" | tee -a gpurun_out/k4_ab.txt
done
done
cp /tmp/keep.so $P
