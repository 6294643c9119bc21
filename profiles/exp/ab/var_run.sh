#!/bin/bash
# times the bench with variant builds on ONE box, alternating: var_run.sh reps "<bench args>" name ...   ("base" = the tree's build)
P=structure-light-reconstructor_amd/libslr_hip.so
cp $P /tmp/keep.so
reps=$1; bargs=$2; shift 2
for r in $(seq $reps); do
for n in "$@"; do
  if [ "$n" = base ]; then cp /tmp/keep.so $P; else cp profiles/exp/ab/so/var_$n.so $P; fi
  line=$(python bench.py --steps 20 --warmup 3 --cpu-baseline 0 --host-io 0 --traffic off $bargs 2>/dev/null | tail -1)
  echo "$n : $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("ms/frame %.4f  " % d["ms_per_frame"] + "  ".join("%s %.1f" % (x["name"].replace("slr_",""), x["avg_us"]) for x in d["kernels"]))')" | tee -a gpurun_out/var_ab.txt
done
done
cp /tmp/keep.so $P
