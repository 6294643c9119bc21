#!/bin/bash
# round 5, batch 2: K4's L2 prefetch distance (SLR_K4_PF, workgroups of an XCD ahead; 0 = off), grouped batch (bench.py default) and
# frame-by-frame match launches (--match-group 1)
mkdir -p gpurun_out/r05c
O=gpurun_out/r05c/k4_prefetch.txt; : > $O
run() { # name extra
  line=$(python bench.py --steps 10 --warmup 3 --cpu-baseline 0 --host-io 0 --traffic off --map-sweep 0 $2 2>/tmp/err.txt | tail -1)
  echo "$1 $2 : $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("ms/frame %.4f  " % d["ms_per_frame"] + "  ".join("%s %.1f" % (x["name"].replace("slr_",""), x["avg_us"]) for x in d["kernels"]), "self_check", d["self_check"]["ok"])' 2>&1 || tail -3 /tmp/err.txt)" | tee -a $O
}
for rep in 1 2; do
  for pf in 0 16 32 48 64 96 128 192; do
    SLR_K4_PF=$pf run pf$pf ""
  done
  for pf in 0 32 64 128; do
    SLR_K4_PF=$pf run pf$pf-single "--match-group 1"
  done
done
