#!/bin/bash
# round 5, batch 12: row pitch of the HBM-resident stacks padded away from a power of two (the caller's choice of layout; default pitch = width)
mkdir -p gpurun_out/r05m; O=gpurun_out/r05m/pitch.txt; : > $O
run() { line=$(python bench.py --steps 10 --warmup 3 --cpu-baseline 0 --host-io 0 --traffic off --map-sweep 0 $2 2>/tmp/err.txt | tail -1)
  echo "$1 $2 : $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("ms/frame %.4f  " % d["ms_per_frame"] + "  ".join("%s %.1f" % (x["name"].replace("slr_",""), x["avg_us"]) for x in d["kernels"]), "self_check", d["self_check"]["ok"])' 2>&1 || tail -3 /tmp/err.txt)" | tee -a $O; }
for rep in 1 2; do for pad in 0 64 128 256 576 1152; do run pad$pad "--pitch-pad $pad"; done; done
