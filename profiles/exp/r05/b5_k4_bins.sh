#!/bin/bash
# round 5, batch 5: the lean K4 with 1/16-wide bins and an unconditional first query step -- bench.py default line, twice
mkdir -p gpurun_out/r05f
O=gpurun_out/r05f/k4_bins.txt; : > $O
run() { # name extra
  line=$(python bench.py --steps 10 --warmup 3 --cpu-baseline 0 --host-io 0 --traffic off --map-sweep 0 $2 2>/tmp/err.txt | tail -1)
  echo "$1 $2 : $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("ms/frame %.4f  " % d["ms_per_frame"] + "  ".join("%s %.1f" % (x["name"].replace("slr_",""), x["avg_us"]) for x in d["kernels"]), "self_check", d["self_check"]["ok"])' 2>&1 || tail -3 /tmp/err.txt)" | tee -a $O
}
for rep in 1 2 3; do run new ""; run new-x87 "--eval-model x87"; run new-single "--match-group 1"; done
