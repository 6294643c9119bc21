#!/bin/bash
# round 5, batch 6: GRAY_ONLY batch -- the two-stream pipeline (default) against one frame after the other
mkdir -p gpurun_out/r05g
O=gpurun_out/r05g/gray_streams.txt; : > $O
run() { # name extra
  line=$(python bench.py --mode gray --steps 10 --warmup 3 --cpu-baseline 0 --host-io 0 --traffic off $2 2>/tmp/err.txt | tail -1)
  echo "$1 $2 : $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("ms/frame %.4f  " % d["ms_per_frame"] + "  ".join("%s %.1f" % (x["name"].replace("slr_",""), x["avg_us"]) for x in d["kernels"]))' 2>&1 || tail -3 /tmp/err.txt)" | tee -a $O
}
for rep in 1 2; do run piped ""; run serial "--batch-streams 1"; done
