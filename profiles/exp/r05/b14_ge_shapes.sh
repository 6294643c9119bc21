#!/bin/bash
# round 5, batch 14: the fused Gray decode's tile shapes and DMA depths on today's kernel (bench.py --mode ge)
mkdir -p gpurun_out/r05p; O=gpurun_out/r05p/ge_shapes.txt; : > $O
run() { line=$(python bench.py --mode ge --steps 10 --warmup 3 --cpu-baseline 0 --host-io 0 --traffic off $2 2>/tmp/err.txt | tail -1)
  echo "$1 $2 : $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("ms/frame %.4f  " % d["ms_per_frame"] + "  ".join("%s %.1f" % (x["name"].replace("slr_",""), x["avg_us"]) for x in d["kernels"]))' 2>&1 || tail -3 /tmp/err.txt)" | tee -a $O; }
for rep in 1 2; do run auto ""; run shape0 "--dma-shape 0"; run shape1 "--dma-shape 1"; run shape3 "--dma-shape 3"; run shape3-depth1 "--dma-shape 3 --dma-depth 1"; run shape1-depth1 "--dma-shape 1 --dma-depth 1"; done
