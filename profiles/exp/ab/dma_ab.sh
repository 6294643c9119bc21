#!/bin/bash
# A/B of the LDS-DMA fused decode's tile shapes and depths against round 1's form (5) on ONE box, alternating.
# usage (repo root, through gpurun): bash profiles/exp/ab/dma_ab.sh [reps]
reps=${1:-2}
out=gpurun_out/dma_ab.txt
: > $out
for r in $(seq $reps); do
  for cfg in ${CFGS:-5_0_1 7_0_1 7_1_1 7_1_2 7_3_1 7_4_1 7_6_1}; do
    set -- ${cfg//_/ }
    line=$(python bench.py --steps 20 --warmup 3 --cpu-baseline 0 --host-io 0 --traffic off --rect-algo $1 --dma-shape $2 --dma-depth $3 2>/dev/null | tail -1)
    echo "algo $1 shape $2 depth $3 : $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); k={x["name"]:x["avg_us"] for x in d["kernels"]}; print("ms/step %.4f  pair %.1f us  match %.1f us" % (d["ms_per_step"], k.get("slr_mf_rectify_decode_pair",0), k.get("slr_mf_match_triangulate",0)))')" | tee -a $out
  done
done
