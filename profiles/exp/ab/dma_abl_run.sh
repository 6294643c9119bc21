#!/bin/bash
# times the ablation builds on ONE box: dma_abl_run.sh "<shape> <depth>" n n n ...   (repo root, through gpurun)
P=structure-light-reconstructor_amd/libslr_hip.so
cp $P /tmp/keep.so
cfg=($1); shift
for n in "$@"; do
  cp profiles/exp/ab/so/abl$n.so $P
  line=$(python bench.py --steps 20 --warmup 3 --cpu-baseline 0 --host-io 0 --traffic off --rect-algo 7 --dma-shape ${cfg[0]} --dma-depth ${cfg[1]} 2>/dev/null | tail -1)
  echo "abl $n shape ${cfg[0]} depth ${cfg[1]} : $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); k={x["name"]:x["avg_us"] for x in d["kernels"]}; print("pair %.1f us" % k.get("slr_mf_rectify_decode_pair",0))')" | tee -a gpurun_out/dma_abl.txt
done
cp /tmp/keep.so $P
