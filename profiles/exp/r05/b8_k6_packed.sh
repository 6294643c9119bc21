#!/bin/bash
# round 5, batch 8: K6 with two pairs per step on packed f32 (v_pk_mul / add / fma) against round 4's kernel
mkdir -p gpurun_out/r05i
O=gpurun_out/r05i/k6_packed.txt; : > $O
P=structure-light-reconstructor_amd/libslr_hip.so
cp $P /tmp/keep.so
run() { # name extra
  line=$(python bench.py --mode gray --steps 10 --warmup 3 --cpu-baseline 0 --host-io 0 --traffic off $2 2>/tmp/err.txt | tail -1)
  echo "$1 $2 : $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("ms/frame %.4f  " % d["ms_per_frame"] + "  ".join("%s %.1f" % (x["name"].replace("slr_",""), x["avg_us"]) for x in d["kernels"]))' 2>&1 || tail -3 /tmp/err.txt)" | tee -a $O
}
for rep in 1 2; do
  cp profiles/exp/ab/so/var_k6base.so $P; run k6base ""; run k6base-serial "--batch-streams 1"
  cp /tmp/keep.so $P; run packed ""; run packed-serial "--batch-streams 1"
done
cp /tmp/keep.so $P
