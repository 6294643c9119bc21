#!/bin/bash
# round 5, batch 7: K6 with the per-ray terms (c = v2.v2, e = v12.v2 in registers under s_set_gpr_idx; a, d per left ray) computed once per ray:
# 4 waves per SIMD (128 VGPRs, 76 bytes of scratch) and 3 waves per SIMD (146 VGPRs) against round 4's kernel (56 VGPRs, 4 waves)
mkdir -p gpurun_out/r05h
O=gpurun_out/r05h/k6_terms.txt; : > $O
P=structure-light-reconstructor_amd/libslr_hip.so
cp $P /tmp/keep.so
run() { # name extra
  line=$(python bench.py --mode gray --steps 10 --warmup 3 --cpu-baseline 0 --host-io 0 --traffic off $2 2>/tmp/err.txt | tail -1)
  echo "$1 $2 : $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("ms/frame %.4f  " % d["ms_per_frame"] + "  ".join("%s %.1f" % (x["name"].replace("slr_",""), x["avg_us"]) for x in d["kernels"]))' 2>&1 || tail -3 /tmp/err.txt)" | tee -a $O
}
for rep in 1 2; do
  for n in k6base k6w4 k6w3; do cp profiles/exp/ab/so/var_$n.so $P; run $n ""; run $n-serial "--batch-streams 1"; done
done
cp /tmp/keep.so $P
