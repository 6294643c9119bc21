#!/bin/bash
# round 5, batch 1: (a) the grouped K4 per row (default again) against the persistent form (SLR_OPT_MF_MATCH_ALGO 7);
# (b) upper bounds of VERDICT r4's items 3 and 8 from ablation builds (wrong results, timing only):
#     abl7 = the decode's phases stored to a 1 MB L2-resident window (a fused decode -> match launch with a per-XCD phase ring, decode side),
#     k4hot = K4 reading the phases of 64 L2-resident rows (the same fusion, match side),
#     abl8 = the map digest read from 64 L2-resident tiles (a map stream of zero bytes), abl9 = abl7 + abl8.
mkdir -p gpurun_out/r05b
O=gpurun_out/r05b/fusion_bounds.txt; : > $O
P=structure-light-reconstructor_amd/libslr_hip.so
cp $P /tmp/keep.so
run() { # name extra
  line=$(python bench.py --steps 10 --warmup 3 --cpu-baseline 0 --host-io 0 --traffic off --map-sweep 0 --self-check 0 $2 2>/tmp/err.txt | tail -1)
  echo "$1 $2 : $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("ms/frame %.4f  " % d["ms_per_frame"] + "  ".join("%s %.1f" % (x["name"].replace("slr_",""), x["avg_us"]) for x in d["kernels"]))' 2>&1 || tail -3 /tmp/err.txt)" | tee -a $O
}
for rep in 1 2; do
  cp /tmp/keep.so $P
  run base ""
  run persist "--match-algo 7"
  run base-near "--maps near-identity"
  for n in abl7 abl8 abl9 k4hot; do
    cp profiles/exp/ab/so/var_$n.so $P
    run $n ""
    [ $n = abl9 ] && run $n-near "--maps near-identity"
  done
done
cp /tmp/keep.so $P
