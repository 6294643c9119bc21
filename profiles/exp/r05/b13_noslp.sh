#!/bin/bash
# round 5, batch 13: the fused decode / K4 compiled without the SLP vectoriser (no v_pk_*_f32 the compiler packs on its own)
mkdir -p gpurun_out/r05n; O=gpurun_out/r05n/noslp.txt; : > $O
P=structure-light-reconstructor_amd/libslr_hip.so; cp $P /tmp/keep.so
run() { line=$(python bench.py --steps 10 --warmup 3 --cpu-baseline 0 --host-io 0 --traffic off --map-sweep 0 $2 2>/tmp/err.txt | tail -1)
  echo "$1 $2 : $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("ms/frame %.4f  " % d["ms_per_frame"] + "  ".join("%s %.1f" % (x["name"].replace("slr_",""), x["avg_us"]) for x in d["kernels"]), "self_check", d["self_check"]["ok"])' 2>&1 || tail -3 /tmp/err.txt)" | tee -a $O; }
for rep in 1 2; do cp /tmp/keep.so $P; run base ""; for n in rdnoslp k4noslp; do cp profiles/exp/ab/so/var_$n.so $P; run $n ""; done; done
cp /tmp/keep.so $P
