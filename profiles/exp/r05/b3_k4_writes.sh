#!/bin/bash
# round 5, batch 3: what this box streams (read-only / write-only / copy), and K4 without its output stores / with L2-hot phases / both
mkdir -p gpurun_out/r05d
profiles/exp/r05/bw > gpurun_out/r05d/bw.txt 2>&1; cat gpurun_out/r05d/bw.txt
O=gpurun_out/r05d/k4_writes.txt; : > $O
P=structure-light-reconstructor_amd/libslr_hip.so
cp $P /tmp/keep.so
run() { # name extra
  line=$(SLR_K4_PF=0 python bench.py --steps 10 --warmup 3 --cpu-baseline 0 --host-io 0 --traffic off --map-sweep 0 --self-check 0 $2 2>/tmp/err.txt | tail -1)
  echo "$1 $2 : $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("ms/frame %.4f  " % d["ms_per_frame"] + "  ".join("%s %.1f" % (x["name"].replace("slr_",""), x["avg_us"]) for x in d["kernels"]))' 2>&1 || tail -3 /tmp/err.txt)" | tee -a $O
}
for rep in 1 2; do
  cp /tmp/keep.so $P; run base ""
  for n in k4nost k4hot k4hotnost; do cp profiles/exp/ab/so/var_$n.so $P; run $n ""; done
done
cp /tmp/keep.so $P
