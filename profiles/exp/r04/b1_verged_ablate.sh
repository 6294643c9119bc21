#!/bin/bash
# round 4, batch 1: where the fused decode's time goes on the maps of the rig verged by 0.2 rad (the new bench default), against the
# near-identity maps: ablation builds (no source traffic / no LDS tap reads / no barriers / every wave in read mode 0) and a PMC pass
mkdir -p gpurun_out
O=gpurun_out/r04_b1.txt; : > $O
P=structure-light-reconstructor_amd/libslr_hip.so
cp $P /tmp/keep.so
run() { # name maps extra
  line=$(python bench.py --steps 10 --warmup 2 --cpu-baseline 0 --host-io 0 --traffic off --map-sweep 0 --maps $2 $3 2>/tmp/err.txt | tail -1)
  echo "$1 maps=$2 $3 : $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("ms/frame %.4f  " % d["ms_per_frame"] + "  ".join("%s %.1f" % (x["name"].replace("slr_",""), x["avg_us"]) for x in d["kernels"]), d["roofline"].get("waves_by_mode"), d["roofline"].get("nofit_tiles"))' 2>&1 || tail -3 /tmp/err.txt)" | tee -a $O
}
for rep in 1 2; do
  cp /tmp/keep.so $P
  run base near-identity
  run base verged
  run base verged:0.3:-0.2
  for n in abl1 abl2 abl3 mode0 mode10; do
    cp profiles/exp/ab/so/var_$n.so $P
    run $n verged
    [ $rep = 1 ] && run $n near-identity
  done
done
cp /tmp/keep.so $P
run base-shape0 verged "--dma-shape 0"
run base-shape1 verged "--dma-shape 1"
run base-depth1 verged "--dma-depth 1"
# the whole default line once (driver form), kept
python bench.py > gpurun_out/r04_b1_default_line.json 2> gpurun_out/r04_b1_default.err
# PMC on the verged rig
export TMPDIR=/tmp; R=$PWD; cd /tmp
for M in verged near-identity; do
for PMC in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY" "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  rm -rf /tmp/pp; timeout 300 rocprofv3 --kernel-trace --pmc $PMC -f csv -d /tmp/pp -o p -- python $R/bench.py --steps 2 --warmup 1 --cpu-baseline 0 --host-io 0 --traffic off --map-sweep 0 --maps $M > /tmp/pp.log 2>&1
  python - <<PY | tee -a $R/$O
import csv,glob,collections
f=glob.glob('/tmp/pp/**/*counter_collection.csv',recursive=True)
acc=collections.defaultdict(list)
if f:
    for r in csv.DictReader(open(f[0])):
        if 'mf_rect_decode_dma' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
print("PMC $M:", "  ".join("%s %.5g" % (k, sum(v)/len(v)) for k,v in sorted(acc.items())))
PY
done
done
