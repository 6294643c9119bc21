#!/bin/bash
# round 4, batch 3: (continuation of b1, whose box took 200 s per bench run) the rest of the ablations and the PMC passes
mkdir -p gpurun_out
O=gpurun_out/r04_b3.txt; : > $O
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
  run base-nosort near-identity "--debug-flags 32"
  run base-nosort verged "--debug-flags 32"
  for n in abl1 abl2 abl3 mode0 mode10; do
    cp profiles/exp/ab/so/var_$n.so $P
    run $n verged
    run $n near-identity
    [ $n = abl2 ] && run $n-nosort verged "--debug-flags 32"
    [ $n = abl2 ] && run $n-nosort near-identity "--debug-flags 32"
  done
done
cp /tmp/keep.so $P
run base-shape0 verged "--dma-shape 0"
run base-shape1 verged "--dma-shape 1"
run base-depth1 verged "--dma-depth 1"
# PMC on the verged rig and the near-identity maps
export TMPDIR=/tmp; R=$PWD; cd /tmp
for M in verged near-identity; do
for PMC in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  rm -rf /tmp/pp; timeout 200 rocprofv3 --kernel-trace --pmc $PMC -f csv -d /tmp/pp -o p -- python $R/bench.py --steps 2 --warmup 1 --cpu-baseline 0 --host-io 0 --traffic off --map-sweep 0 --maps $M > /tmp/pp.log 2>&1
  python - <<PY | tee -a $R/$O
import csv,glob,collections
f=glob.glob('/tmp/pp/**/*counter_collection.csv',recursive=True)
acc=collections.defaultdict(list)
if f:
    for r in csv.DictReader(open(f[0])):
        if 'mf_rect_decode_dma' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
print("PMC $M:", "  ".join("%s %.5g" % (k, sum(v)/len(v)) for k,v in sorted(acc.items())) or open('/tmp/pp.log').read()[-300:])
PY
done
done
