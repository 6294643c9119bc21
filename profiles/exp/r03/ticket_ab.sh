#!/bin/bash
# the fixed-SGPR ticket against the previous library on one box (the previous .so is copied in for the run, not committed)
mkdir -p gpurun_out
L=structure-light-reconstructor_amd/libslr_hip.so
cp $L /tmp/new.so
timeout 600 python profiles/exp/r03/quad_sort_dbg.py > gpurun_out/ticket_dbg.txt 2>&1
O=gpurun_out/ticket_ab.txt; : > $O
for rep in 1 2; do for which in prev new; do
  if [ $which = prev ]; then cp profiles/exp/r03/_prev.so $L; else cp /tmp/new.so $L; fi
  for M in mf ge; do
  python bench.py --mode $M --steps 20 --warmup 3 --cpu-baseline 0 --host-io 0 --traffic off --map-sweep 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$which', d['config']['mode'], 'value', d['value'], r['kernel'], r['avg_launch_us'], r['frac'], [(k['name'],k['avg_us']) for k in d['kernels']])" >> $O
  done
done; done
cp /tmp/new.so $L
cat $O
