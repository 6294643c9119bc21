#!/bin/bash
# same-box A/B of the quad sort: SLR_OPT_DEBUG_FLAGS 32 = every wave on the quads of its own block (the digest of before)
mkdir -p gpurun_out
O=gpurun_out/quad_sort_same_box.txt; : > $O
for rep in 1 2; do for fl in 32 0; do for M in mf ge hybrid; do
  python bench.py --mode $M --steps 10 --warmup 2 --cpu-baseline 0 --host-io 0 --traffic off --map-sweep $([ $M = mf ] && [ $rep = 1 ] && echo 1 || echo 0) --debug-flags $fl 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('flags', d['config']['debug_flags'], d['config']['mode'], 'value', d['value'], r['kernel'], r['avg_launch_us'], r['frac'], [(k['name'],k['avg_us']) for k in d['kernels']], [(m['rig'][8:17], m['decode_us_per_frame']) for m in d.get('realistic_maps') or []])" >> $O
done; done; done
cat $O
