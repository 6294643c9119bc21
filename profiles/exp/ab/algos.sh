#!/bin/bash
# bench.py with each fused-decode form on one box: algos.sh "0 4 ..." [extra bench args]
A="$1"; shift
for r in 1 2; do
for a in $A; do
  python bench.py --steps 20 --rect-algo $a --cpu-baseline 0 "$@" 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d.get('kernels') or []
print('algo $a', d['value'], d['ms_per_step'], ' '.join('%s=%.1f' % (e['name'].replace('slr_',''), e['avg_us']) for e in k))"
done; done
