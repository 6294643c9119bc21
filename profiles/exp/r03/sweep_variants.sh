#!/bin/bash
# usage: sweep_variants.sh name ...   ("base" = the tree's build): default-map pair time + the verged-rig sweep per variant
P=structure-light-reconstructor_amd/libslr_hip.so
cp $P /tmp/keep.so
for n in "$@"; do
  if [ "$n" = base ]; then cp /tmp/keep.so $P; else cp profiles/exp/ab/so/var_$n.so $P; fi
  python bench.py --steps 20 --warmup 3 --cpu-baseline 0 --host-io 0 --traffic off 2>/dev/null | tail -1 | python -c '
import sys, json
d = json.loads(sys.stdin.read())
k = {x["name"]: x["avg_us"] for x in d["kernels"]}
print("'$n' : default %.1f us |" % k.get("slr_mf_rectify_decode_pair", 0), " | ".join("%s %.1f us modes %s" % (r["rig"][8:21], r.get("decode_us_per_frame", 0), r.get("waves_by_mode", [[0]])[0]) for r in d["realistic_maps"]))
'
done
cp /tmp/keep.so $P
