#!/bin/bash
# LDS-exchanged row stores in K4 / K5: the whole GPU suite, then K4 A/B (4 = per-thread stores, 0 = exchange) and the ge / mf bench lines
python -m pytest tests -x -q -m gpu 2>&1 | tail -3
python profiles/exp/r04/k4_ab.py --algos 4,0 --reps 5 2>&1 | tail -3
for m in mf ge; do python bench.py --mode $m --steps 10 --warmup 2 --cpu-baseline 0 --host-io 0 --traffic off --map-sweep 0 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["config"]["mode"], "ms/frame %.4f  " % d["ms_per_frame"] + "  ".join("%s %.1f" % (x["name"].replace("slr_",""), x["avg_us"]) for x in d["kernels"]))'; done
