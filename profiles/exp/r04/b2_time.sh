#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out/r04_b2.txt; : > $O
export SLR_BENCH_TRACE=1
( time python bench.py --steps 10 --warmup 2 --cpu-baseline 0 --host-io 0 --traffic off --map-sweep 0 --maps near-identity ) >> $O 2>&1
( time python bench.py --steps 10 --warmup 2 --cpu-baseline 0 --host-io 0 --traffic off --map-sweep 0 --maps verged ) >> $O 2>&1
