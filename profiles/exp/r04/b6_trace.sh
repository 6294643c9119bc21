#!/bin/bash
mkdir -p gpurun_out
export SLR_AB_TRACE=1
timeout 600 python profiles/exp/r04/ab_inproc.py --variants base --maps near-identity,verged --reps 2 > gpurun_out/r04_b6.txt 2>&1
