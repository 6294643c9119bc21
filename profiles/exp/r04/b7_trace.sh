#!/bin/bash
mkdir -p gpurun_out
export SLR_AB_TRACE=1
S=profiles/exp/ab/so
timeout 900 python profiles/exp/r04/ab_inproc.py --variants base,abl1=$S/var_abl1.so,mode0=$S/var_mode0.so --maps near-identity,verged --flags 0,32 --reps 1 > gpurun_out/r04_b7.txt 2>&1
