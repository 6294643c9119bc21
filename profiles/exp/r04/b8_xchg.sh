#!/bin/bash
# round 4, batch 8: results through LDS (park / unpark) in the fused MF decode: correctness first, then same-box A/B
mkdir -p gpurun_out
python -m pytest tests/test_gpu_rectdma.py tests/test_gpu_fuzz.py "tests/test_gpu_fullsize.py" -x -q -m gpu -k "not gray and not ge_ and not ray" > gpurun_out/r04_b8_tests.txt 2>&1
tail -5 gpurun_out/r04_b8_tests.txt
python profiles/exp/r04/ab_inproc.py --variants base,noxchg=profiles/exp/ab/so/var_noxchg.so --maps near-identity,verged,verged:0.3:-0.2,verged:0.1:-0.1 --reps 5 > gpurun_out/r04_b8_ab.txt 2>&1
grep -A20 medians gpurun_out/r04_b8_ab.txt
python profiles/exp/r04/ab_inproc.py --variants base,noxchg=profiles/exp/ab/so/var_noxchg.so --maps verged --k4 1 --reps 3 > gpurun_out/r04_b8_ab_k4.txt 2>&1
grep -A20 medians gpurun_out/r04_b8_ab_k4.txt
