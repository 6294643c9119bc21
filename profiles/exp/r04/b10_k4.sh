#!/bin/bash
# K4 shapes: parity first (lean tests + the whole-frame forms test), then the same-process A/B
python -m pytest tests/test_gpu_lean.py tests/test_gpu_fullsize.py -x -q -m gpu -k "k4 or mf_match" 2>&1 | tail -5
python profiles/exp/r04/k4_ab.py --algos 4,7 --reps 5 2>&1 | tail -30
