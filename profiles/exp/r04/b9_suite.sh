#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu > gpurun_out/r04_b9_suite.txt 2>&1
tail -8 gpurun_out/r04_b9_suite.txt
python profiles/exp/r04/series_pgm.py 6 > gpurun_out/r04_b9_series.txt 2>&1
tail -8 gpurun_out/r04_b9_series.txt
