#!/bin/bash
mkdir -p gpurun_out/r03
( timeout 1500 python -m pytest tests/test_gpu_hybrid.py -x -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -25 ) > gpurun_out/r03/b12_pytest.txt
rm -f gpurun_out/var_ab.txt
bash profiles/exp/ab/var_run.sh 2 "--mode hybrid" base
mv gpurun_out/var_ab.txt gpurun_out/r03/b12_bench.txt
echo done
