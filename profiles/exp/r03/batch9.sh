#!/bin/bash
mkdir -p gpurun_out/r03
( timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 ) > gpurun_out/r03/b9_pytest.txt
rm -f gpurun_out/var_ab.txt
bash profiles/exp/ab/var_run.sh 2 "--mode ge" base
bash profiles/exp/ab/var_run.sh 1 "--mode gray" base
bash profiles/exp/ab/var_run.sh 1 "" base
mv gpurun_out/var_ab.txt gpurun_out/r03/b9_bench.txt
echo done
