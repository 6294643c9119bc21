#!/bin/bash
mkdir -p gpurun_out/r03
( timeout 2400 python -m pytest tests/test_gpu_rectdma.py tests/test_gpu_fuzz.py tests/test_gpu_soak.py tests/test_gpu_hybrid.py tests/test_gpu_lean.py "tests/test_gpu_fullsize.py" -x -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -12 ) > gpurun_out/r03/b20_pytest.txt
python bench.py --steps 20 --warmup 3 --cpu-baseline 0 --host-io 0 > gpurun_out/r03/b20_bench.json 2> gpurun_out/r03/b20_bench.err
rm -f gpurun_out/var_ab.txt
bash profiles/exp/ab/var_run.sh 2 "--map-sweep 0" base
bash profiles/exp/ab/var_run.sh 1 "--mode ge" base
mv gpurun_out/var_ab.txt gpurun_out/r03/b20_ab.txt
echo done
