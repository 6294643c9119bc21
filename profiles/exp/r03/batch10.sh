#!/bin/bash
mkdir -p gpurun_out/r03
( timeout 1200 python -m pytest "tests/test_gpu_fullsize.py::test_fullsize_verged_rig_maps_from_stereo_rectify" -x -q -m gpu -s 2>&1 | grep -v amdgpu.ids | tail -15 ) > gpurun_out/r03/b10_pytest.txt
python bench.py --steps 20 --warmup 3 --cpu-baseline 0 --host-io 0 > gpurun_out/r03/b10_bench.json 2> gpurun_out/r03/b10_bench.err
echo done
