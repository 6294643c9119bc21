#!/bin/bash
mkdir -p gpurun_out/r03
( timeout 1200 python -m pytest tests/test_gpu_rectdma.py tests/test_gpu_fuzz.py tests/test_gpu_soak.py "tests/test_gpu_fullsize.py::test_fullsize_shipped_pair_launch_both_cameras_vs_oracle" tests/test_gpu_multi.py -x -q -m gpu 2>&1 | tail -15 ) > gpurun_out/r03/b8_pytest.txt
rm -f gpurun_out/var_ab.txt
bash profiles/exp/ab/var_run.sh 3 "" base
mv gpurun_out/var_ab.txt gpurun_out/r03/b8_bench.txt
P=structure-light-reconstructor_amd/libslr_hip.so
cp $P /tmp/keep.so; cp profiles/exp/ab/so/var_clk.so $P
python profiles/exp/r03/clockprobe.py > gpurun_out/r03/b8_clock.txt 2>&1
cp /tmp/keep.so $P
echo done
