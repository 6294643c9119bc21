#!/bin/bash
mkdir -p gpurun_out/r03
( timeout 1500 python -m pytest "tests/test_gpu_fullsize.py::test_batch_entry_gray_only_two_frames_with_spare_planes" tests/test_gpu_multi.py tests/test_gpu_soak.py "tests/test_gpu_rectdma.py::test_dma_form_fullsize_every_shape" -q -m gpu 2>&1 | tail -25 ) > gpurun_out/r03/b2_pytest.txt
P=structure-light-reconstructor_amd/libslr_hip.so
cp $P /tmp/keep.so; cp profiles/exp/ab/so/var_clk.so $P
python profiles/exp/r03/clockprobe.py > gpurun_out/r03/b2_clock.txt 2>&1
cp /tmp/keep.so $P
echo done
