#!/bin/bash
# round 3, GPU batch 1: new parity tests + first perf probes of the fused decode (run from the repo root through gpurun)
mkdir -p gpurun_out/r03
( timeout 1500 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_multi.py tests/test_gpu_soak.py "tests/test_gpu_rectdma.py::test_dma_form_fullsize_every_shape" -x -q -m gpu 2>&1 | tail -25 ) > gpurun_out/r03/b1_pytest.txt
rm -f gpurun_out/var_ab.txt
bash profiles/exp/ab/var_run.sh 2 "" base d3 nowt
mv gpurun_out/var_ab.txt gpurun_out/r03/b1_var_depth3.txt
for pad in 0 64 128 192 256 320 512 1088; do
  bash profiles/exp/ab/var_run.sh 1 "--pitch-pad $pad" base
  sed -i "\$s/^/pad $pad /" gpurun_out/var_ab.txt
done
mv gpurun_out/var_ab.txt gpurun_out/r03/b1_pitch.txt
P=structure-light-reconstructor_amd/libslr_hip.so
cp $P /tmp/keep.so; cp profiles/exp/ab/so/var_clk.so $P
python profiles/exp/r03/clockprobe.py > gpurun_out/r03/b1_clock.txt 2>&1
cp /tmp/keep.so $P
rocm-smi --showclocks > gpurun_out/r03/b1_smi.txt 2>&1
echo done
