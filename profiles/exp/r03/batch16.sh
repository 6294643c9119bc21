#!/bin/bash
mkdir -p gpurun_out/r03
( timeout 2000 python -m pytest tests/test_gpu_lean.py tests/test_gpu_parity.py tests/test_gpu_soak.py tests/test_gpu_fuzz.py "tests/test_gpu_fullsize.py::test_fullsize_mf_match_forms_agree_and_oracle_rows" "tests/test_gpu_fullsize.py::test_fullsize_gray_decode_and_ge" -x -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -8 ) > gpurun_out/r03/b16_pytest.txt
rm -f gpurun_out/var_ab.txt
bash profiles/exp/ab/var_run.sh 2 "" base
bash profiles/exp/ab/var_run.sh 1 "--mode ge" base
mv gpurun_out/var_ab.txt gpurun_out/r03/b16_bench.txt
echo done
