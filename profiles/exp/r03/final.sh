#!/bin/bash
mkdir -p gpurun_out/r03
( timeout 3000 python -m pytest tests -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -12 ) > gpurun_out/r03/final_pytest.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r03/final_smoke.txt 2>&1
bash profiles/run_profile.sh r03 > gpurun_out/run_profile_r03.log 2>&1
python bench.py > gpurun_out/profiles_r03/r03_bench_default_line.json 2> gpurun_out/profiles_r03/r03_bench_default.err
for M in ge gray hybrid; do python bench.py --mode $M --steps 10 --warmup 2 > gpurun_out/profiles_r03/r03_bench_${M}_default_line.json 2>/dev/null; done
echo done
