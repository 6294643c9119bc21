#!/bin/bash
# profile refresh behind a kernel change whose tests already ran (final.sh without the test suite)
mkdir -p gpurun_out/r03
bash profiles/run_profile.sh r03 > gpurun_out/run_profile_r03.log 2>&1
python bench.py > gpurun_out/profiles_r03/r03_bench_default_line.json 2> gpurun_out/profiles_r03/r03_bench_default.err
for M in ge gray hybrid; do python bench.py --mode $M --steps 10 --warmup 2 > gpurun_out/profiles_r03/r03_bench_${M}_default_line.json 2>/dev/null; done
echo done
