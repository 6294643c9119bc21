mkdir -p gpurun_out
export SLR_POISON_OUTPUTS=1
timeout 200 python profiles/exp/soak/lean_soak.py 31 120 > gpurun_out/soak_final.txt 2>&1; echo "soak rc=$?" >> gpurun_out/soak_final.txt
timeout 400 python profiles/exp/r03/quad_sort_fuzz.py > gpurun_out/fuzz_final.txt 2>&1; echo "fuzz rc=$?" >> gpurun_out/fuzz_final.txt
unset SLR_POISON_OUTPUTS
timeout 1500 bash profiles/run_profile.sh r03 > gpurun_out/run_profile_r03.log 2>&1
python bench.py > gpurun_out/bench_default_final.json 2> gpurun_out/bench_default_final.err
tail -3 gpurun_out/soak_final.txt gpurun_out/fuzz_final.txt
tail -c 600 gpurun_out/bench_default_final.json
