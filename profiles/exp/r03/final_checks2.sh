# soak + small-rig fuzz under BOTH poison hooks, then a plain default bench (hooks off)
mkdir -p gpurun_out
export SLR_POISON_OUTPUTS=1 SLR_POISON_SCRATCH=1
timeout 200 python profiles/exp/soak/lean_soak.py 77 90 > gpurun_out/soak_final.txt 2>&1; echo "soak rc=$?" >> gpurun_out/soak_final.txt
timeout 300 python profiles/exp/r03/small_rig_fuzz.py 23 90 > gpurun_out/small_rig_fuzz.txt 2>&1; echo "rc=$?" >> gpurun_out/small_rig_fuzz.txt
unset SLR_POISON_OUTPUTS SLR_POISON_SCRATCH
python bench.py > gpurun_out/bench_default_final.json 2> gpurun_out/bench_default_final.err
tail -n 2 gpurun_out/soak_final.txt; tail -n 2 gpurun_out/small_rig_fuzz.txt; tail -c 300 gpurun_out/bench_default_final.json
