#!/bin/bash
# round 4, batch 5: the new GPU tests (x87 option, rectified config 5, multi proof, disassembly wrapper), then bench lines
mkdir -p gpurun_out
python -m pytest tests/test_gpu_x87.py tests/test_mfn_extension.py tests/test_gpu_disasm.py tests/test_gpu_multi.py -x -q -m gpu > gpurun_out/r04_b5_tests.txt 2>&1
tail -15 gpurun_out/r04_b5_tests.txt
SLR_BENCH_TRACE=1 python bench.py --mode mfn --steps 3 --warmup 1 > gpurun_out/r04_b5_mfn.json 2> gpurun_out/r04_b5_mfn.err
tail -3 gpurun_out/r04_b5_mfn.err; head -c 1500 gpurun_out/r04_b5_mfn.json; echo
SLR_BENCH_TRACE=1 python bench.py > gpurun_out/r04_b5_default.json 2> gpurun_out/r04_b5_default.err
tail -3 gpurun_out/r04_b5_default.err; python -c "
import json; d=json.load(open('gpurun_out/r04_b5_default.json')); print(d['value'], d['ms_per_frame'], d['roofline']['frac'], d['roofline']['avg_launch_us'], d['config']['maps'][:60], [(m['rig'][:24], m.get('decode_us_per_frame')) for m in d['realistic_maps']], d['host_buffers_pcie_inclusive'], d['cpu_baseline']['value'])"
