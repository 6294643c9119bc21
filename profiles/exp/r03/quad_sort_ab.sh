mkdir -p gpurun_out
python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "verged" -s 2>&1 | tail -30 > gpurun_out/t_sort.log
python -m pytest tests/test_gpu_rectdma.py tests/test_gpu_hybrid.py -x -q -m gpu 2>&1 | tail -5 >> gpurun_out/t_sort.log
python bench.py --cpu-baseline 0 --host-io 0 --traffic off > gpurun_out/b_sort_mf.json 2> gpurun_out/b_sort_mf.err
python bench.py --mode ge --cpu-baseline 0 --host-io 0 --traffic off > gpurun_out/b_sort_ge.json 2>> gpurun_out/b_sort_mf.err
