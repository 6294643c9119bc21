mkdir -p gpurun_out
timeout 900 python profiles/exp/r03/quad_sort_dbg.py > gpurun_out/quad_sort_dbg.txt 2>&1
tail -3 gpurun_out/quad_sort_dbg.txt
