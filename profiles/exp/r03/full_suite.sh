mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu 2>&1 | tail -6 > gpurun_out/t_full.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" >> gpurun_out/t_full.log 2>&1
