#!/bin/bash
# the library built with FORMS=all (retired forms compiled in): the tests that ask forms() cover them
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu 2>&1 | tail -6 > gpurun_out/t_allforms.log
