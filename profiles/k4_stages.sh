#!/bin/bash
# NOTE (round 2): the stop hook is compiled only into -DSLR_DEBUG_HOOKS builds: make -C structure-light-reconstructor_amd/csrc clean all CXXFLAGS+=-DSLR_DEBUG_HOOKS
# K4 phase ablation: SLR_DEBUG_K4_STOP=N makes the indexed match kernels return after phase N
# binned form: 1 loads + histogram clear, 4 histogram + scan + scatter, 5 queries, 0 everything
# sorted form: 1 loads+keys, 2 sort, 3 run heads + compaction, 4 bin index, 5 queries, 0 everything
for a in 1 2 3 4 5 0; do echo "stop=$a"; SLR_DEBUG_K4_STOP=$a timeout 120 python profiles/microbench.py 4096 3000 5 2>&1 | grep "mf_match" | awk '{print "   ", $1, $2, $4, $5}'; done
