#!/bin/bash
# builds variants of libslr_hip.so with extra -D flags on ONE kernel file into profiles/exp/ab/so/var_<name>.so (run here, no GPU)
# usage: var_build.sh <file.hip> name "-DFLAG ..." [name "-D..."]...
set -e
cd "$(dirname "$0")/../../.."
C=structure-light-reconstructor_amd/csrc
F=$1; shift
mkdir -p profiles/exp/ab/so
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Iinclude -DSLR_EXPERIMENTS -Wno-unused-function -Wno-inline-asm"
while [ $# -ge 2 ]; do
  n=$1; d=$2; shift 2
  /opt/rocm/bin/hipcc $FL $d -c $C/$F -o /tmp/var_$n.o 2>/dev/null
  objs=""
  for o in slr_capi kernels_decode kernels_rectdma kernels_match kernels_ray kernels_mfn kernels_compact; do
    if [ "$o.hip" = "$F" ]; then objs="$objs /tmp/var_$n.o"; else objs="$objs $C/$o.o"; fi
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o profiles/exp/ab/so/var_$n.so $objs
  echo built var_$n
done
