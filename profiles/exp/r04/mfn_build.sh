#!/bin/bash
# variants of libslr_hip.so with extra -D flags on kernels_mfn.hip into profiles/exp/ab/so/mfn_<name>.so (run here, no GPU)
# usage: mfn_build.sh name "-DFLAG ..." [name "-D..."]...
set -e
cd "$(dirname "$0")/../../.."
C=structure-light-reconstructor_amd/csrc
mkdir -p profiles/exp/ab/so
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Iinclude -DSLR_EXPERIMENTS -Wno-unused-function"
while [ $# -ge 2 ]; do
  n=$1; d=$2; shift 2
  /opt/rocm/bin/hipcc $FL $d -c $C/kernels_mfn.hip -o /tmp/mfn_$n.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o profiles/exp/ab/so/mfn_$n.so $C/slr_capi.o $C/kernels_decode.o $C/kernels_rectdma.o $C/kernels_match.o $C/kernels_ray.o /tmp/mfn_$n.o $C/kernels_compact.o
  echo built mfn_$n
done
