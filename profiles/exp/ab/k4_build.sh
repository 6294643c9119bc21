#!/bin/bash
# builds variants of libslr_hip.so with extra -D flags on kernels_match.hip + slr_capi.hip into profiles/exp/ab/so/ (run here, no GPU)
# usage: k4_build.sh name "-DFLAG ..." [name "-D..."]...
set -e
cd "$(dirname "$0")/../../.."
C=structure-light-reconstructor_amd/csrc
mkdir -p profiles/exp/ab/so
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Iinclude -DSLR_EXPERIMENTS -Wno-unused-function"
while [ $# -ge 2 ]; do
  n=$1; d=$2; shift 2
  /opt/rocm/bin/hipcc $FL $d -c $C/kernels_match.hip -o /tmp/k4_$n.o
  /opt/rocm/bin/hipcc $FL $d -c $C/slr_capi.hip -o /tmp/capi_$n.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o profiles/exp/ab/so/k4_$n.so /tmp/capi_$n.o $C/kernels_decode.o $C/kernels_rectdma.o /tmp/k4_$n.o $C/kernels_ray.o $C/kernels_mfn.o $C/kernels_compact.o
  echo built k4_$n
done
