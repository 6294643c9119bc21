#!/bin/bash
# builds ablation variants of libslr_hip.so (kernels_rectdma.hip with -DSLR_DMA_ABL=n) into profiles/exp/ab/so/ (run here, no GPU)
set -e
cd "$(dirname "$0")/../../.."
C=structure-light-reconstructor_amd/csrc
mkdir -p profiles/exp/ab/so
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Iinclude -DSLR_EXPERIMENTS -Wno-unused-function -Wno-inline-asm"
for n in "$@"; do
  /opt/rocm/bin/hipcc $FL -DSLR_DMA_ABL=$n -c $C/kernels_rectdma.hip -o /tmp/rectdma_abl$n.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o profiles/exp/ab/so/abl$n.so $C/slr_capi.o $C/kernels_decode.o /tmp/rectdma_abl$n.o $C/kernels_match.o $C/kernels_ray.o $C/kernels_mfn.o
  echo built abl$n
done
cp structure-light-reconstructor_amd/libslr_hip.so profiles/exp/ab/so/abl0.so
