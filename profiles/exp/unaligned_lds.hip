// does gfx950 return correct data for a 2-byte-aligned ds_read_b32 (unaligned-ds-access)?  build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 profiles/exp/unaligned_lds.hip -o /tmp/ua && /tmp/ua
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u32a2 __attribute__((aligned(2)));
__global__ void k(const int *idx, unsigned *out)
{
    __shared__ unsigned short tile[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) tile[i] = (unsigned short)(i * 7);
    __syncthreads();
    const int j = idx[threadIdx.x];
    out[threadIdx.x] = *reinterpret_cast<const u32a2 *>(&tile[j]);
}
int main()
{
    int h[256]; unsigned o[256];
    for (int i = 0; i < 256; i++) h[i] = (i * 38 + (i % 3 != 0)) % 4000;
    int *d; unsigned *r;
    hipMalloc(&d, sizeof h); hipMalloc(&r, sizeof o);
    hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, d, r);
    hipMemcpy(o, r, sizeof o, hipMemcpyDeviceToHost);
    int bad = 0, odd = 0;
    for (int i = 0; i < 256; i++) {
        const unsigned e = (unsigned)(unsigned short)(h[i] * 7) | ((unsigned)(unsigned short)((h[i] + 1) * 7) << 16);
        bad += o[i] != e; odd += h[i] & 1;
    }
    printf("unaligned ds_read_b32: %d mismatches of 256 (%d odd indices)\n", bad, odd);
    return bad != 0;
}
