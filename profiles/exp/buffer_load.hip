// raw buffer loads on gfx950: one descriptor for a stack of planes, the plane as a scalar offset, out-of-range -> 0
//   hipcc --offload-arch=gfx950 -O3 profiles/exp/buffer_load.hip -o /tmp/bl && /tmp/bl
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
__global__ void k(const unsigned char *base, unsigned nbytes, unsigned stride, const unsigned *voff, unsigned *out)
{
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, (int)nbytes, 0x00020000);
    const unsigned v = voff[threadIdx.x];
    for (unsigned p = 0; p < 3; p++)
        out[p * 64 + threadIdx.x] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rs, (int)v, (int)(p * stride), 0);
}
int main()
{
    const unsigned stride = 1024, n = 3 * stride;
    unsigned char *h = (unsigned char *)malloc(n), *d;
    for (unsigned i = 0; i < n; i++) h[i] = (unsigned char)(i * 7 + (i >> 8));
    unsigned hv[64], ho[192], *dv, *dout;
    for (int i = 0; i < 64; i++) hv[i] = (i % 5 == 0) ? 0xFFFFFFF0u : (unsigned)(i * 12);   // every 5th lane: out of range
    hipMalloc(&d, n); hipMalloc(&dv, sizeof hv); hipMalloc(&dout, sizeof ho);
    hipMemcpy(d, h, n, hipMemcpyHostToDevice); hipMemcpy(dv, hv, sizeof hv, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, n, stride, dv, dout);
    hipMemcpy(ho, dout, sizeof ho, hipMemcpyDeviceToHost);
    int bad = 0;
    for (unsigned p = 0; p < 3; p++)
        for (int i = 0; i < 64; i++) {
            unsigned e = 0;
            if (hv[i] != 0xFFFFFFF0u) memcpy(&e, h + p * stride + hv[i], 4);
            bad += ho[p * 64 + i] != e;
        }
    printf("raw buffer loads: %d mismatches of 192 (out-of-range lanes must read 0)\n", bad);
    return bad != 0;
}
