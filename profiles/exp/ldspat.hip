// ldspat.hip -- LDS read throughput for the access pattern of a bilinear tap out of a natural [row][x] byte image:
// lane l of a wave reads the dword pair that holds source bytes (4*(l/4) + s .. +1), i.e. 4 neighbouring lanes share a
// pair and consecutive lane quads read OVERLAPPING 8-byte windows at 4-byte steps (mixed 8-byte alignment in one
// instruction).  Which instruction serves that pattern at full rate?
// build: hipcc --offload-arch=gfx950 -O3 -o ldspat ldspat.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// MODE 0: ds_read_b64   1: 2 x ds_read_b32   2: ds_read2_b32 offset1:1   3: ds_read_b64 8 apart (reference)
template <int MODE>
__global__ __launch_bounds__(1024) void pat(unsigned *sink, int iters, int step, int rowskew)
{
    __shared__ __attribute__((aligned(16))) unsigned buf[16384];
    for (int i = threadIdx.x; i < 16384; i += 1024) buf[i] = i * 2654435761u;
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // step = bytes between lane quads (4: the tap pattern; 8: disjoint aligned pairs); rowskew: lanes >= 32 read another row
    unsigned a = (unsigned)(wv * 320 + (lane >> 2) * step + ((lane >> 5) * rowskew));
    if (MODE == 3) a = threadIdx.x * 8;
    unsigned acc = 0;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if constexpr (MODE == 0 || MODE == 3) {
                unsigned long long v;
                asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(a), "i"(u * 5120));
                asm volatile("" : "+v"(v));
                acc += (unsigned)v + (unsigned)(v >> 32);
            } else if constexpr (MODE == 1) {
                unsigned v0, v1;
                asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v0) : "v"(a), "i"(u * 5120));
                asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v1) : "v"(a), "i"(u * 5120 + 4));
                acc += v0 + v1;
            } else {
                unsigned long long v;
                asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(v) : "v"(a), "i"(u * 16), "i"(u * 16 + 1));
                acc += (unsigned)v + (unsigned)(v >> 32);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(acc));
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

int main()
{
    unsigned *sink; CK(hipMalloc(&sink, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 2000;
    struct V { int mode, step, skew; const char *what; };
    const V vs[] = {{3, 8, 0, "ds_read_b64, every lane its own aligned 8 bytes"},
                    {0, 8, 0, "ds_read_b64, lane quads share, 8-byte steps (aligned)"},
                    {0, 4, 0, "ds_read_b64, lane quads share, 4-byte steps (TAP PATTERN)"},
                    {0, 4, 288, "ds_read_b64, tap pattern, upper half-wave on the next row"},
                    {1, 4, 0, "2 x ds_read_b32, tap pattern"},
                    {2, 4, 0, "ds_read2_b32 offset1:1, tap pattern"},
                    {0, 12, 0, "ds_read_b64, 12-byte steps"},
                    {0, 0, 0, "ds_read_b64, whole wave one address"}};
    for (const V &v : vs) {
        float best = 1e9f;
        for (int r = 0; r < 4; r++) {
            CK(hipEventRecord(e0));
            switch (v.mode) {
            case 0: hipLaunchKernelGGL(pat<0>, dim3(256), dim3(1024), 0, 0, sink, iters, v.step, v.skew); break;
            case 1: hipLaunchKernelGGL(pat<1>, dim3(256), dim3(1024), 0, 0, sink, iters, v.step, v.skew); break;
            case 2: hipLaunchKernelGGL(pat<2>, dim3(256), dim3(1024), 0, 0, sink, iters, v.step, v.skew); break;
            default: hipLaunchKernelGGL(pat<3>, dim3(256), dim3(1024), 0, 0, sink, iters, v.step, v.skew); break;
            }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        // 16 waves x iters x 8 pair reads per CU
        const double clk = best * 1e-3 * 2.4e9 / (16.0 * iters * 8);
        printf("%-62s %8.1f us  -> %.2f clk per wave pair-read per CU (2.4 GHz)\n", v.what, best * 1e3, clk);
    }
    return 0;
}
