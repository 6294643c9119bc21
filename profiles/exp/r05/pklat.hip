// Dependent-issue cost of packed f32 on gfx950: C independent fma chains per lane (C = 1, 2, 4, 8), scalar v_fma_f32 against v_pk_fma_f32,
// 4 waves per SIMD (1024 workgroups of 256 threads on 256 CUs), 8192 chain steps.  Prints ns per chain step and the f32-op rate.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -o profiles/exp/r05/pklat profiles/exp/r05/pklat.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int C> __global__ __launch_bounds__(256) void scalar_k(float *out, float x, float y, int n)
{
    float a[C];
#pragma unroll
    for (int i = 0; i < C; i++) a[i] = (float)(threadIdx.x + i);
    for (int it = 0; it < n; it++) {
#pragma unroll
        for (int i = 0; i < C; i++) a[i] = __builtin_fmaf(a[i], x, y);
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < C; i++) s += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int C> __global__ __launch_bounds__(256) void packed_k(float *out, float x, float y, int n)
{
    f2 a[C];
#pragma unroll
    for (int i = 0; i < C; i++) a[i] = f2{(float)(threadIdx.x + i), (float)(threadIdx.x + 8 + i)};
    const f2 xx = {x, x}, yy = {y, y};
    for (int it = 0; it < n; it++) {
#pragma unroll
        for (int i = 0; i < C; i++) a[i] = __builtin_elementwise_fma(a[i], xx, yy);
    }
    f2 s = {0, 0};
#pragma unroll
    for (int i = 0; i < C; i++) s += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y;
}
#define TIME(name, C, PK, ...)                                                                     \
    do {                                                                                           \
        for (int w = 0; w < 3; w++) { __VA_ARGS__; }                                               \
        hipEventRecord(e0, 0);                                                                     \
        for (int w = 0; w < 10; w++) { __VA_ARGS__; }                                              \
        hipEventRecord(e1, 0); hipEventSynchronize(e1);                                            \
        float ms; hipEventElapsedTime(&ms, e0, e1);                                                \
        printf("%-22s chains %d  %7.3f ms  %6.2f ns per instruction slot of a wave  %6.2f T f32-op/s\n", name, C, ms / 10, ms / 10 * 1e6 / n / C, \
               (double)grid * 256 * C * (PK ? 2.0 : 1.0) * n / (ms / 10 * 1e-3) / 1e12);         \
    } while (0)
int main()
{
    const int grid = 256 * 4, n = 8192;            // one workgroup of 4 waves per SIMD... 4 workgroups of 4 waves per CU = 4 waves per SIMD
    float *o; hipMalloc(&o, (size_t)grid * 4 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int g4 = grid * 4;
#define BOTH(C) TIME("scalar v_fma_f32", C, 0, hipLaunchKernelGGL(scalar_k<C>, dim3(g4), dim3(256), 0, 0, o, 1.0001f, 0.5f, n)); \
                TIME("packed v_pk_fma_f32", C, 1, hipLaunchKernelGGL(packed_k<C>, dim3(g4), dim3(256), 0, 0, o, 1.0001f, 0.5f, n));
    { const int grid = g4; BOTH(1) BOTH(2) BOTH(4) BOTH(8) }
    return 0;
}
