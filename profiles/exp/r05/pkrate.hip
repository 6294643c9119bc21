// Does gfx950 issue v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 at the rate of a scalar v_fma_f32 (two f32 operations per lane and issue slot)?
// 16 independent accumulators per lane, 4096 iterations: 16 scalar ops against 8 packed ops per iteration.  HIP-event timed, 4 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o profiles/exp/r05/pkrate profiles/exp/r05/pkrate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int OP> __global__ __launch_bounds__(256) void scalar_k(float *out, float x, float y, int n)
{
    float a[16];
#pragma unroll
    for (int i = 0; i < 16; i++) a[i] = (float)(threadIdx.x + i);
    for (int it = 0; it < n; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            if (OP == 0) a[i] = __builtin_fmaf(a[i], x, y);
            else if (OP == 1) a[i] = a[i] * x;
            else a[i] = a[i] + y;
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) s += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int OP> __global__ __launch_bounds__(256) void packed_k(float *out, float x, float y, int n)
{
    f2 a[8];
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = f2{(float)(threadIdx.x + i), (float)(threadIdx.x + 8 + i)};
    const f2 xx = {x, x}, yy = {y, y};
    for (int it = 0; it < n; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (OP == 0) a[i] = __builtin_elementwise_fma(a[i], xx, yy);
            else if (OP == 1) a[i] = a[i] * xx;
            else a[i] = a[i] + yy;
        }
    }
    f2 s = {0, 0};
#pragma unroll
    for (int i = 0; i < 8; i++) s += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y;
}
#define TIME(name, ...)                                                                            \
    do {                                                                                           \
        for (int w = 0; w < 3; w++) { __VA_ARGS__; }                                               \
        hipEventRecord(e0, 0);                                                                     \
        for (int w = 0; w < 10; w++) { __VA_ARGS__; }                                              \
        hipEventRecord(e1, 0); hipEventSynchronize(e1);                                            \
        float ms; hipEventElapsedTime(&ms, e0, e1);                                                \
        printf("%-28s %8.3f ms  %7.2f T f32-op/s\n", name, ms / 10, (double)grid * 256 * 16.0 * n / (ms / 10 * 1e-3) / 1e12); \
    } while (0)
int main()
{
    const int grid = 256 * 4 * 4, n = 4096;       // 4 workgroups of 4 waves per CU: 4 waves per SIMD
    float *o; hipMalloc(&o, (size_t)grid * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    TIME("scalar v_fma_f32", hipLaunchKernelGGL(scalar_k<0>, dim3(grid), dim3(256), 0, 0, o, 1.0001f, 0.5f, n));
    TIME("packed v_pk_fma_f32", hipLaunchKernelGGL(packed_k<0>, dim3(grid), dim3(256), 0, 0, o, 1.0001f, 0.5f, n));
    TIME("scalar v_mul_f32", hipLaunchKernelGGL(scalar_k<1>, dim3(grid), dim3(256), 0, 0, o, 1.0001f, 0.5f, n));
    TIME("packed v_pk_mul_f32", hipLaunchKernelGGL(packed_k<1>, dim3(grid), dim3(256), 0, 0, o, 1.0001f, 0.5f, n));
    TIME("scalar v_add_f32", hipLaunchKernelGGL(scalar_k<2>, dim3(grid), dim3(256), 0, 0, o, 1.0001f, 0.5f, n));
    TIME("packed v_pk_add_f32", hipLaunchKernelGGL(packed_k<2>, dim3(grid), dim3(256), 0, 0, o, 1.0001f, 0.5f, n));
    return 0;
}
