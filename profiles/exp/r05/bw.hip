// What does this box stream?  float4 copy / read-only / write-only kernels over 1 GiB, several launch shapes, HIP-event timed.
// Build: hipcc --offload-arch=gfx950 -O3 -o profiles/exp/r05/bw profiles/exp/r05/bw.hip      Run on the GPU box: profiles/exp/r05/bw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f4 __attribute__((ext_vector_type(4)));
template <bool NT> __global__ __launch_bounds__(256) void copy_k(const f4 *__restrict__ s, f4 *__restrict__ d, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        if (NT) __builtin_nontemporal_store(__builtin_nontemporal_load(s + i), d + i); else d[i] = s[i];
    }
}
template <bool NT> __global__ __launch_bounds__(256) void read_k(const f4 *__restrict__ s, float *__restrict__ out, size_t n)
{
    f4 a = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) a += NT ? __builtin_nontemporal_load(s + i) : s[i];
    if (a.x + a.y + a.z + a.w == 1.2345f) out[0] = 1;
}
template <bool NT> __global__ __launch_bounds__(256) void write_k(f4 *__restrict__ d, size_t n)
{
    const f4 v = {1, 2, 3, 4};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { if (NT) __builtin_nontemporal_store(v, d + i); else d[i] = v; }
}
#define TIME(name, bytes, ...)                                                                     \
    do {                                                                                           \
        for (int w = 0; w < 20; w++) { __VA_ARGS__; }                                              \
        hipEventRecord(e0, 0);                                                                     \
        for (int w = 0; w < 10; w++) { __VA_ARGS__; }                                              \
        hipEventRecord(e1, 0); hipEventSynchronize(e1);                                            \
        float ms; hipEventElapsedTime(&ms, e0, e1);                                                \
        printf("%-44s %7.1f GB/s\n", name, (double)(bytes) * 10 / (ms * 1e-3) / 1e9);              \
    } while (0)
int main()
{
    const size_t bytes = 1ull << 30, n = bytes / 16;
    f4 *a, *b; float *o;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&o, 64);
    hipMemset(a, 1, bytes); hipMemset(b, 2, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    char nm[128];
    for (unsigned g : {1024u, 2048u, 4096u, 16384u, (unsigned)(n / 256)}) {
        snprintf(nm, sizeof nm, "copy nt,    grid %u", g);      TIME(nm, 2 * bytes, hipLaunchKernelGGL(copy_k<true>, dim3(g), dim3(256), 0, 0, a, b, n));
        snprintf(nm, sizeof nm, "copy plain, grid %u", g);      TIME(nm, 2 * bytes, hipLaunchKernelGGL(copy_k<false>, dim3(g), dim3(256), 0, 0, a, b, n));
        snprintf(nm, sizeof nm, "read nt,    grid %u", g);      TIME(nm, bytes, hipLaunchKernelGGL(read_k<true>, dim3(g), dim3(256), 0, 0, a, o, n));
        snprintf(nm, sizeof nm, "read plain, grid %u", g);      TIME(nm, bytes, hipLaunchKernelGGL(read_k<false>, dim3(g), dim3(256), 0, 0, a, o, n));
        snprintf(nm, sizeof nm, "write nt,   grid %u", g);      TIME(nm, bytes, hipLaunchKernelGGL(write_k<true>, dim3(g), dim3(256), 0, 0, b, n));
        snprintf(nm, sizeof nm, "write plain,grid %u", g);      TIME(nm, bytes, hipLaunchKernelGGL(write_k<false>, dim3(g), dim3(256), 0, 0, b, n));
    }
    TIME("hipMemcpyDtoD", 2 * bytes, hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0));
    return 0;
}
