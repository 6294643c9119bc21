// does gfx950 execute scalar-memory atomics (s_atomic_add ... glc returning the old value through lgkmcnt)?  Every wave takes
// 16 tickets from one counter; the tickets must be a permutation of 0 .. n-1.  Also times it.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
__global__ void take(unsigned *ctr, unsigned *out, unsigned long long *clk)
{
    const unsigned long long t0 = wall_clock64();
    for (int i = 0; i < 16; i++) {
        unsigned v = 1;
        asm volatile("s_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "+s"(v) : "s"(ctr) : "memory");
        if (threadIdx.x == 0) out[blockIdx.x * 16 + i] = v;
    }
    if (threadIdx.x == 0) clk[blockIdx.x] = wall_clock64() - t0;
}
int main()
{
    const int nb = 2048;
    unsigned *ctr, *out; unsigned long long *clk;
    hipMalloc(&ctr, 256); hipMalloc(&out, nb * 16 * 4); hipMalloc(&clk, nb * 8);
    hipMemset(ctr, 0, 256);
    hipLaunchKernelGGL(take, dim3(nb), dim3(64), 0, 0, ctr, out, clk);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
    std::vector<unsigned> h(nb * 16); std::vector<unsigned long long> c(nb); unsigned fin = 0;
    hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(c.data(), clk, nb * 8, hipMemcpyDeviceToHost);
    hipMemcpy(&fin, ctr, 4, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    bool ok = fin == (unsigned)(nb * 16);
    for (size_t i = 0; i < h.size(); i++) ok = ok && h[i] == i;
    std::sort(c.begin(), c.end());
    printf("s_atomic_add glc: final %u (expected %d), tickets %s; 16 round trips per wave: median %.2f us, max %.2f us\n", fin, nb * 16,
           ok ? "are a permutation (WORKS)" : "BROKEN", c[nb / 2] / 100.0, c[nb - 1] / 100.0);
    return ok ? 0 : 2;
}
