// dma_misalign.hip -- does gfx950 LDS-DMA (buffer_load_dwordx4 ... lds) accept a source address that is only 2-byte (or 1-byte)
// aligned, does the data land correctly, and what does it cost?  (A 2-byte-shifted second copy of every source row in LDS would
// let ONE aligned dword hold any horizontal tap pair of the bilinear blend: half the tap reads of kernels_rectdma.hip.)
// build: hipcc --offload-arch=gfx950 -O3 -o dma_misalign dma_misalign.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ void dma16(unsigned voff, __amdgpu_buffer_rsrc_t rsrc, unsigned lds_dst, unsigned soff)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_dst), "s"(soff) : "memory");
}

// every workgroup streams `bytes_per_wg` bytes of src (its own slice) through LDS in 8 KB pieces with source offset +mis,
// copies the LAST piece out for checking, and xors everything so nothing is optimised away
__global__ __launch_bounds__(512) void stream(const uint8_t *src, size_t n, unsigned per_wg, int mis, uint8_t *check, unsigned *sink)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, (int)n, 0x00020000);
    const unsigned wave_off = __builtin_amdgcn_readfirstlane((threadIdx.x >> 6) * 1024u);
    const unsigned base = blockIdx.x * per_wg;
    unsigned acc = 0;
    for (unsigned p = 0; p < per_wg; p += 8192) {
        dma16(base + p + threadIdx.x * 16 + (unsigned)mis, rsrc, wave_off, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
        acc ^= reinterpret_cast<const unsigned *>(lds)[threadIdx.x];
        asm volatile("s_barrier" ::: "memory");
    }
    if (check && blockIdx.x == 0) for (int i = threadIdx.x; i < 8192; i += 512) check[i] = lds[i];
    if (acc == 0x12345678u) sink[0] = acc;
}

int main()
{
    const size_t n = (size_t)1 << 28;                       // 256 MiB
    uint8_t *src, *chk; unsigned *sink;
    CK(hipMalloc(&src, n + 64)); CK(hipMalloc(&chk, 8192)); CK(hipMalloc(&sink, 4));
    std::vector<uint8_t> h(n);
    for (size_t i = 0; i < n; i++) h[i] = (uint8_t)((i * 2654435761u) >> 13);
    CK(hipMemcpy(src, h.data(), n, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const unsigned wgs = 2048, per_wg = (unsigned)(n / wgs);
    for (int mis : {0, 4, 2, 1, 6, 3}) {
        float best = 1e9f;
        for (int it = 0; it < 4; it++) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(stream, dim3(wgs), dim3(512), 8192, 0, src, n, per_wg, mis, chk, sink);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        CK(hipGetLastError());
        std::vector<uint8_t> got(8192);
        CK(hipMemcpy(got.data(), chk, 8192, hipMemcpyDeviceToHost));
        size_t bad = 0;
        const size_t last = (size_t)per_wg - 8192 + mis;    // workgroup 0's last piece
        for (int i = 0; i < 8192; i++) if (got[i] != h[last + i]) bad++;
        printf("source misalignment %d bytes: %.1f us for 256 MiB -> %.2f TB/s, %zu of 8192 checked bytes wrong\n", mis, best * 1e3,
               (double)n / (best * 1e-3) / 1e12, bad);
    }
    return 0;
}
