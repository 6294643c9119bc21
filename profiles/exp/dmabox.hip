// dmabox.hip -- round 2 design probe for the fused rectify+decode: how fast do the SOURCE BOXES arrive when they go
// HBM -> LDS by gfx950 LDS-DMA (buffer_load_dwordx4 ... lds, 16 B per lane, no VGPR staging) in PLANE GROUPS of G planes,
// double-buffered, one barrier per phase (= one plane group of one tile)?  A 14-plane box of a wide tile does not fit LDS
// twice; G planes of a 256 x 16 tile do (2 x 2 x 7.7 KB), so the row segments can be 4x longer than round 1's 128 x 8 form.
// Also: is a 4-byte-aligned ds_read_b64 a fast path (the natural [plane][row][x] LDS image the DMA writes needs it for the
// dword pair of a bilinear tap)?
// build: hipcc --offload-arch=gfx950 -O3 -o dmabox dmabox.hip ; run: ./dmabox [pitch pad]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int NP = 14;

__device__ __forceinline__ void dma16(unsigned voff, __amdgpu_buffer_rsrc_t rsrc, unsigned lds_dst, unsigned soff)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(__builtin_amdgcn_readfirstlane(lds_dst)), "s"(soff) : "memory");
}

// TW x TH destination tile; its source box is (TW + 32) bytes x (TH + 3) rows, x0 16-byte aligned; chunk = 16 bytes.
// consume: 0 = fetch only; 1 = every thread also does the LDS reads of the real kernel (2 ds_read_b64 per plane and pixel)
template <int TW, int TH, int G, int NT, int WGS>
__global__ __launch_bounds__(NT, WGS * NT / 256 > 8 ? 2 : 1) void dmabox(const uint8_t *base, unsigned pstride, int pitch, int W, int H, int tiles_x, int tiles_y,
                                                 unsigned *sink, int consume, int misalign)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    constexpr int C = TW / 16 + 2, BH = TH + 3, E = C * BH, ROUNDS = (E + NT - 1) / NT;
    constexpr int PS = (E + 63) / 64 * 1024;                   // LDS bytes per plane (whole waves)
    constexpr int NPH = (NP + G - 1) / G;
    constexpr int PX = TW * TH / NT;                           // pixels per thread
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, (int)((NP - 1) * pstride + (unsigned)H * pitch), 0x00020000);
    const int T = tiles_x * tiles_y, per = (T + 7) >> 3;
    const int xcd = blockIdx.x & 7, lb = blockIdx.x >> 3, nbx = gridDim.x >> 3;
    const unsigned wave_lds = __builtin_amdgcn_readfirstlane((threadIdx.x >> 6) * 1024u);
    unsigned acc = 0;
    unsigned voff[ROUNDS];
    auto geom = [&](int cur) {
        const int ty = cur / tiles_x, tx = cur - ty * tiles_x;
        const int x0 = tx * TW - 16, y0 = ty * TH - 1;
#pragma unroll
        for (int r = 0; r < ROUNDS; r++) {
            const int e = threadIdx.x + NT * r;
            const int rr = e / C, cc = e - rr * C;
            const int gx = x0 + 16 * cc, gy = y0 + rr;
            const bool in = e < E && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
            voff[r] = in ? (unsigned)gy * pitch + gx : 0xFFFFFFF0u;
        }
    };
    auto issue = [&](int ph, int buf) {
#pragma unroll
        for (int g = 0; g < G; g++) {
            const int p = ph * G + g;
            if (p < NP) {
#pragma unroll
                for (int r = 0; r < ROUNDS; r++)
                    if ((int)__builtin_amdgcn_readfirstlane(threadIdx.x & ~63u) + NT * r < E) dma16(voff[r], rsrc, (unsigned)(buf * G * PS + g * PS + r * NT * 16) + wave_lds, (unsigned)p * pstride);
            }
        }
    };
    int l = lb;
    if (l >= per || xcd * per + l >= T) return;
    geom(xcd * per + l);
    issue(0, 0);
    int k = 0;                                                  // global phase counter (buffer = k & 1)
    for (;;) {
        const int nl = l + nbx;
        const bool has_next = nl < per && xcd * per + nl < T;
#pragma unroll
        for (int ph = 0; ph < NPH; ph++, k++) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (ph + 1 < NPH) issue(ph + 1, (k + 1) & 1);
            else if (has_next) { geom(xcd * per + nl); issue(0, (k + 1) & 1); }
            if constexpr (G * PS + C * 16 + 8 < 32768) if (consume) {
#pragma unroll
                for (int q = 0; q < PX; q++) {
                    // lane -> consecutive source bytes of one row (4 lanes per dword), rows by pass
                    const unsigned a = (unsigned)((k & 1) * G * PS + ((q + (threadIdx.x / TW)) % BH) * (C * 16) + ((threadIdx.x % TW) & ~3u) + (unsigned)misalign);
#pragma unroll
                    for (int g = 0; g < G; g++) {
                        unsigned long long v0, v1;
                        asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v0) : "v"(a), "i"(g * PS));
                        asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v1) : "v"(a), "i"(g * PS + C * 16));
                        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v0), "+v"(v1));
                        acc ^= (unsigned)v0 ^ (unsigned)(v0 >> 32) ^ (unsigned)v1 ^ (unsigned)(v1 >> 32);
                    }
                }
            }
        }
        if (!has_next) break;
        l = nl;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc == 0x12345678u) sink[0] = acc;
}

// throughput of ds_read_b64 at 8-byte vs 4-byte alignment (one workgroup per CU, 1024 threads, everything else idle)
__global__ __launch_bounds__(1024) void lds_align(unsigned *sink, int misalign, int iters)
{
    __shared__ __attribute__((aligned(16))) unsigned buf[8192];
    for (int i = threadIdx.x; i < 8192; i += 1024) buf[i] = i;
    __syncthreads();
    unsigned a = (threadIdx.x & 1023) * 8 + misalign, acc = 0;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            unsigned long long v;
            asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(a), "i"(u * 2048));
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v));
            acc += (unsigned)v + (unsigned)(v >> 32);
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
__global__ __launch_bounds__(1024) void lds_align32(unsigned *sink, int iters)
{
    __shared__ __attribute__((aligned(16))) unsigned buf[8192];
    for (int i = threadIdx.x; i < 8192; i += 1024) buf[i] = i;
    __syncthreads();
    unsigned a = (threadIdx.x & 1023) * 8 + 4, acc = 0;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            unsigned v0, v1;
            asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v0) : "v"(a), "i"(u * 2048));
            asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v1) : "v"(a), "i"(u * 2048 + 4));
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v0), "+v"(v1));
            acc += v0 + v1;
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

static uint8_t *st[2];
static unsigned *sink;
static hipEvent_t e0, e1;

template <int TW, int TH, int G, int NT, int WGS>
static void run(int pitch, unsigned pstride, int cold, int consume, int misalign)
{
    const int W = 4096, H = 3000;
    constexpr int C = TW / 16 + 2, BH = TH + 3, E = C * BH, ROUNDS = (E + NT - 1) / NT, PS = (E + 63) / 64 * 1024;
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    const size_t shmem = (size_t)2 * G * PS;
    if (shmem * WGS > 160 * 1024) { printf("tile %dx%d G %d NT %d wgs %d: LDS %zu x %d too large\n", TW, TH, G, NT, WGS, shmem, WGS); return; }
    auto kern = dmabox<TW, TH, G, NT, WGS>;
    CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    float sum = 0, best = 1e9f; const int reps = 12;
    for (int it = 0; it < reps + 2; it++) {
        const uint8_t *b = st[cold ? it & 1 : 0];
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(256 * WGS), dim3(NT), shmem, 0, b, pstride, pitch, W, H, tiles_x, tiles_y, sink, consume, misalign);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (it >= 2) { sum += ms; if (ms < best) best = ms; }
    }
    CK(hipGetLastError());
    printf("%s tile %3dx%-2d G %2d NT %4d wg/cu %d lds/wg %5.1f KB consume %d mis %d : avg %7.1f us  min %7.1f us -> %.2f TB/s (unique bytes)\n",
           cold ? "cold" : "warm", TW, TH, G, NT, WGS, shmem / 1024.0, consume, misalign, sum / reps * 1e3, best * 1e3,
           (double)W * H * NP / (sum / reps * 1e-3) / 1e12);
}

int main(int argc, char **argv)
{
    const int W = 4096, H = 3000;
    const int pad = argc > 1 ? atoi(argv[1]) : 0;
    const int pitch = W + pad;
    const unsigned pstride = (unsigned)pitch * H;
    for (int i = 0; i < 2; i++) { CK(hipMalloc(&st[i], (size_t)pstride * NP + 256)); CK(hipMemset(st[i], i + 1, (size_t)pstride * NP)); }
    CK(hipMalloc(&sink, 4));
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("pitch %d\n", pitch);
    // LDS alignment probe
    for (int mis = 0; mis <= 4; mis += 4) {
        float best = 1e9f;
        for (int it = 0; it < 5; it++) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(lds_align, dim3(256), dim3(1024), 0, 0, sink, mis, 2000);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        printf("ds_read_b64 misalign %d: %.1f us for 16000 reads per lane -> %.1f TB/s aggregate\n", mis, best * 1e3,
               256.0 * 1024 * 16000 * 8 / (best * 1e-3) / 1e12);
    }
    {
        float best = 1e9f;
        for (int it = 0; it < 5; it++) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(lds_align32, dim3(256), dim3(1024), 0, 0, sink, 2000);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        printf("2 x ds_read_b32 (same bytes): %.1f us -> %.1f TB/s aggregate\n", best * 1e3, 256.0 * 1024 * 16000 * 8 / (best * 1e-3) / 1e12);
    }
    for (int cold = 0; cold < 2; cold++) {
        run<128, 8, 14, 512, 3>(pitch, pstride, cold, 0, 0);
        run<256, 8, 14, 512, 2>(pitch, pstride, cold, 0, 0);
        run<256, 8, 7, 512, 3>(pitch, pstride, cold, 0, 0);
        run<256, 8, 4, 512, 3>(pitch, pstride, cold, 0, 0);
        run<256, 8, 4, 512, 4>(pitch, pstride, cold, 0, 0);
        run<256, 8, 2, 512, 4>(pitch, pstride, cold, 0, 0);
        run<256, 16, 4, 512, 2>(pitch, pstride, cold, 0, 0);
        run<256, 16, 4, 512, 3>(pitch, pstride, cold, 0, 0);
        run<256, 16, 2, 512, 3>(pitch, pstride, cold, 0, 0);
        run<256, 16, 2, 512, 4>(pitch, pstride, cold, 0, 0);
        run<256, 16, 2, 1024, 2>(pitch, pstride, cold, 0, 0);
        run<512, 8, 2, 512, 3>(pitch, pstride, cold, 0, 0);
        run<512, 8, 4, 512, 3>(pitch, pstride, cold, 0, 0);
        run<512, 16, 2, 1024, 2>(pitch, pstride, cold, 0, 0);
        run<128, 16, 4, 256, 4>(pitch, pstride, cold, 0, 0);
        run<128, 16, 2, 256, 6>(pitch, pstride, cold, 0, 0);
        run<256, 16, 2, 512, 3>(pitch, pstride, cold, 1, 0);
        run<256, 16, 2, 512, 3>(pitch, pstride, cold, 1, 4);
        run<256, 16, 4, 512, 3>(pitch, pstride, cold, 1, 4);
        run<256, 8, 4, 512, 4>(pitch, pstride, cold, 1, 4);
    }
    return 0;
}
