// kernels_rectdma.hip -- fused rectify + multi-frequency decode, LDS-DMA form (SLR_OPT_RECT_DECODE_ALGO = 7, what "auto"
// picks when the maps and the stack layout allow it).  gfx950 (MI355X) only.
//
// Round 1's fused kernel staged the 14-plane source box of a 128 x 8 tile through VGPRs: box fetch (short row segments,
// the cost follows the 128-byte lines touched), tap/decode VALU work and LDS traffic ADDED instead of overlapping
// (VERDICT r01, 0.39 of the HBM roofline).  This form changes the decomposition:
//   * the box goes HBM -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds: 16 bytes per lane, no VGPR staging, no commit pass)
//     in its NATURAL [plane][row][x] byte layout with a compile-time row stride, every thread owning one 16-byte chunk;
//     the taps are read back with ds_read_b32 at immediate offsets from one address register, a pixel ahead of their use;
//   * a tile is decoded in 7 PHASES of 2 planes (white/black, then G1/G3 and G2/G4 of each frequency): only 2 planes of a
//     box are resident, so a 256 x 16 (or 512 x 8) tile fits a double / triple LDS buffer and its source row segments are
//     2-4x longer than before (profiles/r02/exp_dmabox.txt: 37 us instead of 54 us to pull one camera's boxes, cold);
//   * the tap state of a thread's 4 or 8 pixels stays in registers across the phases, the per-frequency results collapse
//     to one wrapped phase per pixel as soon as their four planes have passed;
//   * one s_barrier per phase; the DMA of phase k+A is in flight while phase k is decoded (A = 1 or 2); every wave issues
//     the same static sequence of vector-memory operations, so the waits are COUNTED (s_waitcnt vmcnt(N), N from the
//     sequence) and never drain the queue; the map digest of the next tile arrives by DMA as well.
// Results are bit-identical to the other forms (same table-driven decode, same v_perm / v_dot2 blend).
//
// Reference behaviour restated (never copied): stereoRect::doStereoRectify (cv::remap INTER_LINEAR, BORDER_CONSTANT)
// Duke/stereorect.cpp:26-34 fused with MFReconstruct::computeShadows/decodePatterns/getPhase Duke/mfreconstruct.cpp:190-269.
#include "decode_common.hpp"

#include <stdlib.h>

// Experiment switches.  The ablations and probes the measurements of DESIGN.md / LABNOTES.md were taken with -- SLR_DMA_ABL (1 no
// source traffic, 2 no LDS tap reads, 3 no barriers, 4 neither reads nor blend, 5 reads without blend, 6 = 1 + 2,
// 7 phases stored to an L2-resident window, 8 digests from an L2-resident set of tiles, 9 = 7 + 8), SLR_DMA_FORCE_MODE
// (one read mode for every wave: wrong results, timing only) and SLR_DMA_CLOCKPROBE (per-workgroup timeline) -- compile only with
// -DSLR_EXPERIMENTS (profiles/exp/ab/var_build.sh passes it); a production build that names one of them is an error, and every
// other former switch (static priorities, computed weights, flush-now / safe Gray forms, the tunable pair counts) is gone: the
// results are in LABNOTES.md.
#if !defined(SLR_EXPERIMENTS) && (defined(SLR_DMA_ABL) || defined(SLR_DMA_FORCE_MODE) || defined(SLR_DMA_CLOCKPROBE))
#error "SLR_DMA_ABL / SLR_DMA_FORCE_MODE / SLR_DMA_CLOCKPROBE are experiment switches: build with -DSLR_EXPERIMENTS"
#endif

namespace slr {

constexpr unsigned kDmaInvalid = 0x80000000u;    // buffer offset beyond every descriptor's range: loads 0, stores nothing

// destination tile TW x TH decoded by a workgroup of NT threads.  A thread owns QUADS of 4 horizontally adjacent pixels (one or
// two per tile; quad g = pass * NT + thread, see quad_row / quad_col).  The four pixels of a quad almost always take their taps
// from the same 8 source bytes of two (or three) source rows -- see the digest below.
template <int TW, int TH, int NT>
struct DmaGeom {
    static constexpr int NWAVES = NT / 64;
    static constexpr int PX = TW * TH / NT;                     // pixels per thread
    static constexpr int NQ = PX / 4;                           // quads per thread (= passes)
    static constexpr int QPR = TW / 4;                          // quads per tile row
    // A wave's 64 quads form a block of BR tile rows x BC quads (8 x 8 where the tile is 8 rows high): steps of sy run down the
    // image, so a compact block meets fewer of them than a 256-pixel row segment would (more waves in the cheap read mode), and
    // its 8 rows x 8 dwords fall on 64 different LDS banks (row stride 40 or 72 dwords).
    static constexpr int BR = TH >= 8 ? 8 : TH, BC = 64 / BR;
    static_assert(QPR % BC == 0 && TH % BR == 0, "quad blocks");
    // quad g = pass * NT + thread -> tile row, first tile column
    static __device__ __host__ constexpr int quad_row(int g) { return (g / 64) / (QPR / BC) * BR + (g % 64) / BC; }
    static __device__ __host__ constexpr int quad_col(int g) { return ((g / 64) % (QPR / BC) * BC + (g % 64) % BC) * 4; }
    static constexpr int WPR = TW / 64;                         // (waves per tile row of the chunk ownership, unchanged)
    static constexpr int RPP = NWAVES / WPR;
    static constexpr int CMAX = TW / 16 + 2;                    // 16-byte chunks per source row held in LDS
    static constexpr int BHMAX = (TH + 8) * CMAX <= NT ? TH + 8 : NT / CMAX;   // source rows held (one chunk per thread)
    static constexpr int RS = CMAX * 16;                        // LDS row stride (bytes)
    static constexpr int NCH = (CMAX * BHMAX + 63) / 64 * 64;   // chunks of a plane image, whole waves
    static constexpr int PS = NCH * 16;                         // LDS plane stride: the first NCH / 64 waves own a 1 KiB slot each
    static_assert(TW % 64 == 0 && NWAVES % WPR == 0 && RPP >= 1 && PX >= 4 && PX % 4 == 0 && PX <= 8, "tile shape");
    static_assert(BHMAX >= TH + 3, "no room for the bilinear row and a little tilt");
    static_assert(BHMAX * RS <= PS && PS <= 8192 && NCH <= NT, "tap address field (13 bits), one chunk per thread");
};

// shapes (SLR_OPT_RECT_DMA_SHAPE): 0 = 256x16 / 512 threads, 1 = 256x8 / 512, 2 = 256x8 / 256, 3 = 128x16 / 512, 4 = 128x8 / 256,
// 5 = 256x4 / 256, 6 = 128x16 / 256
constexpr int kDmaShapes = 7;
static int dma_shape_tw(int shape) { return shape == 3 || shape == 4 || shape == 6 ? 128 : 256; }
static int dma_shape_th(int shape) { return shape == 0 || shape == 3 || shape == 6 ? 16 : shape == 5 ? 4 : 8; }
// (shapes 2, 4, 5 and 6 are measured-dominated -- DESIGN section 9 -- and compiled with -DSLR_ALL_FORMS only; without it
// slr_set_option refuses them)
#ifdef SLR_ALL_FORMS
#define SLR_DMA_SHAPE_SWITCH(shape, X)                  \
    switch (shape) {                                    \
    case 1:  X(256, 8, 512); break;                     \
    case 2:  X(256, 8, 256); break;                     \
    case 3:  X(128, 16, 512); break;                    \
    case 4:  X(128, 8, 256); break;                     \
    case 5:  X(256, 4, 256); break;                     \
    case 6:  X(128, 16, 256); break;                    \
    default: X(256, 16, 512); break;                    \
    }
#else
#define SLR_DMA_SHAPE_SWITCH(shape, X)                  \
    switch (shape) {                                    \
    case 1:  X(256, 8, 512); break;                     \
    case 3:  X(128, 16, 512); break;                    \
    default: X(256, 16, 512); break;                    \
    }
#endif

// map digest, one dword per destination pixel, [tile][thread][pass][pixel of the quad] (a thread's PX entries are contiguous):
//   [12:2]  dword index of the tap's upper left byte in a plane's LDS image: (sy - y0) * (CMAX * 4) + ((sx - x0) >> 2)
//   [14:13] (sx - x0) & 3
//   [28:16] 4 * (fy << 5 | fx)  (cv::remap's 5-bit fractions: the byte offset of the pixel's entry in the weight tables), or
//           4 * 1024: the sample is 0 (pixel beyond the ragged image edge, or footprint completely outside the source)
//   [31:29] of a quad's entries 0 .. 3: bits 2:0, 5:3, 8:6, 11:9 of the quad's SLOT in the tile, tile row << 6 | quad column --
//           where the decode stores the quad's results (see "quads sorted by class" in dma_tiles_kernel)
// and the QUAD's class, from which the kernel picks a wave-uniform way of reading the taps:
//   class 0: all four pixels' taps lie in source rows r0, r0+1 and in the two dwords c0, c0+1 of those rows -> 4 dword reads per
//            plane serve the whole quad (instead of 4 per pixel); class 1: the same with rows r0 .. r0+2 (the quad straddles a
//            step of sy); class 2: anything else (the per-pixel reads).  With a scale near 1 a quad spans source bytes
//            x .. x+4 -- always inside two dwords -- so class 2 is borders and strongly distorting maps.
//   bit 0   the pixel's dword column minus c0 (0 / 1), bit 1: its sy minus r0 (0 / 1); classes 0 and 1 only
//   bit 15  entry 0: class & 1, entry 1: class >> 1, entry 2: the quad is not the one the thread's position names (it was moved)
//   A zero-sample pixel inside a class 0 / 1 quad carries the quad's base address (r0, c0) in [12:2]: entry 0 always yields it.
// box table: int4 per tile = x0 (multiple of 16, may be negative), y0, chunks per row, rows; z == 0: nothing to fetch
constexpr unsigned kDmaZeroEntry = 1024u << 18;
// An ENTRY of the tile tables is what a workgroup decodes in one go: a whole tile -- or, where the tile's source box is larger than
// the LDS image (the corner tiles of a keystone map), a PART of it: the waves of a workgroup own fixed blocks of the tile
// (DmaGeom::quad_row / quad_col), so a part is a contiguous group of waves (halves, quarters, ... single waves), has its own,
// smaller box, and is decoded by its waves alone while the others only keep the barriers and fetch their chunks.  Entry t < T is
// tile t (or its first part); the other parts take entries T, T + 1, ... in allocation order (an atomic counter, at most T / 2
// of them).  A tile that no split makes fit goes to the gather fix-up list.
// box int4 of an entry: x0 (multiple of 16, may be negative), y0, z = chunks per row | tx << 8, w = rows | ty << 8 | wave mask << 24;
// chunks == 0: nothing to fetch; mask == 0: an unused extra entry (the table is zeroed first).
constexpr unsigned kDmaPromote = 2, kDmaNoPromote = 5;       // flagged pixel positions from which a wave's three-row quads become per-pixel quads
constexpr int kDmaStatExtras = 7;                            // statistics word: extra entries allocated (may overshoot the capacity)
__device__ __host__ constexpr unsigned dma_extra_capacity(unsigned T) { return T / 2; }

template <int TW, int TH, int NT>
__global__ __launch_bounds__(NT) void dma_tiles_kernel(const int16_t *__restrict__ map_xy, const uint16_t *__restrict__ map_frac,
                                                       int W, int H, int tiles_x, int4 *__restrict__ boxes,
                                                       unsigned *__restrict__ digest, unsigned *__restrict__ nofit,
                                                       unsigned *__restrict__ nofit_list, unsigned promote, int sort_quads)
{
    typedef DmaGeom<TW, TH, NT> Gm;
    constexpr int NW = Gm::NWAVES;
    __shared__ int red[NW][4];
    __shared__ int spart[NW][4];                             // per part: x0, y0, fits, entry
    __shared__ int slevel;
    const unsigned T = gridDim.x;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int sxs[Gm::PX], sys[Gm::PX];
    unsigned frs[Gm::PX];
    int gq[Gm::NQ];                                          // the quads this thread's slots decode (DmaGeom numbering)
    auto load_quad = [&](int p) {                            // map entries of quad gq[p]
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int q = 4 * p + i;
            const int row = ty * TH + Gm::quad_row(gq[p]), col = tx * TW + Gm::quad_col(gq[p]) + i;
            sxs[q] = 0x7FFFFFFF; sys[q] = 0; frs[q] = 0;
            if (row < H && col < W) {
                const size_t m = (size_t)row * W + col;
                const int sx = map_xy[2 * m], sy = map_xy[2 * m + 1];
                if (!(sx >= W || sx + 1 < 0 || sy >= H || sy + 1 < 0)) {   // footprints completely outside read 0
                    sxs[q] = sx; sys[q] = sy; frs[q] = map_frac[m] & 1023u;
                }
            }
        }
    };
    int mnx = 0x7FFFFFFF, mxx = -0x7FFFFFFF, mny = 0x7FFFFFFF, mxy = -0x7FFFFFFF;
#pragma unroll
    for (int p = 0; p < Gm::NQ; p++) { gq[p] = p * NT + (int)threadIdx.x; load_quad(p); }
#pragma unroll
    for (int q = 0; q < Gm::PX; q++)
        if (sxs[q] != 0x7FFFFFFF) {
            mnx = sxs[q] < mnx ? sxs[q] : mnx; mxx = sxs[q] > mxx ? sxs[q] : mxx;
            mny = sys[q] < mny ? sys[q] : mny; mxy = sys[q] > mxy ? sys[q] : mxy;
        }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        int t;
        t = __shfl_xor(mnx, d); mnx = t < mnx ? t : mnx;
        t = __shfl_xor(mxx, d); mxx = t > mxx ? t : mxx;
        t = __shfl_xor(mny, d); mny = t < mny ? t : mny;
        t = __shfl_xor(mxy, d); mxy = t > mxy ? t : mxy;
    }
    if (lane == 0) { red[wv][0] = mnx; red[wv][1] = mxx; red[wv][2] = mny; red[wv][3] = mxy; }
    __syncthreads();
    if (threadIdx.x == 0) {
        // the box of waves [w0, w1): z = 0 when they sample nothing; returns whether it fits the LDS image
        auto part_box = [&](int w0, int w1, int4 &b) -> bool {
            int ax = 0x7FFFFFFF, bx = -0x7FFFFFFF, ay = 0x7FFFFFFF, by = -0x7FFFFFFF;
            for (int w = w0; w < w1; w++) {
                ax = red[w][0] < ax ? red[w][0] : ax; bx = red[w][1] > bx ? red[w][1] : bx;
                ay = red[w][2] < ay ? red[w][2] : ay; by = red[w][3] > by ? red[w][3] : by;
            }
            b = make_int4(0, 0, 0, 0);
            if (ax > bx) return true;
            b.x = ax & ~15;                                  // 16-byte aligned origin (also for negative x)
            b.y = ay;
            b.z = (((bx + 1) - b.x) >> 4) + 1;               // chunks covering x0 .. max sx + 1
            b.w = (by + 1) - ay + 1;                         // rows y0 .. max sy + 1
            return b.z <= Gm::CMAX && b.w <= Gm::BHMAX;
        };
        int4 b;
        int level = 1;                                       // parts the tile is decoded in; 0: the gather fix-up
        if (!part_box(0, NW, b)) {
            level = 0;
            for (int L = 2; L <= NW && level == 0; L *= 2) {
                bool all = true;
                for (int p = 0; p < L && all; p++) all = part_box(p * (NW / L), (p + 1) * (NW / L), b);
                if (all) level = L;
            }
        }
        unsigned base = 0;
        if (level > 1) {
            base = atomicAdd(nofit + kDmaStatExtras, (unsigned)(level - 1));
            if (base + (unsigned)(level - 1) > dma_extra_capacity(T)) level = 0;     // (the table is full: its slots stay unused entries)
        }
        if (level == 0) {                                    // rewritten by the fix-up pass (mf_rect_fixup_kernel): nothing to fetch here
            nofit_list[atomicAdd(nofit, 1u)] = blockIdx.x;
            boxes[blockIdx.x] = make_int4(0, 0, tx << 8, ty << 8 | (int)(((1u << NW) - 1u) << 24));
            spart[0][0] = 0; spart[0][1] = 0; spart[0][2] = 0; spart[0][3] = (int)blockIdx.x;
        } else {
            for (int p = 0; p < level; p++) {
                const int w0 = p * (NW / level), w1 = (p + 1) * (NW / level);
                (void)part_box(w0, w1, b);
                const unsigned ent = p == 0 ? blockIdx.x : T + base + (unsigned)(p - 1);
                const unsigned mask = ((1u << (w1 - w0)) - 1u) << w0;
                spart[p][0] = b.x; spart[p][1] = b.y; spart[p][2] = 1; spart[p][3] = (int)ent;
                b.z |= tx << 8; b.w |= ty << 8 | (int)(mask << 24);
                boxes[ent] = b;
            }
        }
        slevel = level;
    }
    __syncthreads();
    const int level = slevel, part = level > 1 ? wv / (NW / level) : 0;
    const int x0 = spart[part][0], y0 = spart[part][1], fits = spart[part][2];
    const size_t ent = (size_t)(unsigned)spart[part][3];
    // the class and base (r0, c0) of the quad in slot p, over its pixels that sample anything
    auto quad_class = [&](int p, int &r0, int &c0) -> unsigned {
        int r1 = -0x7FFFFFFF, c1 = -0x7FFFFFFF;
        r0 = 0x7FFFFFFF; c0 = 0x7FFFFFFF;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int q = 4 * p + i;
            if (fits && sxs[q] != 0x7FFFFFFF) {
                const int rr = sys[q] - y0, cc = (sxs[q] - x0) >> 2;
                r0 = rr < r0 ? rr : r0; r1 = rr > r1 ? rr : r1; c0 = cc < c0 ? cc : c0; c1 = cc > c1 ? cc : c1;
            }
        }
        const bool none = r0 == 0x7FFFFFFF;
        if (none) { r0 = r1 = c0 = c1 = 0; }
        bool wide = r1 - r0 > 1 || c1 - c0 > 1;
#pragma unroll
        for (int i = 0; i < 4; i++) {                       // the RIGHT tap too must lie inside the 8 bytes c0*4 .. c0*4+7
            const int q = 4 * p + i;
            if (fits && sxs[q] != 0x7FFFFFFF && (sxs[q] - x0) - 4 * c0 > 6) wide = true;
        }
        return wide ? 2u : (r1 > r0 ? 1u : 0u);
    };
    // Quads sorted by class.  The decode reads a wave's taps in ONE mode, so a single straddling quad makes its whole wave pay the
    // three-row (or per-pixel) reads -- and the steps of sy run down the image in strips every 1 / slope pixels: on the keystone
    // maps of verged rigs 5-8 % of the quads put 60 % of the waves into the dear modes.  A tile that is decoded whole therefore
    // hands its class 1 / 2 quads to as few waves as hold them (the waves that have the most already), in exchange for plain
    // quads of those waves; every quad carries its slot in the tile (digest bits 31:29), which is where the decode stores it.
    if (sort_quads) {                                        // (kernel argument: block-uniform)
        __shared__ unsigned short lst_a[Gm::NQ * NT], lst_c[Gm::NQ * NT];
        __shared__ int cnt_a[NW * Gm::NQ], cnt_c[NW * Gm::NQ], wbad[NW];
        __shared__ unsigned ssink;
        bool bad[Gm::NQ];
        int nbad = 0;
#pragma unroll
        for (int p = 0; p < Gm::NQ; p++) {
            int r0, c0;
            bad[p] = level == 1 && fits && quad_class(p, r0, c0) != 0u;
            nbad += __popcll(__ballot(bad[p]));
        }
        if (lane == 0) wbad[wv] = nbad;
        __syncthreads();
        if (threadIdx.x == 0) {
            int total = 0, holders = 0;
            for (int w = 0; w < NW; w++) { total += wbad[w]; holders += wbad[w] > 0 ? 1 : 0; }
            const int need = (total + 64 * Gm::NQ - 1) / (64 * Gm::NQ);
            unsigned sink = 0;
            if (need < holders)                              // (otherwise no exchange lowers the number of waves that pay)
                for (int k = 0; k < need; k++) {
                    int best = -1;
                    for (int w = 0; w < NW; w++)
                        if (!((sink >> w) & 1u) && (best < 0 || wbad[w] >= wbad[best])) best = w;
                    sink |= 1u << best;
                }
            ssink = sink;
        }
        __syncthreads();
        const unsigned sink = ssink;
        if (sink != 0u) {                                    // (block-uniform)
            const bool mine = ((sink >> wv) & 1u) != 0;
            unsigned long long ba[Gm::NQ], bc[Gm::NQ];
#pragma unroll
            for (int p = 0; p < Gm::NQ; p++) {
                ba[p] = __ballot(bad[p] && !mine);           // class 1 / 2 quads outside the sink waves ...
                bc[p] = __ballot(!bad[p] && mine);           // ... change places with plain quads inside them, in slot order
                if (lane == 0) { cnt_a[wv * Gm::NQ + p] = __popcll(ba[p]); cnt_c[wv * Gm::NQ + p] = __popcll(bc[p]); }
            }
            __syncthreads();
            int base_a = 0, base_c = 0, total_a = 0;
            for (int j = 0; j < NW * Gm::NQ; j++) {
                if (j < wv * Gm::NQ) { base_a += cnt_a[j]; base_c += cnt_c[j]; }
                total_a += cnt_a[j];
            }
            const unsigned long long below = (1ull << lane) - 1ull;
            int rank[Gm::NQ];
#pragma unroll
            for (int p = 0; p < Gm::NQ; p++) {
                const int ra = base_a + __popcll(ba[p] & below), rc = base_c + __popcll(bc[p] & below);
                rank[p] = -1;
                if (bad[p] && !mine) { lst_a[ra] = (unsigned short)gq[p]; rank[p] = ra; }
                if (!bad[p] && mine) { lst_c[rc] = (unsigned short)gq[p]; rank[p] = rc; }
                base_a += __popcll(ba[p]); base_c += __popcll(bc[p]);
            }
            __syncthreads();
#pragma unroll
            for (int p = 0; p < Gm::NQ; p++) {
                const int own = gq[p];
                if (bad[p] && !mine) gq[p] = lst_c[rank[p]];
                else if (!bad[p] && mine && rank[p] < total_a) gq[p] = lst_a[rank[p]];
                if (gq[p] != own) load_quad(p);
            }
        }
    }
    unsigned tcls = 0;                                      // the thread's read class over its quads (what dma_tap_setup sees)
#pragma unroll
    for (int p = 0; p < Gm::NQ; p++) {
        int r0, c0;
        unsigned cls = quad_class(p, r0, c0);
        // The decode reads a wave's taps in ONE mode.  The three-row mode blends a third row for every pixel POSITION (0 .. 3 of a
        // quad) at which some lane's pixel lies in its quad's lower row pair: +4 VALU instructions per position, plane pair and
        // pixel, in a kernel that is bound by VALU issue.  The per-pixel mode costs LDS reads instead (32 instead of 12 dwords per
        // quad and plane pair: another unit -- as long as not most waves do it).  From `promote` flagged positions on, the wave's
        // straddling quads are filed as class 2, i.e. the wave takes the per-pixel mode.  launch_dma_tiles decides: on maps where
        // few waves straddle (the near-identity bench maps: 27 %) promoting them is worth 2-3 % (120.2 -> 117.4 us); where most
        // do (keystone maps of verged rigs: 59-66 %) the LDS unit saturates (0.1 rad rig 140 -> 158 us), so there nothing is
        // promoted (profiles/exp/r03/b22_promote.txt).
        {
            unsigned flagged = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int q = 4 * p + i;
                const bool low = cls == 1u && fits && sxs[q] != 0x7FFFFFFF && (sys[q] - y0) - r0 == 1;
                flagged += __ballot(low) != 0ull ? 1u : 0u;
            }
            if (flagged >= promote && cls == 1u) cls = 2u;
        }
        tcls = cls > tcls ? cls : tcls;
        if (fits) {                                         // statistics (slr_get_rectify_info): quads per class
            const unsigned long long b1 = __ballot(cls == 1u), b2 = __ballot(cls == 2u);
            if (lane == 0) {
                const unsigned n1 = (unsigned)__popcll(b1), n2 = (unsigned)__popcll(b2);
                atomicAdd(nofit + 1, 64u - n1 - n2);
                if (n1) atomicAdd(nofit + 2, n1);
                if (n2) atomicAdd(nofit + 3, n2);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int q = 4 * p + i;
            unsigned e = kDmaZeroEntry;
            if (fits && sxs[q] != 0x7FFFFFFF) {
                const int bx = sxs[q] - x0, rr = sys[q] - y0;
                e = (unsigned)(rr * (Gm::CMAX * 4) + (bx >> 2)) << 2 | ((unsigned)bx & 3u) << 13 | frs[q] << 18;
                if (cls < 2u) e |= (unsigned)((bx >> 2) - c0) | (unsigned)(rr - r0) << 1;
            } else if (cls < 2u) {
                e |= (unsigned)(r0 * (Gm::CMAX * 4) + c0) << 2;          // zero sample, but the quad's base address
            }
            if (i == 0) e |= (cls & 1u) << 15;
            if (i == 1) e |= (cls >> 1) << 15;
            if (i == 2) e |= gq[p] != p * NT + (int)threadIdx.x ? 1u << 15 : 0u;
            e |= (((unsigned)(Gm::quad_row(gq[p]) << 6 | Gm::quad_col(gq[p]) >> 2) >> (3 * i)) & 7u) << 29;
            digest[ent * (TW * TH) + threadIdx.x * Gm::PX + q] = e;      // (a part's digest holds its own waves' slots only)
        }
    }
    if (fits) {                                             // ... and waves per read mode (the wave-uniform choice of the decode)
        const int mode = __ballot(tcls == 2u) != 0ull ? 2 : (__ballot(tcls == 1u) != 0ull ? 1 : 0);
        if (lane == 0) atomicAdd(nofit + 4 + mode, 1u);
    }
}

static size_t dma_tile_count(int W, int H, int shape)
{
    const int tw = dma_shape_tw(shape), th = dma_shape_th(shape);
    return (size_t)((W + tw - 1) / tw) * ((H + th - 1) / th);
}
// entries the tables have room for: the tiles + the extra parts of split tiles
static size_t dma_entry_capacity(int W, int H, int shape) { const size_t T = dma_tile_count(W, H, shape); return T + dma_extra_capacity((unsigned)T); }
static size_t dma_digest_offset(int W, int H, int shape) { return (dma_entry_capacity(W, H, shape) * sizeof(int4) + 255) & ~(size_t)255; }
static size_t dma_nofit_offset(int W, int H, int shape)
{
    return dma_digest_offset(W, H, shape) + dma_entry_capacity(W, H, shape) * (size_t)(dma_shape_tw(shape) * dma_shape_th(shape)) * 4;
}
static size_t dma_list_offset(int W, int H, int shape) { return dma_nofit_offset(W, H, shape) + 256; }     // tiles that do not fit
size_t dma_tiles_bytes(int W, int H, int shape) { return dma_list_offset(W, H, shape) + dma_tile_count(W, H, shape) * sizeof(unsigned); }
size_t dma_tile_count_of(int W, int H, int shape) { return dma_tile_count(W, H, shape); }
unsigned dma_extra_entries_capacity(int W, int H, int shape) { return dma_extra_capacity((unsigned)dma_tile_count(W, H, shape)); }

hipError_t launch_dma_tiles(const int16_t *map_xy, const uint16_t *map_frac, int W, int H, void *buf, int shape, bool promote,
                            bool sort_quads, unsigned *nofit_host, hipStream_t s)
{
    char *b = reinterpret_cast<char *>(buf);
    int4 *boxes = reinterpret_cast<int4 *>(b);
    unsigned *digest = reinterpret_cast<unsigned *>(b + dma_digest_offset(W, H, shape));
    unsigned *nofit = reinterpret_cast<unsigned *>(b + dma_nofit_offset(W, H, shape));
    unsigned *nofit_list = reinterpret_cast<unsigned *>(b + dma_list_offset(W, H, shape));
    hipError_t e = hipMemsetAsync(nofit, 0, kDmaTileStats * sizeof(unsigned), s);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(boxes, 0, dma_entry_capacity(W, H, shape) * sizeof(int4), s);     // (unused extra entries: no waves, nothing to fetch)
    if (e != hipSuccess) return e;
    const int tiles_x = (W + dma_shape_tw(shape) - 1) / dma_shape_tw(shape);
    const dim3 grid((unsigned)dma_tile_count(W, H, shape));
#define SLR_DMA_X(TW, TH, NT) SLR_LAUNCH((dma_tiles_kernel<TW, TH, NT>), grid, dim3(NT), 0, s, map_xy, map_frac, W, H, tiles_x, boxes, digest, nofit, nofit_list, promote ? kDmaPromote : kDmaNoPromote, sort_quads ? 1 : 0)
    SLR_DMA_SHAPE_SWITCH(shape, SLR_DMA_X)
#undef SLR_DMA_X
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    return hipMemcpyAsync(nofit_host, nofit, kDmaTileStats * sizeof(unsigned), hipMemcpyDeviceToHost, s);   // the caller synchronises the stream
}

// ------------------------------------------------------------------------------------------------------
// the decode kernel
// ------------------------------------------------------------------------------------------------------
struct DmaJob {
    const uint8_t *base;         // plane 0; plane p is base + p * pstride (+ fringe_skip for the fringe planes p >= 2)
    unsigned pstride;
    unsigned fringe_skip;        // bytes between black and the first fringe plane beyond one plane stride (a hybrid stack keeps its
                                 // Gray planes there, BASELINE config 3); 0 for a plain 14-plane stack
    unsigned stack_bytes;        // 13 * pstride + fringe_skip + H * pitch
    const int4 *boxes;
    const unsigned *digest;
    unsigned digest_bytes;
    unsigned entries;            // tiles + the extra parts of split tiles (dma_tiles_kernel)
    float *phase;
    uint8_t *valid;              // null: the flag is folded into the phase (NaN), see launch_mf_decode
};
struct DmaJobs { DmaJob j[kDmaMaxJobs]; };   // the cameras of one frame, or of a group of frames (slr_reconstruct_mf_batch): j[2 f + cam]

// LDS-DMA of 16 bytes per lane: LDS[lds_dst + 16 * lane] = buffer[voff + soff .. + 15] (zeros beyond the descriptor's range).
// M0 carries the LDS base and is compiler-reserved: saved and restored inside the one statement that uses it.
__device__ __forceinline__ void dma16(unsigned voff, __amdgpu_buffer_rsrc_t rsrc, unsigned lds_dst, unsigned soff)
{
    unsigned keep;
    // (lds_dst and soff are wave-uniform; under register pressure hipcc may keep such a value in a VGPR and would then hand the
    //  VGPR to the "s" operand: readfirstlane pins it to the scalar file)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(__builtin_amdgcn_readfirstlane(lds_dst)), "s"(__builtin_amdgcn_readfirstlane(soff)) : "memory");
}

template <int N>
__device__ __forceinline__ void wait_vm()
{
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
}

// planes of phase p: (white, black), then per frequency c (G1, G3) -> d and (G2, G4) -> n   (plane 4c + 2 + s holds G(s+1))
__device__ __host__ constexpr int dma_phase_plane(int p, int g)
{
    return p == 0 ? g : 2 + 4 * ((p - 1) >> 1) + ((p - 1) & 1) + 2 * g;
}

struct DmaTap {                          // per-pixel tap state, shared by all planes
    unsigned a0;                         // LDS address of the dword holding the upper left tap in the plane image at offset 0
    unsigned sel;                        // v_perm selector: bytes (sh, 0, sh+1, 0) -> u16 pair of the two taps
    u16x2 w0, w1;                        // wx0*wy0, wx1*wy0 | wx0*wy1, wx1*wy1, times 64 (see tile_taps_map in kernels_decode.hip)
};

// the 16-bit blend weights of a (fx, fy) fraction pair -- what tile_taps_packed (kernels_decode.hip) computes per pixel; here
// they come from a 1025-entry LDS table indexed by the digest (entry 1024 = zero weights: the sample is 0)
__device__ __forceinline__ void dma_weights(unsigned i, unsigned &w0, unsigned &w1)
{
    const unsigned fx = i & 31u, fy6 = (i >> 5) << 6;                      // fy << 6
    const unsigned wxp = i >= 1024u ? 0u : __umul24(fx, 0xFFFFu) + 32u;    // (32 - fx) | fx << 16
    w0 = __umul24(wxp, 2048u - fy6);
    w1 = __umul24(wxp, fy6);
    w0 = w0 == 0x10000u ? 0xFFFFu : w0;                                   // fx = fy = 0: (S*65535 + 32768) >> 16 == S
}

// The taps of one pixel in the two planes of a phase: 8 dwords (plane 0: row 0 dwords c, c+1, row 1 dwords c, c+1; plane 1
// the same).  hipcc would fold these loads into ds_read2_b32 behind a v_add per plane image (8-bit offsets) and wait for each
// one; here they are ds_read_b32 with 16-bit immediate offsets from ONE address register, issued a pixel ahead of their use
// and retired by a counted lgkmcnt (LDS operations return in order; anything the compiler interleaves only makes the count
// stricter).  A 4-byte-aligned ds_read_b64 is NOT a fast path when the lanes' alignments differ (profiles/r02/exp_ldspat.txt:
// 64 instead of 3 clocks per wave instruction), 2 x ds_read_b32 costs 4.9.
struct DmaRd { unsigned v[8]; };

template <unsigned IMG0, unsigned IMG1, unsigned RS>
__device__ __forceinline__ void dma_rd(DmaRd &r, unsigned addr)
{
    asm volatile("ds_read_b32 %0, %8 offset:%9\n\t"
                 "ds_read_b32 %1, %8 offset:%10\n\t"
                 "ds_read_b32 %2, %8 offset:%11\n\t"
                 "ds_read_b32 %3, %8 offset:%12\n\t"
                 "ds_read_b32 %4, %8 offset:%13\n\t"
                 "ds_read_b32 %5, %8 offset:%14\n\t"
                 "ds_read_b32 %6, %8 offset:%15\n\t"
                 "ds_read_b32 %7, %8 offset:%16"
                 : "=&v"(r.v[0]), "=&v"(r.v[1]), "=&v"(r.v[2]), "=&v"(r.v[3]), "=&v"(r.v[4]), "=&v"(r.v[5]), "=&v"(r.v[6]), "=&v"(r.v[7])
                 : "v"(addr), "n"(IMG0), "n"(IMG0 + 4), "n"(IMG0 + RS), "n"(IMG0 + RS + 4), "n"(IMG1), "n"(IMG1 + 4), "n"(IMG1 + RS),
                   "n"(IMG1 + RS + 4)
                 : "memory");
}
// at most N LDS operations issued after r's loads may still be outstanding
template <int N>
__device__ __forceinline__ void dma_rd_wait(DmaRd &r)
{
    asm volatile("s_waitcnt lgkmcnt(%8)"
                 : "+v"(r.v[0]), "+v"(r.v[1]), "+v"(r.v[2]), "+v"(r.v[3]), "+v"(r.v[4]), "+v"(r.v[5]), "+v"(r.v[6]), "+v"(r.v[7])
                 : "n"(N) : "memory");
}
// blended sample of plane g of a phase: the sample is the HIGH half-word of the result (see tile_sample, kernels_decode.hip)
__device__ __forceinline__ unsigned dma_blend(const DmaRd &r, int g, const DmaTap &k)
{
    const unsigned p0 = __builtin_amdgcn_perm(r.v[4 * g + 1], r.v[4 * g], k.sel);
    const unsigned p1 = __builtin_amdgcn_perm(r.v[4 * g + 3], r.v[4 * g + 2], k.sel);
    unsigned acc = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, p0), k.w0, 512u << 6, false);
    acc = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, p1), k.w1, acc, false);
    return acc;
}

// three source rows of the two planes of a phase (quads of class 1): rows 0, 1 as DmaRd's v[0..7], row 2 in x[0..3]
struct DmaRd3 { DmaRd a; unsigned x[4]; };
template <unsigned IMG0, unsigned IMG1, unsigned RS>
__device__ __forceinline__ void dma_rd_row2(DmaRd3 &r, unsigned addr)
{
    asm volatile("ds_read_b32 %0, %4 offset:%5\n\t"
                 "ds_read_b32 %1, %4 offset:%6\n\t"
                 "ds_read_b32 %2, %4 offset:%7\n\t"
                 "ds_read_b32 %3, %4 offset:%8"
                 : "=&v"(r.x[0]), "=&v"(r.x[1]), "=&v"(r.x[2]), "=&v"(r.x[3])
                 : "v"(addr), "n"(IMG0 + 2 * RS), "n"(IMG0 + 2 * RS + 4), "n"(IMG1 + 2 * RS), "n"(IMG1 + 2 * RS + 4)
                 : "memory");
}
__device__ __forceinline__ void dma_rd3_wait(DmaRd3 &r)
{
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(r.a.v[0]), "+v"(r.a.v[1]), "+v"(r.a.v[2]), "+v"(r.a.v[3]), "+v"(r.a.v[4]), "+v"(r.a.v[5]), "+v"(r.a.v[6]), "+v"(r.a.v[7]),
                   "+v"(r.x[0]), "+v"(r.x[1]), "+v"(r.x[2]), "+v"(r.x[3])
                 :: "memory");
}
// blended sample of plane g over THREE rows (read mode 1): the pixel's two weight pairs sit in the row slots its sy selects, the
// third slot is zero -- an exact no-op in the integer dot product, and no per-register selects
__device__ __forceinline__ unsigned dma_blend3(const DmaRd3 &r, int g, const DmaTap &k)
{
    const unsigned p0 = __builtin_amdgcn_perm(r.a.v[4 * g + 1], r.a.v[4 * g], k.sel);
    const unsigned p1 = __builtin_amdgcn_perm(r.a.v[4 * g + 3], r.a.v[4 * g + 2], k.sel);
    const unsigned p2 = __builtin_amdgcn_perm(r.x[2 * g + 1], r.x[2 * g], k.sel);
    unsigned acc = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, p0), k.w0, 512u << 6, false);
    acc = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, p1), k.w1, acc, false);
    return __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, p2), __builtin_bit_cast(u16x2, k.a0), acc, false);   // (a0: row 2's weights)
}

// How a wave reads the taps of its quads in a tile (wave-uniform, from the quads' classes in the digest): 0 = every quad from its
// 8 bytes x 2 rows, 1 = 8 bytes x 3 rows, 2 = per pixel.  Sets the per-pixel tap state and the quads' base addresses.
template <int PX, int RS, int WT_OFF, int WT1_OFF>
__device__ __forceinline__ int dma_tap_setup(const uint8_t *smem, unsigned lds0, const unsigned *dg, DmaTap tap[PX], unsigned qbase[PX / 4],
                                             unsigned &second, unsigned oslot[PX / 4])
{
    unsigned e[PX];
    unsigned cls = 0;
#pragma unroll
    for (int q = 0; q < PX; q++) e[q] = dg[q];
#pragma unroll
    for (int p = 0; p < PX / 4; p++) {
        const unsigned c = ((e[4 * p] >> 15) & 1u) | ((e[4 * p + 1] >> 14) & 2u);
        cls = c > cls ? c : cls;
        qbase[p] = ((e[4 * p] & 0x1FFCu) - ((e[4 * p] >> 1) & 1u) * (unsigned)RS - (e[4 * p] & 1u) * 4u) | lds0;
        // the quad's slot in the tile (row << 6 | quad column): three bits from the top of each entry, one funnel shift each
        unsigned sl = e[4 * p + 3] >> 29;
        sl = __builtin_amdgcn_alignbit(sl, e[4 * p + 2], 29);
        sl = __builtin_amdgcn_alignbit(sl, e[4 * p + 1], 29);
        oslot[p] = __builtin_amdgcn_alignbit(sl, e[4 * p], 29);
    }
#if defined(SLR_DMA_FORCE_MODE) && SLR_DMA_FORCE_MODE == 10      // timing probe: every wave takes mode 0, all three paths compiled
    const int mode = __builtin_amdgcn_readfirstlane(__ballot(cls == 7u) != 0ull ? 2 : (__ballot(cls == 6u) != 0ull ? 1 : 0));
#elif defined(SLR_DMA_FORCE_MODE)
    const int mode = SLR_DMA_FORCE_MODE == 1 ? __builtin_amdgcn_readfirstlane(__ballot(cls == 2u) != 0ull ? 2 : 1) : SLR_DMA_FORCE_MODE;
#else
    const int mode = __builtin_amdgcn_readfirstlane(__ballot(cls == 2u) != 0ull ? 2 : (__ballot(cls == 1u) != 0ull ? 1 : 0));
#endif
    second = 0;
#pragma unroll
    for (int q = 0; q < PX; q++) {
        const unsigned *wt = reinterpret_cast<const unsigned *>(smem + ((e[q] >> 16) & 0x1FFFu));
        const unsigned sh = (e[q] >> 13) & 3u;
        const unsigned o = mode == 2 ? sh : sh + ((e[q] & 1u) << 2);           // byte offset of the left tap in the 8 bytes read
        tap[q].sel = __umul24(o, 0x10001u) + 0x0C010C00u;
        unsigned w0, w1;
        if constexpr (WT_OFF < 0) { dma_weights((e[q] >> 18) & 0x7FFu, w0, w1); (void)wt; }   // no weight tables in this kernel's LDS
        else { w0 = wt[WT_OFF / 4]; w1 = wt[WT1_OFF / 4]; }
        if (mode == 1) {                                     // weights of rows 0, 1, 2 of the quad (the pixel's own rows: st, st + 1)
            const bool st = ((e[q] >> 1) & 1u) != 0;
            if (__ballot(st) != 0ull) second |= 1u << q;     // (wave-uniform: does ANY lane's pixel q reach into row 2?)
            tap[q].w0 = __builtin_bit_cast(u16x2, st ? 0u : w0);
            tap[q].w1 = __builtin_bit_cast(u16x2, st ? w0 : w1);
            tap[q].a0 = st ? w1 : 0u;
        } else {
            tap[q].a0 = (e[q] & 0x1FFCu) | lds0;
            tap[q].w0 = __builtin_bit_cast(u16x2, w0);
            tap[q].w1 = __builtin_bit_cast(u16x2, w1);
        }
    }
    second = __builtin_amdgcn_readfirstlane(second);          // (what the ballots imply: wave-uniform, the branches on it are scalar)
    return mode;
}

// sd[q] = blended sample of plane image IMG0 minus IMG1 for the thread's PX pixels, by the tile's read mode
template <int PX, unsigned IMG0, unsigned IMG1, unsigned RS>
__device__ __forceinline__ void dma_differences(int mode, const DmaTap tap[PX], const unsigned qbase[PX / 4], unsigned second, int sd[PX])
{
    if (mode == 0) {
        DmaRd r[2];
        dma_rd<IMG0, IMG1, RS>(r[0], qbase[0]);
#pragma unroll
        for (int p = 0; p < PX / 4; p++) {
            if (p + 1 < PX / 4) { dma_rd<IMG0, IMG1, RS>(r[(p + 1) & 1], qbase[p + 1]); dma_rd_wait<8>(r[p & 1]); }
            else dma_rd_wait<0>(r[p & 1]);
#pragma unroll
            for (int i = 0; i < 4; i++)
                sd[4 * p + i] = (int)(dma_blend(r[p & 1], 0, tap[4 * p + i]) >> 16) - (int)(dma_blend(r[p & 1], 1, tap[4 * p + i]) >> 16);
        }
    } else if (mode == 1) {
#pragma unroll
        for (int p = 0; p < PX / 4; p++) {
            DmaRd3 r;
            dma_rd<IMG0, IMG1, RS>(r.a, qbase[p]);
            dma_rd_row2<IMG0, IMG1, RS>(r, qbase[p]);
            dma_rd3_wait(r);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                // a quad that straddles a step of sy does so from some pixel on: the pixels before it never need row 2
                if ((second >> (4 * p + i)) & 1u)
                    sd[4 * p + i] = (int)(dma_blend3(r, 0, tap[4 * p + i]) >> 16) - (int)(dma_blend3(r, 1, tap[4 * p + i]) >> 16);
                else
                    sd[4 * p + i] = (int)(dma_blend(r.a, 0, tap[4 * p + i]) >> 16) - (int)(dma_blend(r.a, 1, tap[4 * p + i]) >> 16);
            }
        }
    } else {
        DmaRd r[2];
        dma_rd<IMG0, IMG1, RS>(r[0], tap[0].a0);
#pragma unroll
        for (int q = 0; q < PX; q++) {
            if (q + 1 < PX) { dma_rd<IMG0, IMG1, RS>(r[(q + 1) & 1], tap[q + 1].a0); dma_rd_wait<8>(r[q & 1]); }
            else dma_rd_wait<0>(r[q & 1]);
            sd[q] = (int)(dma_blend(r[q & 1], 0, tap[q]) >> 16) - (int)(dma_blend(r[q & 1], 1, tap[q]) >> 16);
        }
    }
}

// LDS-DMA operations a wave issues per tile, in program order (the static sequence the counted waits rely on):
//   start of phase p:  [p == 1: PX/4 digest DMAs of the next tile]  2 plane DMAs of phase p + A
// The wait at the top of phase p lets everything issued after the plane DMAs of phase p itself stay in flight.  Only DMAs
// are counted: LDS-DMA and ordinary vector-memory operations are NOT retired in order relative to each other (a count that
// included the output stores let the wait pass with the DMA still pending -- found the hard way), so the stores of a tile
// are issued at the start of the NEXT tile's phase 0, a whole phase before the following wait, and a wave that still has
// one of them outstanding there merely waits a little longer.
constexpr int kDmaDigestPhase = 3;    // the phase of a tile in which the NEXT tile's digest is fetched (the next tile is known from phase 2 on)
template <int PX, int A, int PLANE_DMAS = 2>
__device__ __host__ constexpr int dma_wait_count(int p)
{
    int n = 0;
    for (int j = 1; j < A; j++) { const int ph = ((p - A + j) % 7 + 7) % 7; n += PLANE_DMAS + (ph == kDmaDigestPhase ? PX / 4 : 0); }
    return n;
}

// ---- dynamic tile schedule -----------------------------------------------------------------------------------------------
// Round 2 gave every workgroup a fixed, strided share of its XCD band's tiles.  The per-workgroup timeline (profiles/r03:
// all 768 workgroups start within 1 us, the first finishes after 92 us, the median after 114, the last after 138) showed that
// the kernel's duration was its SLOWEST workgroup's, a fifth above the mean: tiles differ in cost (read modes, borders) and the
// three workgroups of a CU share its SIMDs.  Now a workgroup takes its first tile by its index and every further one from a
// per-(camera, XCD band) ticket counter in global memory.  The ticket is drawn with a SCALAR-memory atomic (s_atomic_add ... glc,
// returned through lgkmcnt): the vector-memory queue, whose counted waits track the LDS-DMA operations, never sees it.  Wave 0
// draws the ticket of the next tile during phase 1 and publishes it through LDS; every wave picks it up behind the barrier of
// phase 2; the next tile's digest and boxes are fetched from phase 3 / 7 - A on.  The counters return to zero in the launch
// that used them: the last workgroup to leave (a second counter) clears them.
// An XCD's pool is not one band of the image but kSubBands of them, dealt round-robin (band g belongs to XCD g % 8) and walked
// in order: the bands at the top and the bottom of the frame are dearer (taller boxes, more three-row waves) than the ones in
// the middle -- with one band per XCD the XCDs finished 112 .. 120 us after the start.  A pool still moves down the image
// through rows that are adjacent in memory (a band is 1 / 32 of the frame), so what its L2 holds of the rows above stays useful.
constexpr int kSubBands = 4;
struct DmaPool {
    int pb;                              // tiles per band
    int per;                             // pool-local indices of a whole pool: kSubBands * pb
    int T, E;                            // tiles; entries (tiles + the extra parts of split tiles, see dma_tiles_kernel)
    int per_x;                           // ... of THIS XCD's pool that are tiles of the image (the last band may be short)
    __device__ __host__ static DmaPool of(int T, int E = 0, int x = 0)
    {
        DmaPool p;
        p.pb = (T + 8 * kSubBands - 1) / (8 * kSubBands); p.per = kSubBands * p.pb; p.T = T; p.E = E > T ? E : T;
        p.per_x = 0;
        for (int k = 0; k < kSubBands; k++) {
            const int left = T - (k * 8 + x) * p.pb;
            p.per_x += left < 0 ? 0 : left > p.pb ? p.pb : left;
        }
        return p;
    }
    // pool-local index l of XCD x -> entry, or -1 behind the pool's last one (and behind every larger l): first the pool's tiles
    // band by band, then its share of the extra entries (dealt round-robin over the XCDs: parts of corner tiles, few)
    __device__ int entry(int x, int l) const
    {
        if (l < per_x) { const int k = l / pb; return (k * 8 + x) * pb + (l - k * pb); }
        const int e = T + (l - per_x) * 8 + x;
        return e < E ? e : -1;
    }
};
constexpr int kSchedStride = 16;                             // one counter per 64-byte line
constexpr int kSchedDone = kDmaMaxJobs * 8 * kSchedStride;   // sched[kSchedDone]: workgroups that have left the kernel
constexpr size_t kSchedBytes = (size_t)(kSchedDone + kSchedStride) * sizeof(unsigned);
size_t dma_sched_bytes() { return kSchedBytes; }

// The next tile's ticket: *p is incremented by a scalar atomic that is issued a phase before its result is needed and NOT waited
// for in between.  Such a result must not live in a register the compiler knows about: to the compiler the asm statement's output
// is complete, so it may copy or spill it at once -- and in the fused Gray decode of 256 x 8 tiles it did copy it, in front of the
// wait, on the path of a wave 0 that decodes nothing of an entry (a part of a split tile): a copy taken before the atomic has
// returned holds the operand (1) instead of the ticket, the workgroup decodes entry first + 1 a second time and the entry the
// ticket named is never decoded (found by poisoning the outputs: whole parts of corner tiles of a verged rig's maps unwritten,
// a different set every run).  So the result goes to a fixed SGPR that the two decode kernels keep out of register allocation
// (amdgpu_num_sgpr(96): s0 .. s89 + VCC / XNACK / flat-scratch accounting), written by dma_ticket_issue and read by dma_ticket_take
// and by nothing else (checked on the disassembly: tests/test_capi_symbols.py).
#define SLR_TICKET_SGPR "s94"
#define SLR_TICKET_KERNEL __attribute__((amdgpu_num_sgpr(96)))
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"            // ("clobber list contains reserved registers": that is the point -- and the
                                                           //  clobber makes the kernel's SGPR count include the register)
__device__ __forceinline__ void dma_ticket_issue(unsigned *p)
{
    asm volatile("s_mov_b32 " SLR_TICKET_SGPR ", 1\n\ts_atomic_add " SLR_TICKET_SGPR ", %0, 0x0 glc" :: "s"(p) : "memory", SLR_TICKET_SGPR);
}
__device__ __forceinline__ unsigned dma_ticket_take()
{
    unsigned v;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_mov_b32 %0, " SLR_TICKET_SGPR : "=s"(v) :: "memory", SLR_TICKET_SGPR);
    return v;
}
#pragma clang diagnostic pop
// a workgroup leaves the kernel (one thread calls this): the last one zeroes the counters for the next launch
__device__ __forceinline__ void dma_sched_leave(unsigned *sched)
{
    unsigned v;                                             // (issued and waited for in one statement: an ordinary value)
    asm volatile("s_mov_b32 %0, 1\n\ts_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=&s"(v) : "s"(sched + kSchedDone) : "memory");
    if (v + 1u == gridDim.x) {
        for (int i = 0; i < kDmaMaxJobs * 8; i++) __hip_atomic_store(sched + i * kSchedStride, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(sched + kSchedDone, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// the wrapped phases' difference as the f32 image of the 2^24-scaled integers (het_pair_q24 with wrapping arithmetic: a
// sentinel operand must not be undefined behaviour, its result is discarded)
__device__ __forceinline__ float dma_pair_q24(int Pa, int Pb)
{
    return (float)(int)(((unsigned)Pa - (unsigned)Pb) + ((Pa > Pb) ? 0u : (unsigned)kQ24TwoPI));
}
// het_finish_q24 with the 2^-24 scaling moved behind the division: every step is the same rounding of the same real
// number (power-of-two scalings are exact, 255 * 2^-24 is a float)
__device__ __forceinline__ float dma_finish_q24(float Fa, float Fb)
{
    constexpr float two_pi_q24 = kTwoPI * 16777216.0f;
    const float F123 = (Fa > Fb) ? (Fa - Fb) : (Fa - Fb + two_pi_q24);
    constexpr float rc = 1.0f / kTwoPI;
    const float q = F123 * rc;
    const float r = __builtin_fmaf(-q, kTwoPI, F123);
    return __builtin_fmaf(r, rc, q) * (255.0f / 16777216.0f);
}
constexpr int kDmaSentinel = 0x7FFFFFFF;        // wrapped phase of the reference's undefined case (n == d == 0), folded mode

// the schedule state of a workgroup (both fused decodes): its pool, the tile after the current one, this thread's box chunk
struct DmaSched {
    unsigned *ctr;                       // the pool's ticket counter
    const int4 *boxes;                   // the camera's box table
    int pitch, W, H, crow, ccol, xcd, first_ticket;
    unsigned tkt_lds;                    // LDS address of the ticket slot
    unsigned wave_bit;                   // 1 << (this wave's index): its bit in an entry's wave mask
    DmaPool pool;
    bool chunk_live;                     // this thread owns a chunk of the plane images
    bool wave0;
    int nxt;                             // the entry after the current one (valid once pick_up() ran in the current tile)
    bool has_next;
    unsigned voff_next;                  // this thread's chunk of that entry's box
    int nxt_ty, nxt_tx;                  // ... its tile
    bool nxt_act;                        // ... and whether this wave decodes anything of it

    // this thread's chunk of entry t's source box: the buffer offset its DMAs fetch from (or the out-of-range offset: zeros);
    // also the entry's tile and whether this wave belongs to the entry's part
    __device__ __forceinline__ unsigned box_voff(int t, int &ty, int &tx, bool &act) const
    {
        // t is wave-uniform: an explicit SCALAR load (hipcc emits a vector load for boxes[t], and waiting for that would drain the
        // vector-memory queue, i.e. the DMAs in flight)
        i32x4 b;
        asm volatile("s_load_dwordx4 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(b) : "s"(boxes + __builtin_amdgcn_readfirstlane(t)) : "memory");
        const int chunks = b.z & 0xFF, rows = b.w & 0xFF;
        tx = (int)((unsigned)b.z >> 8); ty = (int)(((unsigned)b.w >> 8) & 0xFFFFu);
        act = (((unsigned)b.w >> 24) & wave_bit) != 0u;
        const int gx = b.x + 16 * ccol, gy = b.y + crow;
        const bool in = chunk_live && crow < rows && ccol < chunks && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
        return in ? (unsigned)gy * (unsigned)pitch + (unsigned)gx : kDmaInvalid;
    }
    // wave 0, one phase before pick_up(): draw (before the phase's tap loop) and publish (behind it) the next tile's ticket
    __device__ __forceinline__ void draw() const { if (wave0) dma_ticket_issue(ctr); }
    __device__ __forceinline__ void publish() const
    {
        if (!wave0) return;
        const unsigned ticket = dma_ticket_take();
        // every lane writes the same word; the barriers of these kernels are bare s_barrier, so the write must have landed before
        // the next one (explicit DS operations: a generic-pointer access would be a FLAT instruction and drain the vector-memory queue)
        asm volatile("ds_write_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" :: "v"(tkt_lds), "v"(ticket) : "memory");
    }
    // every wave, behind the barrier that follows publish(): the ticket -> the next tile
    __device__ __forceinline__ void pick_up()
    {
        unsigned tk;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(tk) : "v"(tkt_lds) : "memory");
        const int nl = first_ticket + (int)__builtin_amdgcn_readfirstlane(tk);
        const int ne = pool.entry(xcd, nl);
        has_next = ne >= 0;
        nxt = has_next ? ne : 0;
        voff_next = box_voff(nxt, nxt_ty, nxt_tx, nxt_act);
        if (!has_next) { voff_next = kDmaInvalid; nxt_act = false; }
    }
};

// X87: SLR_OPT_EVAL_MODEL = 1 -- the heterodyne tail as the reference's MSVC2010 x87 binary rounds it (het_finish_x87,
// decode_common.hpp; the launcher hands over the x87 variant of the tables): the one place where the two models' decodes differ
template <int TW, int TH, int NT, int A, bool HASVALID, bool X87 = false>
struct DmaDecode {
    typedef DmaGeom<TW, TH, NT> Gm;
    static constexpr int PX = Gm::PX, PS = Gm::PS, RS = Gm::RS, D = A + 1;
    // SPLIT: the workgroup has twice as many threads as a plane image has chunks (128 x 16 tiles on 512 threads): waves 0-3 fetch
    // the first plane of a phase's pair, waves 4-7 the second -- ONE plane DMA per wave and phase instead of two for half the waves
    static constexpr bool SPLIT = 2 * Gm::NCH <= NT;
    static constexpr int PLANE_DMAS = SPLIT ? 1 : 2;
    // dynamic LDS (the kernel has no static LDS, so it starts at LDS address 0 and the tap addresses are 13-bit ORs):
    // D buffers of 2 plane images | digest of the tile | weight tables | decode tables (slr_create's, kLutWords).
    // Waves beyond a plane image's chunks (128 x 16 tiles: waves 4-7) issue no plane DMAs and wait for none -- a wave's counted
    // waits concern its own DMAs only, the barrier behind them covers everybody else's -- so there is no scratch slot for dummy
    // transfers: with it the triple buffer of DMA depth 2 plus the 8 KB weight tables would miss three workgroups per CU by 1 KB.
    static constexpr int DIG_OFF = D * 2 * PS, DIG_BYTES = TW * TH * 4;
    static constexpr int WT_OFF = DIG_OFF + DIG_BYTES, WT1_OFF = WT_OFF + 1026 * 4, WT_BYTES = 2 * 1026 * 4;   // two tables: w0[1025], w1[1025]
    static constexpr int TAP_WT = WT_OFF;
    static constexpr int LUT_OFF = WT_OFF + WT_BYTES;
    static constexpr int TKT_OFF = LUT_OFF + (kLutWords + 1) * 4;        // the next tile's ticket (wave 0 -> everybody)
    static constexpr int LDS_BYTES = TKT_OFF + 12;
    static_assert(WT_OFF <= 65536 && LDS_BYTES <= 160 * 1024, "DMA destinations are 16-bit LDS addresses (M0)");
    static constexpr int kSentinel = kDmaSentinel;

    const uint8_t *smem;
    const float *lut;
    unsigned lds0, wave_off;
    unsigned pslot_off;                  // this wave's 1 KiB slot inside a plane image
    unsigned plane_g;                    // SPLIT: which plane of a phase's pair this wave fetches (wave-uniform)
    bool plane_wave;                     // this wave owns chunks of the plane images (wave-uniform)
    __amdgpu_buffer_rsrc_t rs_stack, rs_dig, rs_phase, rs_valid;
    unsigned pstride, fringe_skip;
    int W, H, black_thr;
    DmaSched sc;                         // schedule (see "dynamic tile schedule")
    // per-tile state
    DmaTap tap[PX];
    unsigned qbase[PX / 4];              // LDS address of the 8-byte window of each quad (read modes 0 and 1)
    unsigned oslot[PX / 4];              // where the quad's results go: tile row << 6 | quad column (the digest's slot field); kept until the flush
    unsigned second;                     // bit q (wave-uniform): some lane's pixel q blends rows 1, 2 of its quad's three (read mode 1)
    int mode;                            // the wave's read mode for this tile (dma_tap_setup)
    // validity.  HASVALID: bit q of ok = pixel q passed the shadow mask and every (n | d) != 0 so far, bit 16 + q = the mask
    // alone.  Folded mode (the flag travels in the phase as a NaN): pm[q] = max over "masked out ? sentinel : INT_MIN" and
    // the wrapped phases so far -- the table returns the sentinel for the undefined case -- so one compare decides at the end.
    unsigned ok;
    int pm[PX];
    int dd[PX];                          // d of the frequency in flight
    int Pk[PX];                          // wrapped phase kept for the next heterodyne step
    float F12[PX];
    // results of the previous tile, stored at the start of the next phase 0 (see dma_wait_count)
    float outv[PX];
    unsigned out_ok;
    int out_ty, out_tx;
    bool out_pending;

    __device__ __forceinline__ void flush() const
    {
#pragma unroll
        for (int p = 0; p < PX / 4; p++) {
            const int row = out_ty * TH + (int)(oslot[p] >> 6), col = out_tx * TW + (int)(oslot[p] & 63u) * 4;
            const bool inb = row < H && col < W;                     // W % 16 == 0: a quad is whole inside or outside
#if defined(SLR_DMA_ABL) && (SLR_DMA_ABL == 7 || SLR_DMA_ABL == 9)
            // ablation (round 5): the phases go to a 1 MB window that stays in the XCD's L2 -- what a fused decode -> match launch
            // with a per-XCD phase ring could save on the decode side at best (wrong results, timing only)
            const unsigned m = ((unsigned)row * (unsigned)W + (unsigned)col) & 0x3FFFFu;
#else
            const unsigned m = (unsigned)row * (unsigned)W + (unsigned)col;
#endif
            if constexpr (HASVALID) {
                const unsigned ok4 = (out_ok >> (4 * p)) & 0xFu;
                __builtin_amdgcn_raw_buffer_store_b32(__umul24(ok4, 0x204081u) & 0x01010101u, rs_valid, inb ? m : kDmaInvalid, 0, 0);
            }
            typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
            const u32x4_t v = {__builtin_bit_cast(unsigned, outv[4 * p]), __builtin_bit_cast(unsigned, outv[4 * p + 1]),
                               __builtin_bit_cast(unsigned, outv[4 * p + 2]), __builtin_bit_cast(unsigned, outv[4 * p + 3])};
            __builtin_amdgcn_raw_buffer_store_b128(v, rs_phase, inb ? m * 4u : kDmaInvalid, 0, 2);
        }
    }

    __device__ __forceinline__ void issue_planes(int p, int buf, unsigned voff) const
    {
#if defined(SLR_DMA_ABL) && (SLR_DMA_ABL == 1 || SLR_DMA_ABL == 6)
        voff = kDmaInvalid;              // ablation: no source traffic (the DMA instructions still issue and zero-fill)
#endif
        if (!plane_wave) return;         // (wave-uniform)
        if constexpr (SPLIT) {
            const unsigned plane = (unsigned)dma_phase_plane(p, 0) + (p == 0 ? plane_g : 2u * plane_g);
            dma16(voff, rs_stack, lds0 + (unsigned)(buf * 2 * PS) + plane_g * (unsigned)PS + pslot_off, plane * pstride + (p == 0 ? 0u : fringe_skip));
        } else {
#pragma unroll
            for (int g = 0; g < 2; g++)
                dma16(voff, rs_stack, lds0 + (unsigned)((buf * 2 + g) * PS) + pslot_off,
                      (unsigned)dma_phase_plane(p, g) * pstride + (p == 0 ? 0u : fringe_skip));
        }
    }
    __device__ __forceinline__ void issue_digest(unsigned tile, bool live) const
    {
#if defined(SLR_DMA_ABL) && (SLR_DMA_ABL == 8 || SLR_DMA_ABL == 9)
        tile = (tile & 63u) * 94u;       // ablation (round 5): 64 tiles' digests, hot in L2 = a map stream of zero bytes (timing only)
#endif
#pragma unroll
        for (int r = 0; r < PX / 4; r++) {
            const unsigned voff = live ? tile * (unsigned)DIG_BYTES + (threadIdx.x + (unsigned)(NT * r)) * 16u : kDmaInvalid;
            dma16(voff, rs_dig, lds0 + (unsigned)(DIG_OFF + r * NT * 16) + wave_off, 0u);
        }
    }

    static __device__ __forceinline__ float pair_q24(int Pa, int Pb) { return dma_pair_q24(Pa, Pb); }
    static __device__ __forceinline__ float finish_q24(float Fa, float Fb)
    {
        if constexpr (X87) return het_finish_x87(Fa, Fb);
        else return dma_finish_q24(Fa, Fb);
    }

    // phase P of a tile whose phase 0 uses LDS buffer K0.  voff_cur: this thread's chunk of the current tile's box.
    template <int K0, int P>
    __device__ __forceinline__ void phase(int ty, int tx, unsigned voff_cur)
    {
        static_assert(7 - A > 2 && kDmaDigestPhase > 2 && kDmaDigestPhase < 7, "the next tile is known from phase 2 on");
        if (plane_wave) wait_vm<dma_wait_count<PX, A, PLANE_DMAS>(P)>();
        else if (P == 0) wait_vm<0>();   // (its share of the tile's digest, issued four phases ago)
#if !defined(SLR_DMA_ABL) || SLR_DMA_ABL != 3
        asm volatile("s_barrier" ::: "memory");
#endif
        if (P == 1) sc.draw();                                        // (scalar memory: invisible to the counted vmcnt waits)
        if constexpr (P == 2) sc.pick_up();                           // the ticket wave 0 drew in phase 1 -> the next tile
        if (P == kDmaDigestPhase) issue_digest((unsigned)sc.nxt, sc.has_next);
        issue_planes((P + A) % 7, (K0 + P + A) % D, P + A >= 7 ? sc.voff_next : voff_cur);
        if constexpr (P == 0) {
            if (out_pending) flush();
            // tap state of the tile's pixels from the digest, and the wave's read mode for the tile
            const unsigned *dg = reinterpret_cast<const unsigned *>(smem + DIG_OFF + threadIdx.x * (PX * 4));
            mode = dma_tap_setup<PX, RS, TAP_WT, WT1_OFF>(smem, lds0, dg, tap, qbase, second, oslot);
            ok = 0;
        }
        constexpr unsigned img0 = (unsigned)(((K0 + P) % D) * 2 * PS), img1 = img0 + PS;
        // sd[q] = sample of the phase's first plane minus its second plane (white - black, G1 - G3, G2 - G4), reads one pixel ahead
        int sd[PX];
#if defined(SLR_DMA_ABL) && (SLR_DMA_ABL == 2 || SLR_DMA_ABL == 6)
#pragma unroll
        for (int q = 0; q < PX; q++) {   // ablation: no LDS tap reads (the blend runs on register junk)
            DmaRd r;
#pragma unroll
            for (int i = 0; i < 8; i++) r.v[i] = tap[q].a0 * (unsigned)(i + 3 + P);
            sd[q] = (int)(dma_blend(r, 0, tap[q]) >> 16) - (int)(dma_blend(r, 1, tap[q]) >> 16);
        }
#elif defined(SLR_DMA_ABL) && SLR_DMA_ABL == 4
#pragma unroll
        for (int q = 0; q < PX; q++) sd[q] = (int)tap[q].a0 + P;   // ablation: neither reads nor blend
#elif defined(SLR_DMA_ABL) && SLR_DMA_ABL == 5
        DmaRd r[2];
        dma_rd<img0, img1, (unsigned)RS>(r[0], tap[0].a0);
#pragma unroll
        for (int q = 0; q < PX; q++) {   // ablation: the reads, but no blend
            if (q + 1 < PX) { dma_rd<img0, img1, (unsigned)RS>(r[(q + 1) & 1], tap[q + 1].a0); dma_rd_wait<8>(r[q & 1]); }
            else dma_rd_wait<0>(r[q & 1]);
            sd[q] = (int)(r[q & 1].v[0] ^ r[q & 1].v[1] ^ r[q & 1].v[2] ^ r[q & 1].v[3] ^ r[q & 1].v[4] ^ r[q & 1].v[5] ^ r[q & 1].v[6] ^ r[q & 1].v[7]) & 255;
        }
#else
        // A wave in its tap loop outranks the waves that are not: without it the SIMD's waves advance through the loop in
        // lockstep (all reading, then all blending); with it one runs ahead and the LDS and VALU work of different waves
        // overlap (157 -> 150 us per pair launch; priority by wave or by workgroup, or only around the read issue: 154-156)
        __builtin_amdgcn_s_setprio(1);
        // (the (G2, G4) phases take the planes in the other order: n = G4 - G2 comes out of the subtraction itself)
        if constexpr (P >= 2 && (P & 1) == 0) dma_differences<PX, img1, img0, (unsigned)RS>(mode, tap, qbase, second, sd);
        else dma_differences<PX, img0, img1, (unsigned)RS>(mode, tap, qbase, second, sd);
        __builtin_amdgcn_s_setprio(0);
#endif
        if (P == 1) sc.publish();                                     // (the ticket has had the tap loop to arrive)
        if constexpr (P == 0) {
#pragma unroll
            for (int q = 0; q < PX; q++) {                         // computeShadows :198-204
                if constexpr (HASVALID) ok |= (sd[q] > black_thr ? 0x10001u : 0u) << q;
                else pm[q] = sd[q] > black_thr ? (int)0x80000000 : kSentinel;
            }
        } else if constexpr ((P & 1) != 0) {
#pragma unroll
            for (int q = 0; q < PX; q++) dd[q] = sd[q];            // d = G1 - G3
        } else {
#pragma unroll
            for (int q = 0; q < PX; q++) {
                int nz;
                const int Pw = wrapped_nd_q24(sd[q], dd[q], lut, nz);    // sd = n = G4 - G2 (see above), dd = d = G1 - G3
                if constexpr (HASVALID) { if (nz == 0) ok &= ~(1u << q); }  // Q5 rule: an undefined P makes the pixel invalid
                if constexpr (P == 2) { Pk[q] = Pw; if constexpr (!HASVALID) pm[q] = pm[q] > Pw ? pm[q] : Pw; }
                else if constexpr (P == 4) { F12[q] = pair_q24(Pk[q], Pw); Pk[q] = Pw; if constexpr (!HASVALID) pm[q] = pm[q] > Pw ? pm[q] : Pw; }
                else {
                    const float ph = finish_q24(F12[q], pair_q24(Pk[q], Pw));
                    if constexpr (HASVALID) {
                        // as mf_pixel_sh: the mask alone decides between the computed phase and 0
                        outv[q] = ((ok >> (q + 16)) & 1u) ? ph : 0.0f;
                    } else {
                        const int m3 = pm[q] > Pw ? pm[q] : Pw;
                        outv[q] = m3 != kSentinel ? ph : kInvalidPhase;    // folded flag
                    }
                }
            }
            if constexpr (P == 6) { out_ok = ok; out_ty = ty; out_tx = tx; out_pending = true; }
        }
    }

    // phase P of an entry this wave decodes nothing of (the entry is a PART of a tile -- see dma_tiles_kernel -- and the wave is
    // not among its waves): the barriers, its share of the DMAs, wave 0's ticket; the previous tile's outputs leave as usual
    template <int K0, int P>
    __device__ __forceinline__ void idle_phase(unsigned voff_cur)
    {
        if (plane_wave) wait_vm<dma_wait_count<PX, A, PLANE_DMAS>(P)>();
        else if (P == 0) wait_vm<0>();
        asm volatile("s_barrier" ::: "memory");
        if (P == 1) sc.draw();
        if constexpr (P == 2) sc.pick_up();
        if (P == kDmaDigestPhase) issue_digest((unsigned)sc.nxt, sc.has_next);
        issue_planes((P + A) % 7, (K0 + P + A) % D, P + A >= 7 ? sc.voff_next : voff_cur);
        if constexpr (P == 0) { if (out_pending) flush(); out_pending = false; }
        if (P == 1) sc.publish();
    }

    template <int K0>
    __device__ __forceinline__ void tile(int ty, int tx, bool act, unsigned voff_cur)
    {
        if (act) {                       // (wave-uniform; every wave of a whole tile)
            phase<K0, 0>(ty, tx, voff_cur);
            phase<K0, 1>(ty, tx, voff_cur);
            phase<K0, 2>(ty, tx, voff_cur);
            phase<K0, 3>(ty, tx, voff_cur);
            phase<K0, 4>(ty, tx, voff_cur);
            phase<K0, 5>(ty, tx, voff_cur);
            phase<K0, 6>(ty, tx, voff_cur);
        } else {
            idle_phase<K0, 0>(voff_cur);
            idle_phase<K0, 1>(voff_cur);
            idle_phase<K0, 2>(voff_cur);
            idle_phase<K0, 3>(voff_cur);
            idle_phase<K0, 4>(voff_cur);
            idle_phase<K0, 5>(voff_cur);
            idle_phase<K0, 6>(voff_cur);
        }
    }
};

// resident workgroups per CU that the LDS allows -> waves per SIMD the register allocation should aim for
template <int LDS_BYTES, int NT>
constexpr int dma_waves_per_simd()
{
    constexpr int wgs = 160 * 1024 / LDS_BYTES, w = wgs * NT / 256;
    return w > 8 ? 8 : w < 1 ? 1 : w;
}

// (a thread's state does not fit the 64 VGPRs of 8 waves per SIMD without spilling: capped at 6 like the Gray kernel; with separate
// valid bytes -- the single-camera entry -- it does not fit the 80 of 6 either: 9 spilled registers whose reloads drain the DMA
// queue, so that variant runs 4 waves per SIMD on 96 registers, 78.5 -> 75.9 us per camera)
template <int LDS_BYTES, int NT>
constexpr int mf_dma_waves() { return dma_waves_per_simd<LDS_BYTES, NT>() > 6 ? 6 : dma_waves_per_simd<LDS_BYTES, NT>(); }

template <int TW, int TH, int NT, int A, bool HASVALID, bool X87 = false>
__global__ __launch_bounds__(NT, (HASVALID ? 4 : mf_dma_waves<DmaDecode<TW, TH, NT, A, HASVALID>::LDS_BYTES, NT>())) SLR_TICKET_KERNEL
void mf_rect_decode_dma_kernel(DmaJobs jobs, int njobs, int pitch, int W, int H, int black_thr, const float *__restrict__ lut_g,
                               int tiles_x, int tiles_y, unsigned *__restrict__ sched)
{
    typedef DmaDecode<TW, TH, NT, A, HASVALID, X87> Dec;
    typedef DmaGeom<TW, TH, NT> Gm;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    Dec d;
    d.smem = smem;
    d.lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)smem);
    if (d.lds0 & 0x1FFFu) __builtin_trap();                 // the tap addresses OR the buffer base in (no static LDS here: 0)
#if defined(SLR_DMA_CLOCKPROBE)
    const unsigned long long probe_entry = wall_clock64();
#endif
    float *lut = reinterpret_cast<float *>(smem + Dec::LUT_OFF);
    d.lut = lut;
    for (int i = threadIdx.x; i < kLutWords; i += NT) lut[i] = lut_g[i];
    if constexpr (Dec::WT_BYTES > 0)
    for (unsigned i = threadIdx.x; i < 1025u; i += NT) {
        unsigned w0, w1;
        dma_weights(i, w0, w1);
        *reinterpret_cast<unsigned *>(smem + Dec::WT_OFF + 4u * i) = w0;
        *reinterpret_cast<unsigned *>(smem + Dec::WT1_OFF + 4u * i) = w1;
    }
    __syncthreads();
    if (!HASVALID && threadIdx.x == 0)                      // the undefined wrapped phase (n == d == 0: lutR -> S = 9, s = 0, sgn 0)
        reinterpret_cast<int *>(lut)[kLutP + (9 << 8)] = Dec::kSentinel;
    __syncthreads();
    // workgroup -> (XCD band, camera, index in the pool): the hardware deals workgroups round-robin to the XCDs, so bits 0-2 are the
    // band; the cameras ALTERNATE over the workgroups of a band.  (Round 2 gave the first half of the grid to camera 0.  The
    // workgroups of a persistent kernel keep the age they were launched with and the SIMDs favour older waves: the first third
    // of the grid runs its tiles in ~6 us, the last third in ~8 us -- so camera 0's tiles were gone after 100 us and camera 1's
    // after 130.  With the cameras interleaved and the tiles drawn by ticket, both pools are served by the same mix.)
    const unsigned nblk = gridDim.x / (unsigned)njobs;      // workgroups per camera (a multiple of 8)
    const int xcd = (int)(blockIdx.x & 7u), nbx = (int)(nblk >> 3);
    const int ji = (int)((blockIdx.x >> 3) % (unsigned)njobs), lb = (int)((blockIdx.x >> 3) / (unsigned)njobs);
    const int T = tiles_x * tiles_y;
    d.sc.pool = DmaPool::of(T, (int)jobs.j[ji].entries, xcd);
    int cur = d.sc.pool.entry(xcd, lb);
    if (cur < 0) {                                          // (whole workgroup) nothing to do
        if (threadIdx.x == 0) dma_sched_leave(sched);
        return;
    }

    d.wave_off = __builtin_amdgcn_readfirstlane((threadIdx.x >> 6) * 1024u);
    {
        const unsigned wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        constexpr unsigned wpp = (unsigned)Gm::NCH / 64u;    // waves per plane image
        d.plane_g = Dec::SPLIT ? wv / wpp : 0u;
        d.pslot_off = (Dec::SPLIT ? wv % wpp : wv) * 1024u;
        d.plane_wave = Dec::SPLIT ? wv < 2u * wpp : wv < wpp;
    }
    d.W = W; d.H = H; d.black_thr = black_thr;
    d.sc.pitch = pitch; d.sc.W = W; d.sc.H = H;
    // this thread's chunk of a box: row crow, 16-byte column ccol
    const int chunk = Dec::SPLIT ? (int)threadIdx.x % Gm::NCH : (int)threadIdx.x;
    d.sc.crow = chunk / Gm::CMAX; d.sc.ccol = chunk - d.sc.crow * Gm::CMAX;
    d.sc.chunk_live = Dec::SPLIT || chunk < Gm::NCH;
    d.sc.xcd = xcd; d.sc.first_ticket = nbx;
    d.sc.wave0 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) == 0u;
    d.sc.wave_bit = 1u << __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    d.sc.tkt_lds = d.lds0 + (unsigned)Dec::TKT_OFF;

#if defined(SLR_DMA_CLOCKPROBE)      // experiment: every workgroup's life on the 100 MHz constant clock (+ shader cycles) -> phase buffer of job 0
    const unsigned long long probe_c0 = clock64(), probe_w0 = wall_clock64();
#endif
    // the schedule: the first tile by index, the others by ticket (tickets count from the band's nbx-th tile)
    d.pstride = jobs.j[ji].pstride; d.fringe_skip = jobs.j[ji].fringe_skip;
    d.rs_stack = __builtin_amdgcn_make_buffer_rsrc((void *)jobs.j[ji].base, 0, (int)jobs.j[ji].stack_bytes, 0x00020000);
    d.rs_dig = __builtin_amdgcn_make_buffer_rsrc((void *)jobs.j[ji].digest, 0, (int)jobs.j[ji].digest_bytes, 0x00020000);
    d.rs_phase = __builtin_amdgcn_make_buffer_rsrc((void *)jobs.j[ji].phase, 0, (int)((unsigned)W * (unsigned)H * 4u), 0x00020000);
    d.rs_valid = __builtin_amdgcn_make_buffer_rsrc((void *)jobs.j[ji].valid, 0, HASVALID ? (int)((unsigned)W * (unsigned)H) : 0, 0x00020000);
    d.sc.boxes = jobs.j[ji].boxes;
    d.sc.ctr = sched + (ji * 8 + xcd) * kSchedStride;
    d.sc.nxt = 0; d.sc.has_next = false; d.sc.voff_next = kDmaInvalid; d.sc.nxt_ty = d.sc.nxt_tx = 0; d.sc.nxt_act = false;
#if defined(SLR_DMA_CLOCKPROBE)
    int it = 0;
#define SLR_DMA_COUNT(x) ((x)++)
#else
#define SLR_DMA_COUNT(x) ((void)0)
#endif
    int cur_ty, cur_tx;
    bool cur_act;
    unsigned voff_cur = d.sc.box_voff(cur, cur_ty, cur_tx, cur_act);
    // prologue = what the last phases of a previous tile would have issued: digest, plane DMAs of phases 0 .. A-1
    d.out_pending = false;
    d.ok = 0;
    d.issue_digest((unsigned)cur, true);
#pragma unroll
    for (int a = 0; a < A; a++) d.issue_planes(a, a % Dec::D, voff_cur);

    for (;;) {
        // the phase-0 buffer index advances by 7 mod D from tile to tile: D tiles per round of this loop
#define SLR_DMA_TILE(K0)                                                                                       \
        {                                                                                                      \
            d.template tile<K0>(cur_ty, cur_tx, cur_act, voff_cur);                                            \
            SLR_DMA_COUNT(it);                                                                                 \
            if (!d.sc.has_next) break;                                                                         \
            cur = d.sc.nxt; voff_cur = d.sc.voff_next;                                                         \
            cur_ty = d.sc.nxt_ty; cur_tx = d.sc.nxt_tx; cur_act = d.sc.nxt_act;                                \
        }
        SLR_DMA_TILE(0)
        SLR_DMA_TILE(7 % Dec::D)
        if constexpr (Dec::D >= 3) SLR_DMA_TILE(14 % Dec::D)
        if constexpr (Dec::D >= 4) SLR_DMA_TILE(21 % Dec::D)
#undef SLR_DMA_TILE
    }
#undef SLR_DMA_COUNT
    // (only what a tile left behind: a wave that was idle in every entry its workgroup decoded -- parts of split tiles as a workgroup's
    //  first and only entries, i.e. small images on the full resident set -- has no results, and its out_ty / oslot / values are
    //  whatever the registers held: stores of junk to junk addresses inside the image, found with poisoned outputs at 1040 x 524)
    if (d.out_pending) d.flush();
    wait_vm<0>();                                           // the dummy DMAs behind the last tile must land before the LDS is released
    if (threadIdx.x == 0) dma_sched_leave(sched);
#if defined(SLR_DMA_CLOCKPROBE)
    if (threadIdx.x == 0) {
        unsigned long long *o = reinterpret_cast<unsigned long long *>(jobs.j[0].phase) + 4 * blockIdx.x;
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        o[0] = probe_entry; o[1] = wall_clock64(); (void)probe_w0; o[2] = clock64() - probe_c0; o[3] = (unsigned long long)it << 32 | (unsigned)ji << 16 | (xcc & 0xFu) << 8 | (unsigned)xcd;
    }
#endif
}

// ------------------------------------------------------------------------------------------------------
// Fix-up pass.  A tile whose source box is larger than the form's LDS image (strong keystone in the image corners: the
// maps of a verged rig tilt the rows there) used to send the WHOLE frame back to round 1's forms -- 1.6 % of the tiles of a
// rig verged by 0.2 rad cost 215 instead of ~140 us.  Now the tile tables list those tiles, the LDS-DMA kernel fetches nothing
// for them (and writes junk), and these kernels rewrite them behind it on the same stream: one workgroup per listed tile,
// every pixel gathered straight from global memory (same make_tap / sample / decode as the direct-gather form).
// ------------------------------------------------------------------------------------------------------
template <int TW, int TH, bool X87 = false>
__global__ __launch_bounds__(256) void mf_rect_fixup_kernel(MfPlanes pl, int pitch, int W, int H, int black_thr,
                                                            const float *__restrict__ lut_g, const int16_t *__restrict__ map_xy,
                                                            const uint16_t *__restrict__ map_frac, const unsigned *__restrict__ list,
                                                            unsigned n, int tiles_x, float *__restrict__ phase, uint8_t *__restrict__ valid)
{
    __shared__ float lut[kLutWords + 1];
    load_lut(lut, lut_g);
    // one pixel per thread, TW * TH / 256 workgroups per listed tile: a tile's pixels are gathered side by side, not one after the
    // other (one workgroup per tile took longer than the whole main kernel: every pixel is 56 dependent-latency byte loads)
    constexpr unsigned kParts = TW * TH / 256;
    for (unsigned i = blockIdx.x; i < n * kParts; i += gridDim.x) {
        const int t = (int)list[i / kParts], ty = t / tiles_x, tx = t - ty * tiles_x;
        {
            const int p = (int)((i % kParts) * 256u + threadIdx.x);
            const int row = ty * TH + p / TW, col = tx * TW + p % TW;
            if (row >= H || col >= W) continue;
            const size_t m = (size_t)row * W + col;
            const Tap tap = make_tap(map_xy[2 * m], map_xy[2 * m + 1], map_frac[m], pitch, W, H);
            int g[SLR_MF_PLANES];
#pragma unroll
            for (int k = 0; k < SLR_MF_PLANES; k++) g[k] = sample(pl.p[k], pitch, W, H, tap);
            int v;
            const float ph = mf_pixel<X87>(g, black_thr, lut, v);
            phase[m] = valid || v ? ph : kInvalidPhase;
            if (valid) valid[m] = (uint8_t)v;
        }
    }
}

template <int TW, int TH>
__global__ __launch_bounds__(256) void gray_rect_fixup_kernel(GrayPlanes pl, int ncol, int nrow, int pitch, int W, int H, int black_thr,
                                                              int white_thr, int scan_w, int scan_h, const int16_t *__restrict__ map_xy,
                                                              const uint16_t *__restrict__ map_frac, const unsigned *__restrict__ list,
                                                              unsigned n, int tiles_x, int32_t *__restrict__ code_x,
                                                              int32_t *__restrict__ code_y, uint8_t *__restrict__ valid)
{
    constexpr unsigned kParts = TW * TH / 256;              // (see mf_rect_fixup_kernel)
    for (unsigned i = blockIdx.x; i < n * kParts; i += gridDim.x) {
        const int t = (int)list[i / kParts], ty = t / tiles_x, tx = t - ty * tiles_x;
        {
            const int p = (int)((i % kParts) * 256u + threadIdx.x);
            const int row = ty * TH + p / TW, col = tx * TW + p % TW;
            if (row >= H || col >= W) continue;
            const size_t m = (size_t)row * W + col;
            const Tap tap = make_tap(map_xy[2 * m], map_xy[2 * m + 1], map_frac[m], pitch, W, H);
            const int mask = sample(pl.p[0], pitch, W, H, tap) - sample(pl.p[1], pitch, W, H, tap) > black_thr ? 1 : 0;   // reconstruct.cpp:218-224
            int gx = 0, gy = 0, err = 0;
            for (int c = 0; c < ncol + nrow; c++) {                          // reconstruct.cpp:387-400 / 349-360
                const int df = sample(pl.p[2 * c + 2], pitch, W, H, tap) - sample(pl.p[2 * c + 3], pitch, W, H, tap);
                err |= ((df < 0 ? -df : df) < white_thr) ? 1 : 0;
                if (c < ncol) gx = (gx << 1) | (df > 0 ? 1 : 0);
                else gy = (gy << 1) | (df > 0 ? 1 : 0);
            }
            gx ^= gx >> 1; gx ^= gx >> 2; gx ^= gx >> 4; gx ^= gx >> 8;     // grayToDec (graycodes.cpp:116-128)
            gy ^= gy >> 1; gy ^= gy >> 2; gy ^= gy >> 4; gy ^= gy >> 8;
            if (nrow > 0) err |= (gy > scan_h || gx > scan_w) ? 1 : 0;       // reconstruct.cpp:364 (Q9 '>')
            else err |= (gx > scan_w) ? 1 : 0;                               // reconstruct.cpp:403
            const int ok = mask & (err ^ 1);
            code_x[m] = ok ? gx : -1;
            if (code_y) code_y[m] = (ok && nrow > 0) ? gy : -1;
            if (valid) valid[m] = (uint8_t)ok;
        }
    }
}

static const unsigned *dma_nofit_list(const void *tiles, int W, int H, int shape)
{
    return reinterpret_cast<const unsigned *>(reinterpret_cast<const char *>(tiles) + dma_list_offset(W, H, shape));
}

// the stack layout this form needs: 14 planes equally spaced in one allocation (buffer addressing: plane = scalar offset) -- or
// white, black and, a whole number of plane strides further on, the 12 fringes (a hybrid stack with its Gray planes in between) --
// 16-byte aligned rows, W a multiple of 16 (a chunk is completely inside or completely outside the image)
static bool dma_job(const MfPlanes &pl, int pitch, int W, int H, float *phase, uint8_t *valid, const void *tiles, int shape, DmaJob &j)
{
    if (!tiles || W % 16 != 0 || pitch % 16 != 0 || ((uintptr_t)pl.p[0] % 16) != 0 || ((uintptr_t)phase % 4) != 0) return false;
    const long long st = (long long)(pl.p[1] - pl.p[0]);
    if (st < (long long)H * pitch || st % 16 != 0) return false;
    const long long skip = (long long)(pl.p[2] - pl.p[0]) - 2 * st;
    if (skip < 0 || skip % st != 0) return false;
    for (int i = 2; i < SLR_MF_PLANES; i++) if ((long long)(pl.p[i] - pl.p[0]) != st * i + skip) return false;
    const long long bytes = st * (SLR_MF_PLANES - 1) + skip + (long long)H * pitch;
    if (bytes >= (1ll << 31) || (long long)W * H * 4 >= (1ll << 31)) return false;
    const char *b = reinterpret_cast<const char *>(tiles);
    j.base = pl.p[0]; j.pstride = (unsigned)st; j.fringe_skip = (unsigned)skip; j.stack_bytes = (unsigned)bytes;
    j.boxes = reinterpret_cast<const int4 *>(b);
    j.digest = reinterpret_cast<const unsigned *>(b + dma_digest_offset(W, H, shape));
    const size_t dg = dma_entry_capacity(W, H, shape) * (size_t)(dma_shape_tw(shape) * dma_shape_th(shape)) * 4;
    if (dg >= (1ull << 31)) return false;
    j.digest_bytes = (unsigned)dg;
    j.entries = (unsigned)dma_tile_count(W, H, shape);      // (+ the extra parts: the launcher adds them)
    j.phase = phase; j.valid = valid;
    return true;
}

template <int TW, int TH, int NT, int A, bool HV, bool X87 = false>
static hipError_t launch_dma_variant(const DmaJobs &j, int njobs, int pitch, int W, int H, int black_thr, const float *lut, unsigned *sched,
                                     hipStream_t s)
{
    typedef DmaDecode<TW, TH, NT, A, HV, X87> Dec;
    auto kern = mf_rect_decode_dma_kernel<TW, TH, NT, A, HV, X87>;
    static DevSlots resident;                             // resident workgroups of this kernel, per device
    int dev = 0;
    (void)hipGetDevice(&dev);
    int res = resident.get(dev);
    if (!res) {
        hipError_t e0 = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, Dec::LDS_BYTES);
        if (e0 != hipSuccess) return e0;
        int per_cu = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, NT, Dec::LDS_BYTES) != hipSuccess || per_cu < 1) per_cu = 1;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
        res = per_cu * cus;
        resident.put(dev, res);
    }
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    const int T = tiles_x * tiles_y, per = DmaPool::of(T).per;
    const int r = (tl_debug.rect_resident > 0 ? tl_debug.rect_resident : res) / njobs;   // tests: few workgroups -> many tiles each
    int nbx = r / 8 < per ? r / 8 : per;
    if (nbx < 1) nbx = 1;
    // The tile tickets start from zero.  The last workgroup of a launch clears them, which is enough as long as every launch runs to
    // its end; a launch that was aborted (or a debugger detaching) would leave them behind and the next launch would silently skip
    // or repeat tiles -- 2 KB of memset in stream order in front of every launch rules that out (ADVICE r3 / VERDICT r4 item 11).
    hipError_t e = hipMemsetAsync(sched, 0, kSchedBytes, s);
    if (e != hipSuccess) return e;
    SLR_LAUNCH(kern, dim3(8u * (unsigned)nbx * (unsigned)njobs), dim3(NT), Dec::LDS_BYTES, s, j, njobs, pitch, W, H, black_thr, lut,
               tiles_x, tiles_y, sched);
    return hipGetLastError();
}

#define SLR_DMA_DEPTH2(TW, TH, NT) 2   // DMA issue distance of SLR_OPT_RECT_DMA_DEPTH = 2 (3 was measured with computed weights: slower, LABNOTES.md)
// one camera (n == 1) or both cameras of a stereo frame (n == 2) in one launch.  *done = false: this form does not apply
// (stack layout, image width), nothing was launched.  depth: DMA issue distance A (1 or 2).
hipError_t launch_mf_rect_decode_dma(const MfPlanes *pl, int n, int pitch, int W, int H, int black_thr, const float *lut,
                                     float *const *phase, uint8_t *const *valid, const void *const *tiles, int shape, int depth,
                                     unsigned *sched, const DmaFixup *fix, bool *done, hipStream_t s, const int *fix_slot)
{
    // fix_slot (n > 2: the cameras of a group of frames, job 2 f + cam): job c's slot in *fix; null: slot c
    *done = false;
    if (shape < 0 || shape >= kDmaShapes || !sched || n < 1 || n > kDmaMaxJobs) return hipSuccess;
    auto slot = [&](int c) { return fix_slot ? fix_slot[c] : c; };
    for (int c = 0; c < n; c++)
        if (fix && fix->nofit[slot(c)] > 0 && (!fix->map_xy[slot(c)] || !fix->map_frac[slot(c)])) return hipSuccess;
    DmaJobs j;
    for (int c = 0; c < n; c++) {
        if (!dma_job(pl[c], pitch, W, H, phase[c], valid[c], tiles[c], shape, j.j[c])) return hipSuccess;
        if (fix) j.j[c].entries += fix->extras[slot(c)];
    }
    for (int c = n; c < kDmaMaxJobs; c++) j.j[c] = j.j[0];
    const bool hv = valid[0] != nullptr;
    for (int c = 1; c < n; c++) if ((valid[c] != nullptr) != hv) return hipSuccess;
    *done = true;
    hipError_t e = hipSuccess;
    // SLR_OPT_EVAL_MODEL = 1: the same kernel with the x87 heterodyne tail (`lut` is then the x87 variant of the tables); it exists at
    // the default DMA distance only (SLR_OPT_RECT_DMA_DEPTH is a tuning knob of the strict build)
    const bool x87 = tl_debug.eval_x87;
#define SLR_DMA_X(TW, TH, NT)                                                                                          \
    e = x87        ? (hv ? launch_dma_variant<TW, TH, NT, SLR_DMA_DEPTH2(TW, TH, NT), true, true>(j, n, pitch, W, H, black_thr, lut, sched, s)     \
                         : launch_dma_variant<TW, TH, NT, SLR_DMA_DEPTH2(TW, TH, NT), false, true>(j, n, pitch, W, H, black_thr, lut, sched, s))   \
      : depth >= 2 ? (hv ? launch_dma_variant<TW, TH, NT, SLR_DMA_DEPTH2(TW, TH, NT), true>(j, n, pitch, W, H, black_thr, lut, sched, s)     \
                         : launch_dma_variant<TW, TH, NT, SLR_DMA_DEPTH2(TW, TH, NT), false>(j, n, pitch, W, H, black_thr, lut, sched, s))   \
                   : (hv ? launch_dma_variant<TW, TH, NT, 1, true>(j, n, pitch, W, H, black_thr, lut, sched, s)              \
                         : launch_dma_variant<TW, TH, NT, 1, false>(j, n, pitch, W, H, black_thr, lut, sched, s))
    SLR_DMA_SHAPE_SWITCH(shape, SLR_DMA_X)
#undef SLR_DMA_X
    // the tiles this form does not hold (none for mild maps): rewritten behind the main kernel
    for (int c = 0; c < n && e == hipSuccess; c++) {
        if (!fix || fix->nofit[slot(c)] == 0) continue;
        const unsigned cnt = fix->nofit[slot(c)];
        const int tiles_x = (W + dma_shape_tw(shape) - 1) / dma_shape_tw(shape);
#define SLR_DMA_X(TW, TH, NT)                                                                                          \
        if (x87) hipLaunchKernelGGL((mf_rect_fixup_kernel<TW, TH, true>), dim3(cnt * (unsigned)(TW * TH / 256) < 16384u ? cnt * (unsigned)(TW * TH / 256) : 16384u), dim3(256), 0, s, pl[c], pitch, W, H, black_thr, \
                           lut, fix->map_xy[slot(c)], fix->map_frac[slot(c)], dma_nofit_list(tiles[c], W, H, shape), cnt, tiles_x, phase[c], valid[c]); \
        else hipLaunchKernelGGL((mf_rect_fixup_kernel<TW, TH>), dim3(cnt * (unsigned)(TW * TH / 256) < 16384u ? cnt * (unsigned)(TW * TH / 256) : 16384u), dim3(256), 0, s, pl[c], pitch, W, H, black_thr, \
                           lut, fix->map_xy[slot(c)], fix->map_frac[slot(c)], dma_nofit_list(tiles[c], W, H, shape), cnt, tiles_x, phase[c], valid[c])
        SLR_DMA_SHAPE_SWITCH(shape, SLR_DMA_X)
#undef SLR_DMA_X
        e = hipGetLastError();
    }
    return e;
}

// ------------------------------------------------------------------------------------------------------
// fused rectify + Gray decode, LDS-DMA form (K1+K3 / K1+K3').  The scheme of the multi-frequency kernel above with the
// Gray stack's phases: (white, black) -> shadow mask, then one (pattern, inverse) pair per code bit, column bits first
// (plane 2c + 2 / 2c + 3, reconstruct.cpp:390-391, 352-353), TWO plane pairs per phase: this kernel has no decode tables and its
// registers (not its LDS) limit it to three workgroups per CU, so the LDS holds 4-plane buffers and a tile needs half the
// barriers.  The pair count 1 + n_col_bits + n_row_bits is a run-time value, so a tile is phase 0 plus a loop over pairs of
// phases (the two LDS buffers alternate; ODD = the phase count is odd and the buffer roles swap from tile to tile).  One DMA distance (A = 1): every phase starts by draining the vector-memory queue,
// which also retires the previous tile's stores.  Same boxes and digest as the MF kernel (one set per camera and shape).
//
// Reference behaviour restated (never copied): stereoRect::doStereoRectify Duke/stereorect.cpp:26-34 fused with
// Reconstruct::computeShadows Duke/reconstruct.cpp:210-227, getProjPixel_GE :381-407 / getProjPixel :325-370 and
// GrayCodes::grayToDec Duke/graycodes.cpp:116-128.
// ------------------------------------------------------------------------------------------------------
struct GrayDmaJob {
    const uint8_t *base;         // plane 0; plane p is base + p * pstride
    unsigned pstride;
    unsigned stack_bytes;        // (planes - 1) * pstride + H * pitch
    const int4 *boxes;
    const unsigned *digest;
    unsigned digest_bytes;
    unsigned entries;            // tiles + the extra parts of split tiles (dma_tiles_kernel)
    int32_t *code_x, *code_y;    // code_y may be null
    uint8_t *valid;              // may be null (the consumer reads validity off code_x == -1)
    float *phase;                // hybrid stacks only (invalid pixels: NaN)
};
struct GrayDmaJobs { GrayDmaJob j[kDmaMaxJobs]; };   // the cameras of one frame or of a group of frames: j[2 f + cam]

constexpr int kGrayDmaWaves1 = 6;     // waves per SIMD the one-pair-per-phase Gray form is compiled for
constexpr int kGrayDmaNpp = 2;        // plane pairs per phase of the fused Gray decode (1 and 2 measured equal)
// HYB (BASELINE config 3, "Gray-code + phase hybrid decode"): the stack carries the 12 fringe planes of the multi-frequency
// method behind the Gray pairs -- white, black, 2 * bits Gray planes, 3 x 4 fringes -- and the tile goes through six more plane
// pairs, (G1, G3) and (G2, G4) of each frequency, decoded exactly as the multi-frequency kernel above decodes them (same tables,
// same integer heterodyne): ONE pass over one box geometry and one digest gives the code image AND the phase image, the shadow
// mask computed once.  The reference's modes are exclusive (mainwindow.h:94); each output equals its own mode's decode
// (reconstruct.cpp:79-97,381-407 and mfreconstruct.cpp:210-269).  Validity travels in-band: code -1 / phase NaN.
// A (round 5): DMA issue distance.  A = 1 is the form above (two buffers, every phase drains the vector-memory queue).  A = 2 is
// the multi-frequency kernel's scheme with this kernel's run-time phase count: THREE buffers of one plane pair each, the planes of
// phase k + 2 issued during phase k, and COUNTED waits -- every wave issues the same static sequence of LDS-DMA operations, so the
// wait at the top of phase k lets exactly the operations issued during phase k - 1 stay in flight (see phase2()).
template <int TW, int TH, int NT, int NPP, bool HYB = false, bool X87 = false /* HYB: the fringe part under SLR_OPT_EVAL_MODEL = 1 */, int A = 1>
struct GrayDma {
    static_assert(NPP == 1 || NPP == 2, "plane pairs per phase");
    static_assert(A == 1 || (A == 2 && NPP == 1), "DMA distance 2: one plane pair per phase, three buffers");
    static constexpr int D = A + 1;                                         // LDS buffers of 2 * NPP plane images
    typedef DmaGeom<TW, TH, NT> Gm;
    static constexpr int PX = Gm::PX, PS = Gm::PS, RS = Gm::RS;
    // SPLIT (twice as many threads as a plane image has chunks): the waves' first half fetches the first half of a phase's
    // 2 * NPP planes, the second half the rest
    static constexpr bool SPLIT = 2 * Gm::NCH <= NT;
    // dynamic LDS from address 0: 2 buffers of 2 * NPP plane images | digest of the tile | weight tables
    static constexpr int DIG_OFF = D * 2 * NPP * PS, DIG_BYTES = TW * TH * 4; // (no scratch slot: waves without chunks issue no DMAs)
    // (HYB with two plane pairs per phase: the blend weights are computed per tile instead of tabulated -- the 8 KB the tables
    //  take would push the workgroup over a third of the CU's LDS)
    static constexpr bool NOWT = HYB && NPP == 2;
    static constexpr int WT_OFF = DIG_OFF + DIG_BYTES, WT1_OFF = WT_OFF + (NOWT ? 0 : 1026 * 4), TAP_WT = NOWT ? -1 : WT_OFF;
    static constexpr int LUT_OFF = WT_OFF + (NOWT ? 0 : 2 * 1026 * 4);      // HYB: the decode tables (slr_create's, kLutWords)
    static constexpr int TKT_OFF = LUT_OFF + (HYB ? (kLutWords + 1) * 4 : 0);   // the next tile's ticket (wave 0 -> everybody)
    static constexpr int LDS_BYTES = TKT_OFF + 8;
    static_assert(WT_OFF <= 65536, "DMA destinations are 16-bit LDS addresses (M0)");

    const uint8_t *smem;
    DmaSched sc;                                             // schedule (see "dynamic tile schedule")
    unsigned lds0, wave_off, pslot_off, plane_g;             // (pslot_off, plane_g: see DmaDecode)
    bool plane_wave, has_cy, has_valid;
    __amdgpu_buffer_rsrc_t rs_stack, rs_dig, rs_cx, rs_cy, rs_valid, rs_phase;
    unsigned pstride;
    int W, H, black_thr, white_thr, ncol, nrow, npairs, nq, scan_w, scan_h;
    // HYB: multi-frequency state (see DmaDecode: folded validity through a running max with the table's sentinel).  The kernel
    // sits at its register budget and its pair loop is a run-time loop, so the state that is never live together shares
    // registers by hand: d of the frequency in flight lives in acc[] and the previous wrapped phase in gxs[] (the code words
    // are finished before the first fringe pair), and the phase to store replaces F12 in fo[].
    const float *lut;
    int pm[PX];
    float fo[PX];
    DmaTap tap[PX];
    unsigned qbase[PX / 4], second;      // (the quad read modes of the MF kernel: dma_tap_setup)
    unsigned oslot[PX / 4];              // the quads' slots in the tile (see the MF kernel)
    int mode;
    unsigned acc[PX], gxs[PX];           // Gray bits of the axis in flight (MSB first); the finished column word
    unsigned flags;                      // bit q: shadow mask of pixel q, bit 8 + q: error
    int out_x[PX], out_y[PX];
    unsigned out_ok;
    int out_ty, out_tx;
    bool out_pending;

    __device__ __forceinline__ void flush() const
    {
        typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int p = 0; p < PX / 4; p++) {
            const int row = out_ty * TH + (int)(oslot[p] >> 6), col = out_tx * TW + (int)(oslot[p] & 63u) * 4;
            const bool inb = row < H && col < W;                     // W % 16 == 0: a quad is whole inside or outside
            const unsigned m = (unsigned)row * (unsigned)W + (unsigned)col;
            const u32x4_t vx = {(unsigned)out_x[4 * p], (unsigned)out_x[4 * p + 1], (unsigned)out_x[4 * p + 2], (unsigned)out_x[4 * p + 3]};
            __builtin_amdgcn_raw_buffer_store_b128(vx, rs_cx, inb ? m * 4u : kDmaInvalid, 0, 2);
            if (!HYB && has_cy) {
                const u32x4_t vy = {(unsigned)out_y[4 * p], (unsigned)out_y[4 * p + 1], (unsigned)out_y[4 * p + 2], (unsigned)out_y[4 * p + 3]};
                __builtin_amdgcn_raw_buffer_store_b128(vy, rs_cy, inb ? m * 4u : kDmaInvalid, 0, 2);
            }
            if (!HYB && has_valid)
                __builtin_amdgcn_raw_buffer_store_b32(__umul24((out_ok >> (4 * p)) & 0xFu, 0x204081u) & 0x01010101u, rs_valid,
                                                      inb ? m : kDmaInvalid, 0, 0);
            if constexpr (HYB) {
                const u32x4_t vp = {__builtin_bit_cast(unsigned, fo[4 * p]), __builtin_bit_cast(unsigned, fo[4 * p + 1]),
                                    __builtin_bit_cast(unsigned, fo[4 * p + 2]), __builtin_bit_cast(unsigned, fo[4 * p + 3])};
                __builtin_amdgcn_raw_buffer_store_b128(vp, rs_phase, inb ? m * 4u : kDmaInvalid, 0, 2);
            }
        }
    }
    // plane g (0 / 1) of pair j: pair 0 = (white, black), pair c = code bit c - 1 (pattern, inverse); HYB, behind the code bits:
    // pair 2f = (G1, G3) and pair 2f + 1 = (G2, G4) of frequency f (fringe plane 4f + s holds G(s+1), mfreconstruct.cpp:239-242)
    __device__ __forceinline__ int pair_plane(int j, int g) const
    {
        const int nb = HYB ? ncol : ncol + nrow;                      // (a hybrid stack carries column bits only)
        if (!HYB || j <= nb) return 2 * j + g;
        const int m = j - 1 - nb;
        return 2 + 2 * nb + 4 * (m >> 1) + (m & 1) + 2 * g;
    }
    // the planes of phase k -- plane pairs NPP * k .. -- into buffer buf
    __device__ __forceinline__ void issue_planes(int k, int buf, unsigned voff) const
    {
        if (!plane_wave) return;         // (wave-uniform)
        const int n = NPP == 1 || 2 * k + 1 < npairs ? 2 * NPP : 2;
        if constexpr (SPLIT) {
            for (int g = (int)plane_g * NPP; g < (int)(plane_g + 1) * NPP && g < n; g++)
                dma16(voff, rs_stack, lds0 + (unsigned)((buf * 2 * NPP + g) * PS) + pslot_off, (unsigned)pair_plane(NPP * k + (g >> 1), g & 1) * pstride);
        } else {
            for (int g = 0; g < n; g++)
                dma16(voff, rs_stack, lds0 + (unsigned)((buf * 2 * NPP + g) * PS) + pslot_off, (unsigned)pair_plane(NPP * k + (g >> 1), g & 1) * pstride);
        }
    }
    // HYB: the difference of fringe pair m (0 .. 5) of a pixel: d = G1 - G3 kept, n = G4 - G2 -> the frequency's wrapped phase,
    // the heterodyne folded in as the frequencies arrive (DmaDecode::phase)
    __device__ __forceinline__ void fringe_step(int q, int m, int sd)
    {
        if (m == 0) pm[q] = ((flags >> q) & 1u) != 0 ? (int)0x80000000 : kDmaSentinel;      // the shadow mask enters the running max
        if ((m & 1) == 0) { acc[q] = (unsigned)sd; return; }                                 // d = G1 - G3
        int nz;
        const int Pw = wrapped_nd_q24(-sd, (int)acc[q], lut, nz);                            // n = G4 - G2
        const int mx = pm[q] > Pw ? pm[q] : Pw;
        pm[q] = mx;
        if (m == 1) gxs[q] = (unsigned)Pw;
        else if (m == 3) { fo[q] = dma_pair_q24((int)gxs[q], Pw); gxs[q] = (unsigned)Pw; }
        else {
            const float F23 = dma_pair_q24((int)gxs[q], Pw);
            float ph;
            if constexpr (X87) ph = het_finish_x87(fo[q], F23);
            else ph = dma_finish_q24(fo[q], F23);
            fo[q] = mx != kDmaSentinel ? ph : kInvalidPhase;
        }
    }
    // the differences of pair j (>= 1) of the thread's pixels go where they belong.  Wave-uniform facts stay out of the per-pixel
    // work: whether the contrast test can fire at all (whiteThreshold's default is 0, SURVEY Q10: it never does), and whether the
    // column word is complete behind this pair -- together 7 of the 10 instructions a code bit used to cost per pixel.
    __device__ __forceinline__ void pair_steps(int j, const int sd[PX])
    {
        if (!HYB || j <= ncol) {
#pragma unroll
            for (int q = 0; q < PX; q++) bit_step(q, sd[q]);
            if (white_thr > 0) {
#pragma unroll
                for (int q = 0; q < PX; q++) flags |= ((sd[q] < 0 ? -sd[q] : sd[q]) < white_thr ? 0x100u : 0u) << q;
            }
            if (j == ncol) {                                 // the column word is complete: the row bits (if any) start from zero
#pragma unroll
                for (int q = 0; q < PX; q++) { gxs[q] = acc[q]; acc[q] = 0u; }
            }
        } else {
#pragma unroll
            for (int q = 0; q < PX; q++) fringe_step(q, j - 1 - ncol, sd[q]);
        }
    }
    __device__ __forceinline__ void issue_digest(unsigned tile, bool live) const
    {
#pragma unroll
        for (int r = 0; r < PX / 4; r++) {
            const unsigned voff = live ? tile * (unsigned)DIG_BYTES + (threadIdx.x + (unsigned)(NT * r)) * 16u : kDmaInvalid;
            dma16(voff, rs_dig, lds0 + (unsigned)(DIG_OFF + r * NT * 16) + wave_off, 0u);
        }
    }
    // code bit from a pair's difference (reconstruct.cpp:387-400 / 349-360): acc = 2 acc + (df > 0), a compare into VCC and an
    // add-with-carry of acc to itself
    __device__ __forceinline__ void bit_step(int q, int df)
    {
        asm("v_cmp_lt_i32 vcc, 0, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(acc[q]) : "v"(df) : "vcc");
    }
    __device__ __forceinline__ void finish(int ty, int tx)
    {
        out_ok = 0;
#pragma unroll
        for (int q = 0; q < PX; q++) {
            unsigned gx = gxs[q], gy = acc[q];                     // grayToDec: running XOR from the MSB == prefix XOR
            gx ^= gx >> 1; gx ^= gx >> 2; gx ^= gx >> 4; gx ^= gx >> 8;
            gy ^= gy >> 1; gy ^= gy >> 2; gy ^= gy >> 4; gy ^= gy >> 8;
            const int x = (int)gx, y = (int)gy;
            bool err = ((flags >> (8 + q)) & 1u) != 0;
            if (!HYB && nrow > 0) err = err || y > scan_h || x > scan_w;   // reconstruct.cpp:364 (Q9 '>')
            else err = err || x > scan_w;                          // reconstruct.cpp:403
            const bool ok = ((flags >> q) & 1u) != 0 && !err;
            out_x[q] = ok ? x : -1;
            if constexpr (!HYB) out_y[q] = (ok && nrow > 0) ? y : -1;
            out_ok |= (ok ? 1u : 0u) << q;
        }
        if constexpr (!HYB) { out_ty = ty; out_tx = tx; out_pending = true; }
    }
    // phase k of a tile in buffer B: plane pairs 2k and 2k + 1 (pair 0 = white, black; pair j = code bit j - 1).  FIRST: k == 0.
    // A pair's sample difference is consumed as soon as it exists; the tap reads run one (pixel, pair) ahead of their use.
    template <int B, bool FIRST>
    __device__ __forceinline__ void phase(int k, int ty, int tx, bool act, unsigned voff_cur)
    {
        if (plane_wave || FIRST) wait_vm<0>();              // (a wave without chunks only waits for its share of the digest)
        asm volatile("s_barrier" ::: "memory");
        // the schedule: wave 0 draws the next tile's ticket around the tap loop of phase 0, every wave picks it up here in phase 1
        // (nq >= 2); the next tile's digest and its first planes are fetched in the tile's last phase
        if constexpr (FIRST) sc.draw();
        else if (k == 1) sc.pick_up();
        const bool last = k + 1 == nq;
        if (last) issue_digest((unsigned)sc.nxt, sc.has_next);   // (every wave is past its digest reads of phase 0)
        issue_planes(last ? 0 : k + 1, B ^ 1, last ? sc.voff_next : voff_cur);
        if constexpr (FIRST) {
            if (out_pending) flush();
            out_pending = false;
        }
        // act (wave-uniform): this wave belongs to the part of the tile that the entry decodes (see dma_tiles_kernel); the other waves
        // keep the barriers, fetch their chunks and (wave 0) draw the next ticket
        if (!act) {
            if constexpr (FIRST) sc.publish();
            return;
        }
        if constexpr (FIRST) {
            const unsigned *dg = reinterpret_cast<const unsigned *>(smem + DIG_OFF + threadIdx.x * (PX * 4));
            mode = dma_tap_setup<PX, RS, TAP_WT, WT1_OFF>(smem, lds0, dg, tap, qbase, second, oslot);
#pragma unroll
            for (int q = 0; q < PX; q++) { acc[q] = 0; gxs[q] = 0; }
            flags = 0;
        }
        constexpr unsigned i00 = (unsigned)(B * 2 * NPP * PS), i01 = i00 + PS, i10 = i00 + 2 * PS, i11 = i00 + 3 * PS;
        __builtin_amdgcn_s_setprio(1);                       // (see the MF kernel: a wave in its tap loop outranks the others)
        {
            int sd[PX];
            dma_differences<PX, i00, i01, (unsigned)RS>(mode, tap, qbase, second, sd);
            if constexpr (FIRST) {
#pragma unroll
                for (int q = 0; q < PX; q++) flags |= (sd[q] > black_thr ? 1u : 0u) << q;   // computeShadows, reconstruct.cpp:218-224
            } else pair_steps(NPP * k, sd);
        }
        if constexpr (NPP == 2) {
            if (2 * k + 1 < npairs) {                        // (a stack's last phase may hold one pair only)
                int sd[PX];
                dma_differences<PX, i10, i11, (unsigned)RS>(mode, tap, qbase, second, sd);
                pair_steps(2 * k + 1, sd);
            }
        }
        __builtin_amdgcn_s_setprio(0);
        if constexpr (FIRST) sc.publish();                   // (the ticket has had the tap loop to arrive)
        // the code words are complete behind the last code-bit pair (HYB: the fringe pairs follow; acc / gxs are free from here)
        const int jlast = NPP * k + (NPP == 2 && 2 * k + 1 < npairs ? 1 : 0);
        if (HYB ? (NPP * k <= ncol && jlast >= ncol && !FIRST) : last) finish(ty, tx);
        if (HYB && last) { out_ty = ty; out_tx = tx; out_pending = true; }
    }
    template <int K>
    __device__ __forceinline__ void tile(int ty, int tx, bool act, unsigned voff_cur)
    {
        phase<K, true>(0, ty, tx, act, voff_cur);
        int k = 1;
        for (; k + 1 < nq; k += 2) {
            phase<K ^ 1, false>(k, ty, tx, act, voff_cur);
            phase<K, false>(k + 1, ty, tx, act, voff_cur);
        }
        if (k < nq) phase<K ^ 1, false>(k, ty, tx, act, voff_cur);
    }

    // ---- A = 2 ------------------------------------------------------------------------------------------------------------------
    // LDS-DMA operations a wave issues, in program order (the static sequence the counted waits rely on), nq >= 4 phases per tile:
    //   phase k:  [k == 2: PX / 4 digest DMAs of the next tile]  PLANE_DMAS plane DMAs of phase k + 2 (of the next tile from nq - 2 on)
    // At the top of phase k the planes of phase k must have landed; what was issued behind them is phase k - 1's share of the
    // sequence: PLANE_DMAS, plus the digest DMAs when k - 1 == 2.  Loads retire in order among themselves; the previous tile's
    // output stores (issued in phase 0, behind that phase's DMAs) are NOT counted, so the wait of phase 1 also retires them (as in
    // the multi-frequency kernel: a count that included them could pass with a DMA still pending).  The ticket of the next tile is
    // drawn in phase 0 and picked up in phase 1, its digest fetched in phase 2 (the digest in LDS is only read in phase 0), its
    // first planes in phases nq - 2 and nq - 1.
    static constexpr int PLANE_DMAS2 = SPLIT ? 1 : 2;
    template <int B, bool FIRST>
    __device__ __forceinline__ void phase2(int k, int ty, int tx, bool act, unsigned voff_cur)
    {
        static_assert(A == 2, "the counted-wait form");
        if (plane_wave) {
            if (k == 3) wait_vm<PLANE_DMAS2 + PX / 4>(); else wait_vm<PLANE_DMAS2>();
        } else if (FIRST) wait_vm<0>();                      // (a wave without chunks only waits for its share of the digest)
        asm volatile("s_barrier" ::: "memory");
        if constexpr (FIRST) sc.draw();
        else if (k == 1) sc.pick_up();
        if (k == 2) issue_digest((unsigned)sc.nxt, sc.has_next);
        {
            const int kp = k + 2;
            const bool nxt_tile = kp >= nq;
            issue_planes(nxt_tile ? kp - nq : kp, (B + 2) % 3, nxt_tile ? sc.voff_next : voff_cur);
        }
        if constexpr (FIRST) {
            if (out_pending) flush();
            out_pending = false;
        }
        if (!act) {
            if constexpr (FIRST) sc.publish();
            return;
        }
        if constexpr (FIRST) {
            const unsigned *dg = reinterpret_cast<const unsigned *>(smem + DIG_OFF + threadIdx.x * (PX * 4));
            mode = dma_tap_setup<PX, RS, TAP_WT, WT1_OFF>(smem, lds0, dg, tap, qbase, second, oslot);
#pragma unroll
            for (int q = 0; q < PX; q++) { acc[q] = 0; gxs[q] = 0; }
            flags = 0;
        }
        constexpr unsigned i00 = (unsigned)(B * 2 * PS), i01 = i00 + PS;
        __builtin_amdgcn_s_setprio(1);
        {
            int sd[PX];
            dma_differences<PX, i00, i01, (unsigned)RS>(mode, tap, qbase, second, sd);
            if constexpr (FIRST) {
#pragma unroll
                for (int q = 0; q < PX; q++) flags |= (sd[q] > black_thr ? 1u : 0u) << q;   // computeShadows, reconstruct.cpp:218-224
            } else pair_steps(k, sd);
        }
        __builtin_amdgcn_s_setprio(0);
        if constexpr (FIRST) sc.publish();
        const bool last = k + 1 == nq;
        if (HYB ? (k == ncol && !FIRST) : last) finish(ty, tx);
        if (HYB && last) { out_ty = ty; out_tx = tx; out_pending = true; }
    }
    // a tile whose phase 0 uses buffer K0; the next tile's phase 0 uses buffer (K0 + nq) % 3
    template <int K0>
    __device__ __forceinline__ void tile2(int ty, int tx, bool act, unsigned voff_cur)
    {
        phase2<K0, true>(0, ty, tx, act, voff_cur);
        int k = 1;
        for (; k + 2 < nq; k += 3) {
            phase2<(K0 + 1) % 3, false>(k, ty, tx, act, voff_cur);
            phase2<(K0 + 2) % 3, false>(k + 1, ty, tx, act, voff_cur);
            phase2<K0, false>(k + 2, ty, tx, act, voff_cur);
        }
        if (k < nq) {
            phase2<(K0 + 1) % 3, false>(k, ty, tx, act, voff_cur);
            if (k + 1 < nq) phase2<(K0 + 2) % 3, false>(k + 1, ty, tx, act, voff_cur);
        }
    }
};

template <int LDS_BYTES, int NT, int NPP>
constexpr int gray_dma_waves()
{
    constexpr int cap = NPP == 1 ? kGrayDmaWaves1 : 6;
    return dma_waves_per_simd<LDS_BYTES, NT>() > cap ? cap : dma_waves_per_simd<LDS_BYTES, NT>();
}

// ROT: how the buffer of a tile's phase 0 moves from tile to tile -- A = 1: the phase count's parity (1 = the two buffers swap
// roles), A = 2: the phase count mod 3
template <int TW, int TH, int NT, int NPP, int ROT, bool HYB = false, bool X87 = false, int A = 1>
__global__ __launch_bounds__(NT, (gray_dma_waves<GrayDma<TW, TH, NT, NPP, HYB, X87, A>::LDS_BYTES, NT, NPP>())) SLR_TICKET_KERNEL
void gray_rect_decode_dma_kernel(GrayDmaJobs jobs, int njobs, int pitch, int W, int H, int black_thr, int white_thr, int ncol, int nrow,
                                 int scan_w, int scan_h, int tiles_x, int tiles_y, unsigned *__restrict__ sched,
                                 const float *__restrict__ lut_g)
{
    typedef GrayDma<TW, TH, NT, NPP, HYB, X87, A> Dec;
    typedef DmaGeom<TW, TH, NT> Gm;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    Dec d;
    d.smem = smem;
    d.lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)smem);
    if (d.lds0 & 0x1FFFu) __builtin_trap();                 // the tap addresses OR the buffer base in (no static LDS here: 0)
    if constexpr (HYB) {                                    // the decode tables, the undefined wrapped phase patched to the sentinel
        float *lut = reinterpret_cast<float *>(smem + Dec::LUT_OFF);
        d.lut = lut;
        for (int i = threadIdx.x; i < kLutWords; i += NT) lut[i] = lut_g[i];
        __syncthreads();
        if (threadIdx.x == 0) reinterpret_cast<int *>(lut)[kLutP + (9 << 8)] = kDmaSentinel;
    }
    if constexpr (!Dec::NOWT)
    for (unsigned i = threadIdx.x; i < 1025u; i += NT) {
        unsigned w0, w1;
        dma_weights(i, w0, w1);
        *reinterpret_cast<unsigned *>(smem + Dec::WT_OFF + 4u * i) = w0;
        *reinterpret_cast<unsigned *>(smem + Dec::WT1_OFF + 4u * i) = w1;
    }
    __syncthreads();
    // workgroup -> (XCD band set, camera, index in the pool): see mf_rect_decode_dma_kernel
    const unsigned nblk = gridDim.x / (unsigned)njobs;      // workgroups per camera (a multiple of 8)
    const int xcd = (int)(blockIdx.x & 7u), nbx = (int)(nblk >> 3);
    const int ji = (int)((blockIdx.x >> 3) % (unsigned)njobs), lb = (int)((blockIdx.x >> 3) / (unsigned)njobs);
    const int T = tiles_x * tiles_y;
    d.sc.pool = DmaPool::of(T, (int)jobs.j[ji].entries, xcd);
    int cur = d.sc.pool.entry(xcd, lb);
    if (cur < 0) {                                          // (whole workgroup) nothing to do
        if (threadIdx.x == 0) dma_sched_leave(sched);
        return;
    }

    d.wave_off = __builtin_amdgcn_readfirstlane((threadIdx.x >> 6) * 1024u);
    {
        const unsigned wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        constexpr unsigned wpp = (unsigned)Gm::NCH / 64u;    // waves per plane image
        d.plane_g = Dec::SPLIT ? wv / wpp : 0u;
        d.pslot_off = (Dec::SPLIT ? wv % wpp : wv) * 1024u;
        d.plane_wave = Dec::SPLIT ? wv < 2u * wpp : wv < wpp;
    }
    d.pstride = jobs.j[ji].pstride;
    d.W = W; d.H = H; d.black_thr = black_thr; d.white_thr = white_thr; d.ncol = ncol; d.nrow = nrow;
    d.npairs = 1 + ncol + nrow + (HYB ? 6 : 0); d.nq = (d.npairs + NPP - 1) / NPP;
    d.scan_w = scan_w; d.scan_h = scan_h;
    d.has_cy = jobs.j[ji].code_y != nullptr; d.has_valid = jobs.j[ji].valid != nullptr;
    const int n4 = (int)((unsigned)W * (unsigned)H * 4u);
    d.rs_stack = __builtin_amdgcn_make_buffer_rsrc((void *)jobs.j[ji].base, 0, (int)jobs.j[ji].stack_bytes, 0x00020000);
    d.rs_dig = __builtin_amdgcn_make_buffer_rsrc((void *)jobs.j[ji].digest, 0, (int)jobs.j[ji].digest_bytes, 0x00020000);
    d.rs_cx = __builtin_amdgcn_make_buffer_rsrc((void *)jobs.j[ji].code_x, 0, n4, 0x00020000);
    if constexpr (!HYB) {
        d.rs_cy = __builtin_amdgcn_make_buffer_rsrc((void *)jobs.j[ji].code_y, 0, d.has_cy ? n4 : 0, 0x00020000);
        d.rs_valid = __builtin_amdgcn_make_buffer_rsrc((void *)jobs.j[ji].valid, 0, d.has_valid ? n4 / 4 : 0, 0x00020000);
    }
    d.rs_phase = __builtin_amdgcn_make_buffer_rsrc((void *)jobs.j[ji].phase, 0, HYB ? n4 : 0, 0x00020000);
    d.sc.boxes = jobs.j[ji].boxes;
    d.sc.pitch = pitch; d.sc.W = W; d.sc.H = H;
    const int chunk = Dec::SPLIT ? (int)threadIdx.x % Gm::NCH : (int)threadIdx.x;
    d.sc.crow = chunk / Gm::CMAX; d.sc.ccol = chunk - d.sc.crow * Gm::CMAX;
    d.sc.chunk_live = Dec::SPLIT || chunk < Gm::NCH;
    d.sc.xcd = xcd; d.sc.first_ticket = nbx;
    d.sc.wave0 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) == 0u;
    d.sc.wave_bit = 1u << __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    d.sc.tkt_lds = d.lds0 + (unsigned)Dec::TKT_OFF;
    d.sc.ctr = sched + (ji * 8 + xcd) * kSchedStride;
    d.sc.nxt = 0; d.sc.has_next = false; d.sc.voff_next = kDmaInvalid; d.sc.nxt_ty = d.sc.nxt_tx = 0; d.sc.nxt_act = false;

    int cur_ty, cur_tx;
    bool cur_act;
    unsigned voff_cur = d.sc.box_voff(cur, cur_ty, cur_tx, cur_act);
    d.out_pending = false;
    d.issue_digest((unsigned)cur, true);                    // prologue = what the last phase(s) of a previous tile would have issued
    d.issue_planes(0, 0, voff_cur);
    if constexpr (A == 2) d.issue_planes(1, 1, voff_cur);

    for (;;) {
#define SLR_GDMA_TILE(K0)                                                                                      \
        {                                                                                                      \
            if constexpr (A == 2) d.template tile2<K0>(cur_ty, cur_tx, cur_act, voff_cur);                     \
            else d.template tile<K0>(cur_ty, cur_tx, cur_act, voff_cur);                                       \
            if (!d.sc.has_next) break;                                                                         \
            cur = d.sc.nxt; voff_cur = d.sc.voff_next;                                                         \
            cur_ty = d.sc.nxt_ty; cur_tx = d.sc.nxt_tx; cur_act = d.sc.nxt_act;                                \
        }
        SLR_GDMA_TILE(0)
        if constexpr (ROT != 0) SLR_GDMA_TILE(ROT % Dec::D)
        if constexpr (A == 2 && ROT != 0) SLR_GDMA_TILE((2 * ROT) % Dec::D)
#undef SLR_GDMA_TILE
    }
    // (only what a tile left behind: a wave that was idle in every entry its workgroup decoded -- parts of split tiles as a workgroup's
    //  first and only entries, i.e. small images on the full resident set -- has no results, and its out_ty / oslot / values are
    //  whatever the registers held: stores of junk to junk addresses inside the image, found with poisoned outputs at 1040 x 524)
    if (d.out_pending) d.flush();
    wait_vm<0>();                                           // the dummy DMAs behind the last tile must land before the LDS is released
    if (threadIdx.x == 0) dma_sched_leave(sched);
}

static bool gray_dma_job(const GrayPlanes &pl, int np, int pitch, int W, int H, int32_t *cx, int32_t *cy, uint8_t *valid, const void *tiles,
                         int shape, GrayDmaJob &j)
{
    if (!tiles || W % 16 != 0 || pitch % 16 != 0 || ((uintptr_t)pl.p[0] % 16) != 0 || ((uintptr_t)cx % 4) != 0 || ((uintptr_t)cy % 4) != 0)
        return false;
    const long long st = (long long)(pl.p[1] - pl.p[0]);
    if (st < (long long)H * pitch || st % 16 != 0) return false;
    for (int i = 2; i < np; i++) if ((long long)(pl.p[i] - pl.p[0]) != st * i) return false;
    const long long bytes = st * (np - 1) + (long long)H * pitch;
    if (bytes >= (1ll << 31) || (long long)W * H * 4 >= (1ll << 31)) return false;
    const char *b = reinterpret_cast<const char *>(tiles);
    j.base = pl.p[0]; j.pstride = (unsigned)st; j.stack_bytes = (unsigned)bytes;
    j.boxes = reinterpret_cast<const int4 *>(b);
    j.digest = reinterpret_cast<const unsigned *>(b + dma_digest_offset(W, H, shape));
    const size_t dg = dma_entry_capacity(W, H, shape) * (size_t)(dma_shape_tw(shape) * dma_shape_th(shape)) * 4;
    if (dg >= (1ull << 31)) return false;
    j.digest_bytes = (unsigned)dg;
    j.entries = (unsigned)dma_tile_count(W, H, shape);      // (+ the extra parts: the launcher adds them)
    j.code_x = cx; j.code_y = cy; j.valid = valid; j.phase = nullptr;
    return true;
}

template <int TW, int TH, int NT, int NPP, int ROT, bool HYB = false, bool X87 = false, int A = 1>
static hipError_t launch_gray_dma_variant(const GrayDmaJobs &j, int njobs, int pitch, int W, int H, int black_thr, int white_thr, int ncol,
                                          int nrow, int scan_w, int scan_h, unsigned *sched, hipStream_t s, const float *lut = nullptr)
{
    typedef GrayDma<TW, TH, NT, NPP, HYB, X87, A> Dec;
    auto kern = gray_rect_decode_dma_kernel<TW, TH, NT, NPP, ROT, HYB, X87, A>;
    static DevSlots resident;                             // resident workgroups of this kernel, per device
    int dev = 0;
    (void)hipGetDevice(&dev);
    int res = resident.get(dev);
    if (!res) {
        hipError_t e0 = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, Dec::LDS_BYTES);
        if (e0 != hipSuccess) return e0;
        int per_cu = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, NT, Dec::LDS_BYTES) != hipSuccess || per_cu < 1) per_cu = 1;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
        res = per_cu * cus;
        resident.put(dev, res);
    }
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    const int T = tiles_x * tiles_y, per = DmaPool::of(T).per;
    const int r = (tl_debug.rect_resident > 0 ? tl_debug.rect_resident : res) / njobs;   // tests: few workgroups -> many tiles each
    int nbx = r / 8 < per ? r / 8 : per;
    if (nbx < 1) nbx = 1;
    hipError_t e = hipMemsetAsync(sched, 0, kSchedBytes, s);         // (the tickets start from zero: see launch_dma_variant)
    if (e != hipSuccess) return e;
    SLR_LAUNCH(kern, dim3(8u * (unsigned)nbx * (unsigned)njobs), dim3(NT), Dec::LDS_BYTES, s, j, njobs, pitch, W, H, black_thr, white_thr,
               ncol, nrow, scan_w, scan_h, tiles_x, tiles_y, sched, lut);
    return hipGetLastError();
}

// one camera (n == 1) or both cameras of a stereo frame (n == 2) in one launch; *done = false: the form does not apply (stack
// layout, image width, a tile shape without a Gray instantiation), nothing was launched
hipError_t launch_gray_rect_decode_dma(const GrayPlanes *pl, int n, int ncol, int nrow, int pitch, int W, int H, int black_thr,
                                       int white_thr, int scan_w, int scan_h, int32_t *const *code_x, int32_t *const *code_y,
                                       uint8_t *const *valid, const void *const *tiles, int shape, int depth, unsigned *sched, const DmaFixup *fix,
                                       bool *done, hipStream_t s, const int *fix_slot)
{
    *done = false;
    if (!sched || n < 1 || n > kDmaMaxJobs) return hipSuccess;
    auto slot = [&](int c) { return fix_slot ? fix_slot[c] : c; };          // (job c's slot in *fix: launch_mf_rect_decode_dma)
    for (int c = 0; c < n; c++)
        if (fix && fix->nofit[slot(c)] > 0 && (!fix->map_xy[slot(c)] || !fix->map_frac[slot(c)])) return hipSuccess;
    if (shape != 1 && shape != 3 && shape != 4 && shape != 5) return hipSuccess;      // the 4-pixels-per-thread shapes
    if (ncol + nrow < 2) return hipSuccess;                 // (a tile needs two phases: the next digest arrives during the second)
    const int np = 2 + 2 * ncol + 2 * nrow;
    GrayDmaJobs j;
    for (int c = 0; c < n; c++) {
        if (!gray_dma_job(pl[c], np, pitch, W, H, code_x[c], code_y[c], valid[c], tiles[c], shape, j.j[c])) return hipSuccess;
        if (fix) j.j[c].entries += fix->extras[slot(c)];
    }
    for (int c = n; c < kDmaMaxJobs; c++) j.j[c] = j.j[0];
    *done = true;
    constexpr int NPP = kGrayDmaNpp;
    const bool odd = ((((1 + ncol + nrow) + NPP - 1) / NPP) & 1) != 0;   // phases (of NPP plane pairs) per tile
    // round 5: the counted-wait form (DMA distance 2, one plane pair per phase) whenever a tile has the four phases it needs;
    // depth 1 (SLR_OPT_RECT_DMA_DEPTH) keeps round 2's form, two plane pairs per phase and a drained queue every phase
    const int nq2 = 1 + ncol + nrow;
    const bool a2 = nq2 >= 4 && depth >= 2;
    hipError_t e = hipSuccess;
#define SLR_GDMA_X(TW, TH, NT)                                                                                                     \
    e = a2 ? (nq2 % 3 == 0 ? launch_gray_dma_variant<TW, TH, NT, 1, 0, false, false, 2>(j, n, pitch, W, H, black_thr, white_thr, ncol, nrow, scan_w, scan_h, sched, s)   \
            : nq2 % 3 == 1 ? launch_gray_dma_variant<TW, TH, NT, 1, 1, false, false, 2>(j, n, pitch, W, H, black_thr, white_thr, ncol, nrow, scan_w, scan_h, sched, s)   \
                           : launch_gray_dma_variant<TW, TH, NT, 1, 2, false, false, 2>(j, n, pitch, W, H, black_thr, white_thr, ncol, nrow, scan_w, scan_h, sched, s))  \
      : odd ? launch_gray_dma_variant<TW, TH, NT, NPP, 1>(j, n, pitch, W, H, black_thr, white_thr, ncol, nrow, scan_w, scan_h, sched, s)    \
            : launch_gray_dma_variant<TW, TH, NT, NPP, 0>(j, n, pitch, W, H, black_thr, white_thr, ncol, nrow, scan_w, scan_h, sched, s)
    switch (shape) {
    case 1:  SLR_GDMA_X(256, 8, 512); break;
#ifdef SLR_ALL_FORMS
    case 4:  SLR_GDMA_X(128, 8, 256); break;
    case 5:  SLR_GDMA_X(256, 4, 256); break;
#endif
    default: SLR_GDMA_X(128, 16, 512); break;
    }
#undef SLR_GDMA_X
    for (int c = 0; c < n && e == hipSuccess; c++) {        // the tiles this form does not hold: rewritten behind the main kernel
        if (!fix || fix->nofit[slot(c)] == 0) continue;
        const unsigned cnt = fix->nofit[slot(c)];
        const int tiles_x = (W + dma_shape_tw(shape) - 1) / dma_shape_tw(shape);
#define SLR_GDMA_X(TW, TH, NT)                                                                                                     \
        hipLaunchKernelGGL((gray_rect_fixup_kernel<TW, TH>), dim3(cnt * (unsigned)(TW * TH / 256) < 16384u ? cnt * (unsigned)(TW * TH / 256) : 16384u), dim3(256), 0, s, pl[c], ncol, nrow, pitch, W, H, \
                           black_thr, white_thr, scan_w, scan_h, fix->map_xy[slot(c)], fix->map_frac[slot(c)], dma_nofit_list(tiles[c], W, H, shape), \
                           cnt, tiles_x, code_x[c], code_y[c], valid[c])
        switch (shape) {
        case 1:  SLR_GDMA_X(256, 8, 512); break;
#ifdef SLR_ALL_FORMS
        case 4:  SLR_GDMA_X(128, 8, 256); break;
        case 5:  SLR_GDMA_X(256, 4, 256); break;
#endif
        default: SLR_GDMA_X(128, 16, 512); break;
        }
#undef SLR_GDMA_X
        e = hipGetLastError();
    }
    return e;
}

constexpr int kHybNpp = 1;            // plane pairs per phase of the hybrid kernel
// BASELINE config 3: one pass over a hybrid stack (white, black, 2 * ncol Gray planes, 12 fringe planes; equally spaced) of one
// camera (n == 1) or both (n == 2): code_x (-1 where invalid) and phase (NaN where invalid).  *done = false: the form does not
// apply (stack layout, image width, maps), nothing was launched -- the caller runs the two separate fused decodes instead.
hipError_t launch_hybrid_rect_decode_dma(const GrayPlanes *pl, int n, int ncol, int pitch, int W, int H, int black_thr, int white_thr,
                                         int scan_w, const float *lut, int32_t *const *code_x, float *const *phase,
                                         const void *const *tiles, int shape, unsigned *sched, const DmaFixup *fix, bool *done,
                                         hipStream_t s)
{
    *done = false;
    if (!sched || !lut || shape != 3 || ncol < 1) return hipSuccess;     // (the default tile shape only)
    const int np = 2 + 2 * ncol + 12;
    if (np > SLR_MAX_GRAY_PLANES) return hipSuccess;
    for (int c = 0; c < n; c++)
        if (fix && fix->nofit[c] > 0 && (!fix->map_xy[c] || !fix->map_frac[c])) return hipSuccess;
    GrayDmaJobs j;
    for (int c = 0; c < n; c++) {
        if (!gray_dma_job(pl[c], np, pitch, W, H, code_x[c], nullptr, nullptr, tiles[c], shape, j.j[c])) return hipSuccess;
        if (((uintptr_t)phase[c] % 16) != 0) return hipSuccess;
        j.j[c].phase = phase[c];
        if (fix) j.j[c].entries += fix->extras[c];
    }
    if (n == 1) j.j[1] = j.j[0];
    *done = true;
    constexpr int NPP = kHybNpp;
    const bool odd = ((((1 + ncol + 6) + NPP - 1) / NPP) & 1) != 0;     // phases (of NPP plane pairs) per tile
    const bool x87 = tl_debug.eval_x87;                     // (`lut` is then the x87 variant of the tables)
    hipError_t e = x87 ? (odd ? launch_gray_dma_variant<128, 16, 512, NPP, 1, true, true>(j, n, pitch, W, H, black_thr, white_thr, ncol, 0, scan_w, 0, sched, s, lut)
                              : launch_gray_dma_variant<128, 16, 512, NPP, 0, true, true>(j, n, pitch, W, H, black_thr, white_thr, ncol, 0, scan_w, 0, sched, s, lut))
                       : (odd ? launch_gray_dma_variant<128, 16, 512, NPP, 1, true>(j, n, pitch, W, H, black_thr, white_thr, ncol, 0, scan_w, 0, sched, s, lut)
                              : launch_gray_dma_variant<128, 16, 512, NPP, 0, true>(j, n, pitch, W, H, black_thr, white_thr, ncol, 0, scan_w, 0, sched, s, lut));
    // tiles the form does not hold: the two gather fix-ups, one per output
    for (int c = 0; c < n && e == hipSuccess; c++) {
        if (!fix || fix->nofit[c] == 0) continue;
        const unsigned cnt = fix->nofit[c], grid = cnt * 8u < 16384u ? cnt * 8u : 16384u;
        const int tiles_x = (W + 127) / 128;
        MfPlanes mp;
        mp.p[0] = pl[c].p[0]; mp.p[1] = pl[c].p[1];
        for (int k = 0; k < 12; k++) mp.p[2 + k] = pl[c].p[2 + 2 * ncol + k];
        hipLaunchKernelGGL((gray_rect_fixup_kernel<128, 16>), dim3(grid), dim3(256), 0, s, pl[c], ncol, 0, pitch, W, H, black_thr, white_thr,
                           scan_w, 0, fix->map_xy[c], fix->map_frac[c], dma_nofit_list(tiles[c], W, H, shape), cnt, tiles_x, code_x[c],
                           (int32_t *)nullptr, (uint8_t *)nullptr);
        if (x87) hipLaunchKernelGGL((mf_rect_fixup_kernel<128, 16, true>), dim3(grid), dim3(256), 0, s, mp, pitch, W, H, black_thr, lut, fix->map_xy[c],
                           fix->map_frac[c], dma_nofit_list(tiles[c], W, H, shape), cnt, tiles_x, phase[c], (uint8_t *)nullptr);
        else hipLaunchKernelGGL((mf_rect_fixup_kernel<128, 16>), dim3(grid), dim3(256), 0, s, mp, pitch, W, H, black_thr, lut, fix->map_xy[c],
                           fix->map_frac[c], dma_nofit_list(tiles[c], W, H, shape), cnt, tiles_x, phase[c], (uint8_t *)nullptr);
        e = hipGetLastError();
    }
    return e;
}

}  // namespace slr
