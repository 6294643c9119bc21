// kernels_decode.hip -- K1 (rectification remap), K2 (multi-frequency phase decode + heterodyne unwrap),
// K3/K3' (Gray-code decode) and the fused rectify+decode variants.  gfx950 (MI355X) only.
//
// Streaming kernels over N separate u8 planes: no MFMA, plain integer/f32 ALU, wide coalesced loads.  PMC shows they
// move the algorithmic HBM bytes (1.00-1.13x); the per-pixel branch chains live in small LDS tables (the reference's
// quotient is an integer) and the bilinear blend uses v_perm / v_dot2, which took K2 to the box's copy
// ceiling.  The fused rectify+decode forms (SLR_OPT_RECT_DECODE_ALGO, include/slr.h) are bound by how the memory
// system serves the short source row segments of a tile (profiles/exp/boxread.hip):
//   mf_rect_decode_lds_kernel<TH, ROUNDS, STRIDED, TWV, PACKED, NT>   persistent workgroups, register prefetch pipeline
//       <8,1,*,2,true,512>  128x8 tiles, 512 threads, map digest   (5, what "auto" picks when the maps fit it)
//       <8,1,*,1,true,256>  64x8 tiles, map digest                 (6)
//       <16,2,*>            64x16 tiles, two prefetch rounds       (2)
//       <8,2,*,2>           128x8 tiles, 256 threads, two rounds   (4)
//   mf_rect_decode_ring_kernel                                      16-row sliding LDS window down tile columns (3)
//   mf_rect_decode_kernel<V>                                        direct gather (1, and the per-tile fallback)
//
// Reference behaviour restated (never copied):
//   K1  stereoRect::doStereoRectify -> cv::remap(CV_16SC2,CV_16UC1,INTER_LINEAR)   Duke/stereorect.cpp:26-34
//   K2  MFReconstruct::computeShadows/decodePatterns/getPhase                      Duke/mfreconstruct.cpp:190-269
//   K3  Reconstruct::computeShadows/decodePatterns_GE/getProjPixel_GE              Duke/reconstruct.cpp:79-97,210-227,381-407
//   K3' Reconstruct::decodePaterns/getProjPixel (col + row bits)                   Duke/reconstruct.cpp:56-74,325-370
//       GrayCodes::grayToDec                                                       Duke/graycodes.cpp:116-128
#include "decode_common.hpp"
#include <type_traits>

#include <stdlib.h>

namespace slr {

// ------------------------------------------------------------------------------------------------------
// K1: standalone remap, 4 destination pixels per thread (W % 4 == 0) or 1 (generic)
// ------------------------------------------------------------------------------------------------------
template <int V>
__global__ __launch_bounds__(256) void remap_kernel(const uint8_t *__restrict__ src, int src_pitch,
                                                    uint8_t *__restrict__ dst, int dst_pitch, int W, int H,
                                                    const int16_t *__restrict__ map_xy,
                                                    const uint16_t *__restrict__ map_frac)
{
    const int gpr = (W + V - 1) / V;
    const unsigned total = (unsigned)gpr * (unsigned)H;
    for (unsigned g = blockIdx.x * 256u + threadIdx.x; g < total; g += gridDim.x * 256u) {
        const int row = g / gpr, col0 = (g - row * gpr) * V;
        const size_t m = (size_t)row * W + col0;
        if (V == 4) {
            const u32x4 xy = *reinterpret_cast<const u32x4 *>(map_xy + 2 * m);
            const u32x2 fr = *reinterpret_cast<const u32x2 *>(map_frac + m);
            const unsigned xyw[4] = {xy.x, xy.y, xy.z, xy.w};
            const unsigned frw[4] = {fr.x & 0xFFFFu, fr.x >> 16, fr.y & 0xFFFFu, fr.y >> 16};
            unsigned out = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const Tap t = make_tap((int)(short)(xyw[i] & 0xFFFFu), (int)(short)(xyw[i] >> 16), frw[i],
                                       src_pitch, W, H);
                out |= (unsigned)sample(src, src_pitch, W, H, t) << (8 * i);
            }
            *reinterpret_cast<unsigned *>(dst + (size_t)row * dst_pitch + col0) = out;
        } else {
            const Tap t = make_tap(map_xy[2 * m], map_xy[2 * m + 1], map_frac[m], src_pitch, W, H);
            dst[(size_t)row * dst_pitch + col0] = (uint8_t)sample(src, src_pitch, W, H, t);
        }
    }
}

hipError_t launch_remap_u8(const uint8_t *src, int src_pitch, uint8_t *dst, int dst_pitch, int W, int H,
                           const int16_t *map_xy, const uint16_t *map_frac, hipStream_t s)
{
    const bool vec = (W % 4 == 0) && (dst_pitch % 4 == 0) && ((uintptr_t)dst % 4 == 0);
    const size_t groups = vec ? (size_t)(W / 4) * H : (size_t)W * H;
    const unsigned blocks = (unsigned)((groups + 255) / 256 < 8192 ? (groups + 255) / 256 : 8192);
    if (vec) SLR_LAUNCH(remap_kernel<4>, dim3(blocks), dim3(256), 0, s, src, src_pitch, dst, dst_pitch, W, H, map_xy, map_frac);
    else     SLR_LAUNCH(remap_kernel<1>, dim3(blocks), dim3(256), 0, s, src, src_pitch, dst, dst_pitch, W, H, map_xy, map_frac);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------
// K2: multi-frequency decode.  NW dwords (4*NW pixels) per thread per plane; NW==0 -> generic 1 px/thread.
// ------------------------------------------------------------------------------------------------------
template <int NW> struct WordVec;
template <> struct WordVec<1> { typedef unsigned type; };
template <> struct WordVec<2> { typedef u32x2 type; };
template <> struct WordVec<4> { typedef u32x4 type; };

__device__ __forceinline__ unsigned word_of(unsigned v, int) { return v; }
__device__ __forceinline__ unsigned word_of(const u32x2 &v, int k) { return v[k]; }
__device__ __forceinline__ unsigned word_of(const u32x4 &v, int k) { return v[k]; }

template <int NW, bool X87 = false>
__global__ __launch_bounds__(256) void mf_decode_kernel(MfPlanes pl, int pitch, int W, int H, int black_thr,
                                                        const float *__restrict__ lut_g,
                                                        float *__restrict__ phase, uint8_t *__restrict__ valid)
{
    __shared__ float lut[kLutWords + 1];
    load_lut(lut, lut_g);
    typedef typename WordVec<NW>::type vec_t;
    constexpr int V = 4 * NW;
    const int gpr = W / V;                                  // launcher guarantees W % V == 0
    const unsigned total = (unsigned)gpr * (unsigned)H;
    auto process = [&](unsigned row, unsigned col0) {
        // 32-bit byte offsets (image < 2 GiB, checked by the C ABI): lets the compiler use the SGPR-base + VGPR-offset
        // addressing form instead of a 64-bit add per plane
        const unsigned so = row * (unsigned)pitch + col0, oo = row * (unsigned)W + col0;
        vec_t w[SLR_MF_PLANES];
#pragma unroll
        for (int p = 0; p < SLR_MF_PLANES; p++)
            w[p] = __builtin_nontemporal_load(reinterpret_cast<const vec_t *>(pl.p[p] + so));
        unsigned vout[NW];
#pragma unroll
        for (int k = 0; k < NW; k++) {
            float ph[4];
            unsigned vw = 0;
#pragma unroll
            for (int b = 0; b < 4; b++) {
                int gpx[SLR_MF_PLANES];
#pragma unroll
                for (int p = 0; p < SLR_MF_PLANES; p++) gpx[p] = (word_of(w[p], k) >> (8 * b)) & 0xFFu;
                int v;
                ph[b] = mf_pixel<X87>(gpx, black_thr, lut, v);
                if (!valid) ph[b] = v ? ph[b] : kInvalidPhase;
                vw |= (unsigned)v << (8 * b);
            }
            f32x4 o; o.x = ph[0]; o.y = ph[1]; o.z = ph[2]; o.w = ph[3];
            __builtin_nontemporal_store(o, reinterpret_cast<f32x4 *>(phase + oo + 4 * k));
            vout[k] = vw;
        }
        if (valid) {
            vec_t vv;
            if constexpr (NW == 1) vv = vout[0];
            else if constexpr (NW == 2) { vv.x = vout[0]; vv.y = vout[1]; }
            else { vv.x = vout[0]; vv.y = vout[1]; vv.z = vout[2]; vv.w = vout[3]; }
            __builtin_nontemporal_store(vv, reinterpret_cast<vec_t *>(valid + oo));
        }
    };
    if (gridDim.y > 1) {                                    // padded rows: image rows blockIdx.y + k gridDim.y, no division
        const unsigned c = blockIdx.x * 256u + threadIdx.x;
        if (c < (unsigned)gpr)
            for (unsigned row = blockIdx.y; row < (unsigned)H; row += gridDim.y) process(row, c * V);
    } else {                                                // flat image (pitch == W): no row arithmetic at all
        for (unsigned g = blockIdx.x * 256u + threadIdx.x; g < total; g += gridDim.x * 256u) {
            if (pitch == W) process(0u, g * V);
            else { const unsigned row = g / gpr; process(row, (g - row * gpr) * V); }
        }
    }
}

// generic (any W / pitch / alignment): one pixel per thread
template <bool X87 = false>
__global__ __launch_bounds__(256) void mf_decode_scalar_kernel(MfPlanes pl, int pitch, int W, int H, int black_thr,
                                                               const float *__restrict__ lut_g,
                                                               float *__restrict__ phase, uint8_t *__restrict__ valid)
{
    __shared__ float lut[kLutWords + 1];
    load_lut(lut, lut_g);
    const unsigned total = (unsigned)W * (unsigned)H;
    for (unsigned g = blockIdx.x * 256u + threadIdx.x; g < total; g += gridDim.x * 256u) {
        const unsigned row = g / W, col = g - row * W;
        const size_t so = (size_t)row * pitch + col;
        int gpx[SLR_MF_PLANES];
#pragma unroll
        for (int p = 0; p < SLR_MF_PLANES; p++) gpx[p] = pl.p[p][so];
        int v;
        const float ph = mf_pixel<X87>(gpx, black_thr, lut, v);
        phase[g] = valid || v ? ph : kInvalidPhase;
        if (valid) valid[g] = (uint8_t)v;
    }
}

// fused K1+K2: V destination pixels per thread (V = 4 when W % 4 == 0, else 1); taps gathered through L1/L2
template <int V, bool X87 = false>
__global__ __launch_bounds__(256) void mf_rect_decode_kernel(MfPlanes pl, int pitch, int W, int H, int black_thr,
                                                             const float *__restrict__ lut_g,
                                                             const int16_t *__restrict__ map_xy,
                                                             const uint16_t *__restrict__ map_frac,
                                                             float *__restrict__ phase, uint8_t *__restrict__ valid)
{
    __shared__ float lut[kLutWords + 1];
    load_lut(lut, lut_g);
    const int gpr = W / V;
    const unsigned total = (unsigned)gpr * (unsigned)H;
    // XCD-aware block order: workgroup b runs on XCD b % 8 (observed dispatch rule, used for speed only).  Give
    // every XCD a contiguous band of the image so that the two source rows shared by vertically adjacent
    // destination rows are served by ONE XCD's L2 instead of being fetched from HBM by two.
    const unsigned nb = gridDim.x, per = (nb + 7) / 8;
    unsigned vb = (blockIdx.x % 8) * per + blockIdx.x / 8;
    if (nb % 8 != 0) vb = blockIdx.x;                       // launcher keeps nb a multiple of 8; safety net
    for (unsigned g = vb * 256u + threadIdx.x; g < total; g += nb * 256u) {
        const unsigned row = g / gpr, col0 = (g - row * gpr) * V;
        const size_t m = (size_t)row * W + col0;
        unsigned xyw[V], frw[V];
        if constexpr (V == 4) {
            const u32x4 xy = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(map_xy + 2 * m));
            const u32x2 fr = __builtin_nontemporal_load(reinterpret_cast<const u32x2 *>(map_frac + m));
            xyw[0] = xy.x; xyw[1] = xy.y; xyw[2] = xy.z; xyw[3] = xy.w;
            frw[0] = fr.x & 0xFFFFu; frw[1] = fr.x >> 16; frw[2] = fr.y & 0xFFFFu; frw[3] = fr.y >> 16;
        } else {
            xyw[0] = (unsigned)(uint16_t)map_xy[2 * m] | ((unsigned)(uint16_t)map_xy[2 * m + 1] << 16);
            frw[0] = map_frac[m];
        }
        float ph[V];
        unsigned vw = 0;
        Tap taps[V];
#pragma unroll
        for (int i = 0; i < V; i++)
            taps[i] = make_tap((int)(short)(xyw[i] & 0xFFFFu), (int)(short)(xyw[i] >> 16), frw[i], pitch, W, H);
        // rectified samples of the V pixels, one packed word (V bytes) per plane
        unsigned packed[SLR_MF_PLANES];
        bool fast = false;
        if constexpr (V == 4) {
            // Fast path (almost every thread of a real map): the four 2x2 footprints share their two source
            // rows and fit an 8-byte window -> two unaligned 8-byte loads per plane instead of 16 byte loads;
            // tap pairs are cut out with v_perm_b32 and blended with v_dot4_u32_u8 (weights <= 32 fit u8).
            // The four footprints of a smooth map cover at most three source rows (sy changes by <= 1 inside
            // the group) and an 8-byte column window: three unaligned 8-byte loads per plane, no branches, so
            // all 42 loads of a thread are in flight together.
            int symin = taps[0].sy, symax = taps[0].sy;
            int sxmin = taps[0].sx, sxmax = taps[0].sx;
            bool inl = taps[0].kind == 0;
#pragma unroll
            for (int i = 1; i < 4; i++) {
                inl = inl && taps[i].kind == 0;
                symin = taps[i].sy < symin ? taps[i].sy : symin;
                symax = taps[i].sy > symax ? taps[i].sy : symax;
                sxmin = taps[i].sx < sxmin ? taps[i].sx : sxmin;
                sxmax = taps[i].sx > sxmax ? taps[i].sx : sxmax;
            }
            fast = inl && (symax - symin) <= 1 && (sxmax - sxmin) <= 6 && (sxmin + 8) <= W;
            if (fast) {
                typedef unsigned long long u64u __attribute__((aligned(1)));
                const int off0 = symin * pitch + sxmin;
                // third row only matters when some footprint starts at symin+1 (then symin+2 <= H-1 because that
                // tap is an inlier); otherwise re-read row symin+1 so the address is always inside the image
                const int off2 = off0 + ((symax > symin) ? 2 : 1) * pitch;
                unsigned sel[4], wx[4];
                bool dy[4];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const unsigned o = (unsigned)(taps[i].sx - sxmin);
                    sel[i] = o | ((o + 1) << 8) | 0x0C0C0000u;           // bytes o, o+1 of the window; 0 above
                    wx[i] = (unsigned)taps[i].wx0 | ((unsigned)taps[i].wx1 << 8);
                    dy[i] = taps[i].sy != symin;
                }
#pragma unroll
                for (int p = 0; p < SLR_MF_PLANES; p++) {
                    const unsigned long long r0 = *reinterpret_cast<const u64u *>(pl.p[p] + off0);
                    const unsigned long long r1 = *reinterpret_cast<const u64u *>(pl.p[p] + off0 + pitch);
                    const unsigned long long r2 = *reinterpret_cast<const u64u *>(pl.p[p] + off2);
                    unsigned w = 0;
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const unsigned q0 = __builtin_amdgcn_perm((unsigned)(r0 >> 32), (unsigned)r0, sel[i]);
                        const unsigned q1 = __builtin_amdgcn_perm((unsigned)(r1 >> 32), (unsigned)r1, sel[i]);
                        const unsigned q2 = __builtin_amdgcn_perm((unsigned)(r2 >> 32), (unsigned)r2, sel[i]);
                        const unsigned h0 = __builtin_amdgcn_udot4(dy[i] ? q1 : q0, wx[i], 0u, false);
                        const unsigned h1 = __builtin_amdgcn_udot4(dy[i] ? q2 : q1, wx[i], 0u, false);
                        const unsigned v = (h0 * (unsigned)taps[i].wy0 + h1 * (unsigned)taps[i].wy1 + 512u) >> 10;
                        w |= v << (8 * i);
                    }
                    packed[p] = w;
                }
            }
        }
        if (!fast) {
#pragma unroll
            for (int p = 0; p < SLR_MF_PLANES; p++) {
                unsigned w = 0;
#pragma unroll
                for (int i = 0; i < V; i++) w |= (unsigned)sample(pl.p[p], pitch, W, H, taps[i]) << (8 * i);
                packed[p] = w;
            }
        }
#pragma unroll
        for (int i = 0; i < V; i++) {
            int gpx[SLR_MF_PLANES];
#pragma unroll
            for (int p = 0; p < SLR_MF_PLANES; p++) gpx[p] = (packed[p] >> (8 * i)) & 0xFFu;
            int v;
            ph[i] = mf_pixel<X87>(gpx, black_thr, lut, v);
            if (!valid) ph[i] = v ? ph[i] : kInvalidPhase;
            vw |= (unsigned)v << (8 * i);
        }
        if constexpr (V == 4) {
            f32x4 o; o.x = ph[0]; o.y = ph[1]; o.z = ph[2]; o.w = ph[3];
            __builtin_nontemporal_store(o, reinterpret_cast<f32x4 *>(phase + m));
            if (valid) __builtin_nontemporal_store(vw, reinterpret_cast<unsigned *>(valid + m));
        } else {
            phase[m] = ph[0];
            if (valid) valid[m] = (uint8_t)vw;
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// cv::initUndistortRectifyMap(M, D, R, P, size, CV_16SC2) on the device (stereorect.cpp:42-43; SURVEY 8c-3 i, 8f-4).
// OpenCV walks a row with RUNNING SUMS (_x += iR00, ...), so column j depends on j-1: one lane owns one destination
// row and walks it left to right -- the f64 sequence is the oracle's, bit for bit -- and 64 rows x 64 columns go
// through LDS so that the stores are coalesced.  One-time work per calibration (~1 ms at 4096 x 3000).
// ------------------------------------------------------------------------------------------------------
struct MapGen { double ir[9], fx, fy, u0, v0, k1, k2, p1, p2, k3; };

__global__ __launch_bounds__(64) void init_rectify_map_kernel(MapGen g, int W, int H, int16_t *__restrict__ map_xy,
                                                              uint16_t *__restrict__ map_frac)
{
    __shared__ unsigned sxy[64][65];
    __shared__ unsigned short sfr[64][66];
    const int lane = threadIdx.x, i = blockIdx.x * 64 + lane;
    double _x = i * g.ir[1] + g.ir[2], _y = i * g.ir[4] + g.ir[5], _w = i * g.ir[7] + g.ir[8];
    for (int j0 = 0; j0 < W; j0 += 64) {
        const int nj = W - j0 < 64 ? W - j0 : 64;
        for (int jj = 0; jj < nj; jj++, _x += g.ir[0], _y += g.ir[3], _w += g.ir[6]) {
            const double w = 1. / _w, x = _x * w, y = _y * w;
            const double x2 = x * x, y2 = y * y;
            const double r2 = x2 + y2, _2xy = 2 * x * y;
            const double kr = 1 + ((g.k3 * r2 + g.k2) * r2 + g.k1) * r2;
            const double u = g.fx * (x * kr + g.p1 * _2xy + g.p2 * (r2 + 2 * x2)) + g.u0;
            const double v = g.fy * (y * kr + g.p1 * (r2 + 2 * y2) + g.p2 * _2xy) + g.v0;
            const long long iu = __double2ll_rn(u * 32), iv = __double2ll_rn(v * 32);   // cvRound: half to even
            sxy[lane][jj] = (unsigned)(unsigned short)(short)(iu >> 5) | ((unsigned)(unsigned short)(short)(iv >> 5) << 16);
            sfr[lane][jj] = (unsigned short)((iv & 31) * 32 + (iu & 31));
        }
        __syncthreads();
        for (int r = 0; r < 64; r++) {                       // row r of the block, 64 consecutive columns per store
            const int row = blockIdx.x * 64 + r;
            if (row < H && lane < nj) {
                const size_t m = (size_t)row * W + j0 + lane;
                reinterpret_cast<unsigned *>(map_xy)[m] = sxy[r][lane];
                map_frac[m] = sfr[r][lane];
            }
        }
        __syncthreads();
    }
}

hipError_t launch_init_rectify_map(const double M[9], const double D[5], const double R[9], const double P[12], int W, int H,
                                   int16_t *map_xy, uint16_t *map_frac, hipStream_t s)
{
    // iR = (P[:, :3] * R)^-1 by the adjugate, in f64 on the host (9 numbers)
    MapGen g;
    double A[9];
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) {
            double t = 0;
            for (int k = 0; k < 3; k++) t += P[r * 4 + k] * R[k * 3 + c];
            A[r * 3 + c] = t;
        }
    const double det = A[0] * (A[4] * A[8] - A[5] * A[7]) - A[1] * (A[3] * A[8] - A[5] * A[6]) +
                       A[2] * (A[3] * A[7] - A[4] * A[6]);
    const double d = 1. / det;
    g.ir[0] = (A[4] * A[8] - A[5] * A[7]) * d; g.ir[1] = (A[2] * A[7] - A[1] * A[8]) * d;
    g.ir[2] = (A[1] * A[5] - A[2] * A[4]) * d; g.ir[3] = (A[5] * A[6] - A[3] * A[8]) * d;
    g.ir[4] = (A[0] * A[8] - A[2] * A[6]) * d; g.ir[5] = (A[2] * A[3] - A[0] * A[5]) * d;
    g.ir[6] = (A[3] * A[7] - A[4] * A[6]) * d; g.ir[7] = (A[1] * A[6] - A[0] * A[7]) * d;
    g.ir[8] = (A[0] * A[4] - A[1] * A[3]) * d;
    g.fx = M[0]; g.fy = M[4]; g.u0 = M[2]; g.v0 = M[5];
    g.k1 = D[0]; g.k2 = D[1]; g.p1 = D[2]; g.p2 = D[3]; g.k3 = D[4];
    SLR_LAUNCH(init_rectify_map_kernel, dim3((unsigned)((H + 63) / 64)), dim3(64), 0, s, g, W, H, map_xy, map_frac);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------
// fused K1+K2, LDS-tiled form (the default): a workgroup rectifies + decodes 64 x 8 destination tiles (64 x 16 with
// SLR_OPT_RECT_DECODE_ALGO = 2), one pixel per lane, TH / 4 passes of four rows.
//   0. (once per map, at slr_set_rectify_maps / slr_init_rectify_maps) tile_boxes_kernel reduces every tile's map
//      entries to the bounding box of its source footprints with wave shuffles (a smooth map turns 64 x 8 into
//      roughly 70 x 10 source pixels);
//   1. the box of all 14 planes goes HBM -> registers -> LDS with coalesced dword loads (14 independent loads per
//      thread and round); everything outside the image is stored as 0, which IS cv::remap's BORDER_CONSTANT, so the
//      border cases need no special code at all.  LDS layout: [row][dword column][plane], i.e. the 14 planes of one
//      source dword sit next to each other -> every tap read is base + immediate (one ds_read2_b64 per plane);
//   2. per pixel the 2x2 footprint of each plane is two dword pairs; v_perm_b32 cuts the two bytes out as a u16
//      pair and two v_dot2_u32_u16 with the 16-bit weights wx*wy*64 accumulate the whole bilinear sum; the sample
//      (dot2(row1, w1, dot2(row0, w0, 512 << 6))) >> 16 == OpenCV's (sum tap*w + 16384) >> 15 exactly, and it stays
//      in the accumulator's high half-word for the consumer (SDWA operand select instead of a shift);
//   3. the same per-pixel decode as K2.
// The workgroups are persistent and software-pipelined (see below); an XCD-banded tile walk keeps horizontally
// adjacent tiles on one L2.  PMC: 64 x 16 tiles fetch 1.01..1.07x the ideal bytes, 64 x 8 tiles 1.13..1.17x (their two
// halo rows are re-read from HBM) but run 7 % faster (5 instead of 4 workgroups per CU).  A box that does not fit
// (wild maps) falls back, per tile, to the direct gather of the generic kernel.
// ------------------------------------------------------------------------------------------------------
constexpr int kTileW = 64, kTileH = 16;

// int4 per tile: x = x0 (multiple of 4), y = y0, z = BW4 (dwords per row), w = BH (rows); z == 0: no footprint
__global__ __launch_bounds__(256) void tile_boxes_kernel(const int16_t *__restrict__ map_xy, int W, int H, int tiles_x,
                                                         int tile_h, int4 *__restrict__ boxes)
{
    __shared__ int red[4][4];
    const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int col = tx * kTileW + lane;
    int mnx = 0x7FFFFFFF, mxx = -0x7FFFFFFF, mny = 0x7FFFFFFF, mxy = -0x7FFFFFFF;
    for (int q = 0; q < tile_h / 4; q++) {
        const int row = ty * tile_h + 4 * q + wv;
        if (row < H && col < W) {
            const size_t m = (size_t)row * W + col;
            const int sx = map_xy[2 * m], sy = map_xy[2 * m + 1];
            if (!(sx >= W || sx + 1 < 0 || sy >= H || sy + 1 < 0)) {       // footprints completely outside read 0
                mnx = sx < mnx ? sx : mnx; mxx = sx > mxx ? sx : mxx;
                mny = sy < mny ? sy : mny; mxy = sy > mxy ? sy : mxy;
            }
        }
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
        for (int w = 1; w < 4; w++) {
            mnx = red[w][0] < mnx ? red[w][0] : mnx; mxx = red[w][1] > mxx ? red[w][1] : mxx;
            mny = red[w][2] < mny ? red[w][2] : mny; mxy = red[w][3] > mxy ? red[w][3] : mxy;
        }
        int4 b = make_int4(0, 0, 0, 0);
        if (mnx <= mxx) {
            b.x = mnx & ~3;                                  // dword-aligned origin (also for negative x)
            b.y = mny;
            b.z = (((mxx + 1) - b.x + 1) + 3) >> 2;          // columns x0 .. mxx+1 in dwords
            b.w = (mxy + 1) - mny + 1;                       // rows    y0 .. mxy+1
        }
        boxes[blockIdx.x] = b;
    }
}

// two tables in one buffer: 64x16 tiles (multi-frequency kernel: 14 planes) then 64x4 tiles (Gray kernel: 22..66 planes)
constexpr int kGrayTileH = 4, kMidTileH = 8;
static size_t tile_count(int W, int H, int tile_h)
{
    return (size_t)((W + kTileW - 1) / kTileW) * ((H + tile_h - 1) / tile_h);
}
static size_t wide_tile_count(int W, int H)
{
    return (size_t)((W + 2 * kTileW - 1) / (2 * kTileW)) * ((H + kMidTileH - 1) / kMidTileH);
}

// the box of the 128 x 8 tile (ty, tx): union of the 64 x 8 table entries (ty, 2 tx) and (ty, 2 tx + 1)
__device__ __forceinline__ int4 union_box(const int4 *__restrict__ boxes, int ty, int tx, int tiles_x64)
{
    const int t0 = ty * tiles_x64 + 2 * tx;
    const int4 a = boxes[t0];
    const int4 b = boxes[2 * tx + 1 < tiles_x64 ? t0 + 1 : t0];
    const int4 l = a.z > 0 ? a : b, r = b.z > 0 ? b : a;               // an empty half takes the other half's box
    const int x0 = l.x < r.x ? l.x : r.x, y0 = l.y < r.y ? l.y : r.y;
    const int x1l = l.x + 4 * l.z, x1r = r.x + 4 * r.z, y1l = l.y + l.w, y1r = r.y + r.w;
    return make_int4(x0, y0, ((x1l > x1r ? x1l : x1r) - x0) >> 2, (y1l > y1r ? y1l : y1r) - y0);
}

// The same buffer also carries a TILED, PRE-DIGESTED copy of the map for the 64 x 8 kernel: the 512 entries of a tile are
// contiguous and ordered (pass, wave, lane) = the order in which the workgroup's threads consume them, so a wave reads
// 256 contiguous bytes per pass instead of 8 x 256 B pieces 16 KB apart (rows of a 4096-wide map are a power of two
// apart: the same HBM channel and bank) -- and an entry is ONE dword instead of the map's 4 + 2 bytes, already relative
// to the tile's source box (tile_boxes_kernel), which is what the kernel would compute from it anyway:
//   [9:0]   dword index of the tap's upper left byte in the box: (sy - y0) * BW4 + ((sx - x0) >> 2)
//   [11:10] (sx - x0) & 3        [16:12] fx        [21:17] fy        (cv::remap's 5-bit fractions)
//   [31]    the sample is 0: pixel beyond the ragged image edge, or footprint completely outside the source
// Tiles whose box does not fit the kernel's LDS budget are decoded by the gather fallback from the caller's map.
constexpr int kMidBudget = 12 * 1024;                      // LDS bytes of a 64 x 8 tile's box: 14 planes x ~72 x 11
__device__ __host__ inline bool mid_box_fits(int BW4, int BH)
{
    return BW4 > 0 && BW4 * BH <= 256 && BW4 * BH * (SLR_MF_PLANES * 4) <= kMidBudget;
}
static size_t tiled_map_offset(int W, int H)
{
    const size_t b = (tile_count(W, H, kTileH) + tile_count(W, H, kGrayTileH) + tile_count(W, H, kMidTileH)) * sizeof(int4);
    return (b + 255) & ~(size_t)255;
}
// behind the two digests: two counters -- tiles with a footprint whose box does NOT fit the 64 x 8 form / the 128 x 8 form
// (those tiles run the per-pixel gather fallback; the C ABI picks the form with fewer of them, see launch_tile_boxes)
static size_t tile_nofit_offset(int W, int H)
{
    return tiled_map_offset(W, H) + (tile_count(W, H, kMidTileH) * 512 + wide_tile_count(W, H) * 1024) * sizeof(unsigned);
}
size_t tile_boxes_bytes(int W, int H)
{
    return tile_nofit_offset(W, H) + 16;
}

__global__ __launch_bounds__(256) void tile_maps_kernel(const int16_t *__restrict__ map_xy, const uint16_t *__restrict__ map_frac,
                                                        int W, int H, int tiles_x, const int4 *__restrict__ boxes8,
                                                        unsigned *__restrict__ pk_t, unsigned *__restrict__ nofit)
{
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int4 box = boxes8[blockIdx.x];
    const bool fits = mid_box_fits(box.z, box.w);
    if (threadIdx.x == 0 && box.z > 0 && !fits) atomicAdd(nofit, 1u);
#pragma unroll
    for (int q = 0; q < kMidTileH / 4; q++) {
        const int row = ty * kMidTileH + 4 * q + wv, col = tx * kTileW + lane;
        unsigned e = 0x80000000u;
        if (fits && row < H && col < W) {
            const size_t m = (size_t)row * W + col;
            const int sx = map_xy[2 * m], sy = map_xy[2 * m + 1];
            const unsigned fr = map_frac[m];
            if (!(sx >= W || sx + 1 < 0 || sy >= H || sy + 1 < 0)) {
                const int bx = sx - box.x, r0 = sy - box.y;
                e = (unsigned)(r0 * box.z + (bx >> 2)) | ((unsigned)bx & 3u) << 10 | (fr & 1023u) << 12;
            }
        }
        pk_t[(size_t)blockIdx.x * 512 + q * 256 + threadIdx.x] = e;
    }
}

// the same digest for the 128 x 8 form: entries relative to the union box, 1024 per tile in (pass, wave, lane) order of
// a 512-thread workgroup (waves 0-3: left 64 columns, waves 4-7: right 64 columns)
constexpr int kWideBudget = 28 * 1024;
__global__ __launch_bounds__(512) void tile_maps_wide_kernel(const int16_t *__restrict__ map_xy, const uint16_t *__restrict__ map_frac,
                                                             int W, int H, int tiles_x128, int tiles_x64,
                                                             const int4 *__restrict__ boxes8, unsigned *__restrict__ pk_t,
                                                             unsigned *__restrict__ nofit)
{
    const int ty = blockIdx.x / tiles_x128, tx = blockIdx.x - ty * tiles_x128;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int4 box = union_box(boxes8, ty, tx, tiles_x64);
    const bool fits = box.z > 0 && box.z * box.w <= 512 && box.z * box.w * (SLR_MF_PLANES * 4) <= kWideBudget;
    if (threadIdx.x == 0 && box.z > 0 && !fits) atomicAdd(nofit, 1u);
#pragma unroll
    for (int q = 0; q < kMidTileH / 4; q++) {
        const int row = ty * kMidTileH + 4 * q + (wv & 3), col = (2 * tx + (wv >> 2)) * kTileW + lane;
        unsigned e = 0x80000000u;
        if (fits && row < H && col < W) {
            const size_t m = (size_t)row * W + col;
            const int sx = map_xy[2 * m], sy = map_xy[2 * m + 1];
            const unsigned fr = map_frac[m];
            if (!(sx >= W || sx + 1 < 0 || sy >= H || sy + 1 < 0)) {
                const int bx = sx - box.x, r0 = sy - box.y;
                e = (unsigned)(r0 * box.z + (bx >> 2)) | ((unsigned)bx & 3u) << 10 | (fr & 1023u) << 12;
            }
        }
        pk_t[(size_t)blockIdx.x * 1024 + q * 512 + threadIdx.x] = e;
    }
}
hipError_t launch_tile_boxes(const int16_t *map_xy, const uint16_t *map_frac, int W, int H, int4 *boxes, unsigned nofit_host[2],
                             hipStream_t s)
{
    unsigned *nofit = reinterpret_cast<unsigned *>(reinterpret_cast<char *>(boxes) + tile_nofit_offset(W, H));
    hipError_t me = hipMemsetAsync(nofit, 0, 16, s);
    if (me != hipSuccess) return me;
    const int tiles_x = (W + kTileW - 1) / kTileW;
    SLR_LAUNCH(tile_boxes_kernel, dim3((unsigned)tile_count(W, H, kTileH)), dim3(256), 0, s, map_xy, W, H, tiles_x,
                       kTileH, boxes);
    SLR_LAUNCH(tile_boxes_kernel, dim3((unsigned)tile_count(W, H, kGrayTileH)), dim3(256), 0, s, map_xy, W, H, tiles_x,
                       kGrayTileH, boxes + tile_count(W, H, kTileH));
    SLR_LAUNCH(tile_boxes_kernel, dim3((unsigned)tile_count(W, H, kMidTileH)), dim3(256), 0, s, map_xy, W, H, tiles_x,
                       kMidTileH, boxes + tile_count(W, H, kTileH) + tile_count(W, H, kGrayTileH));
    unsigned *pk_t = reinterpret_cast<unsigned *>(reinterpret_cast<char *>(boxes) + tiled_map_offset(W, H));
    SLR_LAUNCH(tile_maps_kernel, dim3((unsigned)tile_count(W, H, kMidTileH)), dim3(256), 0, s, map_xy, map_frac, W, H,
                       tiles_x, boxes + tile_count(W, H, kTileH) + tile_count(W, H, kGrayTileH), pk_t, nofit);
    SLR_LAUNCH(tile_maps_wide_kernel, dim3((unsigned)wide_tile_count(W, H)), dim3(512), 0, s, map_xy, map_frac, W, H,
                       (W + 2 * kTileW - 1) / (2 * kTileW), tiles_x, boxes + tile_count(W, H, kTileH) + tile_count(W, H, kGrayTileH),
                       pk_t + tile_count(W, H, kMidTileH) * 512, nofit + 1);
    me = hipGetLastError();
    if (me != hipSuccess) return me;
    me = hipMemcpyAsync(nofit_host, nofit, 2 * sizeof(unsigned), hipMemcpyDeviceToHost, s);   // the caller synchronises the stream
    return me;
}
// one dword of a plane at (gx..gx+3, gy), zero outside the image; gx is a multiple of 4
__device__ __forceinline__ unsigned load_src_dword(const uint8_t *__restrict__ plane, int pitch, int W, int H, int gx, int gy)
{
    if ((unsigned)gy >= (unsigned)H) return 0u;
    const uint8_t *q = plane + (size_t)gy * pitch + gx;
    unsigned v = 0;
#pragma unroll
    for (int b = 0; b < 4; b++)
        if ((unsigned)(gx + b) < (unsigned)W) v |= (unsigned)q[b] << (8 * b);
    return v;
}


template <int NP>
struct TileTaps {                        // per-pixel tap state shared by all planes
    int a0, a1;                          // LDS byte addresses of the dword pair in source rows sy and sy+1 (plane 0)
    unsigned sel;                        // v_perm selector: bytes (sh, 0, sh+1, 0) -> u16 pair of the two taps
    u16x2 w0, w1;                        // wx0*wy0, wx1*wy0 | wx0*wy1, wx1*wy1 (<= 1024 each)
};

// tap state straight from one map entry (xy = x | y << 16, frac = fy << 5 | fx), without the generic Tap:
// ~20 VALU instructions.  Lanes outside the image (inb == false) and footprints completely outside the source get
// zero weights and address 0, i.e. sample 0 (BORDER_CONSTANT).
template <int NP>
__device__ __forceinline__ TileTaps<NP> tile_taps_map(unsigned xy, unsigned frac, bool inb, int W, int H, int x0, int y0,
                                                      int BW4)
{
    TileTaps<NP> k;
    const int sx = (int)(short)(xy & 0xFFFFu), sy = (int)xy >> 16;
    // sx >= W || sx + 1 < 0  <=>  (unsigned)(sx + 1) > (unsigned)W   (same for y)
    const bool out = !inb || (unsigned)(sx + 1) > (unsigned)W || (unsigned)(sy + 1) > (unsigned)H;
    const unsigned fx = frac & 31u, fy = (frac >> 5) & 31u;
    const unsigned wxp = out ? 0u : __umul24(fx, 0xFFFFu) + 32u;          // (32 - fx) | fx << 16
    // weights x 64 so that the blended sample is the HIGH HALF of the accumulator (tile_sample): every product is
    // < 65536 except wx0*wy0*64 at fx = fy = 0, which is 65536 and would carry into the other half; there the other
    // three weights are 0 and 65535 gives the same sample: (S*65535 + 32768) >> 16 == S for S <= 255.
    unsigned w0 = __umul24(wxp, (32u - fy) << 6);
    const unsigned w1 = __umul24(wxp, fy << 6);
    w0 = w0 == 0x10000u ? 0xFFFFu : w0;
    const int bx = out ? 0 : sx - x0, r0 = out ? 0 : sy - y0;
    k.a0 = __mul24(__mul24(r0, BW4) + (bx >> 2), NP * 4);
    k.a1 = k.a0 + __mul24(BW4, NP * 4);
    k.sel = __umul24((unsigned)bx & 3u, 0x10001u) + 0x0C010C00u;          // bytes (sh, 0, sh+1, 0)
    k.w0 = __builtin_bit_cast(u16x2, w0);
    k.w1 = __builtin_bit_cast(u16x2, w1);
    return k;
}

// the same tap state from one pre-digested entry of the tiled map (tile_maps_kernel)
template <int NP>
__device__ __forceinline__ TileTaps<NP> tile_taps_packed(unsigned e, int BW4)
{
    TileTaps<NP> k;
    const unsigned fx = (e >> 12) & 31u, fy6 = (e >> 11) & 0x7C0u;         // fy << 6
    const unsigned wxp = (int)e < 0 ? 0u : __umul24(fx, 0xFFFFu) + 32u;     // (32 - fx) | fx << 16
    unsigned w0 = __umul24(wxp, 2048u - fy6);                             // see tile_taps_map for the x 64 and the 0xFFFF
    const unsigned w1 = __umul24(wxp, fy6);
    w0 = w0 == 0x10000u ? 0xFFFFu : w0;
    k.a0 = (int)__umul24(e & 1023u, NP * 4);
    k.a1 = k.a0 + __mul24(BW4, NP * 4);
    k.sel = __umul24((e >> 10) & 3u, 0x10001u) + 0x0C010C00u;
    k.w0 = __builtin_bit_cast(u16x2, w0);
    k.w1 = __builtin_bit_cast(u16x2, w1);
    return k;
}

// blended sample of plane p: LDS reads are (base + immediate), 2 perms, 2 dot2, 1 shift
template <int NP, int WSHIFT>
__device__ __forceinline__ int tile_sample(const uint8_t *tile, const TileTaps<NP> &k, int p)
{
    const unsigned *q0 = reinterpret_cast<const unsigned *>(tile + k.a0);
    const unsigned *q1 = reinterpret_cast<const unsigned *>(tile + k.a1);
    const unsigned p0 = __builtin_amdgcn_perm(q0[p + NP], q0[p], k.sel);
    const unsigned p1 = __builtin_amdgcn_perm(q1[p + NP], q1[p], k.sel);
    unsigned acc = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, p0), k.w0, 512u << WSHIFT, false);
    acc = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, p1), k.w1, acc, false);
    return (int)acc;                     // the sample is acc >> (10 + WSHIFT); WSHIFT == 6: the high half-word
}

// Software pipeline (round 1 finding: with the frame NOT resident in the 256 MB Infinity Cache -- the real pipeline
// alternates two 172 MB camera stacks -- the non-pipelined form was bound by HBM latency x the few workgroups that
// happen to be in their fill phase: 129 us vs 100 us warm).  The workgroups are persistent; while a tile is being
// decoded out of LDS, the global loads of the NEXT tile's box are already in flight into registers (2 rounds x 14
// planes per thread) and are committed to LDS after the barrier that ends the current tile.  gfx9 has ONE in-order
// counter for vector memory (vmcnt), so the loads a tile's decode itself needs (its 4x2 map entries) are issued
// BEFORE the prefetch, and the prefetch is branch-free (out-of-box elements read a dummy address) so that the
// compiler's s_waitcnt for the map entries is vmcnt(28) on every path and never drains the prefetch.
struct BoxGeom { int x0, y0, BW4, BH; bool any, fits; };

// TWV == 2: the tile is 128 x TH, the union of two horizontally adjacent 64 x TH tiles of the box table (`tile` counts
// 128-wide tiles, tiles_x of them per row; the table has tiles_x64 per row)
template <int NP, int ROUNDS, int TWV, int NT>
__device__ __forceinline__ BoxGeom box_geom(const int4 *__restrict__ boxes, int tile, int budget, int tiles_x, int tiles_x64)
{
    int4 box;
    if constexpr (TWV == 1) box = boxes[tile];
    else { const int ty = tile / tiles_x; box = union_box(boxes, ty, tile - ty * tiles_x, tiles_x64); }
    BoxGeom g;
    g.x0 = box.x; g.y0 = box.y; g.BW4 = box.z; g.BH = box.w;
    g.any = g.BW4 > 0;
    // the LDS budget, and the prefetch registers: at most ROUNDS x NT dwords per plane
    g.fits = g.any && g.BW4 * g.BH <= NT * ROUNDS && g.BW4 * g.BH * (NP * 4) <= budget;
    return g;
}

// One launch serves one camera (njobs == 1) or both cameras of a stereo frame (njobs == 2: the first half of the grid
// decodes job 0, the second half job 1 -- one launch gap and one tail less per frame).
struct RectJob {
    MfPlanes pl;
    unsigned plane_stride;       // > 0: the 14 planes are pl.p[0] + i * plane_stride (one buffer descriptor serves all)
    const int16_t *map_xy;
    const uint16_t *map_frac;
    const int4 *boxes;
    float *phase;
    uint8_t *valid;
    const unsigned *pk_t;        // tiled, pre-digested copy of the map for 64 x 8 tiles (see tile_maps_kernel), or null
};

struct RectJobs { RectJob j[2]; };

// STRIDED: the planes of a job are equally spaced in ONE allocation (every stack this library stages itself, and the
// batch entry point).  The box is then fetched with raw buffer loads: one descriptor per job, the plane as the scalar
// offset -- no per-plane 64-bit address arithmetic -- and everything outside the image is given an out-of-range offset,
// for which the hardware returns 0 (= BORDER_CONSTANT): no select on the way into LDS either.
// NT == 512 (with TWV == 2): eight waves, waves 0-3 decode the left 64 columns of a 128 x 8 tile and waves 4-7 the right
// ones out of ONE box that all 512 threads fetch in one round -- the source row segments are twice as long (fewer,
// fuller cache lines per byte, see profiles/exp/boxread.hip) at the register cost of the 64 x 8 form.
template <int TH, int ROUNDS, bool STRIDED, int TWV = 1, bool PACKED = false, int NT = 256>
__global__ __launch_bounds__(NT, (NT == 512 ? 6 : ROUNDS == 1 && STRIDED ? 5 : 4)) void mf_rect_decode_lds_kernel(RectJobs jobs, int njobs, int pitch, int W, int H,
                                                                 int black_thr, const float *__restrict__ lut_g,
                                                                 int tiles_x, int tiles_y, int budget)
{
    constexpr int NP = SLR_MF_PLANES;
    extern __shared__ __attribute__((aligned(16))) uint8_t tile[];
    __shared__ float lut[kLutWords + 1];
    load_lut(lut, lut_g);
    const unsigned nblk = gridDim.x / (unsigned)njobs;      // workgroups per job (a multiple of 8)
    const bool second = blockIdx.x >= nblk;
    const unsigned bid = second ? blockIdx.x - nblk : blockIdx.x;
    // the job table stays in the kernarg segment and is indexed there (scalar loads on demand): no local copy (a
    // dynamically indexed local array would live in scratch) and no 2 x 19 pointers held in SGPRs
    const int ji = second ? 1 : 0;
    auto plane = [&](int p) -> const uint8_t * {
        if constexpr (STRIDED) return jobs.j[ji].pl.p[0] + (size_t)p * jobs.j[ji].plane_stride;
        else return jobs.j[ji].pl.p[p];
    };
    const unsigned pstride = jobs.j[ji].plane_stride;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void *)jobs.j[ji].pl.p[0], 0, (int)((NP - 1) * pstride + (unsigned)H * (unsigned)pitch), 0x00020000);
    const int16_t *__restrict__ map_xy = jobs.j[ji].map_xy;
    const uint16_t *__restrict__ map_frac = jobs.j[ji].map_frac;
    const int4 *__restrict__ boxes = jobs.j[ji].boxes;
    float *__restrict__ phase = jobs.j[ji].phase;
    uint8_t *__restrict__ valid = jobs.j[ji].valid;
    const unsigned *__restrict__ pk_t = jobs.j[ji].pk_t;
    constexpr bool packed = PACKED;                         // the launcher passes pk_t != null with PACKED only
    static_assert(!PACKED || (TH == kMidTileH && ROUNDS == 1 && NT == 256 * TWV), "the pre-digested maps serve 64 x 8 / 128 x 8 tiles");
    static_assert(NT == 256 || (NT == 512 && TWV == 2), "eight waves <=> 128-wide tiles");
    // Tile schedule.  Workgroup b runs on XCD b % 8 (round-robin dispatch); XCD x owns the band of `per` consecutive
    // tiles (row-major) [x*per, (x+1)*per), and its nbx workgroups walk the band together: in step i they decode the
    // nbx consecutive tiles starting at x*per + i*nbx, so tiles that share source rows meet in one L2.
    const int T = tiles_x * tiles_y, per = (T + 7) >> 3;
    const int xcd = (int)(bid & 7u), lb = (int)(bid >> 3), nbx = (int)(nblk >> 3);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const unsigned lane31 = (unsigned)lane & 31u;

    // element e of a box (row-major dwords) that thread `tid` owns in round r: e = tid + 256 r
    unsigned pre[ROUNDS][NP];
    auto issue = [&](const BoxGeom &g, bool live) {
        const int E = live && g.fits ? g.BH * g.BW4 : 0;
        const float inv = __builtin_amdgcn_rcpf((float)(g.BW4 > 0 ? g.BW4 : 1));   // v_rcp_f32: 1 ulp, see the margin below
#pragma unroll
        for (int r = 0; r < ROUNDS; r++) {
            const int e = (int)threadIdx.x + NT * r;
            const int rr = (int)(((float)e + 0.5f) * inv);  // e / BW4 (exact: e < 2^20, remainder margin 0.5/BW4)
            const int cc = e - rr * g.BW4;
            const int gx = g.x0 + 4 * cc, gy = g.y0 + rr;
            // W % 4 == 0 and gx % 4 == 0: a dword is completely inside or completely outside the image
            const bool in = e < E && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
            // branch-free 32-bit offset: keeps the loads in the scalar-base + 32-bit-VGPR-offset addressing form
            if constexpr (STRIDED) {
                const unsigned off = in ? __umul24((unsigned)gy, (unsigned)pitch) + (unsigned)gx : 0xFFFFFFF0u;   // out of range -> 0
#pragma unroll
                for (int p = 0; p < NP; p++)
                    pre[r][p] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)off, (int)(p * pstride), 0);
            } else {
                const unsigned off = (__umul24((unsigned)gy, (unsigned)pitch) + (unsigned)gx) & (0u - (unsigned)in);
#pragma unroll
                for (int p = 0; p < NP; p++) pre[r][p] = *reinterpret_cast<const unsigned *>(plane(p) + off);
            }
        }
    };
    auto commit = [&](const BoxGeom &g) {                   // same predicate as issue(); outside the image -> 0
        const int E = g.BH * g.BW4;
        const float inv = __builtin_amdgcn_rcpf((float)g.BW4);
#pragma unroll
        for (int r = 0; r < ROUNDS; r++) {
            const int e = (int)threadIdx.x + NT * r;
            if (e < E) {
                const int rr = (int)(((float)e + 0.5f) * inv);
                const int cc = e - rr * g.BW4;
                const int gx = g.x0 + 4 * cc, gy = g.y0 + rr;
                const bool in = STRIDED || ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W);   // STRIDED: zeros came from the load
                u32x2 *dst = reinterpret_cast<u32x2 *>(tile + (size_t)e * (NP * 4));
#pragma unroll
                for (int p = 0; p < NP; p += 2) {
                    u32x2 w2;
                    w2.x = in ? pre[r][p] : 0u; w2.y = in ? pre[r][p + 1] : 0u;
                    dst[p >> 1] = w2;
                }
            }
        }
    };

    if (lb >= per || xcd * per + lb >= T) return;           // (whole workgroup) nothing to do
    int cur = xcd * per + lb;
    const int tiles_x64 = (W + kTileW - 1) / kTileW;
    constexpr int QH = TH / 4, NPASS = NT == 512 ? QH : QH * TWV;   // passes per 64-wide half, passes of a wave per tile
    const int wrow = NT == 512 ? (wv & 3) : wv;             // the wave's row inside a pass
    BoxGeom gc = box_geom<NP, ROUNDS, TWV, NT>(boxes, cur, budget, tiles_x, tiles_x64);
    issue(gc, true);
    for (int it = 1;; it++) {
        if (gc.fits) commit(gc);
        __syncthreads();
        const int nl = lb + it * nbx;
        const bool has_next = nl < per && xcd * per + nl < T;
        const int nxt = has_next ? xcd * per + nl : cur;
        const BoxGeom gn = box_geom<NP, ROUNDS, TWV, NT>(boxes, nxt, budget, tiles_x, tiles_x64);

        const int ty = cur / tiles_x, tx = cur - ty * tiles_x;
        // map entries of all passes first (see the header: they must be older than the prefetch)
        unsigned xy[NPASS], fr[NPASS];
#pragma unroll
        for (int qq = 0; qq < NPASS; qq++) {
            const int hx = NT == 512 ? (wv >> 2) : qq / QH, q = NT == 512 ? qq : qq - hx * QH;
            const int col = (tx * TWV + hx) * kTileW + lane;
            if constexpr (packed) {                         // tiled map: contiguous per 64 x 8 tile, padded -> no bounds
                const unsigned d = (unsigned)cur * (unsigned)(2 * NT) + (unsigned)(q * NT) + threadIdx.x;
                xy[qq] = *reinterpret_cast<const unsigned *>(reinterpret_cast<const char *>(pk_t) + d * 4u);
                fr[qq] = 0;
            } else {
                const int row = ty * TH + 4 * q + wrow;
                const bool inb = row < H && col < W;
                const unsigned m = inb ? (unsigned)row * (unsigned)W + (unsigned)col : 0u;
                xy[qq] = *reinterpret_cast<const unsigned *>(reinterpret_cast<const char *>(map_xy) + m * 4u);
                fr[qq] = *reinterpret_cast<const uint16_t *>(reinterpret_cast<const char *>(map_frac) + m * 2u);
            }
        }
        issue(gn, has_next);
#pragma unroll
        for (int qq = 0; qq < NPASS; qq++) {
            const int hx = NT == 512 ? (wv >> 2) : qq / QH, q = NT == 512 ? qq : qq - hx * QH;
            const int col = (tx * TWV + hx) * kTileW + lane;
            const int row = ty * TH + 4 * q + wrow;
            const bool inb = row < H && col < W;
            const unsigned m = (unsigned)row * (unsigned)W + (unsigned)col;
            int v;
            float ph;
            if (gc.fits) {
                TileTaps<NP> k;
                if constexpr (packed) k = tile_taps_packed<NP>(xy[qq], gc.BW4);
                else k = tile_taps_map<NP>(xy[qq], fr[qq], inb, W, H, gc.x0, gc.y0, gc.BW4);
                int acc[NP];
#pragma unroll
                for (int p = 0; p < NP; p++) acc[p] = tile_sample<NP, 6>(tile, k, p);
                ph = mf_pixel_sh<16>(acc, black_thr, lut, v);
            } else {                                        // wild map: direct gather for this tile
                unsigned rxy = xy[qq], rfr = fr[qq];
                if constexpr (packed) {                     // (the pre-digested entries only serve boxes that fit)
                    const unsigned mm = inb ? m : 0u;
                    rxy = *reinterpret_cast<const unsigned *>(reinterpret_cast<const char *>(map_xy) + mm * 4u);
                    rfr = *reinterpret_cast<const uint16_t *>(reinterpret_cast<const char *>(map_frac) + mm * 2u);
                }
                Tap t = make_tap((int)(short)(rxy & 0xFFFFu), (int)rxy >> 16, rfr, pitch, W, H);
                if (!inb) t.kind = 1;
                int gpx[NP];
#pragma unroll 1
                for (int p = 0; p < NP; p++) gpx[p] = gc.any ? sample(plane(p), pitch, W, H, t) : 0;
                ph = mf_pixel(gpx, black_thr, lut, v);
            }
            // valid bytes of 4 neighbouring lanes -> one dword store by every 4th lane: the wave's ballot, this lane's
            // nibble of it, and a multiply that spreads 4 bits into 4 bytes (bit i -> bit 8i; the partial products of
            // 1 + 2^7 + 2^14 + 2^21 do not overlap)
            if (valid) {
                const unsigned long long bal = __ballot(v != 0);
                const unsigned half = (lane & 32) ? (unsigned)(bal >> 32) : (unsigned)bal;
                const unsigned vw = __umul24((half >> lane31) & 0xFu, 0x204081u) & 0x01010101u;
                if (inb && (lane & 3) == 0) __builtin_nontemporal_store(vw, reinterpret_cast<unsigned *>(valid + m));
            } else {
                ph = v ? ph : kInvalidPhase;                // folded flag (see launch_mf_decode)
            }
            if (inb) __builtin_nontemporal_store(ph, reinterpret_cast<float *>(reinterpret_cast<char *>(phase) + m * 4u));
        }
        if (!has_next) break;
        __syncthreads();                                    // everybody is done reading this tile's LDS
        cur = nxt;
        gc = gn;
    }
}

#ifdef SLR_ALL_FORMS
// ------------------------------------------------------------------------------------------------------
// fused K1+K2, sliding-window form (SLR_OPT_RECT_DECODE_ALGO = 3): the pipelined kernel above re-reads the two halo
// rows of every 64 x 8 tile from HBM (1.13..1.17x the ideal traffic).  Here a workgroup walks DOWN a tile column and
// keeps the source rows in a 16-row LDS ring (slot = source row & 15): the next tile only loads the rows below the
// ones already there.  Work item = (tile column, vertical segment of the XCD's band of tile rows); the items of one
// segment are handed to consecutive workgroups, so horizontally adjacent tiles are decoded at the same time and share
// their source lines in L2.  All tiles of an item use one LDS geometry (the union of their source column ranges).
// ------------------------------------------------------------------------------------------------------
constexpr int kRingRows = 16, kRingBW = 20;               // ring: 16 source rows x <= 20 dwords x 14 planes = 17.9 KB

struct RingTile {                                        // one 64 x 8 tile and the LDS geometry of its item
    int tx, ty, item;
    int x0, bw;                                          // item geometry: source columns [x0, x0 + 4 bw)
    int y0, bh;                                          // this tile's source rows [y0, y0 + bh)
    bool any, ok, valid;                                 // has a footprint / the item fits the ring / tile exists
};

__global__ __launch_bounds__(256, 5) void mf_rect_decode_ring_kernel(RectJobs jobs, int njobs, int pitch, int W, int H,
                                                                     int black_thr, const float *__restrict__ lut_g,
                                                                     int tiles_x, int tiles_y, int seg_len)
{
    constexpr int NP = SLR_MF_PLANES, TH = kMidTileH;
    extern __shared__ __attribute__((aligned(16))) uint8_t tile[];
    __shared__ float lut[kLutWords + 1];
    load_lut(lut, lut_g);
    const unsigned nblk = gridDim.x / (unsigned)njobs;
    const bool second = blockIdx.x >= nblk;
    const unsigned bid = second ? blockIdx.x - nblk : blockIdx.x;
    const int ji = second ? 1 : 0;
    auto plane = [&](int p) -> const uint8_t * { return jobs.j[ji].pl.p[p]; };
    const int16_t *__restrict__ map_xy = jobs.j[ji].map_xy;
    const uint16_t *__restrict__ map_frac = jobs.j[ji].map_frac;
    const int4 *__restrict__ boxes = jobs.j[ji].boxes;
    float *__restrict__ phase = jobs.j[ji].phase;
    uint8_t *__restrict__ valid = jobs.j[ji].valid;
    const int xcd = (int)(bid & 7u), lb = (int)(bid >> 3), nbx = (int)(nblk >> 3);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const unsigned lane31 = (unsigned)lane & 31u;

    // schedule: XCD x owns tile rows [band0, band1); items = columns x segments of seg_len tile rows, segment-major
    const int R = (tiles_y + 7) >> 3;
    const int band0 = xcd * R, band1 = band0 + R < tiles_y ? band0 + R : tiles_y;
    const int nseg = band1 > band0 ? (band1 - band0 + seg_len - 1) / seg_len : 0;
    const int nitems = nseg * tiles_x;

    auto enter_item = [&](int item, RingTile &t) {         // geometry of an item = union over its tiles
        t.item = item;
        t.valid = item < nitems;
        if (!t.valid) return;
        const int seg = item / tiles_x;
        t.tx = item - seg * tiles_x;
        const int r0 = band0 + seg * seg_len, r1 = r0 + seg_len < band1 ? r0 + seg_len : band1;
        int xa = 0x7FFFFFFF, xb = -0x7FFFFFFF, mh = 0;
        for (int r = r0; r < r1; r++) {
            const int4 b = boxes[r * tiles_x + t.tx];
            if (b.z > 0) {
                xa = b.x < xa ? b.x : xa;
                xb = b.x + 4 * b.z > xb ? b.x + 4 * b.z : xb;
                mh = b.w > mh ? b.w : mh;
            }
        }
        t.x0 = xb > xa ? xa : 0;
        t.bw = xb > xa ? (xb - xa) >> 2 : 1;
        t.ok = t.bw <= kRingBW && mh <= kRingRows - 2 && t.bw * mh <= 256;
        t.ty = r0;
    };
    auto load_box = [&](RingTile &t) {
        const int4 b = boxes[t.ty * tiles_x + t.tx];
        t.any = b.z > 0;
        t.y0 = b.y; t.bh = b.w;
    };
    auto advance = [&](const RingTile &c, RingTile &n) {   // the tile after c in this workgroup's walk
        n = c;
        const int seg = c.item / tiles_x;
        const int r1 = band0 + (seg + 1) * seg_len < band1 ? band0 + (seg + 1) * seg_len : band1;
        if (c.ty + 1 < r1) n.ty = c.ty + 1;
        else enter_item(c.item + nbx, n);
        if (n.valid) load_box(n);
    };

    // ring state (uniform): rows [lo, hi) of geometry (rx0, rbw) are in LDS
    int lo = 0, hi = 0, rx0 = 0, rbw = -1;
    struct Ld { int y0, y1, x0, bw; bool on; };
    auto plan = [&](const RingTile &t, int plo, int phi, int px0, int pbw) -> Ld {   // rows to load for tile t
        Ld l; l.on = t.valid && t.ok && t.any; l.x0 = t.x0; l.bw = t.bw; l.y0 = l.y1 = 0;
        if (!l.on) return l;
        const bool cont = pbw == t.bw && px0 == t.x0 && t.y0 >= plo && t.y0 <= phi;
        l.y0 = cont ? (phi > t.y0 ? phi : t.y0) : t.y0;
        l.y1 = t.y0 + t.bh;
        if (l.y1 < l.y0) l.y1 = l.y0;
        return l;
    };
    auto after = [&](const RingTile &t, const Ld &l, int &plo, int &phi, int &px0, int &pbw) {   // ring after the commit
        if (!l.on) return;
        const bool cont = pbw == t.bw && px0 == t.x0 && t.y0 >= plo && t.y0 <= phi;
        phi = cont ? (phi > l.y1 ? phi : l.y1) : l.y1;
        plo = t.y0; px0 = t.x0; pbw = t.bw;
    };

    unsigned pre[NP];
    auto issue = [&](const Ld &l) {
        const int E = l.on ? (l.y1 - l.y0) * l.bw : 0;
        const float inv = __builtin_amdgcn_rcpf((float)(l.bw > 0 ? l.bw : 1));
        const int e = (int)threadIdx.x;
        const int rr = (int)(((float)e + 0.5f) * inv);
        const int cc = e - rr * l.bw;
        const int gx = l.x0 + 4 * cc, gy = l.y0 + rr;
        const bool in = e < E && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
        const unsigned off = (__umul24((unsigned)gy, (unsigned)pitch) + (unsigned)gx) & (0u - (unsigned)in);
#pragma unroll
        for (int p = 0; p < NP; p++) pre[p] = *reinterpret_cast<const unsigned *>(plane(p) + off);
    };
    auto commit = [&](const Ld &l) {
        if (!l.on) return;
        const int E = (l.y1 - l.y0) * l.bw;
        const int e = (int)threadIdx.x;
        if (e < E) {
            const float inv = __builtin_amdgcn_rcpf((float)l.bw);
            const int rr = (int)(((float)e + 0.5f) * inv);
            const int cc = e - rr * l.bw;
            const int gx = l.x0 + 4 * cc, gy = l.y0 + rr;
            const bool in = (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
            u32x2 *dst = reinterpret_cast<u32x2 *>(tile + (size_t)(((gy & (kRingRows - 1)) * l.bw + cc) * (NP * 4)));
#pragma unroll
            for (int p = 0; p < NP; p += 2) {
                u32x2 w2;
                w2.x = in ? pre[p] : 0u; w2.y = in ? pre[p + 1] : 0u;
                dst[p >> 1] = w2;
            }
        }
    };

    RingTile cur;
    enter_item(lb, cur);
    if (!cur.valid) return;
    load_box(cur);
    Ld ld = plan(cur, lo, hi, rx0, rbw);
    issue(ld);
    for (;;) {
        commit(ld);
        after(cur, ld, lo, hi, rx0, rbw);
        __syncthreads();
        RingTile nxt;
        advance(cur, nxt);
        const int col = cur.tx * kTileW + lane;
        unsigned xy[TH / 4], fr[TH / 4];
#pragma unroll
        for (int q = 0; q < TH / 4; q++) {                  // this tile's map entries first (older than the prefetch)
            const int row = cur.ty * TH + 4 * q + wv;
            const bool inb = row < H && col < W;
            const unsigned m = inb ? (unsigned)row * (unsigned)W + (unsigned)col : 0u;
            xy[q] = *reinterpret_cast<const unsigned *>(reinterpret_cast<const char *>(map_xy) + m * 4u);
            fr[q] = *reinterpret_cast<const uint16_t *>(reinterpret_cast<const char *>(map_frac) + m * 2u);
        }
        const Ld ldn = plan(nxt, lo, hi, rx0, rbw);
        issue(ldn);
#pragma unroll
        for (int q = 0; q < TH / 4; q++) {
            const int row = cur.ty * TH + 4 * q + wv;
            const bool inb = row < H && col < W;
            const unsigned m = (unsigned)row * (unsigned)W + (unsigned)col;
            int v;
            float ph;
            if (cur.ok) {
                // tile_taps_map with ring addressing: row slot = source row & 15, columns relative to the item
                TileTaps<NP> k;
                const int sx = (int)(short)(xy[q] & 0xFFFFu), sy = (int)xy[q] >> 16;
                const bool out = !inb || !cur.any || (unsigned)(sx + 1) > (unsigned)W || (unsigned)(sy + 1) > (unsigned)H;
                const unsigned fx = fr[q] & 31u, fy = (fr[q] >> 5) & 31u;
                const unsigned wxp = out ? 0u : __umul24(fx, 0xFFFFu) + 32u;
                unsigned w0 = __umul24(wxp, (32u - fy) << 6);
                const unsigned w1 = __umul24(wxp, fy << 6);
                w0 = w0 == 0x10000u ? 0xFFFFu : w0;
                const int bx = out ? 0 : sx - cur.x0;
                const int s0 = out ? 0 : (sy & (kRingRows - 1)), s1 = out ? 0 : ((sy + 1) & (kRingRows - 1));
                k.a0 = __mul24(__mul24(s0, cur.bw) + (bx >> 2), NP * 4);
                k.a1 = __mul24(__mul24(s1, cur.bw) + (bx >> 2), NP * 4);
                k.sel = __umul24((unsigned)bx & 3u, 0x10001u) + 0x0C010C00u;
                k.w0 = __builtin_bit_cast(u16x2, w0);
                k.w1 = __builtin_bit_cast(u16x2, w1);
                int acc[NP];
#pragma unroll
                for (int p = 0; p < NP; p++) acc[p] = tile_sample<NP, 6>(tile, k, p);
                ph = mf_pixel_sh<16>(acc, black_thr, lut, v);
            } else {                                        // the item does not fit the ring: direct gather
                Tap t = make_tap((int)(short)(xy[q] & 0xFFFFu), (int)xy[q] >> 16, fr[q], pitch, W, H);
                if (!inb) t.kind = 1;
                int gpx[NP];
#pragma unroll 1
                for (int p = 0; p < NP; p++) gpx[p] = cur.any ? sample(plane(p), pitch, W, H, t) : 0;
                ph = mf_pixel(gpx, black_thr, lut, v);
            }
            if (valid) {
                const unsigned long long bal = __ballot(v != 0);
                const unsigned half = (lane & 32) ? (unsigned)(bal >> 32) : (unsigned)bal;
                const unsigned vw = __umul24((half >> lane31) & 0xFu, 0x204081u) & 0x01010101u;
                if (inb && (lane & 3) == 0) __builtin_nontemporal_store(vw, reinterpret_cast<unsigned *>(valid + m));
            } else {
                ph = v ? ph : kInvalidPhase;                // folded flag (see launch_mf_decode)
            }
            if (inb) __builtin_nontemporal_store(ph, reinterpret_cast<float *>(reinterpret_cast<char *>(phase) + m * 4u));
        }
        if (!nxt.valid) break;
        __syncthreads();
        cur = nxt;
        ld = ldn;
    }
}
#endif  // SLR_ALL_FORMS

static unsigned pick_blocks(size_t groups)
{
    // One workgroup per 256 work items (no persistent grid-stride): with a capped grid the last sweep leaves
    // CUs idle (12.3 Mpx / (2048 x 256 x 16 px) = 1.46 sweeps -> 27 % of the machine-time wasted); small blocks
    // let the dispatcher back-fill.  Rounded up to a multiple of 8 for the XCD band mapping of the fused kernel.
    const size_t b = (groups + 255) / 256;
    return (unsigned)(((b ? b : 1) + 7) & ~(size_t)7);
}

// K2's grid: every workgroup first copies the 12 KB decode tables to LDS, so beyond the resident set (8 workgroups of
// 256 on each CU) the grid is capped and the kernel's grid-stride loop takes over -- 12 000 one-shot workgroups at
// 4096x3000 re-read 147 MB of tables from L2 next to 233 MB of pixels (measured: 48.6 instead of 44 us).  Few sweeps
// keep the old back-filling grid (see above); from 4 sweeps on the uneven last sweep costs less than the tables.
static unsigned pick_blocks_k2(size_t groups)
{
    static DevSlots cus_of;
    int dev = 0;
    (void)hipGetDevice(&dev);
    int cus = cus_of.get(dev);
    if (cus == 0) {
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
        cus_of.put(dev, cus);
    }
    const unsigned b = pick_blocks(groups), resident = (unsigned)cus * 8u;
    return b >= 4 * resident ? resident : b;
}

// the LDS-tiled fused kernel needs dword-aligned planes and outputs and 32-bit offsets
static bool rect_lds_ok(const MfPlanes &pl, int pitch, int W, int H, const float *phase, const uint8_t *valid,
                        const int16_t *map_xy, const void *tile_boxes, int rect_algo)
{
    bool aligned = pitch % 4 == 0;
    for (int p = 0; p < SLR_MF_PLANES; p++) aligned = aligned && ((uintptr_t)pl.p[p] % 4 == 0);
    return map_xy && tile_boxes && W % 4 == 0 && aligned && rect_algo != 1 && ((uintptr_t)phase % 16 == 0) &&
           ((uintptr_t)valid % 4 == 0) && (long long)W * H < (1ll << 30) && (long long)H * pitch < (1ll << 32) &&
           pitch < (1 << 24);                                  // (row * pitch is a 24-bit multiply)
}

static hipError_t launch_rect_lds(const RectJob *jobs, int njobs, int pitch, int W, int H, int black_thr,
                                  const float *atan_lut, int rect_algo, hipStream_t s)
{
    int tiles_x = (W + kTileW - 1) / kTileW;
    // (forms 2, 3 and 4 are measured-dominated -- DESIGN section 9 -- and compiled with -DSLR_ALL_FORMS only; without it
    // slr_set_option refuses them)
#ifdef SLR_ALL_FORMS
    if (rect_algo == 3) {                                // sliding-window form
        const int tiles_y8 = (H + kMidTileH - 1) / kMidTileH;
        RectJobs jr;
        jr.j[0] = jobs[0]; jr.j[1] = jobs[njobs - 1];
        const size_t off8 = tile_count(W, H, kTileH) + tile_count(W, H, kGrayTileH);
        jr.j[0].boxes += off8; jr.j[1].boxes += off8;
        const size_t lds = (size_t)kRingRows * kRingBW * SLR_MF_PLANES * 4 + 16;
        static DevSlots ring_cache;
        int dev = 0;
        (void)hipGetDevice(&dev);
        int ring_res = ring_cache.get(dev);
        if (!ring_res) {
            int per_cu = 0, cus = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, mf_rect_decode_ring_kernel, 256, lds) != hipSuccess || per_cu < 1) per_cu = 4;
            if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
            ring_res = per_cu * cus;
            ring_cache.put(dev, ring_res);
        }
        const int res = (tl_debug.rect_resident > 0 ? tl_debug.rect_resident : ring_res) / njobs;
        int nbx = res / 8 > 0 ? res / 8 : 1;
        const int R = (tiles_y8 + 7) / 8;                    // tile rows per XCD band
        if (nbx > R * tiles_x) nbx = R * tiles_x;
        int seg_len = (R * tiles_x + 4 * nbx - 1) / (4 * nbx);   // ~4 items per workgroup
        if (seg_len < 4) seg_len = 4;
        if (seg_len > R) seg_len = R > 0 ? R : 1;
        SLR_LAUNCH(mf_rect_decode_ring_kernel, dim3(8u * (unsigned)nbx * (unsigned)njobs), dim3(256), lds, s, jr, njobs,
                           pitch, W, H, black_thr, atan_lut, tiles_x, tiles_y8, seg_len);
        return hipGetLastError();
    }
    const bool mid = rect_algo != 2;                     // default: 64 x 8 tiles, one prefetch round; 2: 64 x 16, two
    const bool wide8 = rect_algo == 5;                   // 5: 128 x 8 tiles, 512 threads, one round, pre-digested map
    const bool wide = rect_algo == 4 || wide8;           // 4: 128 x 8 tiles (pairs of 64 x 8 table entries), two rounds
#else
    constexpr bool mid = true;                           // 6 (and whatever else arrives here): 64 x 8 tiles, one prefetch round
    const bool wide8 = rect_algo == 5;                   // 5: 128 x 8 tiles, 512 threads, one round, pre-digested map
    const bool wide = wide8;
#endif
    // persistent workgroups: as many as are resident at once (a multiple of 8 per job for the XCD bands), never more
    // than one per tile
    const int th = mid ? kMidTileH : kTileH;
    const int tiles_yy = (H + th - 1) / th;
    if (wide) tiles_x = (W + 2 * kTileW - 1) / (2 * kTileW);
    const int budget = wide8 ? kWideBudget : wide ? 27 * 1024 : mid ? kMidBudget : 24 * 1024;   // 14 planes x ~(tile width + 8) x (th + 7) source bytes
    const size_t box_off = mid ? tile_count(W, H, kTileH) + tile_count(W, H, kGrayTileH) : 0;
    RectJobs j;
    j.j[0] = jobs[0]; j.j[1] = jobs[njobs - 1];
    for (int jq = 0; jq < 2; jq++) {                      // the tiled map copy lives behind the box tables (launch_tile_boxes)
        const char *basep = reinterpret_cast<const char *>(j.j[jq].boxes);
        j.j[jq].pk_t = mid && !wide && !tl_debug.no_tiled_map ? reinterpret_cast<const unsigned *>(basep + tiled_map_offset(W, H)) : nullptr;
        if (wide8) j.j[jq].pk_t = reinterpret_cast<const unsigned *>(basep + tiled_map_offset(W, H)) + tile_count(W, H, kMidTileH) * 512;
    }
    j.j[0].boxes += box_off; j.j[1].boxes += box_off;
    // resident workgroups of this kernel on the CURRENT device (cached per device and kernel variant)
    // strided planes?  (both jobs)
    bool strided = true;
    for (int jq = 0; jq < 2; jq++) {
        const uint8_t *const *pp = j.j[jq].pl.p;
        const long long st = (long long)(pp[1] - pp[0]);
        bool ok = st >= (long long)H * pitch && st * SLR_MF_PLANES < (1ll << 31);
        for (int i = 2; i < SLR_MF_PLANES && ok; i++) ok = (long long)(pp[i] - pp[0]) == st * i;
        j.j[jq].plane_stride = ok ? (unsigned)st : 0u;
        strided = strided && ok;
    }
    if (tl_debug.no_buffer_form) strided = false;         // tests: force the pointer form
    static DevSlots resident_cache[4][2];
    int dev = 0;
    (void)hipGetDevice(&dev);
    DevSlots &resident_of = resident_cache[wide8 ? 3 : wide ? 2 : mid][strided];
    int resident_slot = resident_of.get(dev);
    if (!resident_slot) {
        int per_cu = 0, cus = 0;
        const size_t dyn = (size_t)budget + 16;
        const hipError_t e =
            wide8 ? (strided ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, mf_rect_decode_lds_kernel<kMidTileH, 1, true, 2, true, 512>, 512, dyn)
                             : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, mf_rect_decode_lds_kernel<kMidTileH, 1, false, 2, true, 512>, 512, dyn)) :
#ifdef SLR_ALL_FORMS
            wide ? (strided ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, mf_rect_decode_lds_kernel<kMidTileH, 2, true, 2>, 256, dyn)
                            : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, mf_rect_decode_lds_kernel<kMidTileH, 2, false, 2>, 256, dyn)) :
            !mid ? (strided ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, mf_rect_decode_lds_kernel<kTileH, 2, true>, 256, dyn)
                            : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, mf_rect_decode_lds_kernel<kTileH, 2, false>, 256, dyn)) :
#endif
                  (strided ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, mf_rect_decode_lds_kernel<kMidTileH, 1, true, 1, true>, 256, dyn)
                           : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, mf_rect_decode_lds_kernel<kMidTileH, 1, false, 1, true>, 256, dyn));
        if (e != hipSuccess || per_cu < 1) per_cu = 4;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
        resident_slot = per_cu * cus;
        resident_of.put(dev, resident_slot);
    }
    const int resident_blocks = resident_slot;
    const int T = tiles_x * tiles_yy, per = (T + 7) / 8;
    const int res = (tl_debug.rect_resident > 0 ? tl_debug.rect_resident : resident_blocks) / njobs;   // tests: few workgroups -> many tiles each
    int nbx = res / 8 < per ? res / 8 : per;
    if (nbx < 1) nbx = 1;
    const dim3 grid(8u * (unsigned)nbx * (unsigned)njobs);
#define SLR_RECT_LAUNCH(TH_, R_, S_)                                                                                  \
    SLR_LAUNCH((mf_rect_decode_lds_kernel<TH_, R_, S_>), grid, dim3(256), (size_t)budget + 16, s, j, njobs, pitch, W, \
                       H, black_thr, atan_lut, tiles_x, tiles_yy, budget)
    if (wide8) {
        if (strided) SLR_LAUNCH((mf_rect_decode_lds_kernel<kMidTileH, 1, true, 2, true, 512>), grid, dim3(512), (size_t)budget + 16, s, j, njobs,
                                        pitch, W, H, black_thr, atan_lut, tiles_x, tiles_yy, budget);
        else         SLR_LAUNCH((mf_rect_decode_lds_kernel<kMidTileH, 1, false, 2, true, 512>), grid, dim3(512), (size_t)budget + 16, s, j, njobs,
                                        pitch, W, H, black_thr, atan_lut, tiles_x, tiles_yy, budget);
    }
#ifdef SLR_ALL_FORMS
    else if (wide) {
        if (strided) SLR_LAUNCH((mf_rect_decode_lds_kernel<kMidTileH, 2, true, 2>), grid, dim3(256), (size_t)budget + 16, s, j, njobs,
                                        pitch, W, H, black_thr, atan_lut, tiles_x, tiles_yy, budget);
        else         SLR_LAUNCH((mf_rect_decode_lds_kernel<kMidTileH, 2, false, 2>), grid, dim3(256), (size_t)budget + 16, s, j, njobs,
                                        pitch, W, H, black_thr, atan_lut, tiles_x, tiles_yy, budget);
    }
    else if (!mid) { if (strided) SLR_RECT_LAUNCH(kTileH, 2, true); else SLR_RECT_LAUNCH(kTileH, 2, false); }
#endif
    else if (mid && j.j[0].pk_t) {
        if (strided) SLR_LAUNCH((mf_rect_decode_lds_kernel<kMidTileH, 1, true, 1, true>), grid, dim3(256), (size_t)budget + 16, s, j, njobs,
                                        pitch, W, H, black_thr, atan_lut, tiles_x, tiles_yy, budget);
        else         SLR_LAUNCH((mf_rect_decode_lds_kernel<kMidTileH, 1, false, 1, true>), grid, dim3(256), (size_t)budget + 16, s, j, njobs,
                                        pitch, W, H, black_thr, atan_lut, tiles_x, tiles_yy, budget);
    }
    else { if (strided) SLR_RECT_LAUNCH(kMidTileH, 1, true); else SLR_RECT_LAUNCH(kMidTileH, 1, false); }
#undef SLR_RECT_LAUNCH
    return hipGetLastError();
}

// both cameras of a stereo frame in one launch; *done = false when the LDS-tiled form does not apply (caller then
// launches the cameras one by one through launch_mf_decode)
hipError_t launch_mf_rect_decode_pair(const MfPlanes pl[2], int pitch, int W, int H, int black_thr, const float *atan_lut,
                                      float *const phase[2], uint8_t *const valid[2], const int16_t *const map_xy[2],
                                      const uint16_t *const map_frac[2], const void *const tile_boxes[2], int rect_algo,
                                      bool *done, hipStream_t s)
{
    *done = false;
    if (tl_debug.eval_x87) return hipSuccess;               // SLR_OPT_EVAL_MODEL = 1: per camera, through launch_mf_decode_x87
    for (int c = 0; c < 2; c++)
        if (!rect_lds_ok(pl[c], pitch, W, H, phase[c], valid[c], map_xy[c], tile_boxes[c], rect_algo)) return hipSuccess;
    RectJob jobs[2];
    for (int c = 0; c < 2; c++) jobs[c] = RectJob{pl[c], 0u, map_xy[c], map_frac[c], (const int4 *)tile_boxes[c], phase[c], valid[c], nullptr};
    *done = true;
    return launch_rect_lds(jobs, 2, pitch, W, H, black_thr, atan_lut, rect_algo, s);
}

// SLR_OPT_EVAL_MODEL = 1 (the x87 evaluation of the heterodyne tail, decode_common.hpp): the plain forms only -- the per-pixel
// gather for a rectifying decode, one pixel group per thread otherwise; atan_lut is then the x87 variant of the tables
static hipError_t launch_mf_decode_x87(const MfPlanes &pl, int pitch, int W, int H, int black_thr, const float *atan_lut,
                                       float *phase, uint8_t *valid, const int16_t *map_xy, const uint16_t *map_frac, hipStream_t s)
{
    if (map_xy) {
        const bool vec = (W % 4 == 0) && ((uintptr_t)phase % 16 == 0) && (!valid || (uintptr_t)valid % 4 == 0) &&
                         ((uintptr_t)map_xy % 16 == 0) && ((uintptr_t)map_frac % 8 == 0);
        if (vec) SLR_LAUNCH((mf_rect_decode_kernel<4, true>), dim3(pick_blocks((size_t)(W / 4) * H)), dim3(256), 0, s,
                            pl, pitch, W, H, black_thr, atan_lut, map_xy, map_frac, phase, valid);
        else     SLR_LAUNCH((mf_rect_decode_kernel<1, true>), dim3(pick_blocks((size_t)W * H)), dim3(256), 0, s,
                            pl, pitch, W, H, black_thr, atan_lut, map_xy, map_frac, phase, valid);
        return hipGetLastError();
    }
    bool a4 = (W % 4 == 0) && (pitch % 4 == 0) && pitch == W;
    for (int p = 0; p < SLR_MF_PLANES; p++) a4 = a4 && ((uintptr_t)pl.p[p] % 4 == 0);
    a4 = a4 && ((uintptr_t)phase % 16 == 0) && ((uintptr_t)valid % 4 == 0);
    if (a4) SLR_LAUNCH((mf_decode_kernel<1, true>), dim3(pick_blocks_k2((size_t)(W / 4) * H)), dim3(256), 0, s,
                       pl, pitch, W, H, black_thr, atan_lut, phase, valid);
    else    SLR_LAUNCH((mf_decode_scalar_kernel<true>), dim3(pick_blocks_k2((size_t)W * H)), dim3(256), 0, s,
                       pl, pitch, W, H, black_thr, atan_lut, phase, valid);
    return hipGetLastError();
}

hipError_t launch_mf_decode(const MfPlanes &pl, int pitch, int W, int H, int black_thr, const float *atan_lut,
                            float *phase, uint8_t *valid, const int16_t *map_xy, const uint16_t *map_frac,
                            const void *tile_boxes, int vec_hint, int rect_algo, hipStream_t s)
{
    if (tl_debug.eval_x87) return launch_mf_decode_x87(pl, pitch, W, H, black_thr, atan_lut, phase, valid, map_xy, map_frac, s);
    if (rect_lds_ok(pl, pitch, W, H, phase, valid, map_xy, tile_boxes, rect_algo)) {
        const RectJob job{pl, 0u, map_xy, map_frac, (const int4 *)tile_boxes, phase, valid, nullptr};
        return launch_rect_lds(&job, 1, pitch, W, H, black_thr, atan_lut, rect_algo, s);
    }
    if (map_xy) {
        // the 4-pixel form stores 16-byte phase vectors and loads the map entries of 4 pixels as vectors
        const bool vec = (W % 4 == 0) && ((uintptr_t)phase % 16 == 0) && (!valid || (uintptr_t)valid % 4 == 0) &&
                         ((uintptr_t)map_xy % 16 == 0) && ((uintptr_t)map_frac % 8 == 0);
        if (vec) SLR_LAUNCH(mf_rect_decode_kernel<4>, dim3(pick_blocks((size_t)(W / 4) * H)), dim3(256), 0, s,
                                    pl, pitch, W, H, black_thr, atan_lut, map_xy, map_frac, phase, valid);
        else     SLR_LAUNCH(mf_rect_decode_kernel<1>, dim3(pick_blocks((size_t)W * H)), dim3(256), 0, s,
                                    pl, pitch, W, H, black_thr, atan_lut, map_xy, map_frac, phase, valid);
        return hipGetLastError();
    }
    bool a16 = (W % 16 == 0) && (pitch % 16 == 0), a4 = (W % 4 == 0) && (pitch % 4 == 0);
    for (int p = 0; p < SLR_MF_PLANES; p++) {
        a16 = a16 && ((uintptr_t)pl.p[p] % 16 == 0);
        a4 = a4 && ((uintptr_t)pl.p[p] % 4 == 0);
    }
    a16 = a16 && ((uintptr_t)phase % 16 == 0) && ((uintptr_t)valid % 16 == 0);
    a4 = a4 && ((uintptr_t)phase % 16 == 0) && ((uintptr_t)valid % 4 == 0);
    const bool a8 = a4 && (W % 8 == 0) && (pitch % 8 == 0) && ((uintptr_t)valid % 8 == 0) && [&] {
        for (int p = 0; p < SLR_MF_PLANES; p++) if ((uintptr_t)pl.p[p] % 8) return false;
        return true; }();
    // measured on MI355X at 4096x3000: 4 px/thread 56.8 us, 8 px 58.5 us, 16 px 69.4 us (the 16-px form stores 64-byte
    // strided fragments per lane; the 4-px form writes one contiguous KiB per wave instruction) -> default 4
    if (a16 && vec_hint == 16)
                  SLR_LAUNCH(mf_decode_kernel<4>, dim3(pick_blocks_k2((size_t)(W / 16) * H)), dim3(256), 0, s,
                                     pl, pitch, W, H, black_thr, atan_lut, phase, valid);
    else if (a8 && vec_hint == 8)
                  SLR_LAUNCH(mf_decode_kernel<2>, dim3(pick_blocks_k2((size_t)(W / 8) * H)), dim3(256), 0, s,
                                     pl, pitch, W, H, black_thr, atan_lut, phase, valid);
    else if (a4 && pitch != W && H > 1) {                   // padded rows: (column group, row) grid, rows strided
        const unsigned gx = (unsigned)((W / 4 + 255) / 256), cap = pick_blocks_k2((size_t)(W / 4) * H);
        unsigned gy = (cap + gx - 1) / gx;
        if (gy > (unsigned)H) gy = (unsigned)H;
        if (gy > 65535u) gy = 65535u;
        if (gy < 2u) gy = 2u;                               // gridDim.y > 1 selects this form in the kernel
                  SLR_LAUNCH(mf_decode_kernel<1>, dim3(gx, gy), dim3(256), 0, s,
                                     pl, pitch, W, H, black_thr, atan_lut, phase, valid);
    }
    else if (a4)  SLR_LAUNCH(mf_decode_kernel<1>, dim3(pick_blocks_k2((size_t)(W / 4) * H)), dim3(256), 0, s,
                                     pl, pitch, W, H, black_thr, atan_lut, phase, valid);
    else          SLR_LAUNCH((mf_decode_scalar_kernel<false>), dim3(pick_blocks_k2((size_t)W * H)), dim3(256), 0, s,
                                     pl, pitch, W, H, black_thr, atan_lut, phase, valid);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------
// K3 / K3': Gray decode.  V pixels per thread (4 with dword loads, or 1 generic / rectified-generic).
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int gray_to_binary(int g)
{
    // graycodes.cpp:116-128: running XOR from the MSB == prefix-XOR of the packed Gray word
    g ^= g >> 1; g ^= g >> 2; g ^= g >> 4; g ^= g >> 8;
    return g;
}

template <int V, bool RECT>
__global__ __launch_bounds__(256) void gray_decode_kernel(GrayPlanes pl, int n_col_bits, int n_row_bits, int pitch,
                                                          int W, int H, int black_thr, int white_thr, int scan_w,
                                                          int scan_h, const int16_t *__restrict__ map_xy,
                                                          const uint16_t *__restrict__ map_frac,
                                                          int32_t *__restrict__ code_x, int32_t *__restrict__ code_y,
                                                          uint8_t *__restrict__ valid)
{
    const int gpr = W / V;
    const unsigned total = (unsigned)gpr * (unsigned)H;
    for (unsigned g = blockIdx.x * 256u + threadIdx.x; g < total; g += gridDim.x * 256u) {
        const unsigned row = g / gpr, col0 = (g - row * gpr) * V;
        const size_t so = (size_t)row * pitch + col0, m = (size_t)row * W + col0;
        Tap taps[V];
        if constexpr (RECT) {
#pragma unroll
            for (int i = 0; i < V; i++)
                taps[i] = make_tap(map_xy[2 * (m + i)], map_xy[2 * (m + i) + 1], map_frac[m + i], pitch, W, H);
        }
        // fetch one plane's V pixels as packed bytes
        auto fetch = [&](int plane) -> unsigned {
            const uint8_t *p = pl.p[plane];
            if constexpr (RECT) {
                unsigned r = 0;
#pragma unroll
                for (int i = 0; i < V; i++) r |= (unsigned)sample(p, pitch, W, H, taps[i]) << (8 * i);
                return r;
            } else if constexpr (V == 4) {
                return __builtin_nontemporal_load(reinterpret_cast<const unsigned *>(p + so));
            } else {
                return p[so];
            }
        };
        const unsigned wv = fetch(0), bv = fetch(1);
        int gx[V], gy[V], err[V];
#pragma unroll
        for (int i = 0; i < V; i++) { gx[i] = 0; gy[i] = 0; err[i] = 0; }
        // (the contrast test can only fire with a positive white threshold -- the reference's default is 0, SURVEY Q10 --: a
        // wave-uniform fact, kept out of the per-pixel work; a code bit is then a byte compare and an add of the word to itself)
        const auto bits = [&](int first, int n, int g[V], auto wt) {
            for (int c = 0; c < n; c++) {
                const unsigned a = fetch(first + 2 * c), b = fetch(first + 2 * c + 1);
#pragma unroll
                for (int i = 0; i < V; i++) {
                    const int v1 = (a >> (8 * i)) & 0xFF, v2 = (b >> (8 * i)) & 0xFF;
                    if constexpr (decltype(wt)::value) { const int df = v1 - v2; err[i] |= ((df < 0 ? -df : df) < white_thr) ? 1 : 0; }
                    g[i] = g[i] + g[i] + (v1 > v2 ? 1 : 0);
                }
            }
        };
        if (white_thr > 0) {
            bits(2, n_col_bits, gx, std::true_type{});                       // reconstruct.cpp:387-400
            bits(2 + 2 * n_col_bits, n_row_bits, gy, std::true_type{});      // reconstruct.cpp:349-360
        } else {
            bits(2, n_col_bits, gx, std::false_type{});
            bits(2 + 2 * n_col_bits, n_row_bits, gy, std::false_type{});
        }
        int cx[V], cy[V];
        unsigned vw = 0;
#pragma unroll
        for (int i = 0; i < V; i++) {
            const int mask = ((int)((wv >> (8 * i)) & 0xFF) - (int)((bv >> (8 * i)) & 0xFF) > black_thr) ? 1 : 0;
            const int x = gray_to_binary(gx[i]), y = gray_to_binary(gy[i]);
            int e = err[i];
            if (n_row_bits > 0) e |= (y > scan_h || x > scan_w) ? 1 : 0;   // reconstruct.cpp:364 (Q9 '>')
            else e |= (x > scan_w) ? 1 : 0;                                // reconstruct.cpp:403
            const int ok = mask & (e ^ 1);
            cx[i] = ok ? x : -1;
            cy[i] = (ok && n_row_bits > 0) ? y : -1;
            vw |= (unsigned)ok << (8 * i);
        }
        if constexpr (V == 4) {
            i32x4 o; o.x = cx[0]; o.y = cx[1]; o.z = cx[2]; o.w = cx[3];
            __builtin_nontemporal_store(o, reinterpret_cast<i32x4 *>(code_x + m));
            if (code_y) { i32x4 q; q.x = cy[0]; q.y = cy[1]; q.z = cy[2]; q.w = cy[3];
                          __builtin_nontemporal_store(q, reinterpret_cast<i32x4 *>(code_y + m)); }
            if (valid) __builtin_nontemporal_store(vw, reinterpret_cast<unsigned *>(valid + m));   // (null: code -1 says it all)
        } else {
            code_x[m] = cx[0];
            if (code_y) code_y[m] = cy[0];
            if (valid) valid[m] = (uint8_t)vw;
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// fused K1+K3, LDS-tiled form: the scheme of mf_rect_decode_lds_kernel for a Gray stack of 22..66 planes.  64 x TH
// destination tiles, one pixel per lane, TH/4 passes: TH = 8 when the tile's source box fits the LDS budget for this
// plane count (fill work and halo re-reads per pixel are ~3x lower than with 64 x 4), else TH = 4.  LDS layout
// [row][plane][dword column] (the plane count is a run-time value, so the per-plane step is one add instead of an
// immediate).  The samples stay in the high half-word of the dot-product accumulators (weights x 64, see
// tile_taps_map) and are compared / subtracted in place.
// ------------------------------------------------------------------------------------------------------
template <int TH, bool STRIDED>
__global__ __launch_bounds__(256) void gray_rect_decode_lds_kernel(GrayPlanes pl, unsigned pstride, int n_col_bits, int n_row_bits, int pitch,
                                                                   int W, int H, int black_thr, int white_thr, int scan_w,
                                                                   int scan_h, const int16_t *__restrict__ map_xy,
                                                                   const uint16_t *__restrict__ map_frac,
                                                                   const int4 *__restrict__ boxes,
                                                                   int32_t *__restrict__ code_x, int32_t *__restrict__ code_y,
                                                                   uint8_t *__restrict__ valid, int tiles_x, int tiles_y,
                                                                   int budget)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t tile[];
    const int NP = 2 + 2 * n_col_bits + 2 * n_row_bits;
    const unsigned nb = gridDim.x, per = nb / 8;
    const unsigned vb = (nb % 8 == 0) ? (blockIdx.x % 8) * per + blockIdx.x / 8 : blockIdx.x;
    const int ty = (int)(vb / (unsigned)tiles_x), tx = (int)(vb - (unsigned)ty * tiles_x);
    if (ty >= tiles_y) return;
    const int4 box = boxes[ty * tiles_x + tx];
    const int x0 = box.x, y0 = box.y, BW4 = box.z, BH = box.w;
    const bool any = BW4 > 0;
    const bool fits = any && (long long)BW4 * BH * 4 * NP <= budget;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const unsigned lane31 = (unsigned)lane & 31u;
    const int col = tx * kTileW + lane;
    const int rowstep = BW4 * 4;                            // bytes per (row, plane) line

    if (fits) {
        // W % 4 == 0, x0 % 4 == 0 and dword-aligned planes (launcher): a source dword is inside or outside as a whole
        const int E = BH * BW4;
        const float inv = __builtin_amdgcn_rcpf((float)BW4);
        for (int e = threadIdx.x; e < E; e += 256) {
            const int rr = (int)(((float)e + 0.5f) * inv);
            const int cc = e - rr * BW4;
            const int gx = x0 + 4 * cc, gy = y0 + rr;
            const bool in = (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
            uint8_t *dst = tile + (size_t)(rr * NP) * rowstep + 4 * cc;
            // 16 independent loads in flight per thread, then 16 LDS stores (a plain p-loop waits per load)
            if constexpr (STRIDED) {
                // equally spaced planes: raw buffer loads, plane = scalar offset, outside the image -> hardware zero
                const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
                    (void *)pl.p[0], 0, (int)((unsigned)(NP - 1) * pstride + (unsigned)H * (unsigned)pitch), 0x00020000);
                const unsigned off = in ? (unsigned)gy * (unsigned)pitch + (unsigned)gx : 0xFFFFFFF0u;
                for (int pb = 0; pb < NP; pb += 16) {
                    unsigned v[16];
#pragma unroll
                    for (int i = 0; i < 16; i++)
                        v[i] = (pb + i < NP) ? (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)off, (int)((pb + i) * pstride), 0) : 0u;
#pragma unroll
                    for (int i = 0; i < 16; i++)
                        if (pb + i < NP) *reinterpret_cast<unsigned *>(dst + (pb + i) * rowstep) = v[i];
                }
            } else {
                const unsigned off = in ? (unsigned)gy * (unsigned)pitch + (unsigned)gx : 0u;
                for (int pb = 0; pb < NP; pb += 16) {
                    unsigned v[16];
#pragma unroll
                    for (int i = 0; i < 16; i++)
                        v[i] = (pb + i < NP) ? *reinterpret_cast<const unsigned *>(pl.p[pb + i] + off) : 0u;
#pragma unroll
                    for (int i = 0; i < 16; i++)
                        if (pb + i < NP) *reinterpret_cast<unsigned *>(dst + (pb + i) * rowstep) = in ? v[i] : 0u;
                }
            }
        }
        __syncthreads();
    }
#pragma unroll 1
    for (int q = 0; q < TH / 4; q++) {
        const int row = ty * TH + 4 * q + wv;
        const bool inb = row < H && col < W;
        const unsigned m = (unsigned)row * (unsigned)W + (unsigned)col;
        unsigned xy = 0, fr = 0;
        if (inb) {
            xy = *reinterpret_cast<const unsigned *>(reinterpret_cast<const char *>(map_xy) + m * 4u);
            fr = *reinterpret_cast<const uint16_t *>(reinterpret_cast<const char *>(map_frac) + m * 2u);
        }
        int gx_ = 0, gy_ = 0, err = 0, mask;
        if (fits) {
            // tap state shared by all planes (tile_taps_map with this kernel's LDS layout)
            const int sx = (int)(short)(xy & 0xFFFFu), sy = (int)xy >> 16;
            const bool out = !inb || (unsigned)(sx + 1) > (unsigned)W || (unsigned)(sy + 1) > (unsigned)H;
            const unsigned fx = fr & 31u, fy = (fr >> 5) & 31u;
            const unsigned wxp = out ? 0u : __umul24(fx, 0xFFFFu) + 32u;
            unsigned w0u = __umul24(wxp, (32u - fy) << 6);
            const unsigned w1u = __umul24(wxp, fy << 6);
            w0u = w0u == 0x10000u ? 0xFFFFu : w0u;
            const u16x2 w0 = __builtin_bit_cast(u16x2, w0u), w1 = __builtin_bit_cast(u16x2, w1u);
            const int bx = out ? 0 : sx - x0, r0 = out ? 0 : sy - y0;
            const unsigned sel = __umul24((unsigned)bx & 3u, 0x10001u) + 0x0C010C00u;
            const uint8_t *base = tile + __mul24(__mul24(r0, NP), rowstep) + (bx & ~3);
            const int rowjump = __mul24(NP, rowstep);       // same plane, next source row
            auto fetch = [&](int p) -> unsigned {             // accumulator: the sample is its high half-word
                const unsigned *q0 = reinterpret_cast<const unsigned *>(base + __mul24(p, rowstep));
                const unsigned *q1 = reinterpret_cast<const unsigned *>(base + __mul24(p, rowstep) + rowjump);
                const unsigned p0 = __builtin_amdgcn_perm(q0[1], q0[0], sel);
                const unsigned p1 = __builtin_amdgcn_perm(q1[1], q1[0], sel);
                unsigned acc = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, p0), w0, 512u << 6, false);
                return __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, p1), w1, acc, false);
            };
            const unsigned aw = fetch(0), ab = fetch(1);
            mask = ((int)(aw >> 16) - (int)(ab >> 16) > black_thr) ? 1 : 0;
#pragma unroll 4
            for (int c = 0; c < n_col_bits; c++) {                       // reconstruct.cpp:387-400
                const unsigned a1 = fetch(2 * c + 2), a2 = fetch(2 * c + 3);
                const int df = (int)(a1 >> 16) - (int)(a2 >> 16);
                err |= ((df < 0 ? -df : df) < white_thr) ? 1 : 0;
                gx_ = (gx_ << 1) | (df > 0 ? 1 : 0);
            }
            for (int c = 0; c < n_row_bits; c++) {                       // reconstruct.cpp:349-360
                const unsigned a1 = fetch(2 * c + 2 + 2 * n_col_bits), a2 = fetch(2 * c + 3 + 2 * n_col_bits);
                const int df = (int)(a1 >> 16) - (int)(a2 >> 16);
                err |= ((df < 0 ? -df : df) < white_thr) ? 1 : 0;
                gy_ = (gy_ << 1) | (df > 0 ? 1 : 0);
            }
        } else {                                            // wild map: direct gather for this tile
            Tap t = make_tap((int)(short)(xy & 0xFFFFu), (int)xy >> 16, fr, pitch, W, H);
            if (!inb) t.kind = 1;
            auto fetch = [&](int p) -> int { return any ? sample(pl.p[p], pitch, W, H, t) : 0; };
            const int wv_ = fetch(0), bv = fetch(1);
            mask = (wv_ - bv > black_thr) ? 1 : 0;
            for (int c = 0; c < n_col_bits; c++) {
                const int df = fetch(2 * c + 2) - fetch(2 * c + 3);
                err |= ((df < 0 ? -df : df) < white_thr) ? 1 : 0;
                gx_ = (gx_ << 1) | (df > 0 ? 1 : 0);
            }
            for (int c = 0; c < n_row_bits; c++) {
                const int df = fetch(2 * c + 2 + 2 * n_col_bits) - fetch(2 * c + 3 + 2 * n_col_bits);
                err |= ((df < 0 ? -df : df) < white_thr) ? 1 : 0;
                gy_ = (gy_ << 1) | (df > 0 ? 1 : 0);
            }
        }
        const int x = gray_to_binary(gx_), y = gray_to_binary(gy_);
        if (n_row_bits > 0) err |= (y > scan_h || x > scan_w) ? 1 : 0;   // reconstruct.cpp:364 (Q9 '>')
        else err |= (x > scan_w) ? 1 : 0;                                // reconstruct.cpp:403
        const int ok = mask & (err ^ 1);
        if (valid) {                                        // (null: the consumer reads validity off code_x == -1)
            const unsigned long long bal = __ballot(ok != 0);    // valid bytes of 4 lanes -> one dword (see the MF kernel)
            const unsigned half = (lane & 32) ? (unsigned)(bal >> 32) : (unsigned)bal;
            const unsigned vw = __umul24((half >> lane31) & 0xFu, 0x204081u) & 0x01010101u;
            if (inb && (lane & 3) == 0) __builtin_nontemporal_store(vw, reinterpret_cast<unsigned *>(valid + m));
        }
        if (inb) {
            __builtin_nontemporal_store(ok ? x : -1, code_x + m);
            if (code_y) __builtin_nontemporal_store((ok && n_row_bits > 0) ? y : -1, code_y + m);
        }
    }
}

hipError_t launch_gray_decode(const GrayPlanes &pl, int n_col_bits, int n_row_bits, int pitch, int W, int H,
                              int black_thr, int white_thr, int scan_w, int scan_h, int32_t *code_x,
                              int32_t *code_y, uint8_t *valid, const int16_t *map_xy, const uint16_t *map_frac,
                              const void *tile_boxes, int rect_algo, hipStream_t s)
{
    const int nplanes = 2 + 2 * n_col_bits + 2 * n_row_bits;
    bool aligned = pitch % 4 == 0;
    for (int p = 0; p < nplanes; p++) aligned = aligned && ((uintptr_t)pl.p[p] % 4 == 0);
    if (map_xy && tile_boxes && W % 4 == 0 && aligned && rect_algo != 1 && ((uintptr_t)valid % 4 == 0) &&
        (long long)W * H < (1ll << 30) && (long long)H * pitch < (1ll << 32)) {
        const int tiles_x = (W + kTileW - 1) / kTileW;
        // 64 x 8 tiles when a typical box (72 source bytes x 11 rows) of all planes fits 32 KB, else 64 x 4; the LDS
        // request is sized for a generous box of this plane count (boxes beyond it take the per-tile gather fallback)
        const bool mid = (size_t)nplanes * 72 * 11 <= 32 * 1024 && rect_algo != 2 && !tl_debug.gray_small_tiles;
        const int th = mid ? kMidTileH : kGrayTileH;
        size_t want = (size_t)nplanes * 76 * (mid ? 12 : 7);
        want = want < 8 * 1024 ? 8 * 1024 : (want > 32 * 1024 ? 32 * 1024 : want);
        const int budget = (int)((want + 255) & ~(size_t)255);
        const int tiles_y = (H + th - 1) / th;
        const unsigned blocks = ((unsigned)(tiles_x * tiles_y) + 7u) & ~7u;
        const int4 *boxes = (const int4 *)tile_boxes + tile_count(W, H, kTileH) + (mid ? tile_count(W, H, kGrayTileH) : 0);
        long long st = nplanes > 1 ? (long long)(pl.p[1] - pl.p[0]) : 0;
        bool strided = st >= (long long)H * pitch && st * nplanes < (1ll << 31) && !tl_debug.no_buffer_form;
        for (int i = 2; i < nplanes && strided; i++) strided = (long long)(pl.p[i] - pl.p[0]) == st * i;
#define SLR_GRAY_LDS(TH_, S_)                                                                                          \
        SLR_LAUNCH((gray_rect_decode_lds_kernel<TH_, S_>), dim3(blocks), dim3(256), (size_t)budget + 16, s, pl,     \
                           (unsigned)st, n_col_bits, n_row_bits, pitch, W, H, black_thr, white_thr, scan_w, scan_h, map_xy, \
                           map_frac, boxes, code_x, code_y, valid, tiles_x, tiles_y, budget)
        if (mid) { if (strided) SLR_GRAY_LDS(kMidTileH, true); else SLR_GRAY_LDS(kMidTileH, false); }
        else     { if (strided) SLR_GRAY_LDS(kGrayTileH, true); else SLR_GRAY_LDS(kGrayTileH, false); }
#undef SLR_GRAY_LDS
        return hipGetLastError();
    }
    bool a4 = (W % 4 == 0) && ((uintptr_t)code_x % 16 == 0) && ((uintptr_t)valid % 4 == 0) &&
              (!code_y || (uintptr_t)code_y % 16 == 0);
    if (!map_xy) {
        a4 = a4 && (pitch % 4 == 0);
        for (int p = 0; p < nplanes; p++) a4 = a4 && ((uintptr_t)pl.p[p] % 4 == 0);
    }
#define SLR_GRAY_LAUNCH(V, RECT)                                                                              \
    SLR_LAUNCH((gray_decode_kernel<V, RECT>), dim3(pick_blocks((size_t)(W / V) * H)), dim3(256), 0, s, \
                       pl, n_col_bits, n_row_bits, pitch, W, H, black_thr, white_thr, scan_w, scan_h, map_xy,  \
                       map_frac, code_x, code_y, valid)
    if (map_xy) { if (a4) SLR_GRAY_LAUNCH(4, true); else SLR_GRAY_LAUNCH(1, true); }
    else        { if (a4) SLR_GRAY_LAUNCH(4, false); else SLR_GRAY_LAUNCH(1, false); }
#undef SLR_GRAY_LAUNCH
    return hipGetLastError();
}

}  // namespace slr
