// kernels_mfn.hip -- BUILD EXTENSION, no reference counterpart (BASELINE config 5, SURVEY 8d: "parity unpinned"):
// generalised multi-frequency phase-shift decode for n_freq frequencies x n_step equally spaced shifts with fp16
// image planes and f32 accumulation.  gfx950 (MI355X) only.
//
// The reference is hard-wired to 3 frequencies x 4 steps of u8 (mfreconstruct.cpp:21-22,237-242); this kernel keeps
// its STRUCTURE -- shadow mask from white/black (:190-207), one wrapped phase per frequency, a heterodyne cascade of
// neighbouring differences with the reference's "a > b ? a-b : a-b+2pi" rule (:265-267), output scaled to 0..255
// (:268) so that K4's 0.1 threshold keeps its meaning -- and replaces what cannot generalise:
//   wrapped phase   I_k = A + B cos(phi + 2 pi k / N)  =>  phi = atan2(-sum I_k sin(2 pi k/N), sum I_k cos(2 pi k/N)),
//                   brought to [0, 2 pi)                   (the reference's 4-step integer-quotient atan is Q1/Q2)
//   PI              the f32 nearest to pi                   (the reference's 3.1416f is Q3)
//   cascade         level 0: P_0..P_{F-1};  level l+1: D_i = wrap(D_i - D_{i+1});  the single value of level F-1
// Pure streaming: (2 + F*N) fp16 reads + 4 + 1 bytes written per pixel (73 B/px at 4 x 8) -> HBM-bound.
#include "slr_device.hpp"

#include <hip/hip_fp16.h>
#include <math.h>

#pragma clang diagnostic ignored "-Wpass-failed"   // the run-time (FS = NS = 0) variant cannot fully unroll its plane loops

// SLR_MFN_ABL (timing only, wrong results; LABNOTES section 10 r4-h, section 11 r5-o): 1 no source traffic, 2 no tap reads / blend, 4 no output
// stores, 8 no group barriers, 16 no per-tile box reduction (a fixed full-size box: more traffic than the real boxes, not a bound), 32 no
// per-pixel set-up (what a per-pixel digest could save at best: 4.5 %, profiles/exp/r05/mfn_setup_abl.txt)
#if !defined(SLR_EXPERIMENTS) && defined(SLR_MFN_ABL)
#error "SLR_MFN_ABL is an experiment switch: build with -DSLR_EXPERIMENTS"
#endif

namespace slr {

constexpr float kTruePI = 3.14159265358979323846f;
constexpr float kTrue2PI = 2 * kTruePI;

// The wrapped phase atan2(y, x) brought into [0, 2 pi) -- what every config-5 kernel computes four times per pixel.  The library
// atan2f costs ~40 instructions (argument scaling against overflow, infinities, NaNs) + 3 for the range; the DFT bins of a decode
// are finite and far from both ends of the exponent range, so: t = min / max by v_rcp_f32 (1 ulp), atan t on [0, 1] as t times a
// degree-7 polynomial in t^2 (weighted least-squares minimax fit, |error| < 1.4e-7 in f32 arithmetic, i.e. the library's 2 ulp
// at pi), three selects for the octant, the half plane and the sign: 22 instructions.  Config 5 is a build extension validated
// against an fp64 model within 1e-4 relative + an f32 floor (tests/test_mfn_extension.py); every kernel form uses THIS function, so
// the forms still agree bit for bit.  x = y = 0 gives 0 (the pixel fails the modulation test anyway).
__device__ __forceinline__ float mfn_phase_of(float y, float x)
{
    const float ax = __builtin_fabsf(x), ay = __builtin_fabsf(y);
    const float mx = __builtin_fmaxf(__builtin_fmaxf(ax, ay), 1e-30f), mn = __builtin_fminf(ax, ay);
    const float t = mn * __builtin_amdgcn_rcpf(mx);
    const float u = t * t;
    float q = -4.054523073e-03f;
    q = __builtin_fmaf(q, u, 2.186279185e-02f);
    q = __builtin_fmaf(q, u, -5.591207743e-02f);
    q = __builtin_fmaf(q, u, 9.642177820e-02f);
    q = __builtin_fmaf(q, u, -1.390862167e-01f);
    q = __builtin_fmaf(q, u, 1.994656473e-01f);
    q = __builtin_fmaf(q, u, -3.332985938e-01f);
    q = __builtin_fmaf(q, u, 9.999993443e-01f);
    float a = q * t;                                     // atan t, 0 <= t <= 1
    a = ay > ax ? 0.5f * kTruePI - a : a;                // [0, pi / 2]
    a = x < 0.0f ? kTruePI - a : a;                      // [0, pi]
    return y < 0.0f ? kTrue2PI - a : a;                  // [0, 2 pi)
}
// One frequency's DFT bin of EIGHT equally spaced steps by its butterflies: with a_k = s_k - s_(k+4),
//   C = sum s_k cos(2 pi k / 8) = a_0 + sqrt(1/2) (a_1 - a_3),   S = sum s_k sin(2 pi k / 8) = a_2 + sqrt(1/2) (a_1 + a_3)
// -- 8 instructions per pixel and frequency where the table form spends 32 (16 multiplies and 16 adds: the library is built without
// contraction).  Any common power-of-two scale of the samples scales S and C exactly (the rectifying forms' samples are x 1024).
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
// The rectified sample (x 1024) of a tap pair pair (u0: row sy, u1: row sy + 1) and its weights, and the same SUBTRACTED from a
// running value -- steps 4..7 of an 8-step frequency fold into a_k = s_k - s_(k+4) through the dot products' own accumulator
// (negated weights, exact in binary16): a - t00 w00 - t01 w01 - t10 w10 - t11 w11, one rounding per v_dot2, no separate subtract.
// Every rectifying form uses these two, in this order.
__device__ __forceinline__ float mfn_sample(unsigned u0, unsigned u1, h16x2 w0, h16x2 w1)
{
    float a;
    asm("v_dot2_f32_f16 %0, %1, %2, 0" : "=v"(a) : "v"(u0), "v"(w0));       // (VOP3P with an inline 0: the VOP2 form needs a v_mov first)
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(h16x2, u1), w1, a, false);
}
__device__ __forceinline__ float mfn_sample_sub(float acc, unsigned u0, unsigned u1, h16x2 nw0, h16x2 nw1)   // nw = -w
{
    const float a = __builtin_amdgcn_fdot2(__builtin_bit_cast(h16x2, u0), nw0, acc, false);
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(h16x2, u1), nw1, a, false);
}
constexpr float kSqrtHalf = 0.70710678118654752440f;
__device__ __forceinline__ void mfn_bin8(float a0, float a1, float a2, float a3, float &S, float &C)
{
    C = __builtin_fmaf(kSqrtHalf, a1 - a3, a0);
    S = __builtin_fmaf(kSqrtHalf, a1 + a3, a2);
}

struct MfnPlanes { const uint16_t *p[SLR_MFN_MAX_PLANES]; };
struct MfnTrig { float cs[SLR_MFN_MAX_STEPS], sn[SLR_MFN_MAX_STEPS]; };

__device__ __forceinline__ float h2f(unsigned short b) { return __half2float(__ushort_as_half(b)); }
typedef unsigned u32a2 __attribute__((aligned(2)));      // a dword at a 2-byte aligned address (two adjacent binary16 taps)

// V pixels per thread: 4 (8-byte loads; W % 4 == 0, aligned planes) or 1.  FS x NS != 0: compile-time frequency and
// step counts -- the plane loop unrolls completely and all 2 + FS*NS loads of a thread are in flight at once (the
// run-time loop keeps one load per iteration in flight, which costs ~25 % of the HBM rate at 4 x 8).
template <int V, int FS, int NS>
__global__ __launch_bounds__(256) void mfn_decode_kernel(MfnPlanes pl, MfnTrig tr, int n_freq_rt, int n_step_rt, int pitch, int W,
                                                         int H, float black_thr, float *__restrict__ phase,
                                                         uint8_t *__restrict__ valid)
{
    const int n_freq = FS ? FS : n_freq_rt, n_step = NS ? NS : n_step_rt;
    const unsigned gpr = (unsigned)(W / V);
    const unsigned total = gpr * (unsigned)H;
    for (unsigned g = blockIdx.x * 256u + threadIdx.x; g < total; g += gridDim.x * 256u) {
        const unsigned row = g / gpr, col0 = (g - row * gpr) * V;
        const size_t so = (size_t)row * pitch + col0, oo = (size_t)row * W + col0;
        float wh[V], bk[V];
        auto load = [&](int p, float out[V]) {
            if constexpr (V == 4) {
                const uint2 w = *reinterpret_cast<const uint2 *>(pl.p[p] + so);
                out[0] = h2f((unsigned short)(w.x & 0xFFFFu)); out[1] = h2f((unsigned short)(w.x >> 16));
                out[2] = h2f((unsigned short)(w.y & 0xFFFFu)); out[3] = h2f((unsigned short)(w.y >> 16));
            } else {
                out[0] = h2f(pl.p[p][so]);
            }
        };
        load(0, wh);
        load(1, bk);
        float D[V][SLR_MFN_MAX_FREQ];
        bool ok[V];
        const float mod2 = (0.25f * n_step) * (0.25f * n_step);        // (B_min * N / 2)^2 with B_min = 0.5
#pragma unroll
        for (int v = 0; v < V; v++) ok[v] = wh[v] - bk[v] > black_thr;
#pragma unroll
        for (int f = 0; f < n_freq; f++) {
            float S[V], C[V];
            if constexpr (NS == 8) {
                float a[4][V];
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    float I[V];
                    load(2 + f * 8 + k, I);
#pragma unroll
                    for (int v = 0; v < V; v++) a[k & 3][v] = k < 4 ? I[v] : a[k & 3][v] - I[v];
                }
#pragma unroll
                for (int v = 0; v < V; v++) mfn_bin8(a[0][v], a[1][v], a[2][v], a[3][v], S[v], C[v]);
            } else {
#pragma unroll
                for (int v = 0; v < V; v++) S[v] = C[v] = 0.0f;
#pragma unroll
                for (int k = 0; k < n_step; k++) {
                    float I[V];
                    load(2 + f * n_step + k, I);
#pragma unroll
                    for (int v = 0; v < V; v++) { S[v] += I[v] * tr.sn[k]; C[v] += I[v] * tr.cs[k]; }
                }
            }
#pragma unroll
            for (int v = 0; v < V; v++) {
                const float p = mfn_phase_of(-S[v], C[v]);
                // modulation B = 2/N * |DFT bin| below half a grey level: the phase is noise (cf. Q5) -> invalid
                ok[v] = ok[v] && (S[v] * S[v] + C[v] * C[v] > mod2);
#pragma unroll
                for (int q = 0; q < SLR_MFN_MAX_FREQ; q++) if (q == f) D[v][q] = p;   // static indexing keeps D in registers
            }
        }
#pragma unroll
        for (int lvl = 1; lvl < SLR_MFN_MAX_FREQ; lvl++) {
            if (lvl < n_freq) {
#pragma unroll
                for (int i = 0; i + 1 < SLR_MFN_MAX_FREQ; i++) {
                    if (i + lvl < n_freq) {
#pragma unroll
                        for (int v = 0; v < V; v++) {
                            const float a = D[v][i], b = D[v][i + 1];
                            D[v][i] = (a > b) ? (a - b) : (a - b + kTrue2PI);
                        }
                    }
                }
            }
        }
        float out[V];
        unsigned vw = 0;
#pragma unroll
        for (int v = 0; v < V; v++) {
            out[v] = (wh[v] - bk[v] > black_thr) ? D[v][0] / kTrue2PI * 255 : 0.0f;
            vw |= (ok[v] ? 1u : 0u) << (8 * v);
        }
        if constexpr (V == 4) {
            *reinterpret_cast<float4 *>(phase + oo) = make_float4(out[0], out[1], out[2], out[3]);
            *reinterpret_cast<unsigned *>(valid + oo) = vw;
        } else {
            phase[oo] = out[0];
            valid[oo] = (uint8_t)vw;
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// Rectification for the generalised decode (BASELINE config 5 through the whole path; build extension, parity unpinned).
// cv::remap has no fp16 mode in OpenCV 2.4, so the GEOMETRY is the reference's -- map1 / map2 of cv::initUndistortRectifyMap as
// stereoRect::doStereoRectify applies them (stereorect.cpp:26-34): integer source position, two 5-bit fractions, the four taps
// (sx, sy) .. (sx + 1, sy + 1), BORDER_CONSTANT 0 -- and the ARITHMETIC is f32:
//     w00 = (32 - fx)(32 - fy), w01 = fx (32 - fy), w10 = (32 - fx) fy, w11 = fx fy            (integers, sum 1024)
//     1024 x sample = dot2((t10, t11), (w10, w11), dot2((t00, t01), (w00, w01), 0))              (v_dot2_f32_f16: f32 accumulation)
// (a product of an fp16 tap and a weight of at most 2^10 is exact in f32; the four of them are summed row sy first, then row
// sy + 1), and the sample goes into the DFT sums as f32: nothing is rounded back to fp16.  Fused with mfn_decode_kernel's decode.
// Row bands (one 8192 x 6000 frame over 8 GPUs): a launch decodes destination rows [row0, row0 + rows) into a
// band-sized output; the planes hold SOURCE rows [src_row0, src_row0 + src_rows) only (plane pointers address source row
// src_row0).  A tap outside that window reads 0 like a tap outside the image; slr_rectify_source_rows gives the window a band
// needs, and with it a band's result is the whole frame's, bit for bit.
// This is the per-pixel gather form (the two taps of a source row are one 4-byte load through L1 / L2; a quad of 4 adjacent pixels
// per thread so that neighbouring lanes touch neighbouring lines): the LDS-tiled form of the u8 path is not built for fp16 yet.
// ------------------------------------------------------------------------------------------------------
// Where the taps come from.  A contiguous stack (equally spaced planes, what BASELINE's configurations hold in HBM) is ONE buffer
// descriptor: the plane is the scalar offset of a buffer load, a tap's position a 32-bit vector offset that is computed once per
// pixel -- no 64-bit address per load (the first version computed 272 of them per thread, kept two VGPRs each and spilled 600
// registers; its 34 plane pointers alone were 68 scalar registers, re-read from vector lanes before every load).  Any other
// layout takes the pointer form.  pair(): the two binary16 taps of a source row as one dword (2-byte aligned: buffer and global
// loads take it); half(): one tap.
struct MfnPtrPlanes {
    MfnPlanes pl;
    __device__ __forceinline__ unsigned pair(int p, unsigned e) const { return *reinterpret_cast<const u32a2 *>(pl.p[p] + e); }
    __device__ __forceinline__ unsigned half(int p, unsigned e) const { return pl.p[p][e]; }
};
struct MfnStridedPlanes {
    __amdgpu_buffer_rsrc_t rsrc; unsigned stride_bytes;
    __device__ __forceinline__ unsigned pair(int p, unsigned e) const { return (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rsrc, e * 2u, (unsigned)p * stride_bytes, 0); }
    __device__ __forceinline__ unsigned half(int p, unsigned e) const { return (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rsrc, e * 2u, (unsigned)p * stride_bytes, 0); }
};
struct MfnStridedArg { const uint16_t *base; unsigned stride_bytes, total_bytes; };   // (the descriptor is made in the kernel)
template <int N> struct MfnTrigN { float cs[N], sn[N]; };   // cos / sin (2 pi k / n_step) / 1024, k < n_step (N = 16: any step count)

// the decode of one thread's V pixels; INSIDE: every footprint lies inside the image and the source window -> pair loads, no masks
template <int V, int FS, int NS, bool INSIDE, typename Src, typename Trig>
__device__ __forceinline__ void mfn_rect_quad(const Src &src, const Trig &tr, int n_freq, int n_step, unsigned pitch, float black_thr,
                                              const unsigned off[V], const unsigned inb[V], const h16x2 w0[V], const h16x2 w1[V],
                                              float out[V], unsigned &vw)
{
    unsigned o01[V], o10[V], o11[V], o00[V], k0[V], k1[V];
    if constexpr (!INSIDE) {                             // clamped (always readable) offsets + masks: no branch per tap
#pragma unroll
        for (int v = 0; v < V; v++) {
            o00[v] = (inb[v] & 1u) ? off[v] : 0u; o01[v] = (inb[v] & 2u) ? off[v] + 1u : 0u;
            o10[v] = (inb[v] & 4u) ? off[v] + pitch : 0u; o11[v] = (inb[v] & 8u) ? off[v] + pitch + 1u : 0u;
            k0[v] = ((inb[v] & 1u) ? 0xFFFFu : 0u) | ((inb[v] & 2u) ? 0xFFFF0000u : 0u);
            k1[v] = ((inb[v] & 4u) ? 0xFFFFu : 0u) | ((inb[v] & 8u) ? 0xFFFF0000u : 0u);
        }
    }
    // smp[v] = 1024 x the rectified sample: the four exact products summed by two v_dot2_f32_f16 -- row sy first, then row sy + 1
    auto taps = [&](int p, int v, unsigned &u0, unsigned &u1) {
        if constexpr (INSIDE) { u0 = src.pair(p, off[v]); u1 = src.pair(p, off[v] + pitch); }
        else {
            u0 = (src.half(p, o00[v]) | src.half(p, o01[v]) << 16) & k0[v];
            u1 = (src.half(p, o10[v]) | src.half(p, o11[v]) << 16) & k1[v];
        }
    };
    auto load = [&](int p, float smp[V]) {
#pragma unroll
        for (int v = 0; v < V; v++) {
            unsigned u0, u1;
            taps(p, v, u0, u1);
            smp[v] = mfn_sample(u0, u1, w0[v], w1[v]);
        }
    };
    auto load_sub = [&](int p, float acc[V]) {           // acc -= the sample, through the dot products' accumulator (mfn_sample_sub)
#pragma unroll
        for (int v = 0; v < V; v++) {
            unsigned u0, u1;
            taps(p, v, u0, u1);
            acc[v] = mfn_sample_sub(acc[v], u0, u1, -w0[v], -w1[v]);
        }
    };
    // (the samples stay scaled by 1024 -- an exact power of two: the shadow test compares against 1024 x the threshold and the
    //  DFT uses sin / cos scaled by 1 / 1024 (tr: scaled by the launcher), so S and C are the sums of the UNscaled samples, rounded
    //  exactly as mfn_decode_kernel rounds them; with identity maps the two kernels agree bit for bit)
    float wh[V], bk[V];
    load(0, wh);
    load(1, bk);
    float D[V][SLR_MFN_MAX_FREQ];
    bool ok[V];
    // (NS == 8: the butterflies work on the x 1024 samples themselves, S and C come out x 1024)
    const float mod2 = (0.25f * n_step) * (0.25f * n_step) * (NS == 8 ? 1048576.0f : 1.0f);
    const float thr1024 = black_thr * 1024.0f;
#pragma unroll
    for (int v = 0; v < V; v++) ok[v] = wh[v] - bk[v] > thr1024;
#pragma unroll
    for (int f = 0; f < (FS ? FS : SLR_MFN_MAX_FREQ); f++) {
        if (f < n_freq) {
            float S[V], C[V];
            float a[4][V];
#pragma unroll
            for (int v = 0; v < V; v++) S[v] = C[v] = 0.0f;
#pragma unroll
            for (int k = 0; k < (NS ? NS : SLR_MFN_MAX_STEPS); k++) {
                if (k < n_step) {
                    if constexpr (NS == 8) {
                        if (k < 4) load(2 + f * 8 + k, a[k & 3]); else load_sub(2 + f * 8 + k, a[k & 3]);
                    } else {
                        float I[V];
                        load(2 + f * n_step + k, I);
#pragma unroll
                        for (int v = 0; v < V; v++) { S[v] += I[v] * tr.sn[k]; C[v] += I[v] * tr.cs[k]; }
                    }
                    // four planes' tap loads (2 x 4 x V dwords) are in flight together; later ones must not be hoisted above this
                    // point (all 272 of a 4 x 8 stack at once need 300 registers -- one wave per SIMD --, a whole frequency's 64
                    // still spill at the 128 registers of four waves per SIMD)
                    if ((k & 3) == 3) asm volatile("" ::: "memory");
                }
            }
            asm volatile("" ::: "memory");
            if constexpr (NS == 8) {
#pragma unroll
                for (int v = 0; v < V; v++) mfn_bin8(a[0][v], a[1][v], a[2][v], a[3][v], S[v], C[v]);
            }
#pragma unroll
            for (int v = 0; v < V; v++) {
                const float p = mfn_phase_of(-S[v], C[v]);
                // modulation B = 2/N * |DFT bin| below half a grey level: the phase is noise (cf. Q5) -> invalid
                ok[v] = ok[v] && (S[v] * S[v] + C[v] * C[v] > mod2);
#pragma unroll
                for (int q = 0; q < SLR_MFN_MAX_FREQ; q++) if (q == f) D[v][q] = p;
            }
        }
    }
#pragma unroll
    for (int lvl = 1; lvl < SLR_MFN_MAX_FREQ; lvl++) {
        if (lvl < n_freq) {
#pragma unroll
            for (int i = 0; i + 1 < SLR_MFN_MAX_FREQ; i++) {
                if (i + lvl < n_freq) {
#pragma unroll
                    for (int v = 0; v < V; v++) {
                        const float a = D[v][i], b = D[v][i + 1];
                        D[v][i] = (a > b) ? (a - b) : (a - b + kTrue2PI);
                    }
                }
            }
        }
    }
    vw = 0;
#pragma unroll
    for (int v = 0; v < V; v++) {
        out[v] = (wh[v] - bk[v] > thr1024) ? D[v][0] / kTrue2PI * 255 : 0.0f;
        vw |= (ok[v] ? 1u : 0u) << (8 * v);
    }
}

// Pixels per thread of the specialised 4 x 8 instances: ONE.  Measured at 8192 x 6000 (profiles/exp/r04/mfn_time.py): a quad per thread
// (272 tap loads per iteration, 168 registers, 390 of them spilled) 6.6 / 8.4 ms per camera, one pixel per thread (77 registers, no
// spill) 2.2 / 2.4 ms -- the same at 4, 6 and 8 waves per SIMD, i.e. bound by the taps' path through L1 / L2, not by latency.
constexpr int kMfnRectV = 1;
template <int V, int FS, int NS, typename Arg>
__global__ __launch_bounds__(256, (V == 1 ? 6 : 3)) void mfn_rect_decode_kernel(Arg arg, MfnTrigN<(NS ? NS : SLR_MFN_MAX_STEPS)> tr, int n_freq_rt,
                                                                 int n_step_rt, int pitch, int W, int H, float black_thr,
                                                                 const int16_t *__restrict__ map_xy, const uint16_t *__restrict__ map_frac,
                                                                 int row0, int rows, int src_row0, int src_rows,
                                                                 float *__restrict__ phase, uint8_t *__restrict__ valid)
{
    const int n_freq = FS ? FS : n_freq_rt, n_step = NS ? NS : n_step_rt;
    auto make_src = [&]() {
        if constexpr (sizeof(Arg) == sizeof(MfnStridedArg))
            return MfnStridedPlanes{__builtin_amdgcn_make_buffer_rsrc((void *)arg.base, 0, (int)arg.total_bytes, 0x00020000), arg.stride_bytes};
        else
            return arg;
    };
    const auto src = make_src();
    const unsigned gpr = (unsigned)(W / V);
    const unsigned total = gpr * (unsigned)rows;
    // XCD-aware order (as mf_rect_decode_kernel): workgroup b runs on XCD b % 8; every XCD gets a contiguous band of rows, so the
    // source rows shared by vertically adjacent destination rows are served by one L2
    const unsigned nb = gridDim.x, per = (nb + 7) / 8;
    unsigned vb = (blockIdx.x % 8) * per + blockIdx.x / 8;
    if (nb % 8 != 0) vb = blockIdx.x;
    for (unsigned g = vb * 256u + threadIdx.x; g < total; g += nb * 256u) {
        const unsigned brow = g / gpr, col0 = (g - brow * gpr) * V;
        const size_t m = (size_t)(brow + (unsigned)row0) * W + col0, oo = (size_t)brow * W + col0;
        unsigned off[V];                                 // element offset of the upper left tap inside the plane window
        unsigned inb[V];                                 // bit 0..3: tap (0,0) (0,1) (1,0) (1,1) is readable
        h16x2 w0[V], w1[V];                              // (w00, w01), (w10, w11): integers up to 1024, exact in binary16
#pragma unroll
        for (int v = 0; v < V; v++) {
            const int sx = map_xy[2 * (m + v)], sy = map_xy[2 * (m + v) + 1];
            const unsigned f = map_frac[m + v] & 1023u, fx = f & 31u, fy = f >> 5;
            w0[v] = h16x2{(_Float16)(float)((32u - fx) * (32u - fy)), (_Float16)(float)(fx * (32u - fy))};
            w1[v] = h16x2{(_Float16)(float)((32u - fx) * fy), (_Float16)(float)(fx * fy)};
            const bool x0 = (unsigned)sx < (unsigned)W, x1 = (unsigned)(sx + 1) < (unsigned)W;
            const bool y0 = (unsigned)sy < (unsigned)H && (unsigned)(sy - src_row0) < (unsigned)src_rows;
            const bool y1 = (unsigned)(sy + 1) < (unsigned)H && (unsigned)(sy + 1 - src_row0) < (unsigned)src_rows;
            inb[v] = (x0 && y0 ? 1u : 0u) | (x1 && y0 ? 2u : 0u) | (x0 && y1 ? 4u : 0u) | (x1 && y1 ? 8u : 0u);
            off[v] = (unsigned)((sy - src_row0) * pitch + sx);     // (only used where a tap is readable)
        }
        float out[V];
        unsigned vw;
        const bool inside = (inb[0] & inb[V > 1 ? 1 : 0] & inb[V > 2 ? 2 : 0] & inb[V > 3 ? 3 : 0]) == 15u;
        if (inside) mfn_rect_quad<V, FS, NS, true>(src, tr, n_freq, n_step, (unsigned)pitch, black_thr, off, inb, w0, w1, out, vw);
        else if (src_rows > 0) mfn_rect_quad<V, FS, NS, false>(src, tr, n_freq, n_step, (unsigned)pitch, black_thr, off, inb, w0, w1, out, vw);
        else {                                           // (no source rows at all: every sample is 0)
#pragma unroll
            for (int v = 0; v < V; v++) out[v] = 0.0f;
            vw = 0;
        }
        if constexpr (V == 4) {
            *reinterpret_cast<float4 *>(phase + oo) = make_float4(out[0], out[1], out[2], out[3]);
            *reinterpret_cast<unsigned *>(valid + oo) = vw;
        } else {
            phase[oo] = out[0];
            valid[oo] = (uint8_t)vw;
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// The LDS-tiled form of the rectifying 4 x 8 decode (BASELINE config 5; a contiguous stack, W % 8 == 0).  The gather form above
// sends every tap through L1 / L2 (68 four-byte loads per pixel: 2.2 ms per 8192 x 6000 camera whatever the occupancy); here a
// 256-thread workgroup owns a 64 x 16 destination tile: the tile's source box (bounding box of its map entries, found with wave
// reductions; x0 aligned to 16 bytes, at most 80 x 24 elements: keystone slopes up to ~0.1) is copied plane by plane into LDS with
// 16-byte buffer loads -- chunks outside the image or the source window become zeros, which IS BORDER_CONSTANT -- four planes at a
// time, double buffered (the loads of the next four planes are in flight while these are decoded; one barrier per group); every
// tap pair is two aligned LDS dwords and a v_alignbit.  Same arithmetic, same order as the gather form: bit-identical results (a
// tile whose box does not fit runs the gather code).  A thread's 4 pixels sit in 4 different tile rows, so a wave stores 256
// contiguous bytes of phases.
// ------------------------------------------------------------------------------------------------------
constexpr int kTileW = 64, kTileH = 16, kBoxW = 80, kBoxH = 24, kTileG = 4;      // pixels, box elements, planes per group
constexpr int kTileNT = 512, kTileNW = kTileNT / 64, kTilePx = kTileW * kTileH / kTileNT;   // threads, waves, pixels per thread (2: rows wv, wv + 8)
constexpr int kBoxRowBytes = kBoxW * 2, kBoxPlaneBytes = kBoxRowBytes * kBoxH, kBoxChunks = (kBoxW / 8) * kBoxH;

constexpr int kTileLoads = (kTileG * kBoxChunks + kTileNT - 1) / kTileNT;                    // 16-byte loads per thread and group

__global__ __launch_bounds__(kTileNT, 4) void mfn_rect_tile_kernel(MfnStridedArg arg, MfnTrigN<8> tr, int pitch, int W, int H, float black_thr,
                                                               const int16_t *__restrict__ map_xy, const uint16_t *__restrict__ map_frac,
                                                               int row0, int rows, int src_row0, int src_rows, int tiles_x,
                                                               float *__restrict__ phase, uint8_t *__restrict__ valid)
{
    __shared__ __attribute__((aligned(16))) unsigned char box[2][kTileG][kBoxPlaneBytes];
    __shared__ int red[kTileNW][4];
    const MfnStridedPlanes src{__builtin_amdgcn_make_buffer_rsrc((void *)arg.base, 0, (int)arg.total_bytes, 0x00020000), arg.stride_bytes};
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int tiles_y = (rows + kTileH - 1) / kTileH, ntiles = tiles_x * tiles_y;
    // XCD-banded tile order (workgroup b runs on XCD b % 8): an XCD walks a contiguous band of tile rows
    const unsigned nb = gridDim.x, per = (nb + 7) / 8;
    unsigned vb = (blockIdx.x % 8) * per + blockIdx.x / 8;
    if (nb % 8 != 0) vb = blockIdx.x;
    for (int t = (int)vb; t < ntiles; t += (int)nb) {
        const int ty = t / tiles_x, tx = t - ty * tiles_x;
        const int col = tx * kTileW + lane;
        // this thread's pixels: column col, band rows ty*16 + wv + kTileNW j
        int sx[kTilePx], sy[kTilePx];
        unsigned fr[kTilePx];
        bool live[kTilePx];
        int mnx = 0x7FFFFFFF, mxx = -0x7FFFFFFF, mny = 0x7FFFFFFF, mxy = -0x7FFFFFFF;
#pragma unroll
        for (int j = 0; j < kTilePx; j++) {
            const int brow = ty * kTileH + wv + kTileNW * j;
            live[j] = brow < rows && col < W;
            sx[j] = sy[j] = 0; fr[j] = 0;
            if (live[j]) {
                const size_t m = (size_t)(brow + row0) * W + col;
                sx[j] = map_xy[2 * m]; sy[j] = map_xy[2 * m + 1]; fr[j] = map_frac[m] & 1023u;
                if (!(sx[j] >= W || sx[j] + 1 < 0 || sy[j] >= H || sy[j] + 1 < 0)) {       // footprints completely outside read 0
                    mnx = sx[j] < mnx ? sx[j] : mnx; mxx = sx[j] > mxx ? sx[j] : mxx;
                    mny = sy[j] < mny ? sy[j] : mny; mxy = sy[j] > mxy ? sy[j] : mxy;
                }
            }
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            int v;
            v = __shfl_xor(mnx, d); mnx = v < mnx ? v : mnx;
            v = __shfl_xor(mxx, d); mxx = v > mxx ? v : mxx;
            v = __shfl_xor(mny, d); mny = v < mny ? v : mny;
            v = __shfl_xor(mxy, d); mxy = v > mxy ? v : mxy;
        }
        __syncthreads();                                  // (the previous tile's readers of red[] and box[] are done)
        if (lane == 0) { red[wv][0] = mnx; red[wv][1] = mxx; red[wv][2] = mny; red[wv][3] = mxy; }
        __syncthreads();
#pragma unroll
        for (int w = 0; w < kTileNW; w++) {
            mnx = red[w][0] < mnx ? red[w][0] : mnx; mxx = red[w][1] > mxx ? red[w][1] : mxx;
            mny = red[w][2] < mny ? red[w][2] : mny; mxy = red[w][3] > mxy ? red[w][3] : mxy;
        }
        const bool empty = mnx > mxx;                     // the tile samples nothing
        const int x0 = empty ? 0 : (mnx & ~7), y0 = empty ? 0 : mny;
        const bool fits = empty || ((mxx + 1) - x0 < kBoxW && (mxy + 1) - y0 < kBoxH);
        float out[kTilePx];
        unsigned okm = 0;                                 // bit j: pixel j valid
        if (!fits) {                                      // (block-uniform) a box beyond the LDS image: the gather code, pixel by pixel
#pragma unroll 1
            for (int j = 0; j < kTilePx; j++) {
                out[j] = 0.0f;
                if (!live[j]) continue;
                const unsigned fx = fr[j] & 31u, fy = fr[j] >> 5;
                const h16x2 w0[1] = {h16x2{(_Float16)(float)((32u - fx) * (32u - fy)), (_Float16)(float)(fx * (32u - fy))}};
                const h16x2 w1[1] = {h16x2{(_Float16)(float)((32u - fx) * fy), (_Float16)(float)(fx * fy)}};
                const bool bx0 = (unsigned)sx[j] < (unsigned)W, bx1 = (unsigned)(sx[j] + 1) < (unsigned)W;
                const bool by0 = (unsigned)sy[j] < (unsigned)H && (unsigned)(sy[j] - src_row0) < (unsigned)src_rows;
                const bool by1 = (unsigned)(sy[j] + 1) < (unsigned)H && (unsigned)(sy[j] + 1 - src_row0) < (unsigned)src_rows;
                const unsigned inb[1] = {(bx0 && by0 ? 1u : 0u) | (bx1 && by0 ? 2u : 0u) | (bx0 && by1 ? 4u : 0u) | (bx1 && by1 ? 8u : 0u)};
                const unsigned off[1] = {(unsigned)((sy[j] - src_row0) * pitch + sx[j])};
                float o1[1];
                unsigned vw1 = 0;
                if (src_rows > 0) mfn_rect_quad<1, 4, 8, false>(src, tr, 4, 8, (unsigned)pitch, black_thr, off, inb, w0, w1, o1, vw1);
                else o1[0] = 0.0f;
                out[j] = o1[0];
                okm |= (vw1 & 1u) << j;
            }
        } else {
            // LDS byte offset of the pixel's upper left tap in a plane image (rounded down to its dword), the alignbit shift, weights
            unsigned ta[kTilePx], tsh[kTilePx];
            h16x2 w0[kTilePx], w1[kTilePx];
#pragma unroll
            for (int j = 0; j < kTilePx; j++) {
                const unsigned fx = fr[j] & 31u, fy = fr[j] >> 5;
                w0[j] = h16x2{(_Float16)(float)((32u - fx) * (32u - fy)), (_Float16)(float)(fx * (32u - fy))};
                w1[j] = h16x2{(_Float16)(float)((32u - fx) * fy), (_Float16)(float)(fx * fy)};
                const bool touch = live[j] && !(sx[j] >= W || sx[j] + 1 < 0 || sy[j] >= H || sy[j] + 1 < 0);
                const unsigned b = touch ? (unsigned)((sy[j] - y0) * kBoxRowBytes + (sx[j] - x0) * 2) : 0u;
                ta[j] = b & ~3u; tsh[j] = (b & 2u) * 8u;
                if (!touch) { w0[j] = h16x2{(_Float16)0.0f, (_Float16)0.0f}; w1[j] = w0[j]; }       // every sample 0
            }
            // this thread's chunks of a group: id = threadIdx + 256 i -> (plane of the group, box row, 16-byte column)
            unsigned cvoff[kTileLoads], cdst[kTileLoads];
            bool cin[kTileLoads];
#pragma unroll
            for (int i = 0; i < kTileLoads; i++) {
                const int id = (int)threadIdx.x + kTileNT * i, pg = id / kBoxChunks, w = id - pg * kBoxChunks, r = w / (kBoxW / 8), c = w - r * (kBoxW / 8);
                const int gy = y0 + r, gx = x0 + 8 * c;
                cin[i] = pg < kTileG && !empty && (unsigned)gy < (unsigned)H && (unsigned)(gy - src_row0) < (unsigned)src_rows && (unsigned)gx < (unsigned)W;
                cvoff[i] = cin[i] ? (unsigned)(((gy - src_row0) * pitch + gx) * 2) : 0u;
                cdst[i] = (unsigned)(pg * kBoxPlaneBytes + r * kBoxRowBytes + c * 16);
            }
            typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
            u32x4_t preA[kTileLoads], preB[kTileLoads];      // two groups in flight: the loads of a group get two decode phases of time
            auto fetch = [&](u32x4_t pre[kTileLoads], int first, int count) {   // planes first .. first + count - 1 -> image slots 0 .. count - 1
#pragma unroll
                for (int i = 0; i < kTileLoads; i++) {
                    const int id = (int)threadIdx.x + kTileNT * i, pg = id / kBoxChunks;
                    const bool on = cin[i] && pg < count;
                    // (the plane of the group goes into the VECTOR offset: a per-thread scalar offset makes the compiler loop over its values)
                    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(src.rsrc, cvoff[i] + (unsigned)(pg < count ? pg : 0) * src.stride_bytes,
                                                                            (unsigned)first * src.stride_bytes, 0);
                    pre[i] = on ? v : u32x4_t{0u, 0u, 0u, 0u};
                }
            };
            auto commit = [&](const u32x4_t pre[kTileLoads], int buf) {
#pragma unroll
                for (int i = 0; i < kTileLoads; i++) {
                    const int id = (int)threadIdx.x + kTileNT * i;
                    if (id < kTileG * kBoxChunks) *reinterpret_cast<u32x4_t *>(&box[buf][0][0] + cdst[i]) = pre[i];
                }
            };
            float wh[kTilePx], bk[kTilePx], L0[kTilePx], L1[kTilePx], L2[kTilePx], fin[kTilePx];
            bool ok[kTilePx];
            const float mod2 = 4.0f * 1048576.0f, thr1024 = black_thr * 1024.0f;      // (0.25 * 8)^2, on S and C of the x 1024 samples
            // the rectified samples (x 1024) of plane image `img` for the thread's 4 pixels
            auto samples = [&](const unsigned char *img, float smp[kTilePx], bool sub = false) {   // sub: smp -= the samples
                asm volatile("" ::: "memory");                // (one plane's tap reads at a time: hoisting four planes' reads spills)
#pragma unroll
                for (int j = 0; j < kTilePx; j++) {
                    const unsigned *r0 = reinterpret_cast<const unsigned *>(img + ta[j]);
                    const unsigned *r1 = reinterpret_cast<const unsigned *>(img + ta[j] + kBoxRowBytes);
                    const unsigned u0 = __builtin_amdgcn_alignbit(r0[1], r0[0], tsh[j]);
                    const unsigned u1 = __builtin_amdgcn_alignbit(r1[1], r1[0], tsh[j]);
                    smp[j] = sub ? mfn_sample_sub(smp[j], u0, u1, -w0[j], -w1[j]) : mfn_sample(u0, u1, w0[j], w1[j]);
                }
            };
            // groups: (white, black) in buffer 0, then per frequency its steps 0-3 in buffer 1 (register set A) and 4-7 in buffer 0 (set
            // B); a group's loads are issued two decode phases before they are committed to LDS, one barrier per group
            fetch(preB, 0, 2);
            commit(preB, 0);
            fetch(preA, 2, 4);
            __syncthreads();
            fetch(preB, 6, 4);
            {
                float smp[kTilePx];
                samples(&box[0][0][0], smp);
#pragma unroll
                for (int j = 0; j < kTilePx; j++) wh[j] = smp[j];
                samples(&box[0][1][0], smp);
#pragma unroll
                for (int j = 0; j < kTilePx; j++) { bk[j] = smp[j]; ok[j] = wh[j] - bk[j] > thr1024; L0[j] = L1[j] = L2[j] = fin[j] = 0.0f; }
            }
            commit(preA, 1);
            __syncthreads();
#pragma unroll 1
            for (int f = 0; f < 4; f++) {
                float S[kTilePx], C[kTilePx], a[4][kTilePx];
                if (f < 3) fetch(preA, 2 + 8 * (f + 1), 4);
#pragma unroll
                for (int k = 0; k < 4; k++) samples(&box[1][k][0], a[k]);
                commit(preB, 0);
                __syncthreads();
                if (f < 3) fetch(preB, 2 + 8 * (f + 1) + 4, 4);
#pragma unroll
                for (int k = 4; k < 8; k++) samples(&box[0][k - 4][0], a[k - 4], true);
#pragma unroll
                for (int j = 0; j < kTilePx; j++) {
                    mfn_bin8(a[0][j], a[1][j], a[2][j], a[3][j], S[j], C[j]);
                    const float p = mfn_phase_of(-S[j], C[j]);
                    ok[j] = ok[j] && (S[j] * S[j] + C[j] * C[j] > mod2);
                    // the cascade of neighbouring differences, streamed: level 1 from (P_{f-1}, P_f), level 2 from the last two level-1
                    // values, level 3 from the two level-2 values -- the same operations in the same order as the array form
                    auto wrapd = [](float a, float b) { return (a > b) ? (a - b) : (a - b + kTrue2PI); };
                    const float d1 = wrapd(L0[j], p);                    // (f >= 1)
                    const float d2 = wrapd(L1[j], d1);                   // (f >= 2)
                    const float d3 = wrapd(L2[j], d2);                   // (f == 3)
                    fin[j] = f == 3 ? d3 : fin[j];
                    L2[j] = f >= 2 ? d2 : L2[j];
                    L1[j] = f >= 1 ? d1 : L1[j];
                    L0[j] = p;
                }
                if (f < 3) commit(preA, 1);
                __syncthreads();
            }
#pragma unroll
            for (int j = 0; j < kTilePx; j++) {
                out[j] = (wh[j] - bk[j] > thr1024) ? fin[j] / kTrue2PI * 255 : 0.0f;
                okm |= (ok[j] ? 1u : 0u) << j;
            }
        }
#pragma unroll
        for (int j = 0; j < kTilePx; j++) {
            if (!live[j]) continue;
            const size_t oo = (size_t)(ty * kTileH + wv + kTileNW * j) * W + col;
            phase[oo] = out[j];
            valid[oo] = (uint8_t)((okm >> j) & 1u);
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// The LDS-DMA form of the rectifying 4 x 8 decode (round 4; BASELINE config 5's shipped kernel).  The register-staged tile kernel
// above waits out one fetch latency per plane group (2.0 ms per 8192 x 6000 camera, hardly faster than the gather form): its
// groups are short (0.15 us of arithmetic against ~2 us of latency) and only two are in flight.  Here
//   * the source box goes HBM -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds, 16 bytes per lane, no VGPR staging, no commit pass)
//     into a RING of R slots of one 4-plane group each (15 KB; rows / columns beyond the tile's real box and everything outside the
//     image or the source window are out-of-range lanes: zeros, no traffic -- BORDER_CONSTANT);
//   * the stream of groups runs ACROSS tiles: a persistent workgroup keeps R - 1 groups in flight while it decodes one, also over
//     a tile boundary -- the next tile's map entries (6 bytes per pixel) arrive by DMA as well during the current tile, its box is
//     reduced from them (wave reductions + one LDS round, riding on the group barriers) four groups before its first DMA is due;
//   * every wave issues the same static sequence of DMAs (2 per group, 1 per tile for the map), so the wait at the top of a group
//     is COUNTED (s_waitcnt vmcnt(2 (R - 2)): everything younger than the group stays in flight) and one s_barrier per group orders
//     both the arrival of a group and the release of the slot it overwrites.  LDS-DMA and ordinary vector memory operations do
//     not retire in order relative to each other (kernels_rectdma.hip): the only other vector memory operations are the output
//     stores, which are never waited for (an outstanding one merely makes a wait longer);
//   * R = 3: 52 KB of LDS, three workgroups per CU; 256 threads x 4 pixels (a wave's rows wv, wv + 4, ..) beat 512 x 2 by 4 %: half the
//     barriers and per-tile set-up per pixel.  Ablations (profiles/exp/r04): no source traffic 1.22 ms, no tap reads / blend 1.12-1.2,
//     neither 0.90, also without the group barriers 0.55 -- the DFT + four atan2f + per-tile set-up alone are 40 % of the kernel.
// Arithmetic, order and therefore results are the gather form's, bit for bit (tests/test_mfn_extension.py); a tile whose box
// exceeds 80 x 24 elements is marked (valid bytes 0xFE) and decoded by mfn_rect_fix_kernel's gather code in a second launch.
// ------------------------------------------------------------------------------------------------------
constexpr int kDmaGroupChunks = kTileG * kBoxChunks;                       // 960 16-byte chunks per group
constexpr int kDmaGroupBytes = kDmaGroupChunks * 16;
constexpr int kDmaMapXyBytes = kTileW * kTileH * 4, kDmaMapFrBytes = kTileW * kTileH * 2;
constexpr unsigned kMfnDmaInvalid = 0xFFFFFFF0u;                           // beyond every descriptor's range: loads 0
static_assert(kDmaGroupChunks % 64 == 0 && kDmaGroupChunks <= 1024 - 64, "whole waves of chunks; the DMAs beyond them go to the dump");
static_assert(kDmaMapXyBytes == 4 * 1024 && kDmaMapFrBytes == 2 * 1024, "map DMAs: waves 0-3 the xy entries, 4-5 the fractions, 6-7 dummies");

__device__ __forceinline__ void mfn_dma16(unsigned voff, __amdgpu_buffer_rsrc_t rsrc, unsigned lds_dst, unsigned soff)
{
    unsigned keep;       // (M0 carries the LDS base and is compiler-reserved: saved and restored inside the one statement that uses it)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(__builtin_amdgcn_readfirstlane(lds_dst)), "s"(__builtin_amdgcn_readfirstlane(soff)) : "memory");
}
template <int N> __device__ __forceinline__ void mfn_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

constexpr unsigned kMfnDeferred = 0x80000000u;                             // okm of a tile the DMA kernel leaves to mfn_rect_fix_kernel
constexpr unsigned kMfnDeferredByte = 0xFEu;                               // ... and the valid byte its live pixels carry until then (never a result)

struct MfnDmaArgs {
    const uint16_t *base; unsigned stride_bytes, total_bytes;               // the contiguous stack (MfnStridedArg)
    const int16_t *map_xy; const uint16_t *map_frac; unsigned map_px;       // the maps and their pixel count (descriptor ranges)
};

template <int R, int NT>
__global__ __launch_bounds__(NT, (NT / 64) * (R == 3 ? 3 : 2) / 4) void mfn_rect_dma_kernel(MfnDmaArgs arg, MfnTrigN<8> tr, int pitch, int W, int H, float black_thr,
                                                                                 int row0, int rows, int src_row0, int src_rows, int tiles_x,
                                                                                 float *__restrict__ phase, uint8_t *__restrict__ valid)
{
    constexpr int kRing = R * kDmaGroupBytes, kOffXy = kRing, kOffFr = kOffXy + kDmaMapXyBytes, kOffDump = kOffFr + kDmaMapFrBytes;
    constexpr int kNW = NT / 64, kPX = kTileW * kTileH / NT;          // waves; pixels per thread (tile rows wv + kNW j)
    constexpr int kDPG = (kDmaGroupChunks + NT - 1) / NT;              // DMAs per wave and group (chunk ids >= 960 go to the dump)
    constexpr int kMPW = 8 / kNW;                                      // map DMAs per wave and tile (8 in all: 4 xy, 2 fractions, 2 dummies)
    constexpr int kWait = kDPG * (R - 2);
    static_assert(NT == 512 || NT == 256, "8 or 4 waves");
    __shared__ __attribute__((aligned(16))) unsigned char smem[kOffDump + 1024];
    __shared__ int red[kNW][4];
    const MfnStridedPlanes src{__builtin_amdgcn_make_buffer_rsrc((void *)arg.base, 0, (int)arg.total_bytes, 0x00020000), arg.stride_bytes};
    const __amdgpu_buffer_rsrc_t rs_xy = __builtin_amdgcn_make_buffer_rsrc((void *)arg.map_xy, 0, (int)(arg.map_px * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_fr = __builtin_amdgcn_make_buffer_rsrc((void *)arg.map_frac, 0, (int)(arg.map_px * 2u), 0x00020000);
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)smem);
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int tiles_y = (rows + kTileH - 1) / kTileH, ntiles = tiles_x * tiles_y;
    const unsigned nb = gridDim.x, per = (nb + 7) / 8;
    unsigned vb = (blockIdx.x % 8) * per + blockIdx.x / 8;   // XCD-banded tile order (workgroup b runs on XCD b % 8)
    if (nb % 8 != 0) vb = blockIdx.x;

    // this thread's two chunks of a group: id = threadIdx (+ 512) -> (plane of the group, box row, 16-byte column); ids >= 960: the dump
    int cpg[kDPG], crow[kDPG], ccol[kDPG];
#pragma unroll
    for (int i = 0; i < kDPG; i++) {
        const int id = (int)threadIdx.x + NT * i, pg = id / kBoxChunks, w = id - pg * kBoxChunks;
        cpg[i] = pg; crow[i] = w / (kBoxW / 8); ccol[i] = w - crow[i] * (kBoxW / 8);
    }

    struct Box { int x0, y0, bw, bh; bool dma; bool fits; };   // bw / bh: the box in elements (0: nothing to fetch); dma: fetch it; fits: decode from LDS
    // vector offsets of the thread's two chunks of a group of `box` (plane 0 of the group; the plane group is the scalar offset)
    auto chunk_offsets = [&](const Box &b, unsigned vo[kDPG]) {
#pragma unroll
        for (int i = 0; i < kDPG; i++) {
            const int gy = b.y0 + crow[i], gx = b.x0 + 8 * ccol[i];
            const bool in = b.dma && cpg[i] < kTileG && crow[i] < b.bh && 8 * ccol[i] < b.bw && (unsigned)gy < (unsigned)H &&
                            (unsigned)(gy - src_row0) < (unsigned)src_rows && (unsigned)gx < (unsigned)W;
#if defined(SLR_MFN_ABL) && (SLR_MFN_ABL & 1)
            vo[i] = in && gy == -12345 ? 0u : kMfnDmaInvalid;        // (ablation: no source traffic, the DMAs write zeros)
#else
            vo[i] = in ? (unsigned)(((gy - src_row0) * pitch + gx) * 2) + (unsigned)cpg[i] * src.stride_bytes : kMfnDmaInvalid;
#endif
        }
    };
    // the two DMAs of one group: planes first .. first + count - 1 into ring slot `slot`
    auto issue_group = [&](const unsigned vo[kDPG], int first, int count, int slot) {
        const unsigned sb = lds0 + (unsigned)(slot * kDmaGroupBytes);
#pragma unroll
        for (int i = 0; i < kDPG; i++) {
            const int cb = i * NT + wv * 64;              // the wave's first chunk of this DMA (wave-uniform)
            mfn_dma16(count == 4 || cpg[i] < count ? vo[i] : kMfnDmaInvalid, src.rsrc, cb >= kDmaGroupChunks ? lds0 + (unsigned)kOffDump : sb + (unsigned)(cb * 16),
                      (unsigned)first * src.stride_bytes);
        }
    };
    // the map entries of tile t (none beyond the last tile): one DMA per wave
    auto issue_map = [&](int t) {
        const bool on = t < ntiles;
        const int ty = on ? t / tiles_x : 0, tx = on ? t - ty * tiles_x : 0;
#pragma unroll
        for (int i = 0; i < kMPW; i++) {
            const int m = wv * kMPW + i;                  // 0-3: the xy entries (16 rows x 16 chunks of 4 pixels), 4-5: the fractions (16 rows x 8 chunks of 8 pixels), 6-7: dummies
            if (m < 4) {
                const int id = m * 64 + lane, r = id >> 4, c = id & 15, brow = ty * kTileH + r, col = tx * kTileW + 4 * c;
                const bool in = on && brow < rows && col < W;
                mfn_dma16(in ? (unsigned)(((size_t)(brow + row0) * W + col) * 4) : kMfnDmaInvalid, rs_xy, lds0 + (unsigned)(kOffXy + m * 1024), 0u);
            } else if (m < 6) {
                const int id = (m - 4) * 64 + lane, r = id >> 3, c = id & 7, brow = ty * kTileH + r, col = tx * kTileW + 8 * c;
                const bool in = on && brow < rows && col < W;
                mfn_dma16(in ? (unsigned)(((size_t)(brow + row0) * W + col) * 2) : kMfnDmaInvalid, rs_fr, lds0 + (unsigned)(kOffFr + (m - 4) * 1024), 0u);
            } else
                mfn_dma16(kMfnDmaInvalid, rs_fr, lds0 + (unsigned)kOffDump, 0u);
        }
    };
    // this thread's pixel j of the staged tile: column lane, tile row wv + 8 j
    auto staged_px = [&](int j, int &sx, int &sy, unsigned &fr) {
        const int e = (wv + kNW * j) * kTileW + lane;
        const unsigned xy = *reinterpret_cast<const unsigned *>(smem + kOffXy + 4 * e);
        sx = (int)(short)(xy & 0xFFFFu); sy = (int)xy >> 16;
        fr = (unsigned)*reinterpret_cast<const unsigned short *>(smem + kOffFr + 2 * e) & 1023u;
    };
    // the wave's share of the staged tile's bounding box -> red[wv]
    auto reduce_staged = [&](int t) {
#if defined(SLR_MFN_ABL) && (SLR_MFN_ABL & 16)
        if (t != -12345) return;                          // (ablation, round 5: no per-tile box reduction -- a precomputed box table at best)
#endif
        const bool on = t < ntiles;
        const int ty = on ? t / tiles_x : 0, tx = on ? t - ty * tiles_x : 0, col = tx * kTileW + lane;
        int mnx = 0x7FFFFFFF, mxx = -0x7FFFFFFF, mny = 0x7FFFFFFF, mxy = -0x7FFFFFFF;
#pragma unroll
        for (int j = 0; j < kPX; j++) {
            int sx, sy; unsigned fr;
            staged_px(j, sx, sy, fr);
            const bool live = on && ty * kTileH + wv + kNW * j < rows && col < W;
            if (live && !(sx >= W || sx + 1 < 0 || sy >= H || sy + 1 < 0)) {       // footprints completely outside read 0
                mnx = sx < mnx ? sx : mnx; mxx = sx > mxx ? sx : mxx; mny = sy < mny ? sy : mny; mxy = sy > mxy ? sy : mxy;
            }
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            int v;
            v = __shfl_xor(mnx, d); mnx = v < mnx ? v : mnx;
            v = __shfl_xor(mxx, d); mxx = v > mxx ? v : mxx;
            v = __shfl_xor(mny, d); mny = v < mny ? v : mny;
            v = __shfl_xor(mxy, d); mxy = v > mxy ? v : mxy;
        }
        if (lane == 0) { red[wv][0] = mnx; red[wv][1] = mxx; red[wv][2] = mny; red[wv][3] = mxy; }
    };
    int abl_tile = 0;                                       // (SLR_MFN_ABL & 16 only)
    (void)abl_tile;
    auto box_from_red = [&]() -> Box {
#if defined(SLR_MFN_ABL) && (SLR_MFN_ABL & 16)
        {   // (ablation: the tile's own rectangle plus a margin, as if read from a table: wrong results, timing only)
            const int ty = abl_tile < ntiles ? abl_tile / tiles_x : 0, tx = abl_tile < ntiles ? abl_tile - ty * tiles_x : 0;
            Box b;
            b.x0 = tx * kTileW > 8 ? tx * kTileW - 8 : 0; b.y0 = ty * kTileH + row0 > 2 ? ty * kTileH + row0 - 2 : 0;
            b.bw = kBoxW - 2; b.bh = kBoxH - 2; b.fits = true; b.dma = abl_tile < ntiles;
            return b;
        }
#endif
        int mnx = 0x7FFFFFFF, mxx = -0x7FFFFFFF, mny = 0x7FFFFFFF, mxy = -0x7FFFFFFF;
#pragma unroll
        for (int w = 0; w < kNW; w++) {
            mnx = red[w][0] < mnx ? red[w][0] : mnx; mxx = red[w][1] > mxx ? red[w][1] : mxx;
            mny = red[w][2] < mny ? red[w][2] : mny; mxy = red[w][3] > mxy ? red[w][3] : mxy;
        }
        mnx = __builtin_amdgcn_readfirstlane(mnx); mxx = __builtin_amdgcn_readfirstlane(mxx);      // (wave-uniform: scalar registers)
        mny = __builtin_amdgcn_readfirstlane(mny); mxy = __builtin_amdgcn_readfirstlane(mxy);
        Box b;
        const bool empty = mnx > mxx;                     // the tile samples nothing
        b.x0 = empty ? 0 : (mnx & ~7); b.y0 = empty ? 0 : mny;
        b.bw = empty ? 0 : (mxx + 2) - b.x0; b.bh = empty ? 0 : (mxy + 2) - b.y0;
        b.fits = empty || ((mxx + 1) - b.x0 < kBoxW && (mxy + 1) - b.y0 < kBoxH);
        b.dma = !empty && b.fits;
        return b;
    };

    // prologue: the first tile's map, its box, its first R - 1 groups
    int t = (int)vb;
    issue_map(t);
    mfn_wait_vm<0>();
    __syncthreads();
    reduce_staged(t);
    __syncthreads();
    abl_tile = t;
    Box cur = box_from_red(), nxt = cur;
    unsigned cvo[kDPG], nvo[kDPG];
    chunk_offsets(cur, cvo);
#pragma unroll
    for (int i = 0; i < kDPG; i++) nvo[i] = kMfnDmaInvalid;
    issue_group(cvo, 0, 2, 0);
#pragma unroll
    for (int g = 1; g < R - 1; g++) issue_group(cvo, 2 + 4 * (g - 1), 4, g);
    int s0 = 0;                                           // ring slot of this tile's group 0
    // a tile's results are stored at the top of the NEXT tile's first step, behind that step's DMAs: vmcnt counts stores too, and
    // the counted wait that follows them is a whole group away
    float pend_out[kPX] = {};
    unsigned pend_okm = 0;
    int pend_t = -1;
    auto flush = [&]() {
        if (pend_t < 0) return;
        const int py = pend_t / tiles_x, px = pend_t - py * tiles_x, pcol = px * kTileW + lane;
#pragma unroll
        for (int j = 0; j < kPX; j++) {
            const int brow = py * kTileH + wv + kNW * j;
            if (!(brow < rows && pcol < W)) continue;
            const size_t oo = (size_t)brow * W + pcol;
#if defined(SLR_MFN_ABL) && (SLR_MFN_ABL & 4)
            if (pend_out[j] != 1.2345e-30f) continue;              // (ablation: no output stores)
#endif
            phase[oo] = pend_out[j];
            valid[oo] = pend_okm == kMfnDeferred ? (uint8_t)kMfnDeferredByte : (uint8_t)((pend_okm >> j) & 1u);
        }
        pend_t = -1;
    };

    const float mod2 = 4.0f * 1048576.0f, thr1024 = black_thr * 1024.0f;      // (0.25 * 8)^2, on S and C of the x 1024 samples
#pragma unroll 1
    for (; t < ntiles; t += (int)nb) {
        const int tn = t + (int)nb;
        const int ty = t / tiles_x, tx = t - ty * tiles_x, col = tx * kTileW + lane;
        // this thread's pixels from the staged map entries
        int sx[kPX], sy[kPX];
        unsigned fr[kPX];
        bool live[kPX];
        unsigned ta[kPX], tsh[kPX];
        h16x2 w0[kPX], w1[kPX];
#pragma unroll
        for (int j = 0; j < kPX; j++) {
#if defined(SLR_MFN_ABL) && (SLR_MFN_ABL & 32)
            // (ablation, round 5: no per-pixel set-up -- one LDS dword per pixel as a digest would be, constant weights: timing only)
            const unsigned dg = *reinterpret_cast<const unsigned *>(smem + kOffXy + 4 * ((wv + kNW * j) * kTileW + lane));
            sx[j] = 0; sy[j] = 0; fr[j] = 0;
            live[j] = ty * kTileH + wv + kNW * j < rows && col < W;
            w0[j] = h16x2{(_Float16)512.0f, (_Float16)256.0f}; w1[j] = h16x2{(_Float16)128.0f, (_Float16)128.0f};
            const unsigned b = ((unsigned)((wv + kNW * j + 1) * kBoxRowBytes + (lane + 4) * 2) + (dg & 2u)) & 0xFFFFu;
            ta[j] = b & ~3u; tsh[j] = (b & 2u) * 8u;
#else
            staged_px(j, sx[j], sy[j], fr[j]);
            live[j] = ty * kTileH + wv + kNW * j < rows && col < W;
            const unsigned fx = fr[j] & 31u, fy = fr[j] >> 5;
            w0[j] = h16x2{(_Float16)(float)((32u - fx) * (32u - fy)), (_Float16)(float)(fx * (32u - fy))};
            w1[j] = h16x2{(_Float16)(float)((32u - fx) * fy), (_Float16)(float)(fx * fy)};
            const bool touch = live[j] && !(sx[j] >= W || sx[j] + 1 < 0 || sy[j] >= H || sy[j] + 1 < 0);
            const unsigned b = touch && cur.fits ? (unsigned)((sy[j] - cur.y0) * kBoxRowBytes + (sx[j] - cur.x0) * 2) : 0u;
            ta[j] = b & ~3u; tsh[j] = (b & 2u) * 8u;
            if (!touch) { w0[j] = h16x2{(_Float16)0.0f, (_Float16)0.0f}; w1[j] = w0[j]; }       // every sample 0
#endif
        }
        float out[kPX];
        unsigned okm = 0;                                 // bit j: pixel j valid
        if (!cur.fits) {                                  // (block-uniform) a box beyond the LDS image: the tile is left to the fix-up pass
#pragma unroll
            for (int j = 0; j < kPX; j++) out[j] = 0.0f;
            okm = kMfnDeferred;                               // (every live pixel's valid byte becomes the marker)
        }
        float wh[kPX], bk[kPX], L0[kPX], L1[kPX], L2[kPX], fin[kPX], S[kPX], C[kPX], A[4][kPX];
        bool ok[kPX];
        // the rectified samples (x 1024) of plane image `img` for the thread's pixels
        h16x2 nw0[kPX], nw1[kPX];
#pragma unroll
        for (int j = 0; j < kPX; j++) { nw0[j] = -w0[j]; nw1[j] = -w1[j]; }
        auto samples = [&](const unsigned char *img, float smp[kPX], bool sub) {       // sub: smp -= the samples (mfn_sample_sub)
#if !defined(SLR_MFN_NOFENCE)
            asm volatile("" ::: "memory");                // (one plane's tap reads at a time)
#endif
#if defined(SLR_MFN_ABL) && (SLR_MFN_ABL & 2)
#pragma unroll
            for (int j = 0; j < kPX; j++) smp[j] = __uint_as_float(ta[j] + (unsigned)(uintptr_t)img);    // (ablation: no tap reads, no blend)
            return;
#endif
#pragma unroll
            for (int j = 0; j < kPX; j++) {
                const unsigned *r0 = reinterpret_cast<const unsigned *>(img + ta[j]);
                const unsigned *r1 = reinterpret_cast<const unsigned *>(img + ta[j] + kBoxRowBytes);
                const unsigned u0 = __builtin_amdgcn_alignbit(r0[1], r0[0], tsh[j]);
                const unsigned u1 = __builtin_amdgcn_alignbit(r1[1], r1[0], tsh[j]);
                smp[j] = sub ? mfn_sample_sub(smp[j], u0, u1, nw0[j], nw1[j]) : mfn_sample(u0, u1, w0[j], w1[j]);
            }
        };
        // one step of the group stream: group g of this tile has landed and is decoded; group g + R - 1 (of this tile or the next) is issued
        auto step_top = [&](int g) {
            mfn_wait_vm<kWait>();
#if !(defined(SLR_MFN_ABL) && (SLR_MFN_ABL & 8))
            __syncthreads();                              // (ablation 8: no group barriers -- wrong results, timing only)
#endif
            if (g == 1) issue_map(tn);                    // (before the step's group DMAs: the counted wait stays exact)
            if (g == 5) { abl_tile = tn; nxt = box_from_red(); chunk_offsets(nxt, nvo); }
            const int gi = g + R - 1;                     // the group to issue
            int slot = s0 + gi; slot -= slot >= R ? R : 0; slot -= slot >= R ? R : 0; slot -= slot >= R ? R : 0; slot -= slot >= R ? R : 0;
            if (gi < 9) issue_group(cvo, gi == 0 ? 0 : 2 + 4 * (gi - 1), gi == 0 ? 2 : 4, slot);
            else issue_group(nvo, gi == 9 ? 0 : 2 + 4 * (gi - 10), gi == 9 ? 2 : 4, slot);
            if (g == 0) flush();
            if (g == 4) reduce_staged(tn);
        };
        auto slot_of = [&](int g) -> const unsigned char * {
            int slot = s0 + g; slot -= slot >= R ? R : 0; slot -= slot >= R ? R : 0; slot -= slot >= R ? R : 0; slot -= slot >= R ? R : 0;
            return smem + slot * kDmaGroupBytes;
        };
        step_top(0);
        if (cur.fits) {
            const unsigned char *img = slot_of(0);
            float smp[kPX];
            samples(img, smp, false);
#pragma unroll
            for (int j = 0; j < kPX; j++) wh[j] = smp[j];
            samples(img + kBoxPlaneBytes, smp, false);
#pragma unroll
            for (int j = 0; j < kPX; j++) { bk[j] = smp[j]; ok[j] = wh[j] - bk[j] > thr1024; L0[j] = L1[j] = L2[j] = fin[j] = 0.0f; }
        }
#pragma unroll 1
        for (int f = 0; f < 4; f++) {
#pragma unroll
            for (int hf = 0; hf < 2; hf++) {
                const int g = 1 + 2 * f + hf;
                step_top(g);
                if (cur.fits) {
                    const unsigned char *img = slot_of(g);
#pragma unroll
                    for (int k = 0; k < 4; k++) samples(img + k * kBoxPlaneBytes, A[k], hf == 1);
                }
            }
            if (cur.fits) {
#pragma unroll
                for (int j = 0; j < kPX; j++) {
                    mfn_bin8(A[0][j], A[1][j], A[2][j], A[3][j], S[j], C[j]);
                    const float p = mfn_phase_of(-S[j], C[j]);
                    ok[j] = ok[j] && (S[j] * S[j] + C[j] * C[j] > mod2);
                    // the cascade of neighbouring differences, streamed (mfn_rect_tile_kernel)
                    auto wrapd = [](float a, float b) { return (a > b) ? (a - b) : (a - b + kTrue2PI); };
                    const float d1 = wrapd(L0[j], p);                    // (f >= 1)
                    const float d2 = wrapd(L1[j], d1);                   // (f >= 2)
                    const float d3 = wrapd(L2[j], d2);                   // (f == 3)
                    fin[j] = f == 3 ? d3 : fin[j];
                    L2[j] = f >= 2 ? d2 : L2[j];
                    L1[j] = f >= 1 ? d1 : L1[j];
                    L0[j] = p;
                }
            }
        }
        if (cur.fits) {
#pragma unroll
            for (int j = 0; j < kPX; j++) {
                out[j] = (wh[j] - bk[j] > thr1024) ? fin[j] / kTrue2PI * 255 : 0.0f;
                okm |= (ok[j] ? 1u : 0u) << j;
            }
        }
#pragma unroll
        for (int j = 0; j < kPX; j++) pend_out[j] = out[j];
        pend_okm = okm; pend_t = t;
        cur = nxt;
#pragma unroll
        for (int i = 0; i < kDPG; i++) cvo[i] = nvo[i];
        s0 += 9 % R; s0 -= s0 >= R ? R : 0;
    }
    flush();
    mfn_wait_vm<0>();                                     // the dummy DMAs behind the last tile must land before the LDS is released
}

// second pass of the LDS-DMA form: the tiles it marked (boxes beyond the LDS image; none on BASELINE's rigs), by the gather code.
// A workgroup looks at 64 tiles' marker bytes (the first live pixel of a tile is its upper left one) and decodes the marked ones.
__global__ __launch_bounds__(256) void mfn_rect_fix_kernel(MfnStridedArg arg, MfnTrigN<8> tr, int pitch, int W, int H, float black_thr,
                                                           const int16_t *__restrict__ map_xy, const uint16_t *__restrict__ map_frac,
                                                           int row0, int rows, int src_row0, int src_rows, int tiles_x, int ntiles,
                                                           float *__restrict__ phase, uint8_t *__restrict__ valid)
{
    const MfnStridedPlanes src{__builtin_amdgcn_make_buffer_rsrc((void *)arg.base, 0, (int)arg.total_bytes, 0x00020000), arg.stride_bytes};
    __shared__ unsigned long long marked;
    const int t0 = blockIdx.x * 64;
    if (threadIdx.x < 64) {
        const int t = t0 + (int)threadIdx.x;
        bool m = false;
        if (t < ntiles) {
            const int ty = t / tiles_x, tx = t - ty * tiles_x;
            m = valid[(size_t)(ty * kTileH) * W + tx * kTileW] == (uint8_t)kMfnDeferredByte;
        }
        const unsigned long long b = __ballot(m);
        if (threadIdx.x == 0) marked = b;
    }
    __syncthreads();
    unsigned long long todo = marked;
    while (todo) {
        const int i = __builtin_ctzll(todo);
        todo &= todo - 1;
        const int t = t0 + i, ty = t / tiles_x, tx = t - ty * tiles_x;
#pragma unroll 1
        for (int e = (int)threadIdx.x; e < kTileW * kTileH; e += 256) {
            const int brow = ty * kTileH + e / kTileW, col = tx * kTileW + e % kTileW;
            if (!(brow < rows && col < W)) continue;
            const size_t m = (size_t)(brow + row0) * W + col;
            const int sx = map_xy[2 * m], sy = map_xy[2 * m + 1];
            const unsigned fr = map_frac[m] & 1023u, fx = fr & 31u, fy = fr >> 5;
            const h16x2 gw0[1] = {h16x2{(_Float16)(float)((32u - fx) * (32u - fy)), (_Float16)(float)(fx * (32u - fy))}};
            const h16x2 gw1[1] = {h16x2{(_Float16)(float)((32u - fx) * fy), (_Float16)(float)(fx * fy)}};
            const bool bx0 = (unsigned)sx < (unsigned)W, bx1 = (unsigned)(sx + 1) < (unsigned)W;
            const bool by0 = (unsigned)sy < (unsigned)H && (unsigned)(sy - src_row0) < (unsigned)src_rows;
            const bool by1 = (unsigned)(sy + 1) < (unsigned)H && (unsigned)(sy + 1 - src_row0) < (unsigned)src_rows;
            const unsigned inb[1] = {(bx0 && by0 ? 1u : 0u) | (bx1 && by0 ? 2u : 0u) | (bx0 && by1 ? 4u : 0u) | (bx1 && by1 ? 8u : 0u)};
            const unsigned off[1] = {(unsigned)((sy - src_row0) * pitch + sx)};
            float o1[1] = {0.0f};
            unsigned vw1 = 0;
            if (src_rows > 0) mfn_rect_quad<1, 4, 8, false>(src, tr, 4, 8, (unsigned)pitch, black_thr, off, inb, gw0, gw1, o1, vw1);
            const size_t oo = (size_t)brow * W + col;
            phase[oo] = o1[0];
            valid[oo] = (uint8_t)(vw1 & 1u);
        }
    }
}

// source rows [lo, hi] that destination rows [row0, row0 + rows) read through a map (taps sy and sy + 1, clipped to the image);
// out[0] = min sy, out[1] = max sy + 1 over the pixels whose footprint touches the image (initialised by the launcher)
__global__ __launch_bounds__(256) void map_source_rows_kernel(const int16_t *__restrict__ map_xy, int W, int H, int row0, int rows,
                                                              int *__restrict__ out)
{
    int lo = 0x7FFFFFFF, hi = -0x7FFFFFFF;
    const size_t n = (size_t)rows * W, base = (size_t)row0 * W;
    for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < n; i += (size_t)gridDim.x * 256u) {
        const int sx = map_xy[2 * (base + i)], sy = map_xy[2 * (base + i) + 1];
        if (sx >= W || sx + 1 < 0 || sy >= H || sy + 1 < 0) continue;
        lo = sy < lo ? sy : lo; hi = sy + 1 > hi ? sy + 1 : hi;
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        const int a = __shfl_xor(lo, d), b = __shfl_xor(hi, d);
        lo = a < lo ? a : lo; hi = b > hi ? b : hi;
    }
    if ((threadIdx.x & 63) == 0) {
        if (lo != 0x7FFFFFFF) atomicMin(out, lo);
        if (hi != -0x7FFFFFFF) atomicMax(out + 1, hi);
    }
}

hipError_t launch_map_source_rows(const int16_t *map_xy, int W, int H, int row0, int rows, int *d_out /* 2 ints */, hipStream_t s)
{
    const int init[2] = {0x7FFFFFFF, -0x7FFFFFFF};
    hipError_t e = hipMemcpyAsync(d_out, init, sizeof init, hipMemcpyHostToDevice, s);
    if (e != hipSuccess) return e;
    const size_t n = (size_t)rows * W;
    const unsigned blocks = (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    SLR_LAUNCH(map_source_rows_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, s, map_xy, W, H, row0, rows, d_out);
    return hipGetLastError();
}

hipError_t launch_mfn_rect_decode(const uint16_t *const *planes, int n_freq, int n_step, int pitch, int W, int H, float black_thr,
                                  const int16_t *map_xy, const uint16_t *map_frac, int row0, int rows, int src_row0, int src_rows,
                                  float *phase, uint8_t *valid, hipStream_t s)
{
    MfnPtrPlanes pp;
    const int np = 2 + n_freq * n_step;
    for (int i = 0; i < SLR_MFN_MAX_PLANES; i++) pp.pl.p[i] = i < np ? planes[i] : nullptr;
    // equally spaced planes (a contiguous stack) whose windows end below 4 GiB from the first: one buffer descriptor
    bool strided = planes[1] > planes[0];
    const size_t stride = strided ? (size_t)(planes[1] - planes[0]) : 0;
    for (int i = 1; i < np && strided; i++) strided = planes[i] == planes[0] + stride * (size_t)i;
    const size_t span = (stride * (size_t)(np - 1) + (size_t)pitch * (size_t)(src_rows > 0 ? src_rows : 1)) * 2;
    strided = strided && span < (1ull << 32) && stride * 2 < (1ull << 32);
    const MfnStridedArg sp{planes[0], (unsigned)(stride * 2), (unsigned)span};
    const bool a4 = W % 4 == 0 && (uintptr_t)phase % 16 == 0 && (uintptr_t)valid % 4 == 0;
    constexpr int VQ = kMfnRectV;                            // pixels per thread of the specialised 4 x 8 instances
    MfnTrigN<SLR_MFN_MAX_STEPS> tr;
    MfnTrigN<8> tr8;
    for (int k = 0; k < SLR_MFN_MAX_STEPS; k++) {            // (the kernel's samples are x 1024: the trigonometric factors / 1024)
        const double a = 2.0 * 3.14159265358979323846 * k / (double)n_step;
        tr.cs[k] = k < n_step ? (float)cos(a) * (1.0f / 1024.0f) : 0.0f;
        tr.sn[k] = k < n_step ? (float)sin(a) * (1.0f / 1024.0f) : 0.0f;
        if (k < 8) { tr8.cs[k] = tr.cs[k]; tr8.sn[k] = tr.sn[k]; }
    }
    const bool spec = n_freq == 4 && n_step == 8 && (VQ == 1 || a4);
    const size_t groups = spec ? (size_t)(W / VQ) * rows : a4 ? (size_t)(W / 4) * rows : (size_t)W * rows;
    unsigned blocks = (unsigned)((groups + 255) / 256 < 65536 ? (groups + 255) / 256 : 65536);
    blocks = (blocks + 7u) & ~7u;                            // (a multiple of 8: the XCD-banded order)
#define SLR_MFNR(V, FS, NS, SRC, TR)                                                                                       \
    SLR_LAUNCH((mfn_rect_decode_kernel<V, FS, NS, decltype(SRC)>), dim3(blocks ? blocks : 8), dim3(256), 0, s, SRC, TR, n_freq, n_step, \
               pitch, W, H, black_thr, map_xy, map_frac, row0, rows, src_row0, src_rows, phase, valid)
    if (spec && strided && W % 8 == 0 && ((uintptr_t)planes[0] % 16) == 0 && ((size_t)pitch * 2) % 16 == 0 && (stride * 2) % 16 == 0 &&
        !tl_debug.no_tiled_map && !tl_debug.no_buffer_form && (uintptr_t)map_xy % 16 == 0 && (uintptr_t)map_frac % 16 == 0 &&
        (size_t)W * H * 4 < (1ull << 31)) {                  // the LDS-DMA form (SLR_OPT_DEBUG_FLAGS bit 1: the register-staged tile form)
        const int tiles_x = (W + kTileW - 1) / kTileW, tiles_y = (rows + kTileH - 1) / kTileH;
        // shipped: 256 threads x 4 pixels, ring of 3 (1.32 ms per 8192 x 6000 camera; 512 x 2: 1.39; ring of 4: 1.45 / 1.51).
        // SLR_OPT_DEBUG_FLAGS bit 8: a ring of 4 groups, two workgroups per CU; bit 7: 512 threads x 2 pixels
        const bool ring4 = tl_debug.mfn_ring4, nt256 = !tl_debug.mfn_nt512;
        const int per_cu = ring4 ? 2 : 3;
        unsigned tb = (unsigned)(tiles_x * tiles_y < 256 * per_cu ? tiles_x * tiles_y : 256 * per_cu);
        tb = (tb + 7u) & ~7u;
        const MfnDmaArgs da{planes[0], (unsigned)(stride * 2), (unsigned)span, map_xy, map_frac, (unsigned)((size_t)W * H)};
#define SLR_MFND(RR, NTT) SLR_LAUNCH((mfn_rect_dma_kernel<RR, NTT>), dim3(tb ? tb : 8), dim3(NTT), 0, s, da, tr8, pitch, W, H, black_thr, row0, rows, \
                                     src_row0, src_rows, tiles_x, phase, valid)
        if (ring4 && nt256) SLR_MFND(4, 256); else if (ring4) SLR_MFND(4, 512); else if (nt256) SLR_MFND(3, 256); else SLR_MFND(3, 512);
#undef SLR_MFND
        const int ntiles = tiles_x * tiles_y;
        SLR_LAUNCH(mfn_rect_fix_kernel, dim3((ntiles + 63) / 64), dim3(256), 0, s, sp, tr8, pitch, W, H, black_thr, map_xy, map_frac, row0, rows,
                   src_row0, src_rows, tiles_x, ntiles, phase, valid);
    }
    else if (spec && strided && W % 8 == 0 && ((uintptr_t)planes[0] % 16) == 0 && ((size_t)pitch * 2) % 16 == 0 && (stride * 2) % 16 == 0 &&
        !tl_debug.no_tiled_map) {                            // the register-staged LDS-tiled form (bit 0 keeps the gather form: tests)
        const int tiles_x = (W + kTileW - 1) / kTileW, tiles_y = (rows + kTileH - 1) / kTileH;
        unsigned tb = (unsigned)(tiles_x * tiles_y < 8 * 256 * 4 ? tiles_x * tiles_y : 8 * 256 * 4);
        tb = (tb + 7u) & ~7u;
        SLR_LAUNCH(mfn_rect_tile_kernel, dim3(tb ? tb : 8), dim3(kTileNT), 0, s, sp, tr8, pitch, W, H, black_thr, map_xy, map_frac, row0, rows,
                   src_row0, src_rows, tiles_x, phase, valid);
    }
    else if (spec && strided) SLR_MFNR(VQ, 4, 8, sp, tr8);   // BASELINE config 5, a contiguous stack: the gather form
    else if (spec) SLR_MFNR(VQ, 4, 8, pp, tr8);
    else if (a4) SLR_MFNR(4, 0, 0, pp, tr);
    else SLR_MFNR(1, 0, 0, pp, tr);
#undef SLR_MFNR
    return hipGetLastError();
}

hipError_t launch_mfn_decode(const uint16_t *const *planes, int n_freq, int n_step, int pitch, int W, int H, float black_thr,
                             float *phase, uint8_t *valid, hipStream_t s)
{
    MfnPlanes pl;
    const int np = 2 + n_freq * n_step;
    bool a4 = W % 4 == 0 && pitch % 4 == 0 && (uintptr_t)phase % 16 == 0 && (uintptr_t)valid % 4 == 0;
    for (int i = 0; i < SLR_MFN_MAX_PLANES; i++) {
        pl.p[i] = i < np ? planes[i] : nullptr;
        if (i < np) a4 = a4 && (uintptr_t)planes[i] % 8 == 0;
    }
    MfnTrig tr;
    for (int k = 0; k < SLR_MFN_MAX_STEPS; k++) {
        const double a = 2.0 * 3.14159265358979323846 * k / (double)n_step;
        tr.cs[k] = k < n_step ? (float)cos(a) : 0.0f;
        tr.sn[k] = k < n_step ? (float)sin(a) : 0.0f;
    }
    const size_t groups = a4 ? (size_t)(W / 4) * H : (size_t)W * H;
    const unsigned blocks = (unsigned)((groups + 255) / 256 < 65536 ? (groups + 255) / 256 : 65536);
#define SLR_MFN(V, FS, NS)                                                                                       \
    SLR_LAUNCH((mfn_decode_kernel<V, FS, NS>), dim3(blocks ? blocks : 1), dim3(256), 0, s, pl, tr, n_freq, n_step, \
                       pitch, W, H, black_thr, phase, valid)
    if (a4 && n_freq == 4 && n_step == 8) SLR_MFN(4, 4, 8);          // BASELINE config 5
    else if (a4 && n_freq == 3 && n_step == 4) SLR_MFN(4, 3, 4);     // the reference's own pattern count
    else if (a4) SLR_MFN(4, 0, 0);
    else SLR_MFN(1, 0, 0);
#undef SLR_MFN
    return hipGetLastError();
}

}  // namespace slr
