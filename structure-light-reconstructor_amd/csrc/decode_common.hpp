// decode_common.hpp -- per-pixel math and rectification-tap helpers shared by the decode kernels (kernels_decode.hip,
// kernels_rectdma.hip).  Device code only, gfx950.
#pragma once
#include "slr_device.hpp"

namespace slr {

// native clang vectors (HIP's uint4/float4 are structs and cannot be used with __builtin_nontemporal_*)
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

// ------------------------------------------------------------------------------------------------------
// per-pixel math
// ------------------------------------------------------------------------------------------------------

// mfreconstruct.cpp:246-261.  n = G4-G2, d = G1-G3.  These kernels turned out to be VALU-bound on MI355X (PMC: the
// fused decode issues ~190 integer/f32 instructions per pixel at 16 lanes/clk, HBM traffic already at the algorithmic
// minimum), so the whole branch chain of a wrapped phase is folded into two LDS tables (LDS reads run on their own pipe)
// filled by the host (slr_capi.hip, slr_create):
//   lutR[d + 255] = R | S << 24.  R = 65536 / |d| + 1 (0 for d == 0): with t = n * R (24-bit signed multiply, the upper
//                   byte is ignored) s = t >> 16 (arithmetic) is floor(|n| / |d|) for n >= 0 and ~floor(|n| / |d|)
//                   for n < 0 -- exact for all |n|,|d| <= 255 (tests/test_decode_tables.py; on the device the
//                   exhaustive 511 x 511 image of tests/test_gpu_parity.py).  S = 2 / 9 / 6 for d < 0 / == 0 / > 0.
//   lutP[((S + sgn n) << 8) + s] = P * 2^24 as an integer, P = atanf((float)q) + offset exactly as the f32 expression
//                   of the reference evaluates it (host libm: the device never evaluates a transcendental), q = the C
//                   quotient n / d, offset: d<0 -> PI; d>0 -> (n>0 ? 2PI : 0); d==0 -> (n<0 ? PI/2 : n>0 ? 3PI/2 : 0;
//                   n == 0 is the reference's undefined P, Q5).  The slots (S-2 full, S one entry, S+1 full) of the
//                   three signs of d do not overlap.  Every P is 0 or at least 0.5 in magnitude and below 8, i.e. a
//                   multiple of 2^-24 below 2^27: the integer holds it exactly.
constexpr int kLutR = 0, kLutP = 512, kLutWords = kDecodeLutWords;
// valid == null in a decode launch: the flag is folded into the phase -- invalid pixels get this NaN, which K4 never matches
#define kInvalidPhase __uint_as_float(0x7FC00000u)
constexpr int kQ24TwoPI = (int)(kTwoPI * 16777216.0f);      // 2*PI as the reference's float, times 2^24: an integer

__device__ __forceinline__ int wrapped_nd_q24(int n, int d, const float *lut, int &nz)
{
    const int *luti = reinterpret_cast<const int *>(lut);
    const int e = luti[kLutR + 255 + d];
    const int s = __mul24(n, e) >> 16;
    int sn;                                           // sign() = clamp to [-1, 1] = one v_med3_i32 (hipcc otherwise
    asm("v_med3_i32 %0, %1, -1, 1" : "=v"(sn) : "v"(n));      // emits two compares and two selects)
    nz = n | d;                                       // == 0 <=> the reference leaves P[count] undefined (:254-255, Q5)
    return luti[kLutP + ((int)(((unsigned)e >> 24) + sn) << 8) + s];
}
__device__ __forceinline__ int wrapped_phase_q24(int G1, int G2, int G3, int G4, const float *lut, int &nz)
{
    return wrapped_nd_q24(G4 - G2, G1 - G3, lut, nz);
}

// mfreconstruct.cpp:265-268: P[] are doubles (holding f32 values), P12/P23 computed in f64 and narrowed once, the rest
// f32.  The f64 difference of two P (plus the f64 image of the float 2*PI) is exact, and so is the same expression on
// the 2^24-scaled integers; the one rounding of the f64 -> f32 narrowing is the rounding of v_cvt_f32_i32 (both
// round-to-nearest-even).  The f32 part runs on the scaled values (a power-of-two scale commutes with every rounding:
// nothing is near the subnormal range) and is scaled back once.
// first half: P12 (or P23) of two wrapped phases, as the f32 image of the 2^24-scaled integer difference
__device__ __forceinline__ float het_pair_q24(int Pa, int Pb)
{
    return (float)((Pa - Pb) + ((Pa > Pb) ? 0 : kQ24TwoPI));
}
// second half: P123 and the final scaling
__device__ __forceinline__ float het_finish_q24(float F12, float F23)
{
    constexpr float two_pi_q24 = kTwoPI * 16777216.0f;
    const float F123 = (F12 > F23) ? (F12 - F23) : (F12 - F23 + two_pi_q24);
    const float P123 = F123 * (1.0f / 16777216.0f);
    // P123 / (2*PI) * 255 (:268).  The correctly rounded quotient by a constant without the 10-instruction IEEE division
    // sequence (Markstein): q = x*rc, r = fma(-q, c, x) (exact), q' = fma(r, rc, q) == RN(x / c) when rc = RN(1/c) --
    // checked against x / c for every finite f32 x with 1e-30 <= |x| <= 1e30 (P123 is in (0, 4*PI]).
    constexpr float rc = 1.0f / kTwoPI;
    const float q = P123 * rc;
    const float r = __builtin_fmaf(-q, kTwoPI, P123);
    return __builtin_fmaf(r, rc, q) * 255;
}
__device__ __forceinline__ float heterodyne_q24(int P0, int P1, int P2)
{
    return het_finish_q24(het_pair_q24(P0, P1), het_pair_q24(P1, P2));
}
// SLR_OPT_EVAL_MODEL = 1: mfreconstruct.cpp:267-268 as the reference's own MSVC2010 x87 / fp:precise binary evaluates it (the
// 53-bit x87 stack, one rounding at each store).  The wrapped phases come from the
// x87 variant of lutP (float + float at 53 bits, stored to a double: the UNROUNDED sums, still integers at the 2^24 scale), P12
// and P23 are formed exactly as in the strict model (het_pair_q24: an exact difference, one rounding).  Then
//   P123 = P12 - P23 (+ 2*PI): the operands are f32 images of integers below 2^28, so the sum is exact in int32 and
//          v_cvt_f32_i32 is the ONE rounding of the x87 store (the strict model rounds twice in the + branch);
//   phase = P123 / (2*PI) * 255: quotient and product rounded to 53 bits, one narrowing to f32.  The quotient by the CONSTANT
//          c = (double)(2*PI) is Markstein's sequence in f64 -- q = x*rc, r = fma(-q, c, x) (exact), q' = fma(r, rc, q) == RN(x / c)
//          for rc = RN(1 / c) -- three v_fma_f64-class instructions at the f32 issue rate instead of the ~25 of an IEEE f64 division
//          (round 4's form: +40 % VALU time in the decode).  q' == x / c bit for bit for EVERY value F123 can take (all 2.1e8 integers
//          d of the line above, through the same f32 rounding: tests/test_x87_model.py::test_x87_quotient_by_constant runs the
//          oracle-side restatement of exactly these operations against the division).
//          The 2^24 scale of F123 rides in the constants (c * 2^24 and rc * 2^-24: exact, so every step is the same rounding of the
//          same real number as on the unscaled P123).
constexpr double kX87TwoPIQ24 = (double)kTwoPI * 16777216.0;
constexpr double kX87RcpTwoPIQ24 = (1.0 / (double)kTwoPI) * (1.0 / 16777216.0);   // RN(1 / c) * 2^-24 (constant-folded IEEE division)
__device__ __forceinline__ float het_finish_x87(float F12, float F23)
{
    // (F12, F23 are f32 images of integers: the conversion back is exact.  The fused decodes also pass the junk pairs of a sentinel
    //  wrapped phase, whose result is discarded: saturating conversions and wrapping integer arithmetic, nothing undefined)
    const int d = (int)(((unsigned)__float2int_rz(F12) - (unsigned)__float2int_rz(F23)) + ((F12 > F23) ? 0u : (unsigned)kQ24TwoPI));
    const double x = (double)(float)d;                                   // P123 * 2^24: the ONE rounding of the x87 store, widened
    const double q0 = x * kX87RcpTwoPIQ24;
    const double r = __builtin_fma(-q0, kX87TwoPIQ24, x);
    const double q = __builtin_fma(r, kX87RcpTwoPIQ24, q0);              // == P123 / (2*PI) rounded to 53 bits
    return (float)(q * 255.0);
}
template <bool X87>
__device__ __forceinline__ float heterodyne_ev(int P0, int P1, int P2)
{
    if constexpr (X87) return het_finish_x87(het_pair_q24(P0, P1), het_pair_q24(P1, P2));
    else return heterodyne_q24(P0, P1, P2);
}

// one pixel of K2: g[0]=white g[1]=black g[2..13] fringes.  SH: the samples sit in bits [SH, SH+8) of g[] with zeros
// above (the fused LDS kernel hands over its dot-product accumulators, SH = 16, so that the extraction folds into the
// subtractions as an SDWA operand select instead of 14 shifts).
template <int SH, bool X87 = false>
__device__ __forceinline__ float mf_pixel_sh(const int *gs, int black_thr, const float *lut, int &valid)
{
    int g[SLR_MF_PLANES];
#pragma unroll
    for (int p = 0; p < SLR_MF_PLANES; p++) g[p] = (int)((unsigned)gs[p] >> SH);
    // computeShadows :198-204: (float)white - (float)black > blackThreshold (exact in integers)
    const bool mask = g[0] - g[1] > black_thr;
    int nz0, nz1, nz2;
    const int P0 = wrapped_phase_q24(g[2], g[3], g[4], g[5], lut, nz0);
    const int P1 = wrapped_phase_q24(g[6], g[7], g[8], g[9], lut, nz1);
    const int P2 = wrapped_phase_q24(g[10], g[11], g[12], g[13], lut, nz2);
    const float ph = heterodyne_ev<X87>(P0, P1, P2);
    valid = (mask && nz0 != 0 && nz1 != 0 && nz2 != 0) ? 1 : 0;   // Q5 rule: an undefined P makes the pixel invalid
    return mask ? ph : 0.0f;
}
template <bool X87 = false>
__device__ __forceinline__ float mf_pixel(const int *g, int black_thr, const float *lut, int &valid)
{
    return mf_pixel_sh<0, X87>(g, black_thr, lut, valid);
}

// ------------------------------------------------------------------------------------------------------
// rectification taps (cv::remap fixed-point bilinear, SURVEY 8c-3 ii)
// ------------------------------------------------------------------------------------------------------
struct Tap {
    int off;        // sy*pitch + sx (valid only when inlier)
    int wx0, wx1;   // 32-fx, fx
    int wy0, wy1;   // 32-fy, fy
    int sx, sy;
    int kind;       // 0 = all four taps inside, 1 = fully outside (-> 0), 2 = partial (per-tap checks)
};

__device__ __forceinline__ Tap make_tap(int sx, int sy, unsigned frac, int pitch, int W, int H)
{
    Tap t;
    const int f = frac & 1023, fx = f & 31, fy = f >> 5;
    t.wx0 = 32 - fx; t.wx1 = fx; t.wy0 = 32 - fy; t.wy1 = fy;
    t.sx = sx; t.sy = sy;
    t.off = sy * pitch + sx;
    if ((unsigned)sx < (unsigned)(W - 1) && (unsigned)sy < (unsigned)(H - 1)) t.kind = 0;
    else if (sx >= W || sx + 1 < 0 || sy >= H || sy + 1 < 0) t.kind = 1;
    else t.kind = 2;
    return t;
}

// (s00*w00 + s01*w01 + s10*w10 + s11*w11 + 16384) >> 15 with w = a*b*32  ==  (h0*wy0 + h1*wy1 + 512) >> 10
__device__ __forceinline__ int blend(int s00, int s01, int s10, int s11, const Tap &t)
{
    const int h0 = s00 * t.wx0 + s01 * t.wx1;
    const int h1 = s10 * t.wx0 + s11 * t.wx1;
    return (h0 * t.wy0 + h1 * t.wy1 + 512) >> 10;
}

__device__ __forceinline__ int sample(const uint8_t *__restrict__ p, int pitch, int W, int H, const Tap &t)
{
    if (t.kind == 0) {
        const uint8_t *q = p + t.off;
        return blend(q[0], q[1], q[pitch], q[pitch + 1], t);
    }
    if (t.kind == 1) return 0;
    const bool x0 = (unsigned)t.sx < (unsigned)W, x1 = (unsigned)(t.sx + 1) < (unsigned)W;
    const bool y0 = (unsigned)t.sy < (unsigned)H, y1 = (unsigned)(t.sy + 1) < (unsigned)H;
    const int s00 = (x0 && y0) ? p[t.off] : 0;
    const int s01 = (x1 && y0) ? p[t.off + 1] : 0;
    const int s10 = (x0 && y1) ? p[t.off + pitch] : 0;
    const int s11 = (x1 && y1) ? p[t.off + pitch + 1] : 0;
    return blend(s00, s01, s10, s11, t);
}

__device__ __forceinline__ void load_lut(float *lut, const float *__restrict__ lut_g)
{
    for (int i = threadIdx.x; i < kLutWords; i += blockDim.x) lut[i] = lut_g[i];
    __syncthreads();
}

}  // namespace slr
