// kernels_match.hip -- K4 (multi-frequency phase match + Q-matrix triangulation) and K5 (Gray-code
// epipolar match + Q-matrix triangulation).  gfx950 (MI355X) only.
//
// Both search along one rectified image row, so a row of the right camera lives in LDS and rows are the
// unit of parallelism (one workgroup per row).  Neither is HBM-bound (PMC: VALU issue + LDS): K4 is an exact
// indexed first-match search (LDS radix sort + bin index), K5 an LDS sort + wave-parallel speculative walk that
// reproduces the reference's sequential `kstart` dependency exactly (wavefront shuffles do the prefix-max).
//
// Reference behaviour restated (never copied):
//   K4  MFReconstruct::triangulation                    Duke/mfreconstruct.cpp:272-334
//       Utilities::undistortPoints                      Duke/utilities.cpp:58-94
//   K5  Reconstruct::triangulation_ge (live code)       Duke/reconstruct.cpp:555-611
// f64 is used exactly where the reference uses it (undistortPoints, Q*p); built with -ffp-contract=off so
// no FMA contraction changes a rounding.
#include "slr_device.hpp"

#include <rocprim/block/block_radix_sort.hpp>

// stage ablation of the match kernels (profiles/k4_stages.sh): "leave after stage N, outputs are not written".  Only builds with
// -DSLR_DEBUG_HOOKS contain the exits; the production kernels carry none of them.  SLR_K4_ABL (timing only, wrong results): bit 0 no
// output stores, 1 no table gathers, 2 no slow-path quotients (512 x 8 shapes), 3 the phases of 64 rows from the middle of the frame,
// hot in L2 (round 5: what a fused decode -> match launch could save on the match side -- nothing, profiles/exp/r05/fusion_bounds.txt).
#if !defined(SLR_EXPERIMENTS) && (defined(SLR_K4_ABL) || defined(SLR_DEBUG_HOOKS))
#error "SLR_K4_ABL / SLR_DEBUG_HOOKS are experiment switches: build with -DSLR_EXPERIMENTS"
#endif
#if defined(SLR_DEBUG_HOOKS)
#define SLR_K4_STOP_AT(n) do { if (stop == (n)) return; } while (0)
#define SLR_K4_STOP_AT5(best0) do { if (stop == 5) { if ((best0) == 12345678) has[0] = 1; return; } } while (0)
#else
#define SLR_K4_STOP_AT(n) do { (void)stop; } while (0)
#define SLR_K4_STOP_AT5(best0) do { (void)stop; } while (0)
#endif

#include <math.h>
#include <stdlib.h>

namespace slr {

// utilities.cpp:58-94 (k[4] = 0 -> the k3 term vanishes exactly)
__device__ __forceinline__ void undistort_point(float px, float py, const DevCamera &c, float &ox, float &oy)
{
    double x = px, y = py;
    const double x0 = x = (x - c.cx) * c.ifx;
    const double y0 = y = (y - c.cy) * c.ify;
#pragma unroll 1
    for (int it = 0; it < 5; it++) {
        const double r2 = x * x + y * y;
        const double icdist = 1. / (1 + ((0 * r2 + c.k1) * r2 + c.k0) * r2);
        const double deltaX = 2 * c.k2 * x * y + c.k3 * (r2 + 2 * x * x);
        const double deltaY = c.k2 * (r2 + 2 * y * y) + 2 * c.k3 * x * y;
        x = (x0 - deltaX) * icdist;
        y = (y0 - deltaY) * icdist;
    }
    ox = (float)((double)(float)(x * c.fx) + c.cx);
    oy = (float)((double)(float)(y * c.fy) + c.cy);
}

// p3D = Q * p2D (cv::Mat f64 GEMM, sequential accumulation), X = (float)(xyz / w).
// simple != 0: Q has cv::stereoRectify's pattern [[1,0,0,a],[0,1,0,b],[0,0,0,c],[0,0,d,e]] (checked on the host); the
// products with the structural zeros/ones are exact no-ops (0*x = 0, s + 0 = s, 1*x = x for finite x), so the short
// form gives the same bits with 1 multiply + 3 adds instead of 16 + 16.
__device__ __forceinline__ void reproject(const double *Q, int simple, double p0, double p1, double p2, float X[3])
{
    double r[4];
    if (simple) {
        r[0] = p0 + Q[3];
        r[1] = p1 + Q[7];
        r[2] = Q[11];
        r[3] = Q[14] * p2 + Q[15];
    } else {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            double s = 0;
            s += Q[i * 4 + 0] * p0;
            s += Q[i * 4 + 1] * p1;
            s += Q[i * 4 + 2] * p2;
            s += Q[i * 4 + 3] * 1.0;
            r[i] = s;
        }
    }
    // X = (float)(r.xyz / r.w): three f64 divisions by the same w (~13 double-rate instructions each).  One division
    // y = RN(1/w) and, per component, Markstein's correction  q = RN(r*y), e = fma(-q, w, r) (exact), q' = fma(e, y, q)
    // give the CORRECTLY ROUNDED quotient RN(r/w) -- the same bits -- whenever nothing under/overflows; 4e8 random and
    // adversarial (all-ones mantissas, exact and near-tie quotients) pairs agree with the division on the host.  The
    // exponent guard keeps every intermediate far from the subnormal / overflow range; anything else (incl. exact
    // zeros, w == 0, NaN) takes the three real divisions.
    auto expo = [](double x) -> unsigned { return ((unsigned)__double2hiint(x) >> 20) & 0x7FFu; };
    const unsigned e0 = expo(r[0]), e1 = expo(r[1]), e2 = expo(r[2]), e3 = expo(r[3]);
    const unsigned emin = min(min(e0, e1), min(e2, e3)), emax = max(max(e0, e1), max(e2, e3));
    if (emin >= 1023u - 200u && emax < 1023u + 200u) {
        const double w = r[3], y = 1.0 / w;
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const double q = r[i] * y;
            const double e = __builtin_fma(-q, w, r[i]);
            X[i] = (float)__builtin_fma(e, y, q);
        }
    } else {
        X[0] = (float)(r[0] / r[3]);
        X[1] = (float)(r[1] / r[3]);
        X[2] = (float)(r[2] / r[3]);
    }
}

// matCoordTrans(3x4 f32) * [X;1]  (OpenCV f32 GEMM: f64 accumulate, narrow once)
__device__ __forceinline__ void apply_T(const float *T, float X[3])
{
    float Y[3];
#pragma unroll
    for (int r = 0; r < 3; r++) {
        double s = 0;
        s += (double)T[r * 4 + 0] * (double)X[0];
        s += (double)T[r * 4 + 1] * (double)X[1];
        s += (double)T[r * 4 + 2] * (double)X[2];
        s += (double)T[r * 4 + 3] * 1.0;
        Y[r] = (float)s;
    }
    X[0] = Y[0]; X[1] = Y[1]; X[2] = Y[2];
}

// mfreconstruct.cpp:295 fabs(cam1Pix[0] - cam2Pix[0]) < 0.1.  Strict IEEE (default): the f32 difference is rounded before it
// widens, and |d| < 0.1 (double) <=> |d| < 0.1f for a float d.  SLR_OPT_EVAL_MODEL = 1: the reference's x87 binary keeps the
// difference of the two floats on the 53-bit stack (exact here) and hands it to fabs(double) unrounded.
__device__ __forceinline__ bool phase_match(float a, float b, int x87)
{
    return x87 ? fabs((double)a - (double)b) < 0.1 : fabsf(a - b) < 0.1f;
}

// ------------------------------------------------------------------------------------------------------
// K4 (exact brute-force form): one workgroup per row; right-row phase in LDS with NaN marking "no phase"
// (NaN never passes fabsf(d) < 0.1f), each lane owns left pixels j, every lane sweeps k ascending and
// retires at its first hit (mfreconstruct.cpp:289-327, Q7).  The LDS reads are wave-uniform broadcasts.
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mf_match_kernel(const float *__restrict__ phaseL, const uint8_t *__restrict__ validL,
                                                       const float *__restrict__ phaseR, const uint8_t *__restrict__ validR,
                                                       int W, int H, int row0, DevCalib cal, float *__restrict__ xyz,
                                                       uint8_t *__restrict__ has, int32_t *__restrict__ match_k)
{
    extern __shared__ float phR[];                 // W rounded up to a multiple of 4, NaN padded
    const int row = blockIdx.x + row0;                   // absolute image row (row0: first row of a band)
    const size_t base = (size_t)blockIdx.x * W;
    const int Wp = (W + 3) & ~3;
    for (int k = threadIdx.x; k < Wp; k += 256)
        phR[k] = (k < W && (!validR || validR[base + k])) ? phaseR[base + k] : __builtin_nanf("");
    __syncthreads();

    for (int j0 = 0; j0 < W; j0 += 256) {
        const int j = j0 + threadIdx.x;
        const bool inb = j < W;
        const bool act = inb && (!validL || validL[base + j]);   // (null: invalid pixels carry a NaN phase and never match)
        const float pl = act ? phaseL[base + j] : 0.0f;
        int best = -1;
        bool searching = act;
        for (int k = 0; k < Wp; k += 4) {
            if (!__any(searching)) break;
            const float4 r = *reinterpret_cast<const float4 *>(phR + k);
            const bool c0 = phase_match(pl, r.x, cal.eval_x87), c1 = phase_match(pl, r.y, cal.eval_x87);
            const bool c2 = phase_match(pl, r.z, cal.eval_x87), c3 = phase_match(pl, r.w, cal.eval_x87);
            if (searching && (c0 | c1 | c2 | c3)) {
                best = k + (c0 ? 0 : (c1 ? 1 : (c2 ? 2 : 3)));
                searching = false;
            }
        }
        if (!inb) continue;
        float X[3] = {0.0f, 0.0f, 0.0f};
        if (best >= 0) {
            float ulx, uly, urx, ury;
            undistort_point((float)j, (float)row, cal.cam[0], ulx, uly);      // mfreconstruct.cpp:297
            undistort_point((float)best, (float)row, cal.cam[1], urx, ury);   // :298
            reproject(cal.Q, 0, (double)ulx, (double)uly, cal.eval_x87 ? (double)ulx - (double)urx : (double)(float)(ulx - urx), X);   // :299-311 (literal form)
            if (cal.has_T) apply_T(cal.T, X);                                 // :315-323
        }
        float *o = xyz + 3 * (base + j);
        o[0] = X[0]; o[1] = X[1]; o[2] = X[2];
        has[base + j] = best >= 0 ? 1 : 0;
        if (match_k) match_k[base + j] = best;
    }
}

// ------------------------------------------------------------------------------------------------------
// K4 (exact indexed form).  The sweep above is O(W) per left pixel; the same answer is available in
// O(1) expected because "first k (ascending) with |phiL - phiR[k]| < 0.1" only ever selects, for each DISTINCT
// right phase value, that value's smallest column.  One 1024-thread workgroup per row (4 pixels per thread, all
// row I/O vectorised and issued up front):
//   1. the right row's keys (12-bit phase bin, 4-bit hash of the value, column) are radix-sorted in LDS on the 16
//      group bits only (2 passes, stable -> ascending column inside a group).  Any function of phi is a valid group
//      key: equal phases share it, and colliding distinct values merely yield a few extra run heads;
//   2. run heads (phase differs from its predecessor) are compacted -> (phi, min column) pairs grouped by bin;
//   3. a suffix-min scan builds binfirst[b] = first pair whose bin >= b; two values closer than 0.1001 are at most
//      one 0.25-wide bin apart, so bins [b-1, b+1] of phiL are a superset of its candidates;
//   4. every left pixel applies the reference's own f32 predicate fabsf(phiL - phiR) < 0.1f to that window
//      (~6 pairs, four per LDS round trip) and keeps the smallest column -- identical to the linear sweep (asserted
//      against the oracle AND the sweep kernel on adversarial rows and on the whole 4096x3000 frame);
//   5. Utilities::undistortPoints comes from per-pixel tables, Q*p and the three divisions stay f64.
// Phase split at 4096x3000 (profiles/k4_stages.sh): loads 21 us, sort 66, heads 13, bins 10, queries 43, f64 + stores 55.
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned sortable_key(float f)
{
    const unsigned b = __float_as_uint(f);
    return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float key_to_float(unsigned k)
{
    return __uint_as_float(k ^ ((k >> 31) ? 0x80000000u : 0xFFFFFFFFu));
}

constexpr int kBins = 4096;                              // 0.25-wide bins over [-512, 512), clamped outside
__device__ __forceinline__ int phase_bin(float p)
{
    const float t = fminf(fmaxf((p + 512.0f) * 4.0f, 0.0f), (float)(kBins - 1));
    return (int)t;                                       // t >= 0: truncation == floor
}

// row I/O helpers: thread t owns the IPT consecutive pixels [t*IPT, (t+1)*IPT) ("blocked" arrangement), so a
// row is read and written with a few wide, mutually independent vector accesses per thread
template <int IPT>
__device__ __forceinline__ void load_f32_blocked(const float *__restrict__ p, int k0, int W, bool vec, float out[IPT])
{
    if (vec && IPT >= 4 && k0 + IPT <= W) {
#pragma unroll
        for (int i = 0; i < IPT; i += 4) {
            const float4 v = *reinterpret_cast<const float4 *>(p + k0 + i);
            out[i] = v.x; out[i + 1] = v.y; out[i + 2] = v.z; out[i + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int i = 0; i < IPT; i++) out[i] = (k0 + i < W) ? p[k0 + i] : 0.0f;
    }
}
template <int IPT>
__device__ __forceinline__ void load_u8_blocked(const uint8_t *__restrict__ p, int k0, int W, bool vec, unsigned out[IPT])
{
    if (vec && IPT >= 4 && k0 + IPT <= W) {
#pragma unroll
        for (int i = 0; i < IPT; i += 4) {
            const unsigned v = *reinterpret_cast<const unsigned *>(p + k0 + i);
            out[i] = v & 0xFFu; out[i + 1] = (v >> 8) & 0xFFu; out[i + 2] = (v >> 16) & 0xFFu; out[i + 3] = v >> 24;
        }
    } else {
#pragma unroll
        for (int i = 0; i < IPT; i++) out[i] = (k0 + i < W) ? p[k0 + i] : 0u;
    }
}
// valid bytes of a row, or "all valid" when the caller folded the flags into the phases as NaNs (valid == null: the
// internal phase format of slr_reconstruct_mf*, see launch_mf_decode)
template <int IPT>
__device__ __forceinline__ void load_valid_blocked(const uint8_t *__restrict__ valid, size_t base, int k0, int W, bool vec,
                                                   unsigned out[IPT])
{
    if (valid) load_u8_blocked<IPT>(valid + base, k0, W, vec, out);
    else {
#pragma unroll
        for (int i = 0; i < IPT; i++) out[i] = 1u;
    }
}

// the common tail of the indexed forms: table gathers for all IPT pixels first (independent loads), then the f64
// reprojection and wide stores (mfreconstruct.cpp:297-326)
template <int IPT>
__device__ __forceinline__ void k4_emit(const int best[IPT], size_t base, int k0, int row, int W, bool vec,
                                        /* base: band-local pixel offset of the row; the tables use the absolute row */
                                        const DevCalib &cal, const float2 *__restrict__ undL,
                                        const float *__restrict__ undRx, float *__restrict__ xyz,
                                        uint8_t *__restrict__ has, int32_t *__restrict__ match_k,
                                        const float *Tm = nullptr /* T to use instead of cal.T (see the chunked kernel) */)
{
    const float *T = Tm ? Tm : cal.T;
    // triangulate: gather the table values for all IPT pixels first (independent loads), then the f64 math
    float ulx[IPT], uly[IPT], urx[IPT];
    if (undL) {
#pragma unroll
        for (int i = 0; i < IPT; i++) {
            ulx[i] = uly[i] = urx[i] = 0.0f;
            if (best[i] >= 0) {
                const float2 u = undL[(size_t)row * W + k0 + i];
                ulx[i] = u.x; uly[i] = u.y;
                urx[i] = undRx[(size_t)row * W + best[i]];
            }
        }
    } else {
#pragma unroll 1
        for (int i = 0; i < IPT; i++) {
            ulx[i] = uly[i] = urx[i] = 0.0f;
            if (best[i] >= 0) {
                float ury;
                undistort_point((float)(k0 + i), (float)row, cal.cam[0], ulx[i], uly[i]);
                undistort_point((float)best[i], (float)row, cal.cam[1], urx[i], ury);
            }
        }
    }
    const bool full = vec && IPT >= 4 && k0 + IPT <= W;
#pragma unroll
    for (int i0 = 0; i0 < IPT; i0 += 4) {
        float out[12];
        unsigned hw = 0;
        int mk[4];
#pragma unroll
        for (int q = 0; q < 4 && i0 + q < IPT; q++) {
            const int i = i0 + q;
            float X[3] = {0.0f, 0.0f, 0.0f};
            if (best[i] >= 0) {
                // :299 camPixelUDL.x - camPixelUDR.x stored to a double: rounded to f32 first under strict IEEE, unrounded on x87
                const double disp = cal.eval_x87 ? (double)ulx[i] - (double)urx[i] : (double)(float)(ulx[i] - urx[i]);
                reproject(cal.Q, cal.q_simple, (double)ulx[i], (double)uly[i], disp, X);
                if (cal.has_T) apply_T(T, X);
            }
            out[3 * q] = X[0]; out[3 * q + 1] = X[1]; out[3 * q + 2] = X[2];
            hw |= (best[i] >= 0 ? 1u : 0u) << (8 * q);
            mk[q] = best[i];
        }
        const size_t o = base + k0 + i0;
        if (full) {
            float4 *dst = reinterpret_cast<float4 *>(xyz + 3 * o);
            typedef float f32x4 __attribute__((ext_vector_type(4)));
            f32x4 *d4 = reinterpret_cast<f32x4 *>(dst);                  // streaming output: non-temporal stores
            f32x4 v0 = {out[0], out[1], out[2], out[3]}, v1 = {out[4], out[5], out[6], out[7]}, v2 = {out[8], out[9], out[10], out[11]};
            __builtin_nontemporal_store(v0, d4);
            __builtin_nontemporal_store(v1, d4 + 1);
            __builtin_nontemporal_store(v2, d4 + 2);
            __builtin_nontemporal_store(hw, reinterpret_cast<unsigned *>(has + o));
            if (match_k) *reinterpret_cast<int4 *>(match_k + o) = make_int4(mk[0], mk[1], mk[2], mk[3]);
        } else {
#pragma unroll
            for (int q = 0; q < 4 && i0 + q < IPT; q++) {
                if (k0 + i0 + q >= W) break;
                xyz[3 * (o + q)] = out[3 * q]; xyz[3 * (o + q) + 1] = out[3 * q + 1]; xyz[3 * (o + q) + 2] = out[3 * q + 2];
                has[o + q] = (uint8_t)((hw >> (8 * q)) & 1u);
                if (match_k) match_k[o + q] = mk[q];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// K4 (exact indexed form, hash + counting-sort build).  Same candidate-window query as the sorted form below, but
// the per-bin lists of DISTINCT right phases are built without a sort:
//   A. the reference's phases are heavily quantised (atan of an INTEGER quotient, Q1): a 4096-pixel row holds only a
//      few hundred distinct values.  An LDS open-addressing hash table keyed by the phase bits keeps, per distinct
//      value, its smallest column (atomicCAS claims the slot, atomicMin on the column) -- "first k ascending" can only
//      ever pick that one;
//   B. the pixels that ARE their value's smallest column are the representatives; they are counting-sorted by
//      0.25-wide phase bin: LDS histogram with returning atomics (bin, rank), one block scan over the 4096 bins,
//      scatter of (phi, k) to start[bin] + rank.  The order inside a bin is arbitrary, which is fine because the
//      query takes the MIN column over all pairs that satisfy the predicate.
// The table's LDS is dead once the representatives are known and is reused for the bin index and the pairs.
// ------------------------------------------------------------------------------------------------------
template <int BLOCK, int IPT>
__global__ __launch_bounds__(BLOCK, (BLOCK == 1024 ? 8 : 4)) void mf_match_binned_kernel(const float *__restrict__ phaseL, const uint8_t *__restrict__ validL,
                                                                const float *__restrict__ phaseR, const uint8_t *__restrict__ validR,
                                                                int W, int H, int row0, DevCalib cal, int vec_ok,
                                                                const float2 *__restrict__ undL, const float *__restrict__ undRx,
                                                                float *__restrict__ xyz,
                                                                uint8_t *__restrict__ has, int32_t *__restrict__ match_k)
{
    constexpr int N = BLOCK * IPT;
    constexpr int TS = 2 * N;                            // hash slots (power of two, load factor <= 0.5)
    constexpr int kPer = kBins / BLOCK;                  // bins per thread (kBins is a multiple of BLOCK)
    constexpr unsigned kEmpty = 0xFFFFFFFFu;             // a NaN pattern: never a candidate's phase bits
    __shared__ union {
        struct { unsigned key[TS]; unsigned mink[TS]; } t;               // phase bits -> smallest column
        struct { float2 pk[N]; unsigned binstart[kBins + 1]; } b;        // (phi, column as bits) grouped by bin; bin index
    } sh;
    __shared__ unsigned scan_tmp[BLOCK / 64];

    const int row = blockIdx.x + row0, tid = threadIdx.x;   // absolute image row (row0: first row of a band)
    const size_t base = (size_t)blockIdx.x * W;
    const int k0 = tid * IPT;
    const bool vec = (vec_ok & 1) != 0;
    const int stop = vec_ok >> 8;                        // debug: leave after phase N (SLR_DEBUG_K4_STOP), 0 = run all

    float pr[IPT], pl[IPT];
    unsigned vr[IPT], vl[IPT];
    load_f32_blocked<IPT>(phaseR + base, k0, W, vec, pr);
    load_valid_blocked<IPT>(validR, base, k0, W, vec, vr);
    load_f32_blocked<IPT>(phaseL + base, k0, W, vec, pl);
    load_valid_blocked<IPT>(validL, base, k0, W, vec, vl);
#pragma unroll
    for (int q = 0; q < 2 * IPT; q++) { sh.t.key[tid + q * BLOCK] = kEmpty; sh.t.mink[tid + q * BLOCK] = kEmpty; }
    __syncthreads();
    SLR_K4_STOP_AT(1);

    // A. distinct values and their smallest column
    unsigned slot[IPT];                                  // hash slot of this pixel's value; kEmpty = not a candidate
    bool prev_ok = false;
    unsigned prev_bits = 0;
#pragma unroll
    for (int i = 0; i < IPT; i++) {
        const bool ok = (k0 + i < W) && vr[i] && (pr[i] == pr[i]);   // NaN can never satisfy the predicate
        const unsigned bits = __float_as_uint(pr[i]);
        const bool dup = prev_ok && ok && bits == prev_bits;         // same value one column to the left (flat regions)
        slot[i] = kEmpty;
        if (ok && !dup) {
            unsigned h = (bits * 2654435761u) >> (32 - __builtin_ctz(TS));
            for (;;) {
                const unsigned old = atomicCAS(&sh.t.key[h], kEmpty, bits);
                if (old == kEmpty || old == bits) break;
                h = (h + 1) & (TS - 1);
            }
            atomicMin(&sh.t.mink[h], (unsigned)(k0 + i));
            slot[i] = h;
        }
        prev_ok = ok; prev_bits = bits;
    }
    __syncthreads();
    unsigned repmask = 0;
#pragma unroll
    for (int i = 0; i < IPT; i++)
        if (slot[i] != kEmpty && sh.t.mink[slot[i]] == (unsigned)(k0 + i)) repmask |= 1u << i;
    __syncthreads();                                     // the table is dead from here on
    SLR_K4_STOP_AT(2);

    // B. counting sort of the representatives by phase bin
#pragma unroll
    for (int q = 0; q < kPer; q++) sh.b.binstart[tid + q * BLOCK] = 0u;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < IPT; i++) {
        if (repmask & (1u << i)) {
            const unsigned b = (unsigned)phase_bin(pr[i]);
            slot[i] = (b << 16) | atomicAdd(&sh.b.binstart[b], 1u);
        }
    }
    __syncthreads();
    {
        unsigned c[kPer], sum = 0;
#pragma unroll
        for (int q = 0; q < kPer; q++) { c[q] = sh.b.binstart[tid * kPer + q]; sum += c[q]; }
        unsigned total;
        unsigned excl = wg_exclusive_scan<BLOCK>(sum, 0u, [](unsigned a, unsigned b) { return a + b; }, scan_tmp, &total);
#pragma unroll
        for (int q = 0; q < kPer; q++) { sh.b.binstart[tid * kPer + q] = excl; excl += c[q]; }   // own bins only: no hazard
        if (tid == 0) sh.b.binstart[kBins] = total;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < IPT; i++)
        if (repmask & (1u << i))
            sh.b.pk[sh.b.binstart[slot[i] >> 16] + (slot[i] & 0xFFFFu)] = make_float2(pr[i], __uint_as_float((unsigned)(k0 + i)));
    __syncthreads();
    SLR_K4_STOP_AT(4);

    // queries: exact reference predicate over the <= 3-bin candidate window, smallest column wins
    int best[IPT];
    int qi0[IPT], qi1[IPT];
#pragma unroll
    for (int i = 0; i < IPT; i++) {                      // all window bounds first: independent LDS reads
        const bool act = k0 + i < W && vl[i] && pl[i] == pl[i];
        const int b = act ? phase_bin(pl[i]) : 0;
        qi0[i] = (int)sh.b.binstart[b > 0 ? b - 1 : 0];
        qi1[i] = act ? (int)sh.b.binstart[b + 2 < kBins ? b + 2 : kBins] : 0;
    }
#pragma unroll
    for (int i = 0; i < IPT; i++) {
        unsigned bk = 0xFFFFFFFFu;
        for (int idx = qi0[i]; idx < qi1[i]; idx += 4) {     // 4 candidates per round trip (8-byte (phi, k) pairs)
            float2 c[4];
#pragma unroll
            for (int q = 0; q < 4; q++) c[q] = sh.b.pk[idx + q < N ? idx + q : N - 1];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const bool hit = idx + q < qi1[i] && phase_match(pl[i], c[q].x, cal.eval_x87);
                const unsigned kk = hit ? __float_as_uint(c[q].y) : 0xFFFFFFFFu;
                bk = kk < bk ? kk : bk;
            }
        }
        best[i] = bk == 0xFFFFFFFFu ? -1 : (int)bk;
    }
    SLR_K4_STOP_AT5(best[0]);
    k4_emit<IPT>(best, base, k0, row, W, vec, cal, undL, undRx, xyz, has, match_k);
}

// ------------------------------------------------------------------------------------------------------
// K4 (exact indexed form, lean): mf_match_binned_kernel's index (phases A and B) with a query and a triangulation cut
// down to what the usual call needs -- 4 pixels per thread, aligned rows, undistortion tables present, Q of
// cv::stereoRectify's pattern.  Everything else takes the general kernel above.  K4 is bound by VALU issue (78 % busy, f64 at
// half rate), so the changes are instruction-count changes that keep every bit:
//   * the bin index is shifted by one bin and padded (bs[0] = 0, two copies of the total behind the last bin): one
//     ds_read2_b32 returns both ends of a pixel's window [bin-1, bin+1], no clamping;
//   * the candidate pairs end in NaN sentinels and every entry of the array is a genuine (phase, column) pair of the right row,
//     so the exact predicate may be evaluated on ANY of them without changing the minimum: the loop reads two aligned
//     16-byte words per step from the window's start rounded down to an even pair, without bound tests per candidate;
//   * the general kernel carries both cameras' intrinsics and the whole Q in SGPRs (the table-less path needs them) and
//     spills them to VGPR lanes; this one takes 5 + 12 doubles (K4Lean), T already widened;
//   * matCoordTrans: a product of two floats is exact in f64, so fma(T, X, s) == T * X + s rounded once -- the same value as the
//     reference's multiply-then-add, in half the instructions; s starts from fma(T0, X0, +0.0), which also reproduces 0 + (-0);
//   * the table gathers, the f64 arithmetic and the stores of a thread's 4 pixels are branch-free (unmatched pixels compute on
//     column 0's table entry and are replaced by zeros at the end).
// ------------------------------------------------------------------------------------------------------
struct K4Lean {
    double q3, q7, q11, q14, q15;      // Q's five entries that are not structural zeros / ones
    double T[12];                      // matCoordTrans widened to f64 (exact)
};

// X87 (SLR_OPT_EVAL_MODEL = 1): the two lines of mfreconstruct.cpp:295 / :299 that the reference's x87 binary
// evaluates differently -- the predicate sees the EXACT difference of the two phases (f64: the difference of two floats of this
// range is exact in 53 bits) and the disparity reaches the double unrounded.  The index is the same: its bins only have to
// hold every pair closer than 0.1 + rounding, which a 0.25-wide bin and its neighbours do under either predicate.
// NOHASH (round 6): the index WITHOUT the hash dedup.  The dedup never changes a result -- the query takes the smallest matching column
// whatever else is in its window -- it only bounds the bins of rows with many equal phases.  So every valid right pixel (minus an
// in-thread left neighbour of the same value) goes straight into the counting sort: no 64 KB table to clear, no CAS / atomicMin probes
// (the random-slot LDS traffic behind the kernel's bank conflicts), two barriers fewer per row.  A row in which some bin ends up with
// more than kHeavyBin entries (flat or saturated regions) is rebuilt with the hash dedup in the same workgroup: bit-equal either way.
// MEASURED SLOWER on the reference's own phases (see the launch site): SLR_OPT_MF_MATCH_ALGO = 9 in FORMS=all builds only.
constexpr unsigned kHeavyBin = 32;
template <int BLOCK, bool HAS_T, bool XCHG = false, bool TAME = false, bool X87 = false, bool NOHASH = false>
__global__ __launch_bounds__(BLOCK, (BLOCK == 1024 ? 8 : 4)) void mf_match_lean_kernel(const float *__restrict__ phaseL, const uint8_t *__restrict__ validL,
                                                              const float *__restrict__ phaseR, const uint8_t *__restrict__ validR,
                                                              int W, int H, int row0, K4Lean kc, int stop,
                                                              const float4 *__restrict__ undL, const float *__restrict__ undRx,
                                                              float *__restrict__ xyz,
                                                              uint8_t *__restrict__ has, int32_t *__restrict__ match_k,
                                                              int nframes, size_t frame_px)
{
    constexpr int IPT = 4;
    constexpr int N = BLOCK * IPT;
    constexpr int TS = 2 * N;                            // hash slots (power of two, load factor <= 0.5)
    constexpr int kPer = kBins / BLOCK;                  // bins per thread (kBins is a multiple of BLOCK)
    constexpr unsigned kEmpty = 0xFFFFFFFFu;             // a NaN pattern: never a candidate's phase bits
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    constexpr bool kCntInKey = TS >= kBins + BLOCK;        // the bin counters reuse the key half of the table when they fit it
    __shared__ union {
        struct { unsigned key[TS]; unsigned mink[TS]; } t;               // phase bits -> smallest column
        struct { f32x4 pk2[N / 2 + 3]; unsigned bs[kBins + 3]; } b;      // pairs of (phi, column as bits) grouped by bin (+ sentinels, + a sink); bin index
        struct { unsigned pad[kCntInKey ? 1 : 2 * TS]; unsigned cnt[kCntInKey ? 1 : kBins + BLOCK]; } c;   // (small rows: behind the table)
    } sh;
    __shared__ unsigned scan_tmp[BLOCK / 64];
    static_assert(BLOCK != 1024 || sizeof(sh.b) <= sizeof(sh.t), "the index must fit the dead hash table (two workgroups per CU)");

    // Several frames in one launch (slr_reconstruct_mf_batch): the undistortion tables are per calibration, not per frame -- 12 of the
    // 35 bytes per pixel K4 moves.  Workgroup b runs on XCD b % 8; the workgroups of one XCD take row r of frames 0 .. nframes - 1 one
    // after the other, so every frame after the first finds the row's table entries in that XCD's L2.
    int brow = (int)blockIdx.x, frame = 0;
    if (nframes > 1) {
        const int xcd = (int)blockIdx.x & 7, q = (int)blockIdx.x >> 3;
        frame = q % nframes;
        brow = (q / nframes) * 8 + xcd;
        if (brow >= H) return;                           // (the grid is padded to whole groups of 8 rows)
        const size_t fo = (size_t)frame * frame_px;
        phaseL += fo; phaseR += fo; xyz += 3 * fo; has += fo;
        if (validL) validL += fo;
        if (validR) validR += fo;
        if (match_k) match_k += fo;
    }
    const int row = brow + row0, tid = threadIdx.x;      // absolute image row (row0: first row of a band)
    const size_t base = (size_t)brow * W;
    const int k0 = tid * IPT;
    const bool inrow = k0 < W;                           // W % 4 == 0: a thread's 4 pixels are all inside or all outside

    // a thread's 4 pixels are whole inside or outside the row (W % 4 == 0): plain vector loads, all issued up front
    float pr[IPT] = {0, 0, 0, 0}, pl[IPT] = {0, 0, 0, 0};
    unsigned vr[IPT] = {0, 0, 0, 0}, vl[IPT] = {0, 0, 0, 0};
    if (inrow) {
#if defined(SLR_K4_ABL) && (SLR_K4_ABL & 8)
        // ablation (round 5): the phases of 64 rows only, hot in L2 -- the match side of a fused decode -> match launch at best
        const size_t lbase = (size_t)((brow & 63) + (H > 128 ? H / 2 - 32 : 0)) * W;   // (rows from the middle of the frame: the top rows of a verged rig are half empty)
#else
        const size_t lbase = base;
#endif
        load_f32_blocked<IPT>(phaseR + lbase, k0, k0 + IPT, true, pr);
        load_valid_blocked<IPT>(validR, base, k0, k0 + IPT, true, vr);
        load_f32_blocked<IPT>(phaseL + lbase, k0, k0 + IPT, true, pl);
        load_valid_blocked<IPT>(validL, base, k0, k0 + IPT, true, vl);
    }
    const size_t trow = (size_t)row * W;
    unsigned *const cnt = kCntInKey ? sh.t.key : sh.c.cnt;
    unsigned slot[IPT], bits[IPT], old[IPT];             // hash slot of this pixel's value; kEmpty = not a candidate
    bool cand[IPT];
    unsigned repmask = 0;
#pragma unroll
    for (int i = 0; i < IPT; i++) {
        bits[i] = __float_as_uint(pr[i]);
        const bool ok = inrow && vr[i] && (pr[i] == pr[i]);          // NaN can never satisfy the predicate
        // same value one column to the left (flat regions): that one is (or defers to) the representative
        const bool dup = i > 0 && ok && vr[i - 1] && bits[i] == bits[i - 1];
        cand[i] = ok && !dup;
        slot[i] = (bits[i] * 2654435761u) >> (32 - __builtin_ctz(TS));
        if (NOHASH && cand[i]) repmask |= 1u << i;       // without the dedup every candidate is a representative
    }
    // A. (hash form, or a heavy row of the NOHASH form) distinct values and their smallest column, as mf_match_binned_kernel; the four
    //    first probes are independent LDS atomics in flight together, a collision continues in the loop.  Leaves repmask and the
    //    zeroed bin counters behind a barrier.
    auto hash_dedup = [&]() {
        {
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 e4 = {kEmpty, kEmpty, kEmpty, kEmpty};
            u32x4 *const t4 = reinterpret_cast<u32x4 *>(sh.t.key);   // (key and mink are adjacent: the whole table in 16-byte stores)
#pragma unroll
            for (int q = 0; q < 2 * TS / 4 / BLOCK; q++) t4[tid + q * BLOCK] = e4;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < IPT; i++) old[i] = cand[i] ? atomicCAS(&sh.t.key[slot[i]], kEmpty, bits[i]) : kEmpty;
#pragma unroll
        for (int i = 0; i < IPT; i++) {
            if (cand[i]) {
                unsigned h = slot[i], o = old[i];
                while (o != kEmpty && o != bits[i]) {
                    h = (h + 1) & (TS - 1);
                    o = atomicCAS(&sh.t.key[h], kEmpty, bits[i]);
                }
                atomicMin(&sh.t.mink[h], (unsigned)(k0 + i));
                slot[i] = h;
            }
        }
        __syncthreads();
        // representatives: the pixels that are their value's smallest column.  The key half of the table is dead already and
        // becomes the bin counters (cnt[kBins + tid] = a sink for the thread's non-representatives: the atomics below are
        // unconditional and independent, and never pile up on one address)
        repmask = 0;
        unsigned mk4[IPT];
#pragma unroll
        for (int i = 0; i < IPT; i++) mk4[i] = sh.t.mink[slot[i]];
#pragma unroll
        for (int q = 0; q < kPer; q++) cnt[tid + q * BLOCK] = 0u;
#pragma unroll
        for (int i = 0; i < IPT; i++)
            if (cand[i] && mk4[i] == (unsigned)(k0 + i)) repmask |= 1u << i;
        __syncthreads();                                 // the whole table is dead from here on
    };
    if constexpr (NOHASH) {
#pragma unroll
        for (int q = 0; q < kPer; q++) cnt[tid + q * BLOCK] = 0u;
        __syncthreads();
        SLR_K4_STOP_AT(1);
    } else {
        SLR_K4_STOP_AT(1);                               // (debug-hook builds: before the table clear since round 6)
        hash_dedup();
    }
    SLR_K4_STOP_AT(2);

    // B. counting sort of the representatives by phase bin
    unsigned bin[IPT], rank[IPT];
    float2 *const pk = reinterpret_cast<float2 *>(sh.b.pk2);
#pragma unroll 1
    for (int pass = 0;; pass++) {
#pragma unroll
        for (int i = 0; i < IPT; i++) bin[i] = (repmask >> i) & 1u ? (unsigned)phase_bin(pr[i]) : (unsigned)(kBins + tid);
#pragma unroll
        for (int i = 0; i < IPT; i++) rank[i] = atomicAdd(&cnt[bin[i]], 1u);
        __syncthreads();
        unsigned c[kPer], sum = 0;
#pragma unroll
        for (int q = 0; q < kPer; q++) {
            c[q] = cnt[tid * kPer + q]; sum += c[q];
            if (NOHASH && c[q] > kHeavyBin) sum += 0x10000u;          // (the row's counts stay below 2^16: the heavy-bin count rides above them)
        }
        unsigned total;
        unsigned excl = wg_exclusive_scan<BLOCK>(sum, 0u, [](unsigned a, unsigned b) { return a + b; }, scan_tmp, &total);   // (its barrier: every thread has read its counters)
        if (NOHASH && pass == 0 && (total >> 16) != 0u) {             // workgroup-uniform: some bin is heavy -> this row takes the dedup
            __syncthreads();
            hash_dedup();
            continue;
        }
        excl &= 0xFFFFu; total &= 0xFFFFu;
#pragma unroll
        for (int q = 0; q < kPer; q++) { sh.b.bs[1 + tid * kPer + q] = excl; excl += c[q]; }   // bs lies in the mink half
        if (tid == 0) { sh.b.bs[0] = 0u; sh.b.bs[kBins + 1] = total; sh.b.bs[kBins + 2] = total; }
        break;
    }
    __syncthreads();
    {
        const unsigned total = sh.b.bs[kBins + 1];
        if (tid < 4) pk[total + tid] = make_float2(__builtin_nanf(""), __uint_as_float(0xFFFFFFFFu));   // sentinels
        unsigned st[IPT];
#pragma unroll
        for (int i = 0; i < IPT; i++) st[i] = sh.b.bs[1 + ((repmask >> i) & 1u ? bin[i] : (unsigned)kBins)];   // (non-representatives: the total, unused)
#pragma unroll
        for (int i = 0; i < IPT; i++) {
            const unsigned at = (repmask >> i) & 1u ? st[i] + rank[i] : (unsigned)N + 4u;
            pk[at] = make_float2(pr[i], __uint_as_float((unsigned)(k0 + i)));
        }
    }
    __syncthreads();
    SLR_K4_STOP_AT(4);

    // queries: the reference's predicate on the pairs of bins [b-1, b+1] (and a few neighbours), smallest column wins
    int best[IPT];
    unsigned qa[IPT], qe[IPT];                           // byte offsets into pk: first 16-byte word, end of the window
#pragma unroll
    for (int i = 0; i < IPT; i++) {
        const bool act = inrow && vl[i] && pl[i] == pl[i];
        const int b = phase_bin(pl[i]);                  // (a NaN lands in bin 0; its window is emptied below)
        const unsigned i0 = sh.b.bs[b], i1 = sh.b.bs[b + 3];
        qa[i] = (i0 & ~1u) * 8u;
        qe[i] = act ? i1 * 8u : 0u;
    }
    const char *const pkb = reinterpret_cast<const char *>(sh.b.pk2);
    // X87: the exact-difference predicate needs f64 only where the f32 difference can round.  Sterbenz: for floats with
    // c in [p / 2, 2 p] the difference p - c is exact in f32.  With |p| >= 0.25 every candidate within 0.125 of p lies in that
    // range, so its f32 difference IS the exact one and `fabsf(p - c) < 0.1f` is the x87 predicate bit for bit; a candidate
    // further away than 0.125 fails both predicates (a rounded difference of 0.125 or more never drops below 0.1).  Only pixels
    // with a phase inside (-0.25, 0.25) -- a few per frame -- need the f64 form: a wave that holds one runs it for all its lanes.
#pragma unroll
    for (int i = 0; i < IPT; i++) {
        unsigned bk = 0xFFFFFFFFu;
        const float p = pl[i];
        bool wide = false;
        if constexpr (X87) wide = __ballot(qa[i] < qe[i] && !(fabsf(p) >= 0.25f)) != 0ull;      // (wave-uniform)
        if (X87 && wide) {
            const double pd = (double)p;
            for (unsigned a = qa[i]; a < qe[i]; a += 32u) {
                const f32x4 c0 = *reinterpret_cast<const f32x4 *>(pkb + a);
                const f32x4 c1 = *reinterpret_cast<const f32x4 *>(pkb + a + 16);
                const unsigned h0 = fabs(pd - (double)c0.x) < 0.1 ? __float_as_uint(c0.y) : 0xFFFFFFFFu;
                const unsigned h1 = fabs(pd - (double)c0.z) < 0.1 ? __float_as_uint(c0.w) : 0xFFFFFFFFu;
                const unsigned h2 = fabs(pd - (double)c1.x) < 0.1 ? __float_as_uint(c1.y) : 0xFFFFFFFFu;
                const unsigned h3 = fabs(pd - (double)c1.z) < 0.1 ? __float_as_uint(c1.w) : 0xFFFFFFFFu;
                bk = min(min(bk, h0), min(h1, min(h2, h3)));
            }
        } else {
            for (unsigned a = qa[i]; a < qe[i]; a += 32u) {
                const f32x4 c0 = *reinterpret_cast<const f32x4 *>(pkb + a);
                const f32x4 c1 = *reinterpret_cast<const f32x4 *>(pkb + a + 16);
                const unsigned h0 = fabsf(p - c0.x) < 0.1f ? __float_as_uint(c0.y) : 0xFFFFFFFFu;
                const unsigned h1 = fabsf(p - c0.z) < 0.1f ? __float_as_uint(c0.w) : 0xFFFFFFFFu;
                const unsigned h2 = fabsf(p - c1.x) < 0.1f ? __float_as_uint(c1.y) : 0xFFFFFFFFu;
                const unsigned h3 = fabsf(p - c1.z) < 0.1f ? __float_as_uint(c1.w) : 0xFFFFFFFFu;
                bk = min(min(bk, h0), min(h1, min(h2, h3)));
            }
        }
        best[i] = (int)bk;                               // 0xFFFFFFFF == -1: no match
    }
    SLR_K4_STOP_AT5(best[0]);
    if (!XCHG && !inrow) return;
    if (XCHG) __syncthreads();                           // every wave is done with the index: the row's XYZ goes through its LDS

    // triangulation (mfreconstruct.cpp:297-326), branch-free for the 4 pixels
#if defined(SLR_K4_ABL) && (SLR_K4_ABL & 2)
    const f32x4 ua = {pl[0], pl[1], pl[2], pl[3]}, ub = {pr[0], pr[1], pr[2], pr[3]};
    float urx[IPT];
#pragma unroll
    for (int i = 0; i < IPT; i++) urx[i] = (float)best[i];
#else
    const size_t tk0 = trow + (inrow ? k0 : 0);          // (XCHG: threads beyond the row stay for the barriers)
    const f32x4 ua = *reinterpret_cast<const f32x4 *>(undL + tk0 / 2);
    const f32x4 ub = *(reinterpret_cast<const f32x4 *>(undL + tk0 / 2) + 1);
    float urx[IPT];
#pragma unroll
    for (int i = 0; i < IPT; i++) urx[i] = undRx[trow + (inrow && best[i] >= 0 ? best[i] : 0)];
#endif
    const float ulx[IPT] = {ua.x, ua.z, ub.x, ub.z}, uly[IPT] = {ua.y, ua.w, ub.y, ub.w};
    float out[12];
    unsigned hw = 0;
    unsigned bad = 0;                                    // pixels whose quotients need the real divisions
#pragma unroll
    for (int i = 0; i < IPT; i++) {
        // p3D = Q * [ulx, uly, ulx - urx, 1] with Q's structural zeros and ones dropped (reproject(): exact no-ops)
        const double r0 = (double)ulx[i] + kc.q3, r1 = (double)uly[i] + kc.q7, r2 = kc.q11;
        const double w = kc.q14 * (X87 ? (double)ulx[i] - (double)urx[i] : (double)(float)(ulx[i] - urx[i])) + kc.q15;   // :299
        // X = (float)(r / w), guarded by the exponents (anything unusual takes the real divisions below)
        auto expo = [](double x) -> unsigned { return ((unsigned)__double2hiint(x) >> 20) & 0x7FFu; };
        if constexpr (TAME) {
            // (round 4) q3, q7, q11 are host-checked (finite, not -0, zero or within 2^+-100): r0 and r1 are zero or within
            // 2^-152 .. 2^129 whenever the table entries are finite, and a zero numerator goes through the quotient sequence
            // exactly (+-0 with the division's sign) -- only w's exponent and the entries' finiteness need a look
            const bool fin = __builtin_isfinite(ulx[i]) && __builtin_isfinite(uly[i]);
            if (!(expo(w) - (1023u - 200u) < 400u && fin) && best[i] >= 0) bad |= 1u << i;
        } else {
            const unsigned e0 = expo(r0), e1 = expo(r1), e2 = expo(r2), e3 = expo(w);
            const unsigned emin = min(min(e0, e1), min(e2, e3)), emax = max(max(e0, e1), max(e2, e3));
            if (!(emin >= 1023u - 200u && emax < 1023u + 200u) && best[i] >= 0) bad |= 1u << i;
        }
        // The three quotients r/w exactly as three IEEE divisions compute them on this GPU: the compiler expands x/w into
        // y = rcp(w) refined by two Newton steps, q = x*y, q' = fma(fma(-w, q, x), y, q) (plus operand scaling and a fix-up
        // that are the identity inside the exponent guard) -- y depends on w alone, so it is computed once for the three.
        double y = __builtin_amdgcn_rcp(w);
        y = __builtin_fma(y, __builtin_fma(-w, y, 1.0), y);
        y = __builtin_fma(y, __builtin_fma(-w, y, 1.0), y);
        const double r[3] = {r0, r1, r2};
        float X[3];
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const double q = r[c] * y;
            const double e = __builtin_fma(-w, q, r[c]);
            X[c] = (float)__builtin_fma(e, y, q);
        }
        out[3 * i] = X[0]; out[3 * i + 1] = X[1]; out[3 * i + 2] = X[2];
    }
    if (__builtin_expect(bad != 0, 0)) {                 // exact zeros, w == 0, NaN, extreme exponents: the three real divisions
#pragma unroll 1
        for (int i = 0; i < IPT; i++) {
            if (!((bad >> i) & 1u)) continue;
            const double r0 = (double)ulx[i] + kc.q3, r1 = (double)uly[i] + kc.q7, r2 = kc.q11;
            const double w = kc.q14 * (X87 ? (double)ulx[i] - (double)urx[i] : (double)(float)(ulx[i] - urx[i])) + kc.q15;
            const float X0 = (float)(r0 / w), X1 = (float)(r1 / w), X2 = (float)(r2 / w);
#pragma unroll
            for (int j = 0; j < IPT; j++)
                if (j == i) { out[3 * j] = X0; out[3 * j + 1] = X1; out[3 * j + 2] = X2; }
        }
    }
#pragma unroll
    for (int i = 0; i < IPT; i++) {
        if constexpr (HAS_T) {                           // matCoordTrans(3x4 f32) * [X;1]: f64 accumulate, narrow once
            const double X0 = (double)out[3 * i], X1 = (double)out[3 * i + 1], X2 = (double)out[3 * i + 2];
#pragma unroll
            for (int c = 0; c < 3; c++) {
                double s = __builtin_fma(kc.T[c * 4], X0, 0.0);
                s = __builtin_fma(kc.T[c * 4 + 1], X1, s);
                s = __builtin_fma(kc.T[c * 4 + 2], X2, s);
                out[3 * i + c] = (float)(s + kc.T[c * 4 + 3]);
            }
        }
        const bool m = best[i] >= 0;
        out[3 * i] = m ? out[3 * i] : 0.0f; out[3 * i + 1] = m ? out[3 * i + 1] : 0.0f; out[3 * i + 2] = m ? out[3 * i + 2] : 0.0f;
        hw |= (m ? 1u : 0u) << (8 * i);
    }
    const size_t o = base + k0;
    f32x4 *d4 = reinterpret_cast<f32x4 *>(xyz + 3 * o);  // streaming output: non-temporal stores
    const f32x4 v0 = {out[0], out[1], out[2], out[3]}, v1 = {out[4], out[5], out[6], out[7]}, v2 = {out[8], out[9], out[10], out[11]};
#if defined(SLR_K4_ABL) && (SLR_K4_ABL & 1)
    if (out[0] + out[5] + out[10] != 1.2345e-30f) return;
#endif
    if constexpr (XCHG) {
        // the row's XYZ transposed through the dead index: a wave's store instruction then writes 1 KB of consecutive bytes
        // instead of 16 bytes per lane at a 48-byte stride
        f32x4 *const x4 = reinterpret_cast<f32x4 *>(&sh);
        x4[3 * tid] = v0; x4[3 * tid + 1] = v1; x4[3 * tid + 2] = v2;
        __syncthreads();
        f32x4 *const r4 = reinterpret_cast<f32x4 *>(xyz + 3 * base);
        const int n4 = 3 * W / 4;                        // 16-byte words of the row
#pragma unroll
        for (int q = 0; q < 3; q++)
            if (tid + q * BLOCK < n4) __builtin_nontemporal_store(x4[tid + q * BLOCK], r4 + tid + q * BLOCK);
        if (!inrow) return;
    } else {
        __builtin_nontemporal_store(v0, d4);
        __builtin_nontemporal_store(v1, d4 + 1);
        __builtin_nontemporal_store(v2, d4 + 2);
    }
    __builtin_nontemporal_store(hw, reinterpret_cast<unsigned *>(has + o));
    if (match_k) *reinterpret_cast<int4 *>(match_k + o) = make_int4(best[0], best[1], best[2], best[3]);
}

// ------------------------------------------------------------------------------------------------------
// K4, lean form, PERSISTENT (round 5; the grouped launches of slr_reconstruct_mf_batch: no valid bytes, no match columns).
// mf_match_lean_kernel<1024, ., true, ., .> with the rows of a launch walked by 2 resident workgroups per CU instead of one
// workgroup per row: with the undistortion tables coming from L2 (frames grouped) the kernel is no longer waiting for HBM but for
// its own VALU work (79 % busy) -- and for the 16 bytes per pixel of phases at the head of every row, which nothing overlapped
// inside a workgroup.  Here a workgroup issues the loads of its NEXT row's phases (8 registers) before it starts on the current
// one, so they arrive during ~13 us of index build and queries; the launch / argument-load / tail of 24000 short workgroups
// goes as well.  Same arithmetic, same LDS layout, same results (tests/test_gpu_lean.py: == the per-row form, bit for bit).
// Workgroup w runs on XCD w % 8 and takes the items q = w / 8, w / 8 + G / 8, ... of that XCD: item q = (row (q / nframes) * 8 + xcd,
// frame q % nframes) -- a row of all frames on the same XCD, close in time, as in the per-row form.
// ------------------------------------------------------------------------------------------------------
#ifdef SLR_ALL_FORMS   // (measured: 105 against 76 us per frame; SLR_OPT_MF_MATCH_ALGO 7 exists in `make FORMS=all` builds only)
template <bool HAS_T, bool TAME, bool X87>
__global__ __launch_bounds__(1024, 8) void mf_match_lean_persist_kernel(const float *__restrict__ phaseL0, const float *__restrict__ phaseR0,
                                                                        int W, int H, int row0, K4Lean kc,
                                                                        const float4 *__restrict__ undL, const float *__restrict__ undRx,
                                                                        float *__restrict__ xyz0, uint8_t *__restrict__ has0,
                                                                        int nframes, size_t frame_px)
{
    constexpr int BLOCK = 1024, IPT = 4, N = BLOCK * IPT, TS = 2 * N, kPer = kBins / BLOCK;
    constexpr unsigned kEmpty = 0xFFFFFFFFu;
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    __shared__ union {
        struct { unsigned key[TS]; unsigned mink[TS]; } t;               // phase bits -> smallest column
        struct { f32x4 pk2[N / 2 + 3]; unsigned bs[kBins + 3]; } b;      // pairs of (phi, column as bits) grouped by bin (+ sentinels, + a sink); bin index
    } sh;
    __shared__ unsigned scan_tmp[BLOCK / 64];
    static_assert(sizeof(sh.b) <= sizeof(sh.t), "the index must fit the dead hash table (two workgroups per CU)");

    const int tid = threadIdx.x, k0 = tid * IPT;
    const bool inrow = k0 < W;                           // W % 4 == 0: a thread's 4 pixels are all inside or all outside
    const int xcd = (int)blockIdx.x & 7, lw = (int)blockIdx.x >> 3, nl = (int)gridDim.x >> 3;
    const int items = ((H - xcd + 7) / 8) * nframes;     // rows xcd, xcd + 8, ... of every frame
    auto item_of = [&](int q, int &brow, size_t &fo) { brow = (q / nframes) * 8 + xcd; fo = (size_t)(q % nframes) * frame_px; };

    // pr / pl: the current row's phases.  Each is re-loaded for the NEXT row right behind its last use in the current one (pr: the
    // scatter of the pairs; pl: the queries) -- into the same registers, so the prefetch costs none (a second set of 8 spilled at
    // the 64 registers of two 1024-thread workgroups per CU)
    float pr[IPT] = {0, 0, 0, 0}, pl[IPT] = {0, 0, 0, 0};
    if (lw < items && inrow) {
        int brow; size_t fo;
        item_of(lw, brow, fo);
        load_f32_blocked<IPT>(phaseR0 + fo + (size_t)brow * W, k0, k0 + IPT, true, pr);
        load_f32_blocked<IPT>(phaseL0 + fo + (size_t)brow * W, k0, k0 + IPT, true, pl);
    }
#pragma unroll 1
    for (int q = lw; q < items; q += nl) {
        int brow; size_t fo;
        item_of(q, brow, fo);
        float *const xyz = xyz0 + 3 * fo;
        uint8_t *const has = has0 + fo;
        const int row = brow + row0;
        const size_t base = (size_t)brow * W, trow = (size_t)row * W;
        const bool more = q + nl < items && inrow;
        size_t nbase = 0;                                // the next item's row, as an offset into the phase arrays
        if (more) { int nb; size_t nfo; item_of(q + nl, nb, nfo); nbase = nfo + (size_t)nb * W; }
        {
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 e4 = {kEmpty, kEmpty, kEmpty, kEmpty};
            u32x4 *const t4 = reinterpret_cast<u32x4 *>(sh.t.key);   // (key and mink are adjacent: the whole table in 16-byte stores)
#pragma unroll
            for (int j = 0; j < 2 * TS / 4 / BLOCK; j++) t4[tid + j * BLOCK] = e4;
        }
        __syncthreads();

        // A. distinct values and their smallest column (as mf_match_lean_kernel)
        unsigned slot[IPT], bits[IPT], old[IPT];
        bool cand[IPT];
#pragma unroll
        for (int i = 0; i < IPT; i++) {
            bits[i] = __float_as_uint(pr[i]);
            const bool ok = inrow && (pr[i] == pr[i]);               // NaN (= no phase) can never satisfy the predicate
            const bool dup = i > 0 && ok && bits[i] == bits[i - 1];
            cand[i] = ok && !dup;
            slot[i] = (bits[i] * 2654435761u) >> (32 - __builtin_ctz(TS));
        }
#pragma unroll
        for (int i = 0; i < IPT; i++) old[i] = cand[i] ? atomicCAS(&sh.t.key[slot[i]], kEmpty, bits[i]) : kEmpty;
#pragma unroll
        for (int i = 0; i < IPT; i++) {
            if (cand[i]) {
                unsigned h = slot[i], o = old[i];
                while (o != kEmpty && o != bits[i]) {
                    h = (h + 1) & (TS - 1);
                    o = atomicCAS(&sh.t.key[h], kEmpty, bits[i]);
                }
                atomicMin(&sh.t.mink[h], (unsigned)(k0 + i));
                slot[i] = h;
            }
        }
        __syncthreads();
        unsigned *const cnt = sh.t.key;                      // the key half is dead: bin counters (+ one sink per thread)
        unsigned repmask = 0;
        unsigned mk4[IPT];
#pragma unroll
        for (int i = 0; i < IPT; i++) mk4[i] = sh.t.mink[slot[i]];
#pragma unroll
        for (int j = 0; j < kPer; j++) cnt[tid + j * BLOCK] = 0u;
#pragma unroll
        for (int i = 0; i < IPT; i++)
            if (cand[i] && mk4[i] == (unsigned)(k0 + i)) repmask |= 1u << i;
        __syncthreads();                                     // the whole table is dead from here on

        // B. counting sort of the representatives by phase bin
        unsigned bin[IPT], rank[IPT];
#pragma unroll
        for (int i = 0; i < IPT; i++) bin[i] = (repmask >> i) & 1u ? (unsigned)phase_bin(pr[i]) : (unsigned)(kBins + tid);
#pragma unroll
        for (int i = 0; i < IPT; i++) rank[i] = atomicAdd(&cnt[bin[i]], 1u);
        __syncthreads();
        float2 *const pk = reinterpret_cast<float2 *>(sh.b.pk2);
        {
            unsigned c[kPer], sum = 0;
#pragma unroll
            for (int j = 0; j < kPer; j++) { c[j] = cnt[tid * kPer + j]; sum += c[j]; }
            unsigned total;
            unsigned excl = wg_exclusive_scan<BLOCK>(sum, 0u, [](unsigned a, unsigned b) { return a + b; }, scan_tmp, &total);
#pragma unroll
            for (int j = 0; j < kPer; j++) { sh.b.bs[1 + tid * kPer + j] = excl; excl += c[j]; }   // bs lies in the mink half
            if (tid == 0) { sh.b.bs[0] = 0u; sh.b.bs[kBins + 1] = total; sh.b.bs[kBins + 2] = total; }
        }
        __syncthreads();
        {
            const unsigned total = sh.b.bs[kBins + 1];
            if (tid < 4) pk[total + tid] = make_float2(__builtin_nanf(""), __uint_as_float(0xFFFFFFFFu));   // sentinels
            unsigned st[IPT];
#pragma unroll
            for (int i = 0; i < IPT; i++) st[i] = sh.b.bs[1 + ((repmask >> i) & 1u ? bin[i] : (unsigned)kBins)];
#pragma unroll
            for (int i = 0; i < IPT; i++) {
                const unsigned at = (repmask >> i) & 1u ? st[i] + rank[i] : (unsigned)N + 4u;
                pk[at] = make_float2(pr[i], __uint_as_float((unsigned)(k0 + i)));
            }
        }
        if (more) load_f32_blocked<IPT>(phaseR0 + nbase, k0, k0 + IPT, true, pr);     // the next row's right phases: in flight from here
        __syncthreads();

        // queries
        int best[IPT];
        unsigned qa[IPT], qe[IPT];
#pragma unroll
        for (int i = 0; i < IPT; i++) {
            const bool act = inrow && pl[i] == pl[i];
            const int b = phase_bin(pl[i]);
            const unsigned i0 = sh.b.bs[b], i1 = sh.b.bs[b + 3];
            qa[i] = (i0 & ~1u) * 8u;
            qe[i] = act ? i1 * 8u : 0u;
        }
        const char *const pkb = reinterpret_cast<const char *>(sh.b.pk2);
#pragma unroll
        for (int i = 0; i < IPT; i++) {
            unsigned bk = 0xFFFFFFFFu;
            const float p = pl[i];
            bool wide = false;
            if constexpr (X87) wide = __ballot(qa[i] < qe[i] && !(fabsf(p) >= 0.25f)) != 0ull;      // (see mf_match_lean_kernel: Sterbenz)
            if (X87 && wide) {
                const double pd = (double)p;
                for (unsigned a = qa[i]; a < qe[i]; a += 32u) {
                    const f32x4 c0 = *reinterpret_cast<const f32x4 *>(pkb + a);
                    const f32x4 c1 = *reinterpret_cast<const f32x4 *>(pkb + a + 16);
                    const unsigned h0 = fabs(pd - (double)c0.x) < 0.1 ? __float_as_uint(c0.y) : 0xFFFFFFFFu;
                    const unsigned h1 = fabs(pd - (double)c0.z) < 0.1 ? __float_as_uint(c0.w) : 0xFFFFFFFFu;
                    const unsigned h2 = fabs(pd - (double)c1.x) < 0.1 ? __float_as_uint(c1.y) : 0xFFFFFFFFu;
                    const unsigned h3 = fabs(pd - (double)c1.z) < 0.1 ? __float_as_uint(c1.w) : 0xFFFFFFFFu;
                    bk = min(min(bk, h0), min(h1, min(h2, h3)));
                }
            } else {
                for (unsigned a = qa[i]; a < qe[i]; a += 32u) {
                    const f32x4 c0 = *reinterpret_cast<const f32x4 *>(pkb + a);
                    const f32x4 c1 = *reinterpret_cast<const f32x4 *>(pkb + a + 16);
                    const unsigned h0 = fabsf(p - c0.x) < 0.1f ? __float_as_uint(c0.y) : 0xFFFFFFFFu;
                    const unsigned h1 = fabsf(p - c0.z) < 0.1f ? __float_as_uint(c0.w) : 0xFFFFFFFFu;
                    const unsigned h2 = fabsf(p - c1.x) < 0.1f ? __float_as_uint(c1.y) : 0xFFFFFFFFu;
                    const unsigned h3 = fabsf(p - c1.z) < 0.1f ? __float_as_uint(c1.w) : 0xFFFFFFFFu;
                    bk = min(min(bk, h0), min(h1, min(h2, h3)));
                }
            }
            best[i] = (int)bk;
        }
        __syncthreads();                                     // every wave is done with the index: the row's XYZ goes through its LDS

        // triangulation (mfreconstruct.cpp:297-326), branch-free for the 4 pixels (as mf_match_lean_kernel)
        const size_t tk0 = trow + (inrow ? k0 : 0);
        const f32x4 ua = *reinterpret_cast<const f32x4 *>(undL + tk0 / 2);
        const f32x4 ub = *(reinterpret_cast<const f32x4 *>(undL + tk0 / 2) + 1);
        float urx[IPT];
#pragma unroll
        for (int i = 0; i < IPT; i++) urx[i] = undRx[trow + (inrow && best[i] >= 0 ? best[i] : 0)];
        const float ulx[IPT] = {ua.x, ua.z, ub.x, ub.z}, uly[IPT] = {ua.y, ua.w, ub.y, ub.w};
        float out[12];
        unsigned hw = 0;
        unsigned bad = 0;
#pragma unroll
        for (int i = 0; i < IPT; i++) {
            const double r0 = (double)ulx[i] + kc.q3, r1 = (double)uly[i] + kc.q7, r2 = kc.q11;
            const double w = kc.q14 * (X87 ? (double)ulx[i] - (double)urx[i] : (double)(float)(ulx[i] - urx[i])) + kc.q15;   // :299
            auto expo = [](double x) -> unsigned { return ((unsigned)__double2hiint(x) >> 20) & 0x7FFu; };
            if constexpr (TAME) {
                const bool fin = __builtin_isfinite(ulx[i]) && __builtin_isfinite(uly[i]);
                if (!(expo(w) - (1023u - 200u) < 400u && fin) && best[i] >= 0) bad |= 1u << i;
            } else {
                const unsigned e0 = expo(r0), e1 = expo(r1), e2 = expo(r2), e3 = expo(w);
                const unsigned emin = min(min(e0, e1), min(e2, e3)), emax = max(max(e0, e1), max(e2, e3));
                if (!(emin >= 1023u - 200u && emax < 1023u + 200u) && best[i] >= 0) bad |= 1u << i;
            }
            double y = __builtin_amdgcn_rcp(w);
            y = __builtin_fma(y, __builtin_fma(-w, y, 1.0), y);
            y = __builtin_fma(y, __builtin_fma(-w, y, 1.0), y);
            const double r[3] = {r0, r1, r2};
            float X[3];
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const double qq = r[c] * y;
                const double e = __builtin_fma(-w, qq, r[c]);
                X[c] = (float)__builtin_fma(e, y, qq);
            }
            out[3 * i] = X[0]; out[3 * i + 1] = X[1]; out[3 * i + 2] = X[2];
        }
        if (__builtin_expect(bad != 0, 0)) {                 // exact zeros, w == 0, NaN, extreme exponents: the three real divisions
#pragma unroll 1
            for (int i = 0; i < IPT; i++) {
                if (!((bad >> i) & 1u)) continue;
                const double r0 = (double)ulx[i] + kc.q3, r1 = (double)uly[i] + kc.q7, r2 = kc.q11;
                const double w = kc.q14 * (X87 ? (double)ulx[i] - (double)urx[i] : (double)(float)(ulx[i] - urx[i])) + kc.q15;
                const float X0 = (float)(r0 / w), X1 = (float)(r1 / w), X2 = (float)(r2 / w);
#pragma unroll
                for (int j = 0; j < IPT; j++)
                    if (j == i) { out[3 * j] = X0; out[3 * j + 1] = X1; out[3 * j + 2] = X2; }
            }
        }
#pragma unroll
        for (int i = 0; i < IPT; i++) {
            if constexpr (HAS_T) {                           // matCoordTrans(3x4 f32) * [X;1]: f64 accumulate, narrow once
                const double X0 = (double)out[3 * i], X1 = (double)out[3 * i + 1], X2 = (double)out[3 * i + 2];
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    double s_ = __builtin_fma(kc.T[c * 4], X0, 0.0);
                    s_ = __builtin_fma(kc.T[c * 4 + 1], X1, s_);
                    s_ = __builtin_fma(kc.T[c * 4 + 2], X2, s_);
                    out[3 * i + c] = (float)(s_ + kc.T[c * 4 + 3]);
                }
            }
            const bool m = best[i] >= 0;
            out[3 * i] = m ? out[3 * i] : 0.0f; out[3 * i + 1] = m ? out[3 * i + 1] : 0.0f; out[3 * i + 2] = m ? out[3 * i + 2] : 0.0f;
            hw |= (m ? 1u : 0u) << (8 * i);
        }
        const f32x4 v0 = {out[0], out[1], out[2], out[3]}, v1 = {out[4], out[5], out[6], out[7]}, v2 = {out[8], out[9], out[10], out[11]};
        {
            f32x4 *const x4 = reinterpret_cast<f32x4 *>(&sh);
            x4[3 * tid] = v0; x4[3 * tid + 1] = v1; x4[3 * tid + 2] = v2;
            if (more) load_f32_blocked<IPT>(phaseL0 + nbase, k0, k0 + IPT, true, pl);     // ... and its left phases (behind the f64 work: registers)
            __syncthreads();
            f32x4 *const r4 = reinterpret_cast<f32x4 *>(xyz + 3 * base);
            const int n4 = 3 * W / 4;                        // 16-byte words of the row
#pragma unroll
            for (int j = 0; j < 3; j++)
                if (tid + j * BLOCK < n4) __builtin_nontemporal_store(x4[tid + j * BLOCK], r4 + tid + j * BLOCK);
        }
        if (inrow) __builtin_nontemporal_store(hw, reinterpret_cast<unsigned *>(has + base + k0));
        __syncthreads();                                     // the exchange space is read: the next row clears the table
    }
}

#ifdef SLR_ALL_FORMS   // (measured, not faster than the 1024 x 4 form: SLR_OPT_MF_MATCH_ALGO 5 / 6 exist in `make FORMS=all` builds only)
#endif  // SLR_ALL_FORMS

// ------------------------------------------------------------------------------------------------------
// K4 (exact indexed form, lean, round 4): the same index as mf_match_lean_kernel -- hash dedup, representatives counting-sorted
// into 0.25-wide bins, exact predicate on the window's pairs -- cut for instruction-level instead of wave-level parallelism:
//   * 512 threads x 8 pixels (two runs of 4: columns 4t.. and 2048 + 4t.., so that a wave's global accesses keep the 1024-thread
//     form's shape): a row is 8 waves instead of 16 (half the barrier population, an 8-wave DPP scan), a thread keeps 8
//     independent LDS atomics / reads in flight where the 1024-thread form keeps 4;
//   * TS (hash slots) = 8192 -> 64 KB, two rows per CU at up to 128 VGPRs; TS = 6144 -> 48 KB, THREE rows per CU at 80 VGPRs:
//     three workgroups in different stages fill each other's barrier waits (the 1024-thread form is pinned to two by the
//     CU's 32 wave slots).  A slot is (hash >> 16) * TS >> 16; linear probing wraps by compare;
//   * the query reads the first 8 pairs of EVERY pixel's window unconditionally (4 ds_read_b128 per pixel, two pixels = 8 reads
//     in flight, no loop-carried branch between them; 8 NaN sentinels close the array) -- 95 % of the windows of a scene's rows end there;
//     a loop continues for the rest.  An inactive pixel queries with a NaN (never matches);
//   * the exponent guard of the shared-reciprocal quotients looks at w only: r0 = ulx + q3 and r1 = uly + q7 are sums of a
//     finite f32 and a host-checked double (K4Lean is only built when q3, q7, q11 are finite, not -0 and zero or within
//     2^+-100), so they are zero or within 2^-152 .. 2^129 whenever the table entries are finite -- one v_cmp_class each;
//   * the hash table is cleared with 16-byte stores.
// Bit-equal to mf_match_lean_kernel / the general forms / the oracle (tests/test_gpu_lean.py, test_gpu_fullsize.py).
// ------------------------------------------------------------------------------------------------------
template <bool HAS_T, int TS, int WPS>
__global__ __launch_bounds__(512, WPS) void mf_match_lean8_kernel(const float *__restrict__ phaseL, const uint8_t *__restrict__ validL,
                                                              const float *__restrict__ phaseR, const uint8_t *__restrict__ validR,
                                                              int W, int H, int row0, K4Lean kc, int stop,
                                                              const float4 *__restrict__ undL, const float *__restrict__ undRx,
                                                              float *__restrict__ xyz,
                                                              uint8_t *__restrict__ has, int32_t *__restrict__ match_k)
{
    constexpr int BLOCK = 512, IPT = 8;
    constexpr int N = BLOCK * IPT;                       // 4096 pixels
    constexpr int kPer = kBins / BLOCK;                  // bins per thread
    constexpr unsigned kEmpty = 0xFFFFFFFFu;             // a NaN pattern: never a candidate's phase bits
    constexpr int kSent = 8;                             // NaN pairs behind the last candidate
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    static_assert(TS >= kBins + BLOCK && TS % (2 * BLOCK) == 0 && TS <= 65536, "bin counters live in the key half; 16-byte clears");
    constexpr bool kPow2 = (TS & (TS - 1)) == 0;
    __shared__ union {
        struct { unsigned key[TS]; unsigned mink[TS]; } t;                         // phase bits -> smallest column
        struct { f32x4 pk2[(N + kSent + 2) / 2]; unsigned bs[kBins + 3]; } b;      // (phi, column) pairs by bin + sentinels + a sink; bin index
    } sh;
    __shared__ unsigned scan_tmp[BLOCK / 64];
    static_assert(sizeof(sh.b.pk2) >= (size_t)(kBins + BLOCK) * 4, "the bin index must not overlap the live bin counters");

    const int row = blockIdx.x + row0, tid = threadIdx.x;
    const size_t base = (size_t)blockIdx.x * W;
    // a thread's 8 pixels are two runs of 4: columns [4 tid, 4 tid + 4) and [2048 + 4 tid, ...) -- every global access of a wave is
    // then the 1024-thread form's (16 bytes per lane at a 16-byte stride in, 48 bytes per lane at a 48-byte stride out); 8
    // CONSECUTIVE pixels per thread cost 86 instead of 11 us of stores (96-byte lane stride)
    const int kg[2] = {4 * tid, N / 2 + 4 * tid};
    const bool ing[2] = {kg[0] < W, kg[1] < W};          // W % 4 == 0: a run is all inside or all outside
#define SLR_KCOL(i) (kg[(i) >> 2] + ((i) & 3))

    float pr[IPT], pl[IPT];
    unsigned vr[IPT], vl[IPT];
#pragma unroll
    for (int i = 0; i < IPT; i++) { pr[i] = 0.0f; pl[i] = 0.0f; vr[i] = 0u; vl[i] = 0u; }
#pragma unroll
    for (int g = 0; g < 2; g++)
        if (ing[g]) {
            load_f32_blocked<4>(phaseR + base, kg[g], kg[g] + 4, true, pr + 4 * g);
            load_valid_blocked<4>(validR, base, kg[g], kg[g] + 4, true, vr + 4 * g);
            load_f32_blocked<4>(phaseL + base, kg[g], kg[g] + 4, true, pl + 4 * g);
            load_valid_blocked<4>(validL, base, kg[g], kg[g] + 4, true, vl + 4 * g);
        }
    const size_t trow = (size_t)row * W;
    {
        const u32x4 e4 = {kEmpty, kEmpty, kEmpty, kEmpty};
        u32x4 *const t4 = reinterpret_cast<u32x4 *>(sh.t.key);
#pragma unroll
        for (int q = 0; q < 2 * TS / 4 / BLOCK; q++) t4[tid + q * BLOCK] = e4;
    }
    __syncthreads();
    SLR_K4_STOP_AT(1);

    // A. distinct values and their smallest column
    unsigned slot[IPT], old[IPT];
    unsigned candm = 0;
#pragma unroll
    for (int i = 0; i < IPT; i++) {
        const unsigned bits = __float_as_uint(pr[i]);
        const bool ok = ing[i >> 2] && vr[i] && (pr[i] == pr[i]);
        const bool dup = (i & 3) > 0 && ok && vr[i - 1] && bits == __float_as_uint(pr[i - 1]);
        candm |= (ok && !dup ? 1u : 0u) << i;
        const unsigned h = bits * 2654435761u;
        slot[i] = kPow2 ? h >> (32 - __builtin_ctz(TS)) : ((h >> 16) * (unsigned)TS) >> 16;
    }
#pragma unroll
    for (int i = 0; i < IPT; i++) old[i] = (candm >> i) & 1u ? atomicCAS(&sh.t.key[slot[i]], kEmpty, __float_as_uint(pr[i])) : kEmpty;
#pragma unroll
    for (int i = 0; i < IPT; i++) {
        if ((candm >> i) & 1u) {
            const unsigned bits = __float_as_uint(pr[i]);
            unsigned h = slot[i], o = old[i];
            while (o != kEmpty && o != bits) {
                h = kPow2 ? (h + 1) & (TS - 1) : (h + 1 == (unsigned)TS ? 0u : h + 1);
                o = atomicCAS(&sh.t.key[h], kEmpty, bits);
            }
            atomicMin(&sh.t.mink[h], (unsigned)SLR_KCOL(i));
            slot[i] = h;
        }
    }
    __syncthreads();
    unsigned *const cnt = sh.t.key;                      // the key half is dead: bin counters (+ one sink per thread)
    unsigned repmask = 0;
    {
        unsigned mk[IPT];
#pragma unroll
        for (int i = 0; i < IPT; i++) mk[i] = sh.t.mink[slot[i]];
        {
            const u32x4 z4 = {0u, 0u, 0u, 0u};
            u32x4 *const c4 = reinterpret_cast<u32x4 *>(cnt);
#pragma unroll
            for (int q = 0; q < kPer / 4; q++) c4[tid + q * BLOCK] = z4;
            cnt[kBins + tid] = 0u;
        }
#pragma unroll
        for (int i = 0; i < IPT; i++)
            if (((candm >> i) & 1u) && mk[i] == (unsigned)SLR_KCOL(i)) repmask |= 1u << i;
    }
    __syncthreads();                                     // the whole table is dead from here on
    SLR_K4_STOP_AT(2);

    // B. counting sort of the representatives by phase bin
    unsigned bin[IPT], rank[IPT];
#pragma unroll
    for (int i = 0; i < IPT; i++) bin[i] = (repmask >> i) & 1u ? (unsigned)phase_bin(pr[i]) : (unsigned)(kBins + tid);
#pragma unroll
    for (int i = 0; i < IPT; i++) rank[i] = atomicAdd(&cnt[bin[i]], 1u);
    __syncthreads();
    float2 *const pk = reinterpret_cast<float2 *>(sh.b.pk2);
    {
        unsigned c[kPer], sum = 0;
        const u32x4 *const c4 = reinterpret_cast<const u32x4 *>(cnt + tid * kPer);
#pragma unroll
        for (int q = 0; q < kPer / 4; q++) {
            const u32x4 v = c4[q];
            c[4 * q] = v.x; c[4 * q + 1] = v.y; c[4 * q + 2] = v.z; c[4 * q + 3] = v.w;
            sum += v.x + v.y + v.z + v.w;
        }
        unsigned total;
        unsigned excl = wg_exclusive_scan<BLOCK>(sum, 0u, [](unsigned a, unsigned b) { return a + b; }, scan_tmp, &total);
#pragma unroll
        for (int q = 0; q < kPer; q++) { sh.b.bs[1 + tid * kPer + q] = excl; excl += c[q]; }
        if (tid == 0) { sh.b.bs[0] = 0u; sh.b.bs[kBins + 1] = total; sh.b.bs[kBins + 2] = total; }
    }
    __syncthreads();
    {
        const unsigned total = sh.b.bs[kBins + 1];
        if (tid < kSent) pk[total + tid] = make_float2(__builtin_nanf(""), __uint_as_float(0xFFFFFFFFu));   // sentinels
        unsigned st[IPT];
#pragma unroll
        for (int i = 0; i < IPT; i++) st[i] = sh.b.bs[1 + ((repmask >> i) & 1u ? bin[i] : (unsigned)kBins)];
#pragma unroll
        for (int i = 0; i < IPT; i++) {
            const unsigned at = (repmask >> i) & 1u ? st[i] + rank[i] : (unsigned)(N + kSent);   // (the sink)
            pk[at] = make_float2(pr[i], __uint_as_float((unsigned)SLR_KCOL(i)));
        }
    }
    __syncthreads();
    SLR_K4_STOP_AT(4);

    // queries: the reference's predicate on the pairs of bins [b-1, b+1] (and a few neighbours), smallest column wins
    int best[IPT];
    unsigned qa[IPT], qe[IPT];                           // byte offsets into pk: first 16-byte word, end of the window
    float pq[IPT];
#pragma unroll
    for (int i = 0; i < IPT; i++) {
        const bool act = ing[i >> 2] && vl[i] && pl[i] == pl[i];
        const int b = phase_bin(pl[i]);                  // (a NaN lands in bin 0; it queries with a NaN and an empty window)
        const unsigned i0 = sh.b.bs[b], i1 = sh.b.bs[b + 3];
        qa[i] = (i0 & ~1u) * 8u;
        qe[i] = act ? i1 * 8u : 0u;
        pq[i] = act ? pl[i] : __builtin_nanf("");
    }
    const char *const pkb = reinterpret_cast<const char *>(sh.b.pk2);
    auto hit4 = [](float p, const f32x4 &c0, const f32x4 &c1) -> unsigned {
        const unsigned h0 = fabsf(p - c0.x) < 0.1f ? __float_as_uint(c0.y) : 0xFFFFFFFFu;
        const unsigned h1 = fabsf(p - c0.z) < 0.1f ? __float_as_uint(c0.w) : 0xFFFFFFFFu;
        const unsigned h2 = fabsf(p - c1.x) < 0.1f ? __float_as_uint(c1.y) : 0xFFFFFFFFu;
        const unsigned h3 = fabsf(p - c1.z) < 0.1f ? __float_as_uint(c1.w) : 0xFFFFFFFFu;
        return min(min(h0, h1), min(h2, h3));
    };
#pragma unroll
    for (int i = 0; i < IPT; i += 2) {                   // two pixels' first 8 pairs each: 8 independent 16-byte reads
        f32x4 c[2][4];
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int q = 0; q < 4; q++) c[j][q] = *reinterpret_cast<const f32x4 *>(pkb + qa[i + j] + 16 * q);
#pragma unroll
        for (int j = 0; j < 2; j++) {
            unsigned bk = min(hit4(pq[i + j], c[j][0], c[j][1]), hit4(pq[i + j], c[j][2], c[j][3]));
            for (unsigned a = qa[i + j] + 64u; a < qe[i + j]; a += 32u) {
                const f32x4 d0 = *reinterpret_cast<const f32x4 *>(pkb + a);
                const f32x4 d1 = *reinterpret_cast<const f32x4 *>(pkb + a + 16);
                bk = min(bk, hit4(pq[i + j], d0, d1));
            }
            best[i + j] = (int)bk;                       // 0xFFFFFFFF == -1: no match
        }
    }
    SLR_K4_STOP_AT5(best[0]);

    // triangulation (mfreconstruct.cpp:297-326), branch-free, 4 pixels at a time.  All table reads before the first store (vmcnt
    // counts stores too: a later run's loads would wait for an earlier run's stores to reach memory)
    float urx[IPT];
    f32x4 ul4[IPT / 2];
#pragma unroll
    for (int g = 0; g < IPT; g += 4) {
        const bool in = ing[g >> 2];
#pragma unroll
        for (int i = 0; i < 4; i++) urx[g + i] = undRx[trow + (in && best[g + i] >= 0 ? best[g + i] : 0)];
        const f32x4 *const u4 = reinterpret_cast<const f32x4 *>(undL + (trow + (in ? kg[g >> 2] : 0)) / 2);
        ul4[g / 2] = u4[0]; ul4[g / 2 + 1] = u4[1];
    }
#pragma unroll
    for (int g = 0; g < IPT; g += 4) {
        const f32x4 ua = ul4[g / 2], ub = ul4[g / 2 + 1];
        float out[12];
        unsigned hw = 0;
        const float ulx[4] = {ua.x, ua.z, ub.x, ub.z}, uly[4] = {ua.y, ua.w, ub.y, ub.w};
        unsigned bad = 0;                                // pixels whose quotients need the real divisions
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const double r0 = (double)ulx[i] + kc.q3, r1 = (double)uly[i] + kc.q7, r2 = kc.q11;
            const double w = kc.q14 * (double)(float)(ulx[i] - urx[g + i]) + kc.q15;
            const unsigned ew = ((unsigned)__double2hiint(w) >> 20) & 0x7FFu;
            const bool fin = __builtin_isfinite(ulx[i]) && __builtin_isfinite(uly[i]);
            if (!(ew - (1023u - 200u) < 400u && fin) && best[g + i] >= 0) bad |= 1u << i;
            double y = __builtin_amdgcn_rcp(w);          // the three IEEE quotients from one refined reciprocal (mf_match_lean_kernel)
            y = __builtin_fma(y, __builtin_fma(-w, y, 1.0), y);
            y = __builtin_fma(y, __builtin_fma(-w, y, 1.0), y);
            const double r[3] = {r0, r1, r2};
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const double q = r[c] * y;
                const double e = __builtin_fma(-w, q, r[c]);
                out[3 * i + c] = (float)__builtin_fma(e, y, q);
            }
        }
#if defined(SLR_K4_ABL) && (SLR_K4_ABL & 4)
        bad = 0;
#endif
        if (__builtin_expect(bad != 0, 0)) {             // w == 0, NaN, extreme exponents, non-finite table entries: the real divisions
#pragma unroll 1
            for (int i = 0; i < 4; i++) {
                if (!((bad >> i) & 1u)) continue;
                float ux = 0.0f, uy = 0.0f, ur = 0.0f;
#pragma unroll
                for (int j = 0; j < 4; j++)
                    if (j == i) { ux = ulx[j]; uy = uly[j]; ur = urx[g + j]; }
                const double r0 = (double)ux + kc.q3, r1 = (double)uy + kc.q7, r2 = kc.q11;
                const double w = kc.q14 * (double)(float)(ux - ur) + kc.q15;
                const float X0 = (float)(r0 / w), X1 = (float)(r1 / w), X2 = (float)(r2 / w);
#pragma unroll
                for (int j = 0; j < 4; j++)
                    if (j == i) { out[3 * j] = X0; out[3 * j + 1] = X1; out[3 * j + 2] = X2; }
            }
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            if constexpr (HAS_T) {                       // matCoordTrans(3x4 f32) * [X;1]: f64 accumulate, narrow once
                const double X0 = (double)out[3 * i], X1 = (double)out[3 * i + 1], X2 = (double)out[3 * i + 2];
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    double s = __builtin_fma(kc.T[c * 4], X0, 0.0);
                    s = __builtin_fma(kc.T[c * 4 + 1], X1, s);
                    s = __builtin_fma(kc.T[c * 4 + 2], X2, s);
                    out[3 * i + c] = (float)(s + kc.T[c * 4 + 3]);
                }
            }
            const bool m = best[g + i] >= 0;
            out[3 * i] = m ? out[3 * i] : 0.0f; out[3 * i + 1] = m ? out[3 * i + 1] : 0.0f; out[3 * i + 2] = m ? out[3 * i + 2] : 0.0f;
            hw |= (m ? 1u : 0u) << (8 * i);
        }
#if defined(SLR_K4_ABL) && (SLR_K4_ABL & 1)
        if (out[0] + out[5] + out[10] != 1.2345e-30f) continue;
#endif
        if (!ing[g >> 2]) continue;
        const size_t o = base + kg[g >> 2];
        f32x4 *d4 = reinterpret_cast<f32x4 *>(xyz + 3 * o);  // streaming output: non-temporal stores
        const f32x4 v0 = {out[0], out[1], out[2], out[3]}, v1 = {out[4], out[5], out[6], out[7]}, v2 = {out[8], out[9], out[10], out[11]};
        __builtin_nontemporal_store(v0, d4);
        __builtin_nontemporal_store(v1, d4 + 1);
        __builtin_nontemporal_store(v2, d4 + 2);
        __builtin_nontemporal_store(hw, reinterpret_cast<unsigned *>(has + o));
        if (match_k) *reinterpret_cast<int4 *>(match_k + o) = make_int4(best[g], best[g + 1], best[g + 2], best[g + 3]);
    }
#undef SLR_KCOL
}

#endif  // SLR_ALL_FORMS

// K4 for rows wider than 4096 pixels.  The hash table of a whole 8192-pixel right row needs 128 KB of LDS (one 1024-thread
// workgroup per CU, 8 pixels and too many registers per thread: 2.3x the time per pixel of a 4096-wide row).  "Smallest
// column k with |phiL - phiR[k]| < 0.1" decomposes over CHUNKS of the right row: the answer lies in the first chunk
// (ascending columns) that holds any candidate.  So the workgroup keeps the 4096-pixel geometry (64 KB table, two
// workgroups per CU, 4 pixels per thread): for every right chunk, in ascending order, it builds the binned index exactly
// as mf_match_binned_kernel does, and every left chunk queries it for the pixels that have no match yet.  Between right
// chunks a thread parks the state of its 4 pixels of a left chunk (u16 each: the match column, 0xFFFF = still open,
// 0xFFFE = can never match) in the first 8 bytes of its own 48-byte span of the XYZ output row, which it overwrites with
// the result after the last right chunk -- nothing but best[] would otherwise have to live across the index builds, and
// the kernel sits at the 64-VGPR limit of two 1024-thread workgroups per CU.  W <= 32768 (columns fit 15 bits).
__global__ __launch_bounds__(1024, 8) void mf_match_chunked_kernel(const float *__restrict__ phaseL, const uint8_t *__restrict__ validL,
                                                                   const float *__restrict__ phaseR, const uint8_t *__restrict__ validR,
                                                                   int W, int H, int row0, DevCalib cal, int vec_ok,
                                                                   const float2 *__restrict__ undL, const float *__restrict__ undRx,
                                                                   float *__restrict__ xyz,
                                                                   uint8_t *__restrict__ has, int32_t *__restrict__ match_k,
                                                                   const int *__restrict__ defer /* null, or defer[0] rows listed behind it: only those */)
{
    constexpr int BLOCK = 1024, IPT = 4, N = BLOCK * IPT;
    constexpr int TS = 2 * N;
    constexpr int kPer = kBins / BLOCK;
    constexpr unsigned kEmpty = 0xFFFFFFFFu;
    constexpr unsigned kOpen = 0xFFFFu, kNever = 0xFFFEu;
    if (defer && (int)blockIdx.x >= defer[0]) return;    // (the rows mf_match_wide_kernel left to this kernel: usually none)
    const int brow = defer ? defer[1 + blockIdx.x] : (int)blockIdx.x;
    __shared__ union {
        struct { unsigned key[TS]; unsigned mink[TS]; } t;
        struct { float2 pk[N]; unsigned binstart[kBins + 1]; } b;
    } sh;
    __shared__ unsigned scan_tmp[BLOCK / 64];

    const int row = brow + row0, tid = threadIdx.x;
    const size_t base = (size_t)brow * W;
    const bool vec = (vec_ok & 1) != 0;
    const int nch = (W + N - 1) / N;

#pragma unroll 1
    for (int rc = 0; rc < nch; rc++) {
        const int k0 = rc * N + tid * IPT;               // right columns of this thread in this chunk
        {
            float pr[IPT];
            unsigned vr[IPT];
            load_f32_blocked<IPT>(phaseR + base, k0, W, vec, pr);
            load_valid_blocked<IPT>(validR, base, k0, W, vec, vr);
#pragma unroll
            for (int q = 0; q < 2 * IPT; q++) { sh.t.key[tid + q * BLOCK] = kEmpty; sh.t.mink[tid + q * BLOCK] = kEmpty; }
            __syncthreads();
            // A. distinct values of the chunk and their smallest column
            unsigned slot[IPT];
            bool prev_ok = false;
            unsigned prev_bits = 0;
#pragma unroll
            for (int i = 0; i < IPT; i++) {
                const bool ok = (k0 + i < W) && vr[i] && (pr[i] == pr[i]);
                const unsigned bits = __float_as_uint(pr[i]);
                const bool dup = prev_ok && ok && bits == prev_bits;
                slot[i] = kEmpty;
                if (ok && !dup) {
                    unsigned h = (bits * 2654435761u) >> (32 - __builtin_ctz(TS));
                    for (;;) {
                        const unsigned old = atomicCAS(&sh.t.key[h], kEmpty, bits);
                        if (old == kEmpty || old == bits) break;
                        h = (h + 1) & (TS - 1);
                    }
                    atomicMin(&sh.t.mink[h], (unsigned)(k0 + i));
                    slot[i] = h;
                }
                prev_ok = ok; prev_bits = bits;
            }
            __syncthreads();
            unsigned repmask = 0;
#pragma unroll
            for (int i = 0; i < IPT; i++)
                if (slot[i] != kEmpty && sh.t.mink[slot[i]] == (unsigned)(k0 + i)) repmask |= 1u << i;
            __syncthreads();
            // B. counting sort of the representatives by phase bin
#pragma unroll
            for (int q = 0; q < kPer; q++) sh.b.binstart[tid + q * BLOCK] = 0u;
            __syncthreads();
#pragma unroll
            for (int i = 0; i < IPT; i++) {
                if (repmask & (1u << i)) {
                    const unsigned b = (unsigned)phase_bin(pr[i]);
                    slot[i] = (b << 16) | atomicAdd(&sh.b.binstart[b], 1u);
                }
            }
            __syncthreads();
            {
                unsigned c[kPer], sum = 0;
#pragma unroll
                for (int q = 0; q < kPer; q++) { c[q] = sh.b.binstart[tid * kPer + q]; sum += c[q]; }
                unsigned total;
                unsigned excl = wg_exclusive_scan<BLOCK>(sum, 0u, [](unsigned a, unsigned b) { return a + b; }, scan_tmp, &total);
#pragma unroll
                for (int q = 0; q < kPer; q++) { sh.b.binstart[tid * kPer + q] = excl; excl += c[q]; }
                if (tid == 0) sh.b.binstart[kBins] = total;
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < IPT; i++)
                if (repmask & (1u << i))
                    sh.b.pk[sh.b.binstart[slot[i] >> 16] + (slot[i] & 0xFFFFu)] = make_float2(pr[i], __uint_as_float((unsigned)(k0 + i)));
            __syncthreads();
        }
        // every left chunk queries this index for its still open pixels
#pragma unroll 1
        for (int lc = 0; lc < nch; lc++) {
            const int j0 = lc * N + tid * IPT;
            if (j0 >= W) continue;                       // (per thread; no barrier inside this loop)
            unsigned *park = reinterpret_cast<unsigned *>(xyz + 3 * (base + j0));   // 2 dwords of this thread's own span
            float pl[IPT];
            load_f32_blocked<IPT>(phaseL + base, j0, W, vec, pl);
            unsigned st[IPT];
            if (rc == 0) {
                unsigned vl[IPT];
                load_valid_blocked<IPT>(validL, base, j0, W, vec, vl);
#pragma unroll
                for (int i = 0; i < IPT; i++) st[i] = (j0 + i < W && vl[i] && pl[i] == pl[i]) ? kOpen : kNever;
            } else {
                const unsigned a = park[0], b = park[1];
                st[0] = a & 0xFFFFu; st[1] = a >> 16; st[2] = b & 0xFFFFu; st[3] = b >> 16;
            }
            int qi0[IPT], qi1[IPT];
#pragma unroll
            for (int i = 0; i < IPT; i++) {
                const bool act = st[i] == kOpen;
                const int b = act ? phase_bin(pl[i]) : 0;
                qi0[i] = (int)sh.b.binstart[b > 0 ? b - 1 : 0];
                qi1[i] = act ? (int)sh.b.binstart[b + 2 < kBins ? b + 2 : kBins] : 0;
            }
#pragma unroll
            for (int i = 0; i < IPT; i++) {
                unsigned bk = 0xFFFFFFFFu;
                for (int idx = qi0[i]; idx < qi1[i]; idx += 4) {
                    float2 c[4];
#pragma unroll
                    for (int q = 0; q < 4; q++) c[q] = sh.b.pk[idx + q < N ? idx + q : N - 1];
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const bool hit = idx + q < qi1[i] && phase_match(pl[i], c[q].x, cal.eval_x87);
                        const unsigned kk = hit ? __float_as_uint(c[q].y) : 0xFFFFFFFFu;
                        bk = kk < bk ? kk : bk;
                    }
                }
                if (bk != 0xFFFFFFFFu) st[i] = bk;
            }
            if (rc + 1 < nch) {
                park[0] = st[0] | st[1] << 16;
                park[1] = st[2] | st[3] << 16;
            } else {
                int best[IPT];
#pragma unroll
                for (int i = 0; i < IPT; i++) best[i] = st[i] < kNever ? (int)st[i] : -1;
                // (an empty asm makes T opaque per chunk: without it the compiler hoists the twelve f32->f64 conversions
                // of apply_T out of the loops and then spills them)
                float Tl[12];
#pragma unroll
                for (int i = 0; i < 12; i++) { Tl[i] = cal.T[i]; asm volatile("" : "+s"(Tl[i])); }
                k4_emit<IPT>(best, base, j0, row, W, vec, cal, undL, undRx, xyz, has, match_k, Tl);
            }
        }
        __syncthreads();                                 // the index is rebuilt for the next right chunk
    }
}

// ------------------------------------------------------------------------------------------------------
// K4 for aligned rows of 4097 .. 8192 pixels (round 5; BASELINE config 5's 8192-pixel rows).  The chunked kernel above builds
// the hash + bin index twice and queries it four times per 8192-pixel row, with the general kernel's code: 0.85 ms per
// 8192 x 6000 frame.  This one keeps the WHOLE right row in one index that fits half a CU's LDS:
//   * no hash: config 5's phases come from atan2 of a DFT bin -- continuous values, practically all distinct -- so "one
//     representative per distinct value" buys nothing there; every valid right pixel (minus a pixel whose left neighbour has the
//     same bits) goes into the counting sort by 0.25-wide bin.  Duplicates are merely more pairs in a window: the query takes
//     the smallest column over everything that satisfies the reference's predicate, exactly as before;
//   * the (phase, column) pairs of 8192 pixels are 64 KB, the bin index is 16-bit (offsets <= 8192): 72 KB, two 1024-thread
//     workgroups per CU; the 32-bit bin counters of the build live in the pairs' space (dead before the scatter);
//   * a thread owns two runs of 4 pixels, columns 4t.. and 4096 + 4t.., so that every global access keeps the 1024 x 4 form's
//     shape (16 bytes per lane at a 16- or 48-byte stride: see the store-stride note at mf_match_lean_kernel);
//   * window query, NaN sentinels, K4Lean constants and the shared-reciprocal quotients as in mf_match_lean_kernel.
// mfreconstruct.cpp:272-334.  Bit-equal to the sweep, the chunked kernel and the oracle (tests/test_gpu_parity.py, W = 5000 / 8192).
// ------------------------------------------------------------------------------------------------------
template <bool HAS_T, bool X87>
__device__ __forceinline__ void k4_lean_tri4(const int best[4], bool inrun, size_t trow, int k0, const K4Lean &kc,
                                             const float4 *__restrict__ undL, const float *__restrict__ undRx, float out[12], unsigned &hw)
{
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const size_t tk0 = trow + (inrun ? k0 : 0);
    const f32x4 ua = *reinterpret_cast<const f32x4 *>(undL + tk0 / 2);
    const f32x4 ub = *(reinterpret_cast<const f32x4 *>(undL + tk0 / 2) + 1);
    float urx[4];
#pragma unroll
    for (int i = 0; i < 4; i++) urx[i] = undRx[trow + (inrun && best[i] >= 0 ? best[i] : 0)];
    const float ulx[4] = {ua.x, ua.z, ub.x, ub.z}, uly[4] = {ua.y, ua.w, ub.y, ub.w};
    hw = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const double r0 = (double)ulx[i] + kc.q3, r1 = (double)uly[i] + kc.q7, r2 = kc.q11;
        const double w = kc.q14 * (X87 ? (double)ulx[i] - (double)urx[i] : (double)(float)(ulx[i] - urx[i])) + kc.q15;   // :299
        auto expo = [](double x) -> unsigned { return ((unsigned)__double2hiint(x) >> 20) & 0x7FFu; };
        const unsigned e0 = expo(r0), e1 = expo(r1), e2 = expo(r2), e3 = expo(w);
        const unsigned emin = min(min(e0, e1), min(e2, e3)), emax = max(max(e0, e1), max(e2, e3));
        float X[3];
        if (emin >= 1023u - 200u && emax < 1023u + 200u) {   // (see mf_match_lean_kernel: the compiler's own division sequence, y shared)
            double y = __builtin_amdgcn_rcp(w);
            y = __builtin_fma(y, __builtin_fma(-w, y, 1.0), y);
            y = __builtin_fma(y, __builtin_fma(-w, y, 1.0), y);
            const double r[3] = {r0, r1, r2};
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const double q = r[c] * y;
                const double e = __builtin_fma(-w, q, r[c]);
                X[c] = (float)__builtin_fma(e, y, q);
            }
        } else {
            X[0] = (float)(r0 / w); X[1] = (float)(r1 / w); X[2] = (float)(r2 / w);
        }
        if constexpr (HAS_T) {                           // matCoordTrans(3x4 f32) * [X;1]: f64 accumulate, narrow once
            const double X0 = (double)X[0], X1 = (double)X[1], X2 = (double)X[2];
#pragma unroll
            for (int c = 0; c < 3; c++) {
                double s = __builtin_fma(kc.T[c * 4], X0, 0.0);
                s = __builtin_fma(kc.T[c * 4 + 1], X1, s);
                s = __builtin_fma(kc.T[c * 4 + 2], X2, s);
                X[c] = (float)(s + kc.T[c * 4 + 3]);
            }
        }
        const bool m = best[i] >= 0;
        out[3 * i] = m ? X[0] : 0.0f; out[3 * i + 1] = m ? X[1] : 0.0f; out[3 * i + 2] = m ? X[2] : 0.0f;
        hw |= (m ? 1u : 0u) << (8 * i);
    }
}

constexpr unsigned kWideHeavyBin = 96;                   // pairs in one 0.25-wide bin beyond which mf_match_wide_kernel hands the row over
template <bool HAS_T, bool X87>
__global__ __launch_bounds__(1024, 8) void mf_match_wide_kernel(const float *__restrict__ phaseL, const uint8_t *__restrict__ validL,
                                                                const float *__restrict__ phaseR, const uint8_t *__restrict__ validR,
                                                                int W, int H, int row0, K4Lean kc,
                                                                const float4 *__restrict__ undL, const float *__restrict__ undRx,
                                                                float *__restrict__ xyz, uint8_t *__restrict__ has, int32_t *__restrict__ match_k,
                                                                int *__restrict__ defer /* null, or [0] = count (zeroed), [1..] = rows left to the chunked kernel */)
{
    constexpr int BLOCK = 1024, HALF = 4096, N = 2 * HALF;
    constexpr int kPer = kBins / BLOCK;
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    __shared__ union {
        f32x4 pk2[N / 2 + 4];                            // pairs of (phi, column as bits) grouped by bin, + 4 sentinels, + a sink for non-candidates
        unsigned cnt[kBins + BLOCK];                     // the build's bin counters (+ one sink per thread): dead before the scatter
    } sh;
    __shared__ unsigned short bs[kBins + 4];             // bs[0] = 0, bs[1 + b] = first pair of bin b, two copies of the total behind the last bin
    __shared__ unsigned scan_tmp[BLOCK / 64];
    static_assert(sizeof(sh) + sizeof(bs) + sizeof(scan_tmp) <= 80 * 1024, "two workgroups per CU");

    const int row = (int)blockIdx.x + row0, tid = (int)threadIdx.x;   // absolute image row (row0: first row of a band)
    const size_t base = (size_t)blockIdx.x * W, trow = (size_t)row * W;
    const int k0[2] = {4 * tid, HALF + 4 * tid};
    const bool in[2] = {k0[0] < W, k0[1] < W};           // W % 4 == 0: a run is whole inside or outside the row

    // the right row
    float pr[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    unsigned vr[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
    for (int r = 0; r < 2; r++)
        if (in[r]) {
            load_f32_blocked<4>(phaseR + base, k0[r], k0[r] + 4, true, pr[r]);
            load_valid_blocked<4>(validR, base, k0[r], k0[r] + 4, true, vr[r]);
        }
#pragma unroll
    for (int q = 0; q < (kBins + BLOCK) / BLOCK; q++) sh.cnt[tid + q * BLOCK] = 0u;
    __syncthreads();
    unsigned bin[2][4], rank[2][4];
    unsigned cmask = 0;
#pragma unroll
    for (int r = 0; r < 2; r++)
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const bool ok = in[r] && vr[r][i] && (pr[r][i] == pr[r][i]);            // NaN can never satisfy the predicate
            // the same value one column to the left: that pixel has the smaller column and the same predicate outcome
            const bool dup = i > 0 && ok && vr[r][i - 1] && __float_as_uint(pr[r][i]) == __float_as_uint(pr[r][i - 1]);
            const bool cand = ok && !dup;
            cmask |= (cand ? 1u : 0u) << (4 * r + i);
            bin[r][i] = cand ? (unsigned)phase_bin(pr[r][i]) : (unsigned)(kBins + tid);
        }
#pragma unroll
    for (int r = 0; r < 2; r++)
#pragma unroll
        for (int i = 0; i < 4; i++) rank[r][i] = atomicAdd(&sh.cnt[bin[r][i]], 1u);
    __syncthreads();
    {
        unsigned c[kPer], sum = 0;
#pragma unroll
        for (int q = 0; q < kPer; q++) {
            c[q] = sh.cnt[tid * kPer + q]; sum += c[q];
            if (c[q] > kWideHeavyBin) sum += 0x10000u;   // (a row's counts stay below 2^14: the number of overfull bins rides above them)
        }
        unsigned total;
        unsigned excl = wg_exclusive_scan<BLOCK>(sum, 0u, [](unsigned a, unsigned b) { return a + b; }, scan_tmp, &total);   // (its barrier: every counter has been read)
        // This index has no dedup of equal phases: a flat or saturated row would put thousands of pairs into one bin and every query
        // of that window would scan them all -- O(W^2) per row.  Such a row is left to the chunked kernel (hash dedup,
        // O(W)), which the host launches behind this one on the listed rows; workgroup-uniform, nothing of the row is written here.
        if (defer && (total >> 16) != 0u) {
            if (tid == 0) defer[1 + atomicAdd(defer, 1)] = (int)blockIdx.x;
            return;
        }
        excl &= 0xFFFFu; total &= 0xFFFFu;
#pragma unroll
        for (int q = 0; q < kPer; q++) { bs[1 + tid * kPer + q] = (unsigned short)excl; excl += c[q]; }
        if (tid == 0) { bs[0] = 0; bs[kBins + 1] = (unsigned short)total; bs[kBins + 2] = (unsigned short)total; }
    }
    __syncthreads();                                     // the counters are dead: their space becomes the pairs
    float2 *const pk = reinterpret_cast<float2 *>(sh.pk2);
    {
        const unsigned total = bs[kBins + 1];
        if (tid < 4) pk[total + tid] = make_float2(__builtin_nanf(""), __uint_as_float(0xFFFFFFFFu));   // sentinels
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const bool cand = ((cmask >> (4 * r + i)) & 1u) != 0;
                const unsigned at = cand ? (unsigned)bs[1 + bin[r][i]] + rank[r][i] : (unsigned)N + 4u + (unsigned)(tid & 3);
                pk[at] = make_float2(pr[r][i], __uint_as_float((unsigned)(k0[r] + i)));
            }
    }
    __syncthreads();

    // queries + triangulation, one run of 4 left pixels after the other
    const char *const pkb = reinterpret_cast<const char *>(sh.pk2);
#pragma unroll 1
    for (int r = 0; r < 2; r++) {
        const int kr = r * HALF + 4 * tid;               // (not k0[r]: a run-time index would put the array into scratch)
        if (kr >= W) continue;                           // (no barrier behind this point)
        float pl[4];
        unsigned vl[4];
        load_f32_blocked<4>(phaseL + base, kr, kr + 4, true, pl);
        load_valid_blocked<4>(validL, base, kr, kr + 4, true, vl);
        int best[4];
        unsigned qa[4], qe[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const bool act = vl[i] && pl[i] == pl[i];
            const int b = phase_bin(pl[i]);              // (a NaN lands in bin 0; its window is emptied below)
            const unsigned i0 = bs[b], i1 = bs[b + 3];
            qa[i] = (i0 & ~1u) * 8u;
            qe[i] = act ? i1 * 8u : 0u;
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            unsigned bk = 0xFFFFFFFFu;
            const float p = pl[i];
            bool wide = false;
            if constexpr (X87) wide = __ballot(qa[i] < qe[i] && !(fabsf(p) >= 0.25f)) != 0ull;      // (see mf_match_lean_kernel: Sterbenz)
            if (X87 && wide) {
                const double pd = (double)p;
                for (unsigned a = qa[i]; a < qe[i]; a += 32u) {
                    const f32x4 c0 = *reinterpret_cast<const f32x4 *>(pkb + a);
                    const f32x4 c1 = *reinterpret_cast<const f32x4 *>(pkb + a + 16);
                    const unsigned h0 = fabs(pd - (double)c0.x) < 0.1 ? __float_as_uint(c0.y) : 0xFFFFFFFFu;
                    const unsigned h1 = fabs(pd - (double)c0.z) < 0.1 ? __float_as_uint(c0.w) : 0xFFFFFFFFu;
                    const unsigned h2 = fabs(pd - (double)c1.x) < 0.1 ? __float_as_uint(c1.y) : 0xFFFFFFFFu;
                    const unsigned h3 = fabs(pd - (double)c1.z) < 0.1 ? __float_as_uint(c1.w) : 0xFFFFFFFFu;
                    bk = min(min(bk, h0), min(h1, min(h2, h3)));
                }
            } else {
                for (unsigned a = qa[i]; a < qe[i]; a += 32u) {
                    const f32x4 c0 = *reinterpret_cast<const f32x4 *>(pkb + a);
                    const f32x4 c1 = *reinterpret_cast<const f32x4 *>(pkb + a + 16);
                    const unsigned h0 = fabsf(p - c0.x) < 0.1f ? __float_as_uint(c0.y) : 0xFFFFFFFFu;
                    const unsigned h1 = fabsf(p - c0.z) < 0.1f ? __float_as_uint(c0.w) : 0xFFFFFFFFu;
                    const unsigned h2 = fabsf(p - c1.x) < 0.1f ? __float_as_uint(c1.y) : 0xFFFFFFFFu;
                    const unsigned h3 = fabsf(p - c1.z) < 0.1f ? __float_as_uint(c1.w) : 0xFFFFFFFFu;
                    bk = min(min(bk, h0), min(h1, min(h2, h3)));
                }
            }
            best[i] = (int)bk;                           // 0xFFFFFFFF == -1: no match
        }
        float out[12];
        unsigned hw;
        k4_lean_tri4<HAS_T, X87>(best, true, trow, kr, kc, undL, undRx, out, hw);
        const size_t o = base + kr;
        f32x4 *d4 = reinterpret_cast<f32x4 *>(xyz + 3 * o);  // streaming output: non-temporal stores
        const f32x4 v0 = {out[0], out[1], out[2], out[3]}, v1 = {out[4], out[5], out[6], out[7]}, v2 = {out[8], out[9], out[10], out[11]};
        __builtin_nontemporal_store(v0, d4);
        __builtin_nontemporal_store(v1, d4 + 1);
        __builtin_nontemporal_store(v2, d4 + 2);
        __builtin_nontemporal_store(hw, reinterpret_cast<unsigned *>(has + o));
        if (match_k) *reinterpret_cast<int4 *>(match_k + o) = make_int4(best[0], best[1], best[2], best[3]);
    }
}

template <int BLOCK, int IPT>
__global__ __launch_bounds__(BLOCK, (BLOCK == 1024 ? 8 : 4)) void mf_match_sorted_kernel(const float *__restrict__ phaseL, const uint8_t *__restrict__ validL,
                                                              const float *__restrict__ phaseR, const uint8_t *__restrict__ validR,
                                                              int W, int H, int row0, DevCalib cal, int vec_ok,
                                                              const float2 *__restrict__ undL, const float *__restrict__ undRx,
                                                              float *__restrict__ xyz,
                                                              uint8_t *__restrict__ has, int32_t *__restrict__ match_k)
{
    constexpr int N = BLOCK * IPT;
    typedef rocprim::block_radix_sort<unsigned, BLOCK, IPT> Sort;  // keys only: (group << 16) | k
    __shared__ union {
        typename Sort::storage_type sort;
        struct { float2 pk[N]; } d;                      // distinct values: (phi, min column as bits)
    } sh;
    __shared__ float phR[N];                             // right-row phases, gathered by column after the sort
    __shared__ int scan_tmp[BLOCK / 64];
    __shared__ unsigned last_key[BLOCK];
    __shared__ int n_distinct;
    __shared__ unsigned short binfirst[kBins + 1];
    __shared__ unsigned chunk_min[BLOCK];
    __shared__ unsigned scanu_tmp[BLOCK / 64];

    const int row = blockIdx.x + row0, tid = threadIdx.x;   // absolute image row (row0: first row of a band)
    const size_t base = (size_t)blockIdx.x * W;
    const int k0 = tid * IPT;
    const bool vec = (vec_ok & 1) != 0;
    const int stop = vec_ok >> 8;                        // debug: leave after phase N (SLR_DEBUG_K4_STOP), 0 = run all

    // all row loads of this thread are issued here, before the sort, and are independent of each other
    float pr[IPT], pl[IPT];
    unsigned vr[IPT], vl[IPT];
    load_f32_blocked<IPT>(phaseR + base, k0, W, vec, pr);
    load_valid_blocked<IPT>(validR, base, k0, W, vec, vr);
    load_f32_blocked<IPT>(phaseL + base, k0, W, vec, pl);
    load_valid_blocked<IPT>(validL, base, k0, W, vec, vl);

    // Sort key: any function of phi works for correctness (equal phases share a key, and the stable order keeps
    // ascending k inside a key).  16 bits = 12-bit bin + 4-bit hash of the value: two 8-bit radix passes instead
    // of four, and the result is already grouped by bin for the candidate index below.  Distinct values that
    // collide on a key merely produce a few extra run heads.
    unsigned keys[IPT];
#pragma unroll
    for (int i = 0; i < IPT; i++) {                      // blocked arrangement: stable order == ascending k
        const bool ok = (k0 + i < W) && vr[i] && (pr[i] == pr[i]);   // NaN can never satisfy the predicate
        const unsigned b = __float_as_uint(pr[i]);
        const unsigned h4 = (b ^ (b >> 4) ^ (b >> 8) ^ (b >> 12) ^ (b >> 16) ^ (b >> 20)) & 0xFu;
        keys[i] = ok ? ((((unsigned)phase_bin(pr[i]) << 4) | h4) << 16) | (unsigned)(k0 + i) : 0xFFFFFFFFu;
        phR[k0 + i] = pr[i];
    }
    SLR_K4_STOP_AT(1);
    Sort().sort(keys, sh.sort, 16, 32);
    __syncthreads();                                     // phR visible; everybody is done with sh.sort
    SLR_K4_STOP_AT(2);
    unsigned phb[IPT];                                   // phase bits of the sorted items (0xFFFFFFFF = none)
#pragma unroll
    for (int i = 0; i < IPT; i++) phb[i] = keys[i] != 0xFFFFFFFFu ? __float_as_uint(phR[keys[i] & 0xFFFFu]) : 0xFFFFFFFFu;
    last_key[tid] = phb[IPT - 1];
    __syncthreads();
    unsigned prev = tid ? last_key[tid - 1] : 0xFFFFFFFFu;
    int heads = 0;
    unsigned headmask = 0;
#pragma unroll
    for (int i = 0; i < IPT; i++) {
        const bool h = phb[i] != 0xFFFFFFFFu && phb[i] != prev;      // run head: value differs from its predecessor
        prev = phb[i];
        headmask |= h ? (1u << i) : 0u;
        heads += h ? 1 : 0;
    }
    int total;
    int pos = wg_exclusive_scan<BLOCK>(heads, 0, [](int a, int b) { return a + b; }, scan_tmp, &total);
#pragma unroll
    for (int i = 0; i < IPT; i++)
        if (headmask & (1u << i)) { sh.d.pk[pos] = make_float2(__uint_as_float(phb[i]), __uint_as_float(keys[i] & 0xFFFFu)); pos++; }
    if (tid == 0) n_distinct = total;
    for (int b = tid; b <= kBins; b += BLOCK) binfirst[b] = 0xFFFFu;
    __syncthreads();
    const int nd = n_distinct;
    SLR_K4_STOP_AT(3);

    // Bin index over the distinct values: bin(phi) = clamp(floor((phi + 512) * 4)) is monotone and two values
    // closer than 0.1001 land at most one bin apart, so bins [b-1, b+1] of phiL hold a superset of its
    // candidates.  binfirst[b] = first distinct index whose bin is >= b (suffix-min fill), binfirst[kBins] = nd.
    for (int i = tid; i < nd; i += BLOCK) {
        const int b = phase_bin(sh.d.pk[i].x);
        if (i == 0 || phase_bin(sh.d.pk[i - 1].x) != b) binfirst[b] = (unsigned short)i;
    }
    __syncthreads();
    {
        constexpr int kPer = kBins / BLOCK;              // bins per thread (kBins is a multiple of BLOCK)
        unsigned m = 0xFFFFu;
#pragma unroll
        for (int q = kPer - 1; q >= 0; q--) { const unsigned v = binfirst[tid * kPer + q]; m = v < m ? v : m; }
        chunk_min[tid] = m;                              // min over this thread's bins
        __syncthreads();
        // exclusive suffix-min over threads = exclusive prefix-min in reversed thread order
        // (0xFFFF is the identity here: the bins' first indices are 16-bit)
        unsigned carry = wg_exclusive_scan<BLOCK>(chunk_min[BLOCK - 1 - tid], 0xFFFFu, [](unsigned a, unsigned b) { return a < b ? a : b; },
                                                  scanu_tmp, (unsigned *)nullptr);
        __syncthreads();
        chunk_min[BLOCK - 1 - tid] = carry;
        __syncthreads();
        carry = chunk_min[tid];
        carry = carry < (unsigned)nd ? carry : (unsigned)nd;     // nothing later -> end sentinel
#pragma unroll
        for (int q = kPer - 1; q >= 0; q--) {
            const unsigned v = binfirst[tid * kPer + q];
            carry = v < carry ? v : carry;
            binfirst[tid * kPer + q] = (unsigned short)carry;
        }
        if (tid == 0) binfirst[kBins] = (unsigned short)nd;
    }
    __syncthreads();

    SLR_K4_STOP_AT(4);
    // queries: exact reference predicate over the <= 3-bin candidate window, smallest column wins
    int best[IPT];
    int qi0[IPT], qi1[IPT];
#pragma unroll
    for (int i = 0; i < IPT; i++) {                      // all window bounds first: independent LDS reads
        const bool act = k0 + i < W && vl[i] && pl[i] == pl[i];
        const int b = act ? phase_bin(pl[i]) : 0;
        qi0[i] = binfirst[b > 0 ? b - 1 : 0];
        qi1[i] = act ? (int)binfirst[b + 2 < kBins ? b + 2 : kBins] : 0;
    }
#pragma unroll
    for (int i = 0; i < IPT; i++) {
        unsigned bk = 0xFFFFFFFFu;
        for (int idx = qi0[i]; idx < qi1[i]; idx += 4) {     // 4 candidates per round trip (8-byte (phi, k) pairs)
            float2 c[4];
#pragma unroll
            for (int q = 0; q < 4; q++) c[q] = sh.d.pk[idx + q < N ? idx + q : N - 1];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const bool hit = idx + q < qi1[i] && phase_match(pl[i], c[q].x, cal.eval_x87);
                const unsigned kk = hit ? __float_as_uint(c[q].y) : 0xFFFFFFFFu;
                bk = kk < bk ? kk : bk;
            }
        }
        best[i] = bk == 0xFFFFFFFFu ? -1 : (int)bk;
    }

    SLR_K4_STOP_AT5(best[0]);
    k4_emit<IPT>(best, base, k0, row, W, vec, cal, undL, undRx, xyz, has, match_k);
}


// Utilities::undistortPoints depends only on the pixel position and the camera, so its 5 f64 fixed-point
// iterations (utilities.cpp:83-91) are evaluated once per (calibration, image size) into tables -- bit-identical
// values, ~550 f64-heavy instructions per matched pixel removed from K4.  left: (x,y); right: x only (:299).
__global__ __launch_bounds__(256) void undistort_table_kernel(DevCalib cal, int W, int H, float2 *__restrict__ undL,
                                                              float *__restrict__ undRx)
{
    const unsigned total = (unsigned)W * (unsigned)H;
    for (unsigned g = blockIdx.x * 256u + threadIdx.x; g < total; g += gridDim.x * 256u) {
        const unsigned row = g / W, col = g - row * W;
        float x, y, rx, ry;
        undistort_point((float)col, (float)row, cal.cam[0], x, y);
        undistort_point((float)col, (float)row, cal.cam[1], rx, ry);
        undL[g] = make_float2(x, y);
        undRx[g] = rx;
    }
}

hipError_t launch_undistort_tables(const DevCalib &cal, int W, int H, float *undL_xy, float *undRx, hipStream_t s)
{
    const size_t n = (size_t)W * H;
    const unsigned blocks = (unsigned)((n + 255) / 256);
    SLR_LAUNCH(undistort_table_kernel, dim3(blocks), dim3(256), 0, s, cal, W, H, (float2 *)undL_xy, undRx);
    return hipGetLastError();
}

// algo: 0 = auto (lean binned form when it applies, else the general binned form, the chunked form for wide rows, the sweep
// beyond), 1 = sweep, 2 = sorted indexed form, 3 = general binned indexed form
// can launch_mf_match take several frames in one launch (the lean 1024-thread kernel on aligned rows of 2049..4096 pixels)?
bool mf_match_batches_frames(const float *phaseL, const float *phaseR, const float *xyz, const uint8_t *has, int W, const DevCalib &cal,
                             int algo, const float *undL_xy, const float *undRx, size_t frame_px)
{
    return (algo == 0 || algo == 4 || algo == 7 || algo == 8 || algo == 9) && undL_xy && undRx && cal.q_simple && W > 2048 && W <= 4096 && W % 4 == 0 &&
           (uintptr_t)undL_xy % 16 == 0 && (uintptr_t)phaseL % 16 == 0 && (uintptr_t)phaseR % 16 == 0 && (uintptr_t)xyz % 16 == 0 &&
           (uintptr_t)has % 4 == 0 && frame_px % 4 == 0;
}

hipError_t launch_mf_match(const float *phaseL, const uint8_t *validL, const float *phaseR, const uint8_t *validR,
                           int W, int H, int row0, const DevCalib &cal, float *xyz, uint8_t *has, int32_t *match_k,
                           int algo, const float *undL_xy, const float *undRx, hipStream_t s, int nframes, size_t frame_px, int *defer)
{
    if (nframes > 1 && (validL || validR || match_k || !mf_match_batches_frames(phaseL, phaseR, xyz, has, W, cal, algo, undL_xy, undRx, frame_px)))
        return hipErrorInvalidValue;                     // (the caller asks mf_match_batches_frames first)
    const float2 *undL = (const float2 *)undL_xy;
    if (algo != 1 && W <= 256 * 32) {
#ifdef SLR_DEBUG_HOOKS
        const int k4_stop = tl_debug.k4_stop;                // phase ablation (profiles/k4_stages.sh): outputs are NOT written
#else
        const int k4_stop = 0;
#endif
        const int vec_ok = (k4_stop << 8) | (int)((W % 4 == 0) && ((uintptr_t)phaseL % 16 == 0) && ((uintptr_t)phaseR % 16 == 0) &&
                           ((uintptr_t)validL % 4 == 0) && ((uintptr_t)validR % 4 == 0) && ((uintptr_t)xyz % 16 == 0) &&
                           ((uintptr_t)has % 4 == 0) && (!match_k || (uintptr_t)match_k % 16 == 0));
    // (the sorted form, algo 2, is measured-dominated by the binned one: compiled with -DSLR_ALL_FORMS only)
#ifdef SLR_ALL_FORMS
#define SLR_SORTED(BLOCK, IPT)                                                                                     \
    do {                                                                                                           \
        if (algo == 2)                                                                                             \
            SLR_LAUNCH((mf_match_sorted_kernel<BLOCK, IPT>), dim3(H), dim3(BLOCK), 0, s, phaseL, validL,   \
                               phaseR, validR, W, H, row0, cal, vec_ok, undL, undRx, xyz, has, match_k);           \
        else                                                                                                       \
            SLR_LAUNCH((mf_match_binned_kernel<BLOCK, IPT>), dim3(H), dim3(BLOCK), 0, s, phaseL, validL,   \
                               phaseR, validR, W, H, row0, cal, vec_ok, undL, undRx, xyz, has, match_k);           \
    } while (0)
#else
#define SLR_SORTED(BLOCK, IPT)                                                                                     \
    SLR_LAUNCH((mf_match_binned_kernel<BLOCK, IPT>), dim3(H), dim3(BLOCK), 0, s, phaseL, validL, phaseR, validR, W, H, row0, cal, \
               vec_ok, undL, undRx, xyz, has, match_k)
#endif
        // the usual call (aligned rows of 513..1024 or 2049..4096 pixels, tables, stereoRectify's Q): the lean kernel
        if ((algo == 0 || (algo >= 4 && algo <= 9)) && (vec_ok & 1) && undL && undRx && cal.q_simple && ((W > 512 && W <= 1024) || (W > 2048 && W <= 4096)) &&
            (uintptr_t)undL % 16 == 0 && ((size_t)W * sizeof(float2)) % 16 == 0) {
            K4Lean kc;
            kc.q3 = cal.Q[3]; kc.q7 = cal.Q[7]; kc.q11 = cal.Q[11]; kc.q14 = cal.Q[14]; kc.q15 = cal.Q[15];
            for (int i = 0; i < 12; i++) kc.T[i] = (double)cal.T[i];
            const float4 *undL4 = (const float4 *)undL_xy;
            const unsigned grid = nframes > 1 ? (unsigned)(((H + 7) / 8) * 8 * nframes) : (unsigned)H;   // (frames batched: whole groups of 8 rows)
            const bool x87 = cal.eval_x87 != 0;              // SLR_OPT_EVAL_MODEL = 1: the same kernels, predicate and disparity of the x87 binary
#define SLR_LEAN1(BLOCK, T_, X_) SLR_LAUNCH((mf_match_lean_kernel<BLOCK, T_, false, false, X_>), dim3(grid), dim3(BLOCK), 0, s, phaseL, validL, phaseR, validR, \
                                  W, H, row0, kc, k4_stop, undL4, undRx, xyz, has, match_k, nframes, frame_px)
#define SLR_LEAN(BLOCK)                                                                                                          \
    do {                                                                                                                         \
        if (cal.has_T) { if (x87) SLR_LEAN1(BLOCK, true, true); else SLR_LEAN1(BLOCK, true, false); }                             \
        else { if (x87) SLR_LEAN1(BLOCK, false, true); else SLR_LEAN1(BLOCK, false, false); }                                     \
    } while (0)
            // round 4, rows of 2049..4096 pixels: algo 0 = 1024 threads x 4 pixels with the row's XYZ stored through LDS (whole
            // kilobytes per store instruction: 86 vs 91 us in the batch); 4 = the same without the exchange (round 2); 5 / 6 = 512
            // threads x 8 pixels with two / three rows per CU (measured: 98 / 94 us, not faster; their quotient guard needs a tame Q)
            auto tame = [](double q, bool zero_ok) {
                return (q == 0.0 && zero_ok && !signbit(q)) || (isfinite(q) && fabs(q) >= 0x1p-100 && fabs(q) <= 0x1p100);
            };
            const bool q_tame = tame(kc.q3, true) && tame(kc.q7, true) && tame(kc.q11, false);
#ifdef SLR_ALL_FORMS
            if (W > 2048 && (algo == 5 || algo == 6) && q_tame && !x87) {
#define SLR_LEAN8(TS, WPS)                                                                                                       \
    do {                                                                                                                         \
        if (cal.has_T) SLR_LAUNCH((mf_match_lean8_kernel<true, TS, WPS>), dim3(H), dim3(512), 0, s, phaseL, validL, phaseR, validR, \
                                  W, H, row0, kc, k4_stop, undL4, undRx, xyz, has, match_k);                                     \
        else SLR_LAUNCH((mf_match_lean8_kernel<false, TS, WPS>), dim3(H), dim3(512), 0, s, phaseL, validL, phaseR, validR,        \
                        W, H, row0, kc, k4_stop, undL4, undRx, xyz, has, match_k);                                               \
    } while (0)
                if (algo == 5) SLR_LEAN8(8192, 4); else SLR_LEAN8(6144, 6);
#undef SLR_LEAN8
                return hipGetLastError();
            }
#endif
#ifdef SLR_ALL_FORMS
            if (W > 2048 && algo == 7 && nframes > 1 && !validL && !validR && !match_k && k4_stop == 0) {
                // round 5: the grouped launch as a PERSISTENT kernel (two resident workgroups per CU walk the rows, the next row's
                // phases prefetched).  Measured (profiles/exp/r05/k4_persist_ab.txt): 105 us per frame against 77 for the per-row
                // launch -- SLR_OPT_MF_MATCH_ALGO = 7 only, never picked by auto
                static DevSlots cus_of;
                int dev = 0;
                (void)hipGetDevice(&dev);
                int cus = cus_of.get(dev);
                if (!cus) {
                    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 8) cus = 256;
                    cus_of.put(dev, cus);
                }
                const long long items = (long long)((H + 7) / 8) * 8 * nframes;
                unsigned pg = (unsigned)(2 * cus) & ~7u;          // two 1024-thread workgroups per CU, whole groups of 8 (XCDs)
                if ((long long)pg > items) pg = (unsigned)((items + 7) / 8 * 8);
#define SLR_LEANP(T_, TAME_, X_) SLR_LAUNCH((mf_match_lean_persist_kernel<T_, TAME_, X_>), dim3(pg), dim3(1024), 0, s, phaseL, phaseR, W, H, row0, kc, \
                                            undL4, undRx, xyz, has, nframes, frame_px)
#define SLR_LEANP2(T_, TAME_) do { if (x87) SLR_LEANP(T_, TAME_, true); else SLR_LEANP(T_, TAME_, false); } while (0)
                if (cal.has_T) { if (q_tame) SLR_LEANP2(true, true); else SLR_LEANP2(true, false); }
                else { if (q_tame) SLR_LEANP2(false, true); else SLR_LEANP2(false, false); }
#undef SLR_LEANP2
#undef SLR_LEANP
                return hipGetLastError();
            }
#endif
            if (W > 2048 && algo != 4) {                     // 1024 x 4 with the row's XYZ stored through LDS (round 4)
                // round 6: SLR_OPT_MF_MATCH_ALGO 9 (FORMS=all builds) = the index WITHOUT the hash dedup (heavy rows rebuilt with it inside the
                // kernel).  Measured on the bench's frames: 90 us against 81 with the dedup -- the reference's integer-quotient atan
                // quantises the phases so hard that a row of 4064 valid pixels holds ~1500 distinct values, some of them 100-300 times
                // (profiles/exp/r06/k4_nohash.txt): every row is "heavy", the dedup is doing real work.  Kept for phases that are not
                // quantised like that; never picked by auto.
#ifdef SLR_ALL_FORMS
                const bool nohash = algo == 9;
#else
                constexpr bool nohash = false;
#endif
#define SLR_LEANX3(T_, TAME_, X_, N_) SLR_LAUNCH((mf_match_lean_kernel<1024, T_, true, TAME_, X_, N_>), dim3(grid), dim3(1024), 0, s, phaseL, validL, phaseR, validR, \
                                      W, H, row0, kc, k4_stop, undL4, undRx, xyz, has, match_k, nframes, frame_px)
#ifdef SLR_ALL_FORMS
#define SLR_LEANX2(T_, TAME_, X_) do { if (nohash) SLR_LEANX3(T_, TAME_, X_, true); else SLR_LEANX3(T_, TAME_, X_, false); } while (0)
#else
#define SLR_LEANX2(T_, TAME_, X_) do { (void)nohash; SLR_LEANX3(T_, TAME_, X_, false); } while (0)
#endif
#define SLR_LEANX(T_, TAME_) do { if (x87) SLR_LEANX2(T_, TAME_, true); else SLR_LEANX2(T_, TAME_, false); } while (0)
                if (cal.has_T) { if (q_tame) SLR_LEANX(true, true); else SLR_LEANX(true, false); }
                else { if (q_tame) SLR_LEANX(false, true); else SLR_LEANX(false, false); }
#undef SLR_LEANX
#undef SLR_LEANX2
#undef SLR_LEANX3
                return hipGetLastError();
            }
            if (W <= 1024) SLR_LEAN(256); else SLR_LEAN(1024);
#undef SLR_LEAN
#undef SLR_LEAN1
            return hipGetLastError();
        }
        if (nframes > 1) return hipErrorInvalidValue;
        // wide rows: 1024 threads x few pixels each -> 16 waves per row hide the serial LDS chains of a thread
        if (W <= 256) SLR_SORTED(256, 1);
        else if (W <= 512) SLR_SORTED(256, 2);
        else if (W <= 1024) SLR_SORTED(256, 4);
        else if (W <= 2048) SLR_SORTED(1024, 2);
        else if (W <= 4096) SLR_SORTED(1024, 4);
#ifdef SLR_ALL_FORMS
        else if (algo == 2) SLR_SORTED(1024, 8);
#endif
        else if ((algo == 0) && (vec_ok & 1) && undL && undRx && cal.q_simple && W <= 8192 && (uintptr_t)undL % 16 == 0 &&
                 ((size_t)W * sizeof(float2)) % 16 == 0) {   // rows of 4097 .. 8192 pixels (config 5): the whole right row in one index (round 5)
            K4Lean kc;
            kc.q3 = cal.Q[3]; kc.q7 = cal.Q[7]; kc.q11 = cal.Q[11]; kc.q14 = cal.Q[14]; kc.q15 = cal.Q[15];
            for (int i = 0; i < 12; i++) kc.T[i] = (double)cal.T[i];
            const float4 *undL4 = (const float4 *)undL_xy;
#define SLR_WIDE(T_, X_) SLR_LAUNCH((mf_match_wide_kernel<T_, X_>), dim3(H), dim3(1024), 0, s, phaseL, validL, phaseR, validR, W, H, row0, kc, undL4, undRx, \
                                    xyz, has, match_k, defer)
            if (defer) { const hipError_t e = hipMemsetAsync(defer, 0, sizeof(int), s); if (e != hipSuccess) return e; }
            if (cal.has_T) { if (cal.eval_x87) SLR_WIDE(true, true); else SLR_WIDE(true, false); }
            else { if (cal.eval_x87) SLR_WIDE(false, true); else SLR_WIDE(false, false); }
#undef SLR_WIDE
            if (defer)                                   // the rows with an overfull bin (flat / saturated regions), usually none: workgroups beyond the count leave at once
                SLR_LAUNCH(mf_match_chunked_kernel, dim3(H), dim3(1024), 0, s, phaseL, validL, phaseR, validR, W, H, row0,
                                   cal, vec_ok, undL, undRx, xyz, has, match_k, (const int *)defer);
        }
        else                                                 // wider rows: the right row in chunks of 4096 columns
            SLR_LAUNCH(mf_match_chunked_kernel, dim3(H), dim3(1024), 0, s, phaseL, validL, phaseR, validR, W, H, row0,
                               cal, vec_ok, undL, undRx, xyz, has, match_k, (const int *)nullptr);
#undef SLR_SORTED
        return hipGetLastError();
    }
    if (algo != 1 && W <= 4096 * 8) {                       // up to 32768 columns: up to 8 chunks
        const int vec_ok = (int)((W % 4 == 0) && ((uintptr_t)phaseL % 16 == 0) && ((uintptr_t)phaseR % 16 == 0) &&
                           ((uintptr_t)validL % 4 == 0) && ((uintptr_t)validR % 4 == 0) && ((uintptr_t)xyz % 16 == 0) &&
                           ((uintptr_t)has % 4 == 0) && (!match_k || (uintptr_t)match_k % 16 == 0));
        SLR_LAUNCH(mf_match_chunked_kernel, dim3(H), dim3(1024), 0, s, phaseL, validL, phaseR, validR, W, H, row0,
                           cal, vec_ok, undL, undRx, xyz, has, match_k, (const int *)nullptr);
        return hipGetLastError();
    }
    const size_t lds = (size_t)((W + 3) & ~3) * sizeof(float);
    SLR_LAUNCH(mf_match_kernel, dim3(H), dim3(256), lds, s, phaseL, validL, phaseR, validR, W, H, row0, cal,
                       xyz, has, match_k);
    return hipGetLastError();
}

constexpr uint8_t kGeDeferred = 0xFE;             // has[] marker: "row left to the general K5 kernel" (never a result value)

// ------------------------------------------------------------------------------------------------------
// K5: one workgroup per row.
//   1. keys (code<<16 | k) of the valid right pixels are radix-sorted in LDS on the code bits (stable) -> for every
//      code an ascending list of columns, with a direct list-head table first[code].
//   2. The reference carries `kstart` from match to match (reconstruct.cpp:556,561,604): pixel j searches its code's
//      list from the column of the LAST match of any earlier pixel -- a sequential dependency along the row.  It is
//      resolved as a fixed point over the whole row at once: thread t owns IPT consecutive left pixels and walks
//      them sequentially (exact, given its incoming kstart ks_in(t)); ks_in(t) is the exclusive prefix max of the
//      threads' last matches (one block scan).  Round 1 speculates ks_in = 0.  kstart only ever grows, and a thread's
//      results stay valid as long as its new ks_in does not pass its FIRST match, so on the usual monotone rows
//      nothing is redone: one walk + one scan.  Thread t is final after at most t+1 rounds; a fixed point equals the
//      sequential answer (induction over t), which the adversarial tests check against the oracle.
// ------------------------------------------------------------------------------------------------------
template <int IPT>
__global__ __launch_bounds__(256) void ge_match_kernel(const int32_t *__restrict__ codeL, const uint8_t *__restrict__ validL,
                                                       const int32_t *__restrict__ codeR, const uint8_t *__restrict__ validR,
                                                       int W, int H, int TC, DevCalib cal,
                                                       const uint8_t *__restrict__ whiteL, const uint8_t *__restrict__ whiteR,
                                                       float *__restrict__ xyz, uint8_t *__restrict__ has,
                                                       uint8_t *__restrict__ color, int32_t *__restrict__ match_k,
                                                       int deferred_only)
{
    // deferred_only: run behind ge_match_lean_kernel, for the rows it marked (has[first pixel of the row] == kGeDeferred)
    if (deferred_only && has[(size_t)blockIdx.x * W] != kGeDeferred) return;
    constexpr int N = 256 * IPT;
    typedef rocprim::block_radix_sort<unsigned, 256, IPT> Sort;
    __shared__ union { typename Sort::storage_type sort; unsigned S[N]; } sh;
    __shared__ short mk[N];                        // match column per left pixel (-1 = none), written by the walk
    __shared__ int scan_tmp[256 / 64];
    extern __shared__ unsigned short first[];      // TC list heads: first[code] = index in S of the code's smallest
    unsigned *S = sh.S;                            // column (0xFFFF = none)
    const int row = blockIdx.x;
    const size_t base = (size_t)row * W;
    // keys (code << 16 | k) in blocked order (= ascending k); a stable radix sort on the 16 code bits alone keeps the
    // columns of every code ascending
    unsigned keys[IPT];
#pragma unroll
    for (int i = 0; i < IPT; i++) {
        const int k = threadIdx.x * IPT + i;
        // (validR == null: the Gray decode was launched without a valid plane, code -1 marks the invalid pixels)
        const int cr = k < W ? codeR[base + k] : -1;
        // codes are Gray-decoded projector columns: [0, 65535] (SLR_MAX_GRAY_BITS = 16, include/slr.h).  Anything else cannot
        // be packed next to the column and is treated as "no code" instead of aliasing another one.
        keys[i] = (k < W && (validR ? validR[base + k] != 0 : true) && (unsigned)cr <= 0xFFFFu) ? (((unsigned)cr << 16) | (unsigned)k) : 0xFFFFFFFFu;
    }
    for (int t = threadIdx.x; t < TC; t += 256) first[t] = 0xFFFFu;
    Sort().sort(keys, sh.sort, 16, 32);
    __syncthreads();                               // everybody is done with sh.sort before S aliases it
#pragma unroll
    for (int i = 0; i < IPT; i++) S[threadIdx.x * IPT + i] = keys[i];
    __syncthreads();
    for (int i = threadIdx.x; i < N; i += 256) {
        const unsigned v = S[i];
        if (v != 0xFFFFFFFFu && (v >> 16) < (unsigned)TC && (i == 0 || (S[i - 1] >> 16) != (v >> 16)))
            first[v >> 16] = (unsigned short)i;
    }
    __syncthreads();

    // smallest column >= ks in code c's list, or -1
    auto find = [&](int c, int ks) -> int {
        const unsigned target = ((unsigned)c << 16) | (unsigned)ks;
        int lo = 0, hi = N;                        // first index with S[idx] >= target
        if (c < TC) {
            // direct list head, then a short forward scan over the code's ascending columns; only long lists
            // (degenerate rows) fall back to the binary search, restricted to [lo, N)
            const unsigned f = first[c];
            lo = f == 0xFFFFu ? N : (int)f;
            int steps = 0;
            while (lo < N && S[lo] < target && steps < 6) { lo++; steps++; }
            if (lo >= N || S[lo] >= target) hi = lo;
        }
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (S[mid] < target) lo = mid + 1; else hi = mid;
        }
        if (lo < N) {
            const unsigned v = S[lo];
            if ((v >> 16) == (unsigned)c) return (int)(v & 0xFFFFu);
        }
        return -1;
    };

    const int j0 = threadIdx.x * IPT;
    int cl[IPT];                                   // this thread's left codes (-1 = no code)
#pragma unroll
    for (int i = 0; i < IPT; i++) {
        const int j = j0 + i;
        cl[i] = (j < W && (!validL || validL[base + j])) ? codeL[base + j] : -1;   // (null validL: invalid pixels hold -1)
        if ((unsigned)cl[i] > 0xFFFFu) cl[i] = -1;                                // outside [0, 65535]: no code (see the keys above)
    }
    int ks_in = 0, lm = -1, fm = 0x7FFFFFFF;       // incoming kstart, last and first match of this thread
    bool dirty = true;
    for (;;) {
        if (dirty) {
            int ks = ks_in;                        // reconstruct.cpp:556 / :604
            lm = -1; fm = 0x7FFFFFFF;
#pragma unroll
            for (int i = 0; i < IPT; i++) {
                int m = -1;
                if (cl[i] >= 0) {
                    m = find(cl[i], ks);
                    if (m >= 0) { ks = m; lm = m; if (fm == 0x7FFFFFFF) fm = m; }
                }
                if (j0 + i < W) mk[j0 + i] = (short)m;     // W <= 32768: columns fit 15 bits, -1 = no match
            }
        }
        const int exc = wg_exclusive_scan<256>(lm, -1, [](int a, int b) { return a > b ? a : b; }, scan_tmp, (int *)nullptr);
        const int new_in = exc > 0 ? exc : 0;
        dirty = new_in > fm;                       // kstart passed this thread's first match: its walk must be redone
        ks_in = new_in > ks_in ? new_in : ks_in;
        if (!__syncthreads_or(dirty ? 1 : 0)) break;
    }
    __syncthreads();
    // all four waves: reprojection + coalesced stores
    for (int j = threadIdx.x; j < W; j += 256) {
        const int m = mk[j];
        float X[3] = {0.0f, 0.0f, 0.0f};
        int col = 0;
        if (m >= 0) {
            reproject(cal.Q, cal.q_simple, (double)j, (double)row, (double)(j - m), X);   // reconstruct.cpp:570-582
            if (cal.has_T) apply_T(cal.T, X);
            if (color) col = ((int)whiteL[base + j] + (int)whiteR[base + m]) / 2;   // :598
        }
        float *o = xyz + 3 * (base + j);
        o[0] = X[0]; o[1] = X[1]; o[2] = X[2];
        has[base + j] = m >= 0 ? 1 : 0;
        if (color) color[base + j] = (uint8_t)col;
        if (match_k) match_k[base + j] = m;
    }
}

// ------------------------------------------------------------------------------------------------------
// K5, lean form for the usual call (aligned rows of 2049..4096 pixels, codes below 8192, Q of cv::stereoRectify's pattern): the same
// walk over per-code ascending column lists, but the lists come from a COUNTING sort in LDS instead of a block radix sort --
// histogram with returning atomics (arrival rank), one scan over the 8192 counters, scatter, and an insertion sort of every
// list by the thread that owns its code (a camera sees a projector column in a few pixels of a row: lists of 1-4 entries) --
// on 1024 threads x 4 pixels (two workgroups per CU, 32 waves) instead of 256 x 16 (12 waves per CU).  Rows this form is not
// made for -- a code >= 8192 or a list longer than 32 columns -- are marked in has[] and done by ge_match_kernel in a second
// launch (deferred_only).  The triangulation is K4-lean's: one refined reciprocal for the three quotients, fma form of T.
// Reference: Reconstruct::triangulation_ge Duke/reconstruct.cpp:555-611.
// ------------------------------------------------------------------------------------------------------
template <bool HAS_T, int TC>                          // TC: the list-head table's codes (8192, or 4096 when the caller's codes are known to fit)
__global__ __launch_bounds__(1024, 8) void ge_match_lean_kernel(const int32_t *__restrict__ codeL, const uint8_t *__restrict__ validL,
                                                                const int32_t *__restrict__ codeR, const uint8_t *__restrict__ validR,
                                                                int W, int H, K4Lean kc,
                                                                const uint8_t *__restrict__ whiteL, const uint8_t *__restrict__ whiteR,
                                                                float *__restrict__ xyz, uint8_t *__restrict__ has,
                                                                uint8_t *__restrict__ color, int32_t *__restrict__ match_k)
{
    constexpr int BLOCK = 1024, IPT = 4, N = BLOCK * IPT, kPer = TC / BLOCK, kMaxList = 32;
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    __shared__ union {
        struct { unsigned start[TC + BLOCK + 1]; unsigned short S[N]; } l;
        f32x4 x4[3 * BLOCK];                               // the row's XYZ on its way out (round 4: whole kilobytes per store instruction)
    } sh;
    unsigned *const start = sh.l.start;                    // counters, then list starts (start[TC] = total); [TC + 1 + tid]: atomics' sinks
    unsigned short *const S = sh.l.S;                      // the lists: columns grouped by code, ascending inside a code
    __shared__ union { unsigned u[BLOCK / 64]; int i[BLOCK / 64]; } scan_tmp;

    const int row = blockIdx.x, tid = threadIdx.x;
    const size_t base = (size_t)row * W;
    const int k0 = tid * IPT;
    const bool inrow = k0 < W;                             // W % 4 == 0

    int cr[IPT] = {-1, -1, -1, -1}, cl[IPT] = {-1, -1, -1, -1};
    if (inrow) {
        const i32x4 a = *reinterpret_cast<const i32x4 *>(codeR + base + k0), b = *reinterpret_cast<const i32x4 *>(codeL + base + k0);
        unsigned va = 0x01010101u, vb = 0x01010101u;       // (null valid: the decode wrote code -1 for invalid pixels)
        if (validR) va = *reinterpret_cast<const unsigned *>(validR + base + k0);
        if (validL) vb = *reinterpret_cast<const unsigned *>(validL + base + k0);
#pragma unroll
        for (int i = 0; i < IPT; i++) {
            const int x = i == 0 ? a.x : i == 1 ? a.y : i == 2 ? a.z : a.w, y = i == 0 ? b.x : i == 1 ? b.y : i == 2 ? b.z : b.w;
            // codes outside [0, 65535] cannot be packed next to a column by the general kernel and are "no code" there: same here
            cr[i] = ((va >> (8 * i)) & 0xFFu) && (unsigned)x <= 0xFFFFu ? x : -1;
            cl[i] = ((vb >> (8 * i)) & 0xFFu) && (unsigned)y <= 0xFFFFu ? y : -1;
        }
    }
    bool defer = false;
#pragma unroll
    for (int i = 0; i < IPT; i++) defer = defer || cr[i] >= TC || cl[i] >= TC;
#pragma unroll
    for (int q = 0; q < kPer; q++) start[tid + q * BLOCK] = 0u;
    if (__syncthreads_or(defer ? 1 : 0)) {                 // (also orders the zeroing before the atomics)
        if (tid == 0) has[base] = kGeDeferred;
        return;
    }
    // counting sort of the right row's columns by code
    unsigned rank[IPT];
#pragma unroll
    for (int i = 0; i < IPT; i++) rank[i] = atomicAdd(&start[cr[i] >= 0 ? cr[i] : TC + 1 + tid], 1u);
    __syncthreads();
    unsigned c[kPer], excl, total;
    {
        unsigned sum = 0, longest = 0;
#pragma unroll
        for (int q = 0; q < kPer; q++) { c[q] = start[tid * kPer + q]; sum += c[q]; longest = c[q] > longest ? c[q] : longest; }
        excl = wg_exclusive_scan<BLOCK>(sum, 0u, [](unsigned a, unsigned b) { return a + b; }, scan_tmp.u, &total);
        unsigned e = excl;
#pragma unroll
        for (int q = 0; q < kPer; q++) { start[tid * kPer + q] = e; e += c[q]; }
        if (tid == 0) start[TC] = total;
        defer = longest > (unsigned)kMaxList;
    }
    if (__syncthreads_or(defer ? 1 : 0)) {
        if (tid == 0) has[base] = kGeDeferred;
        return;
    }
#pragma unroll
    for (int i = 0; i < IPT; i++)
        if (cr[i] >= 0) S[start[cr[i]] + rank[i]] = (unsigned short)(k0 + i);
    __syncthreads();
    {   // ascending columns inside every list (the arrival order of the atomics is arbitrary).  A camera sees a projector column in
        // 1-4 pixels of a row: the thread reads the first four entries of each of its 8 lists at once (independent LDS reads,
        // one latency instead of a chain of dependent ones per list: the insertion sort was 22 of the kernel's 87 us), sorts
        // them in registers with 0xFFFF behind the list's end, and writes back what it changed; longer lists take the loop.
        unsigned lo[kPer], e[kPer][4];
        {
            unsigned at = excl;
#pragma unroll
            for (int q = 0; q < kPer; q++) { lo[q] = at; at += c[q]; }
        }
#pragma unroll
        for (int q = 0; q < kPer; q++)
#pragma unroll
            for (int i = 0; i < 4; i++) e[q][i] = (unsigned)i < c[q] ? (unsigned)S[lo[q] + i < (unsigned)N ? lo[q] + i : (unsigned)N - 1u] : 0xFFFFu;
#pragma unroll
        for (int q = 0; q < kPer; q++) {
            if (c[q] < 2u) continue;
            if (c[q] <= 4u) {
                unsigned a0 = e[q][0], a1 = e[q][1], a2 = e[q][2], a3 = e[q][3];
                const auto cx = [](unsigned &x, unsigned &y) { const unsigned lo_ = x < y ? x : y, hi_ = x < y ? y : x; x = lo_; y = hi_; };
                cx(a0, a1); cx(a2, a3); cx(a0, a2); cx(a1, a3); cx(a1, a2);
                S[lo[q]] = (unsigned short)a0; S[lo[q] + 1] = (unsigned short)a1;
                if (c[q] > 2u) S[lo[q] + 2] = (unsigned short)a2;
                if (c[q] > 3u) S[lo[q] + 3] = (unsigned short)a3;
            } else {
                for (unsigned i = lo[q] + 1; i < lo[q] + c[q]; i++) {
                    const unsigned short v = S[i];
                    unsigned j = i;
                    while (j > lo[q] && S[j - 1] > v) { S[j] = S[j - 1]; j--; }
                    S[j] = v;
                }
            }
        }
    }
    __syncthreads();

    // smallest column >= ks in code cc's list, or -1
    auto find = [&](int cc, int ks) -> int {
        unsigned lo = start[cc];
        const unsigned hi = start[cc + 1];
        while (lo < hi && (int)S[lo] < ks) lo++;
        return lo < hi ? (int)S[lo] : -1;
    };
    int m[IPT];
    int ks_in = 0, lm = -1, fm = 0x7FFFFFFF;               // incoming kstart, last and first match of this thread
    bool dirty = true;
    for (;;) {                                             // the fixed point of ge_match_kernel (see there)
        if (dirty) {
            int ks = ks_in;                                // reconstruct.cpp:556 / :604
            lm = -1; fm = 0x7FFFFFFF;
#pragma unroll
            for (int i = 0; i < IPT; i++) {
                m[i] = -1;
                if (cl[i] >= 0) {
                    m[i] = find(cl[i], ks);
                    if (m[i] >= 0) { ks = m[i]; lm = m[i]; if (fm == 0x7FFFFFFF) fm = m[i]; }
                }
            }
        }
        const int exc = wg_exclusive_scan<BLOCK>(lm, -1, [](int a, int b) { return a > b ? a : b; }, scan_tmp.i, (int *)nullptr);
        const int new_in = exc > 0 ? exc : 0;
        dirty = new_in > fm;                               // kstart passed this thread's first match: its walk must be redone
        ks_in = new_in > ks_in ? new_in : ks_in;
        if (!__syncthreads_or(dirty ? 1 : 0)) break;
    }
    // (the loop's last barrier: every thread is done with the lists; threads beyond the row stay for the exchange's barrier)

    // triangulation of the thread's 4 pixels (reconstruct.cpp:570-603), branch-free
    float out[12];
    unsigned hw = 0, cw = 0, bad = 0;
    unsigned wl4 = 0;
    if (color && inrow) wl4 = *reinterpret_cast<const unsigned *>(whiteL + base + k0);
#pragma unroll
    for (int i = 0; i < IPT; i++) {
        const int j = k0 + i, mm = m[i] >= 0 ? m[i] : 0;
        const double r0 = (double)j + kc.q3, r1 = (double)row + kc.q7, r2 = kc.q11;
        const double w = kc.q14 * (double)(j - mm) + kc.q15;
        auto expo = [](double x) -> unsigned { return ((unsigned)__double2hiint(x) >> 20) & 0x7FFu; };
        const unsigned e0 = expo(r0), e1 = expo(r1), e2 = expo(r2), e3 = expo(w);
        const unsigned emin = min(min(e0, e1), min(e2, e3)), emax = max(max(e0, e1), max(e2, e3));
        if (!(emin >= 1023u - 200u && emax < 1023u + 200u) && m[i] >= 0) bad |= 1u << i;
        double y = __builtin_amdgcn_rcp(w);                // the three IEEE quotients from one refined reciprocal (mf_match_lean_kernel)
        y = __builtin_fma(y, __builtin_fma(-w, y, 1.0), y);
        y = __builtin_fma(y, __builtin_fma(-w, y, 1.0), y);
        const double r[3] = {r0, r1, r2};
#pragma unroll
        for (int cc = 0; cc < 3; cc++) {
            const double q = r[cc] * y;
            const double e = __builtin_fma(-w, q, r[cc]);
            out[3 * i + cc] = (float)__builtin_fma(e, y, q);
        }
        if (color && m[i] >= 0) cw |= (unsigned)(((int)((wl4 >> (8 * i)) & 0xFFu) + (int)whiteR[base + mm]) / 2) << (8 * i);   // :598
    }
    if (__builtin_expect(bad != 0, 0)) {                   // exact zeros, w == 0, extreme exponents: the three real divisions
#pragma unroll 1
        for (int i = 0; i < IPT; i++) {
            if (!((bad >> i) & 1u)) continue;
            const int j = k0 + i;
            const double r0 = (double)j + kc.q3, r1 = (double)row + kc.q7, r2 = kc.q11;
            const double w = kc.q14 * (double)(j - m[i]) + kc.q15;
            const float X0 = (float)(r0 / w), X1 = (float)(r1 / w), X2 = (float)(r2 / w);
#pragma unroll
            for (int jj = 0; jj < IPT; jj++)
                if (jj == i) { out[3 * jj] = X0; out[3 * jj + 1] = X1; out[3 * jj + 2] = X2; }
        }
    }
#pragma unroll
    for (int i = 0; i < IPT; i++) {
        if constexpr (HAS_T) {                             // matCoordTrans(3x4 f32) * [X;1]: f64 accumulate, narrow once
            const double X0 = (double)out[3 * i], X1 = (double)out[3 * i + 1], X2 = (double)out[3 * i + 2];
#pragma unroll
            for (int cc = 0; cc < 3; cc++) {
                double sacc = __builtin_fma(kc.T[cc * 4], X0, 0.0);
                sacc = __builtin_fma(kc.T[cc * 4 + 1], X1, sacc);
                sacc = __builtin_fma(kc.T[cc * 4 + 2], X2, sacc);
                out[3 * i + cc] = (float)(sacc + kc.T[cc * 4 + 3]);
            }
        }
        const bool mt = m[i] >= 0;
        out[3 * i] = mt ? out[3 * i] : 0.0f; out[3 * i + 1] = mt ? out[3 * i + 1] : 0.0f; out[3 * i + 2] = mt ? out[3 * i + 2] : 0.0f;
        hw |= (mt ? 1u : 0u) << (8 * i);
    }
    const size_t o = base + k0;
    f32x4 *d4 = reinterpret_cast<f32x4 *>(xyz + 3 * o);
    const f32x4 v0 = {out[0], out[1], out[2], out[3]}, v1 = {out[4], out[5], out[6], out[7]}, v2 = {out[8], out[9], out[10], out[11]};
    (void)d4;
    sh.x4[3 * tid] = v0; sh.x4[3 * tid + 1] = v1; sh.x4[3 * tid + 2] = v2;
    __syncthreads();
    {
        f32x4 *const r4 = reinterpret_cast<f32x4 *>(xyz + 3 * base);
        const int n4 = 3 * W / 4;                          // 16-byte words of the row
#pragma unroll
        for (int q = 0; q < 3; q++)
            if (tid + q * BLOCK < n4) __builtin_nontemporal_store(sh.x4[tid + q * BLOCK], r4 + tid + q * BLOCK);
    }
    if (!inrow) return;
    __builtin_nontemporal_store(hw, reinterpret_cast<unsigned *>(has + o));
    if (color) __builtin_nontemporal_store(cw, reinterpret_cast<unsigned *>(color + o));
    if (match_k) *reinterpret_cast<int4 *>(match_k + o) = make_int4(m[0], m[1], m[2], m[3]);
}

hipError_t launch_ge_match(const int32_t *codeL, const uint8_t *validL, const int32_t *codeR, const uint8_t *validR,
                           int W, int H, const DevCalib &cal, const uint8_t *whiteL, const uint8_t *whiteR,
                           float *xyz, uint8_t *has, uint8_t *color, int32_t *match_k, int code_bound, hipStream_t s)
{
    const int TC = 8192;                           // codes below this use the direct list-head table (16 KB LDS)
    int deferred_only = 0;
    const bool a16 = ((uintptr_t)codeL | (uintptr_t)codeR | (uintptr_t)xyz | (uintptr_t)match_k) % 16 == 0;
    const bool a4 = ((uintptr_t)validL | (uintptr_t)validR | (uintptr_t)has | (uintptr_t)whiteL | (uintptr_t)color) % 4 == 0;
    if (W > 2048 && W <= 4096 && W % 4 == 0 && a16 && a4 && cal.q_simple && !tl_debug.no_ge_lean) {
        K4Lean kc;
        kc.q3 = cal.Q[3]; kc.q7 = cal.Q[7]; kc.q11 = cal.Q[11]; kc.q14 = cal.Q[14]; kc.q15 = cal.Q[15];
        for (int i = 0; i < 12; i++) kc.T[i] = (double)cal.T[i];
        // code_bound: every valid code is below it (0 = unknown): a 12-bit Gray stack needs half the table, half the counters per
        // thread to clear, scan and own (a code at or above the table's size only defers its row to the general kernel)
#define SLR_GE_LEAN(T_, TC_) SLR_LAUNCH((ge_match_lean_kernel<T_, TC_>), dim3(H), dim3(1024), 0, s, codeL, validL, codeR, validR, W, H, kc, whiteL, whiteR, xyz, has, color, match_k)
        const bool small_table = code_bound > 0 && code_bound <= 4096;
        if (cal.has_T) { if (small_table) SLR_GE_LEAN(true, 4096); else SLR_GE_LEAN(true, 8192); }
        else           { if (small_table) SLR_GE_LEAN(false, 4096); else SLR_GE_LEAN(false, 8192); }
#undef SLR_GE_LEAN
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        deferred_only = 1;                          // the rows it marked (codes >= 8192, long lists) follow in the general kernel
    }
#define SLR_GE(IPT)                                                                                                    \
    SLR_LAUNCH(ge_match_kernel<IPT>, dim3(H), dim3(256), (size_t)TC * 2, s, codeL, validL, codeR, validR, W, H, TC, \
                       cal, whiteL, whiteR, xyz, has, color, match_k, deferred_only)
    if (W <= 256) SLR_GE(1);
    else if (W <= 512) SLR_GE(2);
    else if (W <= 1024) SLR_GE(4);
    else if (W <= 2048) SLR_GE(8);
    else if (W <= 4096) SLR_GE(16);
    else if (W <= 8192) SLR_GE(32);
    else if (W <= 16384) SLR_GE(64);
    else return hipErrorInvalidValue;              // the C ABI rejects wider Gray rows (LDS: keys + matches + heads)
#undef SLR_GE
    return hipGetLastError();
}

}  // namespace slr
