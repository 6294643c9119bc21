// kernels_ray.hip -- GRAY_ONLY mode: K3' bucket scatter (counting sort by projector pixel) and K6 ray-ray
// midpoint triangulation with PointCloudImage sum/count accumulation; plus the PointCloudImage adaptors.
// gfx950 (MI355X) only.
//
// Reference behaviour restated (never copied):
//   scatter   Reconstruct::decodePaterns  camPixels[ac(x,y)].push_back(Point(col,row))   Duke/reconstruct.cpp:61-72
//             ac(x,y) = x*scan_h + y                                                     Duke/reconstruct.h:94-97
//   K6        Reconstruct::triangulation                                                 Duke/reconstruct.cpp:417-481
//             cam2WorldSpace                                                             Duke/reconstruct.cpp:310-322
//             Utilities::pixelToImageSpace / normalize / line_lineIntersection           Duke/utilities.cpp:47-56,19-28,399-425
//             PointCloudImage::addPoint / setPoint / getPoint                            Duke/pointcloudimage.cpp:28-37,56-67,86-97
//
// The reference pushes camera pixels into per-projector-pixel vectors while walking the camera image
// column-major; a bucket's traversal order (c1 outer, c2 inner) fixes the f32 accumulation order.  Here the
// buckets are built as a counting sort -- histogram with returning atomics (rank inside the bucket, arbitrary
// order), one exclusive scan over both cameras' histograms, scatter to offs[bucket]+rank -- and the push order is
// restored inside each bucket by sorting its (few) pixels on (col,row), which IS the column-major walk order.
// Buckets are numbered by their PointCloudImage cell  j*scan_w + i  (bucket ac = i*scan_h + j  <->  cell (j,i)).
// Round 3: the whole-path entries decode inside the histogram kernel (gray_decode_count_kernel); K6 takes its cells from a work
// list ordered by bucket lengths, made inside the scatter's launches (ray_scatter_list_kernel), and runs one wave per 64 cells
// with the buckets in registers and per-lane LDS columns (ray_triangulate_small_kernel), longer buckets through LDS 256 cells
// at a time (ray_triangulate_staged_kernel); DESIGN section 4 has the measurements.
#include "slr_device.hpp"
#include "decode_common.hpp"

#include <rocprim/device/device_scan.hpp>
#include <math.h>
#include <type_traits>
#include <utility>

namespace slr {

constexpr uint32_t kNoBucket = 0xFFFFFFFFu;

// pass 1: per pixel -> cell id (or kNoBucket) and its arrival rank inside that cell
__global__ __launch_bounds__(256) void ray_count_kernel(const int32_t *__restrict__ code_x, const int32_t *__restrict__ code_y,
                                                        const uint8_t *__restrict__ valid, unsigned npix, int scan_w,
                                                        int scan_h, uint32_t *__restrict__ cnt,
                                                        uint32_t *__restrict__ cell_of, uint32_t *__restrict__ rank_of)
{
    const unsigned nb = (unsigned)scan_w * (unsigned)scan_h;
    const unsigned lane = threadIdx.x & 63u;
    // (the trip count is wave-uniform: the loop body uses wave-wide operations)
    for (unsigned p0 = blockIdx.x * 256u + (threadIdx.x & ~63u); p0 < npix; p0 += gridDim.x * 256u) {
        const unsigned p = p0 + lane;
        unsigned cell = kNoBucket, rank = 0;
        if (p < npix && valid[p]) {
            const unsigned long long k = (unsigned long long)code_x[p] * (unsigned)scan_h + (unsigned)code_y[p];
            if (k < nb) {                                      // Q9: ac >= scan_w*scan_h is an OOB write -> dropped
                const unsigned i = (unsigned)k / (unsigned)scan_h, j = (unsigned)k - i * (unsigned)scan_h;
                cell = j * (unsigned)scan_w + i;               // (a code_y >= scan_h aliases into column i+1.., as ac does)
            }
        }
        // A camera sees a projector pixel in a run of adjacent pixels of a row, i.e. in adjacent lanes: one atomic per RUN of
        // equal cells (its first lane adds the run's length and hands the others their places) instead of one per pixel.
        // The order inside a bucket is arbitrary here (ray_triangulate sorts it), only the ranks must be a permutation.
        const unsigned prev = __shfl_up(cell, 1);
        const bool head = lane == 0 || cell != prev;
        const unsigned long long H = __ballot(head);
        const unsigned long long upto = H & (~0ull >> (63u - lane));                  // heads at or below this lane
        const unsigned start = 63u - (unsigned)__builtin_clzll(upto);                 // (lane 0 is a head: never empty)
        const unsigned long long above = start == 63u ? 0ull : H & ~((2ull << start) - 1ull);
        const unsigned len = (above ? (unsigned)__builtin_ctzll(above) : 64u) - start;
        unsigned first = 0;
        if (head && cell != kNoBucket) first = atomicAdd(&cnt[cell], len);
        first = __shfl(first, (int)start);
        rank = first + (lane - start);
        if (p < npix) {
            cell_of[p] = cell;
            rank_of[p] = cell != kNoBucket ? rank : 0u;
        }
    }
}

// pass 1 fused into the GRAY_ONLY decode (round 3): gray_decode_kernel<4, false>'s loop (kernels_decode.hip; reconstruct.cpp:
// 349-407) followed by ray_count_kernel's cell / rank assignment, so that the codes never travel through HBM (9 bytes per pixel
// written and read again) and one launch per camera is gone.  A thread decodes 4 consecutive pixels, so a run of equal cells
// spans pixels of a thread AND lanes: heads are found per pixel (the first pixel against the previous lane's last), a head's
// run ends at the next head in its thread or at the first head of the next lane that has one, and the pixels in front of a
// thread's first head take their places from the last head of the nearest lane below that has one.  One atomic per run.
__global__ __launch_bounds__(256) void gray_decode_count_kernel(GrayPlanes pl, int n_col_bits, int n_row_bits, int pitch, int W, int H,
                                                                int black_thr, int white_thr, int scan_w, int scan_h,
                                                                uint32_t *__restrict__ cnt, uint32_t *__restrict__ cell_of,
                                                                uint32_t *__restrict__ rank_of)
{
    const unsigned gpr = (unsigned)W / 4u, total = gpr * (unsigned)H, nb = (unsigned)scan_w * (unsigned)scan_h;
    const unsigned lane = threadIdx.x & 63u;
    // (the trip count is wave-uniform: the loop body uses wave-wide operations)
    for (unsigned g0 = blockIdx.x * 256u + (threadIdx.x & ~63u); g0 < total; g0 += gridDim.x * 256u) {
        const unsigned g = g0 + lane;
        const bool live = g < total;
        const unsigned gg = live ? g : total - 1;
        const unsigned row = gg / gpr, col0 = (gg - row * gpr) * 4u;
        const size_t so = (size_t)row * pitch + col0, m = (size_t)row * W + col0;
        auto fetch = [&](int plane) -> unsigned { return __builtin_nontemporal_load(reinterpret_cast<const unsigned *>(pl.p[plane] + so)); };
        const unsigned wv = fetch(0), bv = fetch(1);
        int gx[4] = {0, 0, 0, 0}, gy[4] = {0, 0, 0, 0}, err[4] = {0, 0, 0, 0};
        // (the contrast test can only fire with a positive white threshold -- the reference's default is 0, SURVEY Q10 --: a
        // wave-uniform fact, kept out of the per-pixel work; a code bit is then a byte compare and an add of the word to itself)
        const auto bits = [&](int first, int n, int g[4], auto wt) {
            for (int c = 0; c < n; c++) {
                const unsigned a = fetch(first + 2 * c), b = fetch(first + 2 * c + 1);
#pragma unroll
                for (int i = 0; i < 4; i++) {
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
        unsigned cell[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int mask = ((int)((wv >> (8 * i)) & 0xFF) - (int)((bv >> (8 * i)) & 0xFF) > black_thr) ? 1 : 0;
            int x = gx[i], y = gy[i];
            x ^= x >> 1; x ^= x >> 2; x ^= x >> 4; x ^= x >> 8;     // graycodes.cpp:116-128
            y ^= y >> 1; y ^= y >> 2; y ^= y >> 4; y ^= y >> 8;
            const int e = err[i] | ((y > scan_h || x > scan_w) ? 1 : 0);   // reconstruct.cpp:364 (Q9 '>')
            cell[i] = kNoBucket;
            if (live && (mask & (e ^ 1))) {
                const unsigned long long k = (unsigned long long)(unsigned)x * (unsigned)scan_h + (unsigned)y;
                if (k < nb) {                                  // Q9: ac >= scan_w*scan_h is an OOB write -> dropped
                    const unsigned ci = (unsigned)k / (unsigned)scan_h, cj = (unsigned)k - ci * (unsigned)scan_h;
                    cell[i] = cj * (unsigned)scan_w + ci;
                }
            }
        }
        // runs of equal cells over the wave's 256 pixels (pixel p = 4 lane + i)
        const unsigned prev = __shfl_up(cell[3], 1);
        const bool head[4] = {lane == 0 || cell[0] != prev, cell[1] != cell[0], cell[2] != cell[1], cell[3] != cell[2]};
        const int first_head = head[0] ? 0 : head[1] ? 1 : head[2] ? 2 : head[3] ? 3 : 4;
        const int last_head = head[3] ? 3 : head[2] ? 2 : head[1] ? 1 : head[0] ? 0 : -1;
        const unsigned long long any = __ballot(last_head >= 0);                    // (lane 0 is in it)
        const unsigned long long above = lane == 63u ? 0ull : any & ~((2ull << lane) - 1ull);
        const unsigned next_lane = above ? (unsigned)__builtin_ctzll(above) : 64u;
        const unsigned next_first = (unsigned)__shfl(first_head, (int)(next_lane & 63u));
        const unsigned end_after = next_lane < 64u ? 4u * next_lane + next_first : 256u;   // where the run of this thread's last head ends
        unsigned first[4] = {0, 0, 0, 0};                       // per head: the place of its run's first pixel
#pragma unroll
        for (int i = 0; i < 4; i++)
            if (head[i] && cell[i] != kNoBucket) {
                unsigned end = end_after;
#pragma unroll
                for (int j = 3; j > i; j--) if (head[j]) end = 4u * lane + (unsigned)j;
                first[i] = atomicAdd(&cnt[cell[i]], end - (4u * lane + (unsigned)i));
            }
        const unsigned my_last_first = last_head == 3 ? first[3] : last_head == 2 ? first[2] : last_head == 1 ? first[1] : first[0];
        const unsigned long long below = any & ((1ull << lane) - 1ull);
        const unsigned from_lane = below ? 63u - (unsigned)__builtin_clzll(below) : 0u;
        const unsigned from_first = (unsigned)__shfl(my_last_first, (int)from_lane);
        const unsigned from_pos = 4u * from_lane + (unsigned)__shfl(last_head, (int)from_lane);
        unsigned rank[4];
        unsigned cur_first = from_first, cur_pos = from_pos;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            if (head[i]) { cur_first = first[i]; cur_pos = 4u * lane + (unsigned)i; }
            rank[i] = cell[i] != kNoBucket ? cur_first + (4u * lane + (unsigned)i - cur_pos) : 0u;
        }
        if (live) {
            i32x4 o; o.x = (int)cell[0]; o.y = (int)cell[1]; o.z = (int)cell[2]; o.w = (int)cell[3];
            i32x4 q; q.x = (int)rank[0]; q.y = (int)rank[1]; q.z = (int)rank[2]; q.w = (int)rank[3];
            *reinterpret_cast<i32x4 *>(cell_of + m) = o;
            *reinterpret_cast<i32x4 *>(rank_of + m) = q;
        }
    }
}

// whether gray_decode_count_kernel applies: dword loads of 4 pixels, 16-byte stores
bool ray_decode_count_applies(const GrayPlanes &pl, int nplanes, int n_row_bits, int pitch, int W, int H)
{
    if (n_row_bits < 1 || W % 4 != 0 || pitch % 4 != 0 || (long long)W * H >= (1ll << 30) || (long long)H * pitch >= (1ll << 32)) return false;
    for (int p = 0; p < nplanes; p++) if ((uintptr_t)pl.p[p] % 4 != 0) return false;
    return true;
}

hipError_t launch_gray_decode_count(const GrayPlanes &pl, int n_col_bits, int n_row_bits, int pitch, int W, int H, int black_thr,
                                    int white_thr, int scan_w, int scan_h, uint32_t *cnt, uint32_t *cell_of, uint32_t *rank_of,
                                    hipStream_t s)
{
    const size_t groups = (size_t)(W / 4) * H;
    const unsigned blocks = (unsigned)((groups + 255) / 256 < 16384 ? (groups + 255) / 256 : 16384);
    SLR_LAUNCH(gray_decode_count_kernel, dim3(blocks), dim3(256), 0, s, pl, n_col_bits, n_row_bits, pitch, W, H, black_thr, white_thr,
               scan_w, scan_h, cnt, cell_of, rank_of);
    return hipGetLastError();
}

// pass 2 (after the scan): item (col<<16 | row) -> items[offs[cell] + rank]
__global__ __launch_bounds__(256) void ray_scatter_kernel(const uint32_t *__restrict__ cell_of, const uint32_t *__restrict__ rank_of,
                                                          int W, const uint32_t *__restrict__ offs, uint32_t *__restrict__ items)
{
    const unsigned col = blockIdx.x * 256u + threadIdx.x, row = blockIdx.y;
    if (col >= (unsigned)W) return;
    const unsigned p = row * (unsigned)W + col;
    const unsigned cell = cell_of[p];
    if (cell != kNoBucket) items[offs[cell] + rank_of[p]] = (col << 16) | row;
}

hipError_t launch_ray_count(const int32_t *code_x, const int32_t *code_y, const uint8_t *valid, int W, int H,
                            int scan_w, int scan_h, uint32_t *cnt, uint32_t *cell_of, uint32_t *rank_of, hipStream_t s)
{
    const size_t n = (size_t)W * H;
    const unsigned blocks = (unsigned)((n + 255) / 256 < 16384 ? (n + 255) / 256 : 16384);
    SLR_LAUNCH(ray_count_kernel, dim3(blocks), dim3(256), 0, s, code_x, code_y, valid, (unsigned)n, scan_w,
                       scan_h, cnt, cell_of, rank_of);
    return hipGetLastError();
}

hipError_t launch_ray_scatter(const uint32_t *cell_of, const uint32_t *rank_of, int W, int H, const uint32_t *offs,
                              uint32_t *items, hipStream_t s)
{
    SLR_LAUNCH(ray_scatter_kernel, dim3((W + 255) / 256, H), dim3(256), 0, s, cell_of, rank_of, W, offs, items);
    return hipGetLastError();
}

size_t ray_scan_temp_bytes(size_t n)
{
    size_t bytes = 0;
    (void)rocprim::exclusive_scan(nullptr, bytes, (const uint32_t *)nullptr, (uint32_t *)nullptr, 0u, n, rocprim::plus<uint32_t>());
    return bytes;
}

hipError_t launch_ray_scan(const uint32_t *cnt, uint32_t *offs, size_t n, void *temp, size_t temp_bytes, hipStream_t s)
{
    return rocprim::exclusive_scan(temp, temp_bytes, cnt, offs, 0u, n, rocprim::plus<uint32_t>(), s);
}

// ---- per camera-pixel unit ray (reconstruct.cpp:440-445) ----------------------------------------------
__device__ __forceinline__ void undistort_point_ray(float px, float py, const DevCamera &c, float &ox, float &oy)
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

// cam2WorldSpace: p <- (float)(-(R^T t)) + (float)(R^T p), f64 accumulation inside each product
__host__ __device__ __forceinline__ void cam2world(const DevCamera &c, float p[3])
{
    float out[3];
#pragma unroll
    for (int r = 0; r < 3; r++) {
        double s = 0, s2 = 0;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            s += (double)c.R[k * 3 + r] * (double)c.t[k];
            s2 += (double)c.R[k * 3 + r] * (double)p[k];
        }
        out[r] = (float)(s * -1.0) + (float)s2;
    }
    p[0] = out[0]; p[1] = out[1]; p[2] = out[2];
}

// x87 != 0 (SLR_OPT_EVAL_MODEL = 1): the two compound float expressions of this function as the reference's
// MSVC2010 x87 binary evaluates them -- (p - cc) / fc with one rounding at the store (utilities.cpp:51-52), the sum of squares
// at 53 bits, rounded once when it is passed to sqrt(float) (utilities.cpp:21).  cam2WorldSpace is OpenCV library code (its GEMM
// accumulates in f64 either way) and undistortPoints is f64 in the source.
__device__ __forceinline__ void pixel_ray(uint32_t item, const DevCamera &c, const float pos[3], float ray[3], int x87 = 0)
{
    float ux, uy, pt[3];
    undistort_point_ray((float)(item >> 16), (float)(item & 0xFFFFu), c, ux, uy);   // item = col<<16 | row
    if (x87) {
        pt[0] = (float)(((double)ux - (double)c.ccx) / (double)c.fcx);
        pt[1] = (float)(((double)uy - (double)c.ccy) / (double)c.fcy);
    } else {
        pt[0] = (ux - c.ccx) / c.fcx;              // utilities.cpp:51-53
        pt[1] = (uy - c.ccy) / c.fcy;
    }
    pt[2] = 1.0f;
    cam2world(c, pt);
    ray[0] = pos[0] - pt[0]; ray[1] = pos[1] - pt[1]; ray[2] = pos[2] - pt[2];
    // utilities.cpp:21-25: sqrt(float) overload, max(0.000001, mag) in f64, divide by the f32 narrowing
    const float ss = x87 ? (float)((double)ray[0] * (double)ray[0] + (double)ray[1] * (double)ray[1] + (double)ray[2] * (double)ray[2])
                         : ray[0] * ray[0] + ray[1] * ray[1] + ray[2] * ray[2];
    const double mag = (double)sqrtf(ss);
    const float dv = (float)(0.000001 > mag ? 0.000001 : mag);
    ray[0] /= dv; ray[1] /= dv; ray[2] /= dv;
}

// per-(calibration, W, H) unit-ray tables: pixel_ray depends on the camera pixel only, so it is evaluated once
// per calibration and cached by the context (like K4's undistortion tables), not once per pair per frame
__global__ __launch_bounds__(256) void ray_table_kernel(DevCalib cal, int W, int H, float *__restrict__ raysL,
                                                        float *__restrict__ raysR)
{
    float posL[3] = {0, 0, 0}, posR[3] = {0, 0, 0};
    cam2world(cal.cam[0], posL);
    cam2world(cal.cam[1], posR);
    const unsigned col = blockIdx.x * 256u + threadIdx.x, row = blockIdx.y;
    if (col >= (unsigned)W) return;
    const size_t o = ((size_t)row * W + col) * 3;
    float r[3];
    pixel_ray((col << 16) | row, cal.cam[0], posL, r, cal.eval_x87);
    raysL[o] = r[0]; raysL[o + 1] = r[1]; raysL[o + 2] = r[2];
    pixel_ray((col << 16) | row, cal.cam[1], posR, r, cal.eval_x87);
    raysR[o] = r[0]; raysR[o + 1] = r[1]; raysR[o + 2] = r[2];
}

hipError_t launch_ray_tables(const DevCalib &cal, int W, int H, float *raysL, float *raysR, hipStream_t s)
{
    SLR_LAUNCH(ray_table_kernel, dim3((W + 255) / 256, H), dim3(256), 0, s, cal, W, H, raysL, raysR);
    return hipGetLastError();
}

__device__ __forceinline__ float dot3(const float a[3], const float b[3])
{
    float s = 0;
    s += a[0] * b[0];
    s += a[1] * b[1];
    s += a[2] * b[2];
    return s;
}

// Three IEEE quotients over ONE denominator (utilities.cpp:414-415 divides b, c and a by denom).  hipcc expands an f32 division
// into v_div_scale x2, v_rcp, two Newton steps on the reciprocal, the quotient with two fma corrections (the last one as
// v_div_fmas) and v_div_fixup; the scale instructions leave their operands alone (and fmas / fixup reduce to fma / identity)
// when the denominator and the numerator are normal numbers of moderate exponent (ISA: no scaling unless an operand or the
// quotient comes near the denormals, the exponents differ by 96 or more, or the numerator's biased exponent is <= 23).  Inside
// that range the refined reciprocal is the same for the three divisions, so it is computed once and every quotient takes the
// same five instructions the compiler's sequence would have run: bit-identical results, 18 instructions and one v_rcp instead
// of 33 and three.  Outside the range (b below 2^-60, among them the zeros; a or c from 2^12 on) the plain division runs.  -DSLR_RAY_PLAIN_DIV builds the plain divisions everywhere.
// (b, c, a over denom = a c - b b with |denom| >= 0.1: a and c are sums of squares, so a, c < 2^12 bounds denom and |b| by 2^24
// from above and a, c by 2^-16 from below; only b can be small, and from 2^-60 on its quotient (>= 2^-84) and the remainders
// of the two corrections (>= 2^-108) are normal numbers.)
__device__ __forceinline__ void div3(float b, float c, float a, float den, float &qb, float &qc, float &qa)
{
#ifndef SLR_RAY_PLAIN_DIV
    if (fabsf(b) >= 0x1p-60f && fmaxf(a, c) < 0x1p12f) {     // (NaNs fail the comparisons)
        float r = __builtin_amdgcn_rcpf(den);
        const float e = __builtin_fmaf(-den, r, 1.0f);
        r = __builtin_fmaf(e, r, r);
        const auto quot = [&](float n) {
            float q = n * r;
            float rem = __builtin_fmaf(-den, q, n);
            q = __builtin_fmaf(rem, r, q);
            rem = __builtin_fmaf(-den, q, n);
            return __builtin_fmaf(rem, r, q);
        };
        qb = quot(b); qc = quot(c); qa = quot(a);
        return;
    }
#endif
    qb = b / den; qc = c / den; qa = a / den;
}

// Vec3f::dot under the x87 model (utilities.cpp:404-408): s += a[i] * b[i] with the product exact on the 53-bit stack and ONE
// rounding per step at the store -- fma in f64 (the product of two floats is exact there), narrowed
__device__ __forceinline__ float dot3_x87(const float a[3], const float b[3])
{
    float s = 0;
    s = (float)__builtin_fma((double)a[0], (double)b[0], (double)s);
    s = (float)__builtin_fma((double)a[1], (double)b[1], (double)s);
    s = (float)__builtin_fma((double)a[2], (double)b[2], (double)s);
    return s;
}

// utilities.cpp:399-425.  X87 (SLR_OPT_EVAL_MODEL = 1): the dot products as above, denom = a*c - b*b rounded once (:412), s and t
// with their quotients and products at 53 bits and one rounding each (:417-418) -- plain f64 arithmetic, a parity mode; the
// Point3f operators of :420-424 store every f32 operation in both models.
template <bool X87 = false>
__device__ __forceinline__ bool line_line(const float p1[3], const float v1[3], const float p2[3], const float v2[3],
                                          float out[3])
{
    const float v12[3] = {p1[0] - p2[0], p1[1] - p2[1], p1[2] - p2[2]};
    float s, t;
    if constexpr (X87) {
        const float a = dot3_x87(v1, v1), c = dot3_x87(v2, v2), b = dot3_x87(v1, v2);
        const float d = dot3_x87(v12, v1), e = dot3_x87(v12, v2);
        const float denom = (float)((double)a * (double)c - (double)b * (double)b);
        if (fabsf(denom) < 0.1f) return false;
        const double bq = (double)b / (double)denom, cq = (double)c / (double)denom, aq = (double)a / (double)denom;
        s = (float)(bq * (double)e - cq * (double)d);
        t = (float)(-bq * (double)d + aq * (double)e);
    } else {
    const float a = dot3(v1, v1), c = dot3(v2, v2), b = dot3(v1, v2);
    const float d = dot3(v12, v1), e = dot3(v12, v2);
    const float denom = a * c - b * b;
    if (fabsf(denom) < 0.1f) return false;
    float bq, cq, aq;
    div3(b, c, a, denom, bq, cq, aq);
    s = bq * e - cq * d;
    t = -bq * d + aq * e;
    }
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float u = p1[k] + s * v1[k];
        const float w = p2[k] + t * v2[k];
        out[k] = 0.5f * (u + w);
    }
    return true;
}

// Utilities::line_lineIntersection over arrays (utilities.cpp:399-425): n lines through p1 along v1[i] against n lines through p2
// along v2[i] -- the function K6 runs per pixel pair, exposed so that it (and its shared-reciprocal divisions) can be checked on
// rays K6's calibrated unit rays never produce: perpendicular, nearly parallel, tiny, huge
template <bool X87>
__global__ __launch_bounds__(256) void line_line_kernel(size_t n, const float *__restrict__ p1, const float *__restrict__ p2,
                                                        const float *__restrict__ v1, const float *__restrict__ v2,
                                                        float *__restrict__ out, uint8_t *__restrict__ ok)
{
    const float a[3] = {p1[0], p1[1], p1[2]}, b[3] = {p2[0], p2[1], p2[2]};
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256ull) {
        const float r1[3] = {v1[3 * i], v1[3 * i + 1], v1[3 * i + 2]}, r2[3] = {v2[3 * i], v2[3 * i + 1], v2[3 * i + 2]};
        float X[3] = {0.0f, 0.0f, 0.0f};
        const bool hit = line_line<X87>(a, r1, b, r2, X);
        out[3 * i] = hit ? X[0] : 0.0f; out[3 * i + 1] = hit ? X[1] : 0.0f; out[3 * i + 2] = hit ? X[2] : 0.0f;
        ok[i] = hit ? 1 : 0;
    }
}

hipError_t launch_line_line(size_t n, const float *p1, const float *p2, const float *v1, const float *v2, float *out, uint8_t *ok,
                            hipStream_t s)
{
    const unsigned blocks = (unsigned)((n + 255) / 256 < 8192 ? ((n + 255) / 256 ? (n + 255) / 256 : 1) : 8192);
    if (tl_debug.eval_x87) SLR_LAUNCH(line_line_kernel<true>, dim3(blocks), dim3(256), 0, s, n, p1, p2, v1, v2, out, ok);
    else SLR_LAUNCH(line_line_kernel<false>, dim3(blocks), dim3(256), 0, s, n, p1, p2, v1, v2, out, ok);
    return hipGetLastError();
}

// restore the reference's push order inside one bucket: ascending (col,row) == ascending packed item
__device__ __forceinline__ void sort_bucket(uint32_t *__restrict__ a, unsigned lo, unsigned hi)
{
    for (unsigned i = lo + 1; i < hi; i++) {
        const uint32_t v = a[i];
        unsigned j = i;
        while (j > lo && a[j - 1] > v) { a[j] = a[j - 1]; j--; }
        a[j] = v;
    }
}

__device__ __forceinline__ void load_ray(const float *__restrict__ rays, uint32_t item, int W, float ray[3])
{
    const size_t o = ((size_t)(item & 0xFFFFu) * W + (item >> 16)) * 3;     // item = col<<16 | row
    ray[0] = rays[o]; ray[1] = rays[o + 1]; ray[2] = rays[o + 2];
}

// ---- K6 work list ----------------------------------------------------------------------------------------------------
// A lane's (c1 outer, c2 inner) loops run len_L x len_R pairs and a wave runs max len_L x max len_R of its 64 lanes: with one
// lane per consecutive cell that is 144-256 iterations for a mean of ~105 pairs (a camera pixel row sees a projector pixel in
// 9, 12 or now and then 16 of its pixels), 59 % of the VALU work was masked-off lanes (PMC: 287 M wave instructions for 120 M
// pairs).  So the cells are first ordered by (len_L, len_R) -- a counting sort over 32 x 32 keys, longest first so that the
// heaviest waves start first, and those with a bucket of more than 16 items (the staged form's) before all others -- and the
// triangulation kernels take their cells from that list: the lanes of a wave have the same trip counts (lengths above 31 share
// a key and diverge as before).  Cells without pairs get their zeros here and are not listed.  The sort is not stable across
// workgroups (places inside a key are handed out by atomics); cells are independent, so the results do not depend on it.
// It needs the bucket offsets only, not the items: its workgroups ride in the scatter kernels' launches (ray_scatter_list_kernel).
// (Sorting each 4096-cell segment on its own, one kernel and no global atomics, made the list in the same 34 us and the
// triangulation 25 % slower: waves straddling two keys, heavy waves starting late.  profiles/exp/r03/k6_segment_sort_form.hip.txt)
constexpr unsigned kKeyLen = 32, kKeys = kKeyLen * kKeyLen;
constexpr unsigned kNetLen = 16;                             // buckets of up to 16 items: the sorting-network form of K6
constexpr unsigned kBins = 2 * kKeys;                        // cells with a bucket beyond kNetLen items first, then the others
constexpr unsigned kKeyCells = 4096;                         // cells per workgroup of the list work (16 per thread): few workgroups, so that
                                                             // they leave the wave slots of the launch to the scatter they ride with

__device__ __forceinline__ unsigned cell_key(const uint32_t *__restrict__ offs, unsigned nb, unsigned b)
{
    const unsigned lenL = offs[b + 1] - offs[b], lenR = offs[nb + b + 1] - offs[nb + b];
    if (lenL == 0 || lenR == 0) return kBins;                // no pairs
    const unsigned kl = lenL < kKeyLen ? lenL : kKeyLen - 1, kr = lenR < kKeyLen ? lenR : kKeyLen - 1;
    const unsigned big = lenL > kNetLen || lenR > kNetLen ? 0u : kKeys;
    return big + kKeys - 1 - (kl * kKeyLen + kr);            // longest first
}

// One LDS atomic per wave and distinct key (a wave of consecutive cells holds a handful of keys; one atomic per cell piles
// 256 of them on the same word): *place = this lane's place among the wave's lanes of the same key, counted from the value
// the counter had.  Lanes with key == kBins (no pairs) stay out.
__device__ __forceinline__ void wave_count_key(uint32_t *cnt, unsigned key, unsigned *place)
{
    const unsigned lane = threadIdx.x & 63u;
    unsigned long long todo = __ballot(key < kBins);
    while (todo) {
        const unsigned k = (unsigned)__builtin_amdgcn_readlane((int)key, (int)__builtin_ctzll(todo));
        const unsigned long long same = __ballot(key == k);
        unsigned first = 0;
        if (lane == (unsigned)__builtin_ctzll(same)) first = atomicAdd(&cnt[k], (unsigned)__builtin_popcountll(same));
        first = (unsigned)__builtin_amdgcn_readlane((int)first, (int)__builtin_ctzll(same));
        if (key == k && place) *place = first + (unsigned)__builtin_popcountll(same & ((1ull << lane) - 1ull));
        todo &= ~same;
    }
}

// hist[k] = cells with key k, hist[kBins] = listed cells in all, hist[kBins + 1] = those of them with a bucket beyond kNetLen
__device__ __forceinline__ void ray_key_hist_block(unsigned blk, const uint32_t *__restrict__ offs, unsigned nb, uint32_t *__restrict__ hist,
                                                   float *__restrict__ xyz_sum, uint8_t *__restrict__ count)
{
    __shared__ uint32_t h[kBins + 2];
    for (unsigned k = threadIdx.x; k < kBins + 2; k += 256) h[k] = 0;
    unsigned key[kKeyCells / 256];
#pragma unroll
    for (unsigned m = 0; m < kKeyCells / 256; m++) {
        const unsigned b = blk * kKeyCells + m * 256 + threadIdx.x;
        key[m] = b < nb ? cell_key(offs, nb, b) : kBins;
        if (b < nb && key[m] == kBins) { xyz_sum[3 * (size_t)b] = 0.0f; xyz_sum[3 * (size_t)b + 1] = 0.0f; xyz_sum[3 * (size_t)b + 2] = 0.0f; count[b] = 0; }
    }
    __syncthreads();
#pragma unroll
    for (unsigned m = 0; m < kKeyCells / 256; m++) {
        wave_count_key(h, key[m], nullptr);
        const unsigned long long some = __ballot(key[m] < kBins), big = __ballot(key[m] < kKeys);
        if ((threadIdx.x & 63u) == 0 && some) atomicAdd(&h[kBins], (unsigned)__builtin_popcountll(some));
        if ((threadIdx.x & 63u) == 0 && big) atomicAdd(&h[kBins + 1], (unsigned)__builtin_popcountll(big));
    }
    __syncthreads();
    for (unsigned k = threadIdx.x; k < kBins + 2; k += 256) if (h[k]) atomicAdd(&hist[k], h[k]);
}

// order[start(key) + place] = cell; cursor[] (zeroed) hands out the places of a key, one range per workgroup and key
__device__ __forceinline__ void ray_key_order_block(unsigned blk, const uint32_t *__restrict__ offs, unsigned nb, const uint32_t *__restrict__ hist,
                                                    uint32_t *__restrict__ cursor, uint32_t *__restrict__ order)
{
    __shared__ uint32_t start[kBins], cnt[kBins], part[4];
    const unsigned t = threadIdx.x;
    unsigned key[kKeyCells / 256], place[kKeyCells / 256];
#pragma unroll
    for (unsigned m = 0; m < kKeyCells / 256; m++) {
        const unsigned b = blk * kKeyCells + m * 256 + t;
        key[m] = b < nb ? cell_key(offs, nb, b) : kBins;
        place[m] = 0;
    }
    uint32_t mine[kBins / 256], run = 0;                     // exclusive prefix of hist: 8 consecutive keys per thread
#pragma unroll
    for (unsigned k = 0; k < kBins / 256; k++) { mine[k] = run; run += hist[t * (kBins / 256) + k]; cnt[t * (kBins / 256) + k] = 0; }
    const uint32_t before = wg_exclusive_scan<256>(run, 0u, [](uint32_t x, uint32_t y) { return x + y; }, part, (uint32_t *)nullptr);
#pragma unroll
    for (unsigned k = 0; k < kBins / 256; k++) start[t * (kBins / 256) + k] = before + mine[k];
    __syncthreads();
#pragma unroll
    for (unsigned m = 0; m < kKeyCells / 256; m++) wave_count_key(cnt, key[m], &place[m]);
    __syncthreads();
    for (unsigned k = t; k < kBins; k += 256) if (cnt[k]) start[k] += atomicAdd(&cursor[k], cnt[k]);
    __syncthreads();
#pragma unroll
    for (unsigned m = 0; m < kKeyCells / 256; m++)
        if (key[m] < kBins) order[start[key[m]] + place[m]] = blk * kKeyCells + m * 256 + t;
}

// K3' pass 2 and K6's work list in ONE launch per camera: the list needs the bucket offsets only, and its workgroups wait on
// LDS atomics and global latencies while the scatter's stream 16 bytes per pixel -- the first `list_blocks` workgroups of the
// launch do the list's part (PHASE 0, with the left camera's scatter: the key histogram; PHASE 1, with the right camera's: the
// order), all others scatter a row segment of 256 pixels.  (On a second stream the two kernels overlapped poorly: scatter + list
// 126 us for 85 + 45 alone.)
template <int PHASE>
__global__ __launch_bounds__(256) void ray_scatter_list_kernel(const uint32_t *__restrict__ cell_of, const uint32_t *__restrict__ rank_of, int W,
                                                               const uint32_t *__restrict__ offs_cam, uint32_t *__restrict__ items,
                                                               const uint32_t *__restrict__ offs, unsigned nb, uint32_t *__restrict__ hist,
                                                               uint32_t *__restrict__ cursor, uint32_t *__restrict__ order,
                                                               float *__restrict__ xyz_sum, uint8_t *__restrict__ count, unsigned list_blocks)
{
    if (blockIdx.x < list_blocks) {
        if constexpr (PHASE == 0) ray_key_hist_block(blockIdx.x, offs, nb, hist, xyz_sum, count);
        else ray_key_order_block(blockIdx.x, offs, nb, hist, cursor, order);
        return;
    }
    const unsigned gx = ((unsigned)W + 255u) / 256u, sb = blockIdx.x - list_blocks;
    const unsigned row = sb / gx, col = (sb - row * gx) * 256u + threadIdx.x;
    if (col >= (unsigned)W) return;
    const unsigned p = row * (unsigned)W + col;
    const unsigned cell = cell_of[p];
    if (cell != kNoBucket) items[offs_cam[cell] + rank_of[p]] = (col << 16) | row;
}

// One projector pixel's pairs, (c1 outer, c2 inner), accumulated like PointCloudImage::setPoint / addPoint do.
// what K6 takes from the calibration: the camera centres (cam2WorldSpace of the origin, reconstruct.cpp:239-240) and the scan's
// transfer matrix widened to f64 (exact), so that it stays in scalar registers
struct TriArgs {
    double T[12];
    float pos[2][3];
    int has_T;
};

struct CellSum {
    float sum[3] = {0.0f, 0.0f, 0.0f};
    unsigned num = 0;
    __device__ __forceinline__ void add(const TriArgs &cal, float X[3])
    {
        if (cal.has_T) {
            float Y[3];
#pragma unroll
            for (int r = 0; r < 3; r++) {
                double s = 0;
                s += cal.T[r * 4 + 0] * (double)X[0];
                s += cal.T[r * 4 + 1] * (double)X[1];
                s += cal.T[r * 4 + 2] * (double)X[2];
                s += cal.T[r * 4 + 3] * 1.0;
                Y[r] = (float)s;
            }
            X[0] = Y[0]; X[1] = Y[1]; X[2] = Y[2];
        }
        // num == 0: setPoint (also after the uchar counter has wrapped, pointcloudimage.cpp:95), else addPoint -- as selects
        const bool first = num == 0;
        const float s0 = X[0] + sum[0], s1 = X[1] + sum[1], s2 = X[2] + sum[2];
        sum[0] = first ? X[0] : s0; sum[1] = first ? X[1] : s1; sum[2] = first ? X[2] : s2;
        num = (num + 1) & 0xFFu;                   // (0 + 1 = 1 as well)
    }
};

// K6, one thread per listed projector pixel (bucket == PointCloudImage cell j*scan_w + i), 256 consecutive entries of the work
// list per workgroup.  A lane's loop over its (c1, c2) pairs is a serial chain (the f32 accumulation order is the reference's)
// and the cells of a workgroup are no longer neighbours, so what a pair needs must not be a per-lane walk through global
// memory (tried: 1.45 ms, 78 % of the wave cycles waiting): the workgroup brings its cells' buckets into LDS TOGETHER -- every
// cell lane copies its two buckets' items (equal lengths across the wave: no divergence), every staged item finds its place
// inside its bucket by counting the smaller ones (the reference's push order == ascending col<<16|row), fetches its unit ray
// from the table (all gathers of the workgroup in flight at once, up to 24 per lane) and leaves it at its sorted place -- and
// the pair loops read LDS only.  The raw items lie where the rays go afterwards, the rays wait in registers across the barrier
// in between: 79 KB of LDS, two workgroups per CU.  Cells are staged in runs of as many as fit 3072 items per camera (256 x 12:
// one run unless the lengths are 13 and more); a single cell beyond that is listed for the general kernel below.
// Batcher's odd-even merge sort for 16 keys: 63 compare-exchanges, as (a, b) pairs (checked against all 2^16 0/1 inputs when it
// was generated); every index is a template constant, so the keys stay in registers
template <unsigned A, unsigned B>
__device__ __forceinline__ void net_ce(uint32_t v[kNetLen])
{
    const uint32_t a = v[A], b = v[B];
    v[A] = a < b ? a : b;
    v[B] = a < b ? b : a;
}
template <unsigned... P>
struct NetPairs {
    static constexpr unsigned p[sizeof...(P)] = {P...};
    template <size_t... I>
    static __device__ __forceinline__ void run(uint32_t v[kNetLen], std::index_sequence<I...>)
    {
        const int dummy[] = {(net_ce<p[2 * I], p[2 * I + 1]>(v), 0)...};
        (void)dummy;
    }
};
template <unsigned... P>
__device__ __forceinline__ void net_run(uint32_t v[kNetLen])
{
    static_assert(sizeof...(P) == 126, "63 compare-exchanges");
    NetPairs<P...>::run(v, std::make_index_sequence<sizeof...(P) / 2>{});
}
__device__ __forceinline__ void sort16(uint32_t v[kNetLen])
{
    net_run<0, 1, 2, 3, 0, 2, 1, 3, 1, 2, 4, 5, 6, 7, 4, 6, 5, 7, 5, 6, 0, 4, 2, 6, 2, 4, 1, 5, 3, 7, 3, 5, 1, 2, 3, 4, 5, 6, 8, 9, 10, 11, 8, 10, 9, 11, 9, 10, 12, 13,
            14, 15, 12, 14, 13, 15, 13, 14, 8, 12, 10, 14, 10, 12, 9, 13, 11, 15, 11, 13, 9, 10, 11, 12, 13, 14, 0, 8, 4, 12, 4, 8, 2, 10, 6, 14, 6, 10, 2, 4,
            6, 8, 10, 12, 1, 9, 5, 13, 5, 9, 3, 11, 7, 15, 7, 11, 3, 5, 7, 9, 11, 13, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14>(v);
}

constexpr unsigned kTriCells = 256;                          // cells == threads per workgroup
constexpr unsigned kTriPer = 12;                             // staged items per camera and thread
constexpr unsigned kTriCap = kTriCells * kTriPer;

// The usual case: every bucket of a wave's 64 cells has up to 16 items.  Every cell lane takes its own two buckets: items into
// registers (16 loads in flight), sorting network, the rays of the sorted items (16 more).  The LEFT bucket's rays stay in
// registers -- the outer loop is unrolled over its 16 slots, and the lanes of a wave have equal bucket lengths, so a slot's test
// is a uniform branch -- and the RIGHT bucket's rays, which the inner loop walks, go to LDS slot-major (item k of lane j at
// k * 64 + j: a wave's access touches 64 different banks), where only the lane that wrote them reads them: LDS as indexable
// registers.  12 rows of it (9 KB per wave: 16 waves per CU at <= 128 VGPRs); the rays of items 12..15 of a long right bucket stay in
// registers and are picked by selects.  No barrier, no second copy of the items, no scan; a workgroup is ONE wave (nothing is
// shared, and the dispatcher balances at that grain).  (The first staged forms -- both buckets of 256 cells in LDS, 74-81 KB --
// ran 8 waves per CU, and their lanes waited half of their lives.)
constexpr unsigned kSmallCells = 64, kSmallRows = 12;
template <bool X87>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void ray_triangulate_small_kernel(
    const uint32_t *__restrict__ offs, const uint32_t *__restrict__ items, TriArgs cal, unsigned nb, int W, const float *__restrict__ raysL,
    const float *__restrict__ raysR, const uint32_t *__restrict__ order, const uint32_t *__restrict__ hist, float *__restrict__ xyz_sum,
    uint8_t *__restrict__ count)
{
    __shared__ float s_ray[3][kSmallRows * kSmallCells];     // [component][k * 64 + lane]
    const float pos[2][3] = {{cal.pos[0][0], cal.pos[0][1], cal.pos[0][2]}, {cal.pos[1][0], cal.pos[1][1], cal.pos[1][2]}};
    const unsigned t = threadIdx.x;
    const unsigned listed = hist[kBins], first = hist[kBins + 1];   // the list starts with the staged form's cells
    const unsigned g = first + blockIdx.x * kSmallCells + t;
    if (g - t >= listed) return;                             // (the grid covers every cell, the list only those with pairs)
    const bool mine = g < listed;
    unsigned b = 0, goff[2] = {0, 0}, len[2] = {0, 0};
    if (mine) {
        b = order[g];
        goff[0] = offs[b]; len[0] = offs[b + 1] - goff[0];
        goff[1] = offs[(size_t)nb + b]; len[1] = offs[(size_t)nb + b + 1] - goff[1];
    }
    CellSum acc;
    if (mine) {
        float ex[kNetLen - kSmallRows][3];                   // the right bucket's rays 12..15
        {
            uint32_t v[kNetLen];
#pragma unroll
            for (unsigned k = 0; k < kNetLen; k++) v[k] = items[goff[1] + k] | (k < len[1] ? 0u : 0xFFFFFFFFu);   // (no pixel has this item; `items` is padded)
            sort16(v);
            float rv[kSmallRows][3];                         // (slots behind the bucket's end fetch its first ray again: no branches)
#pragma unroll
            for (unsigned k = 0; k < kSmallRows; k++) load_ray(raysR, k < len[1] ? v[k] : v[0], W, rv[k]);
#pragma unroll
            for (unsigned k = kSmallRows; k < kNetLen; k++) load_ray(raysR, k < len[1] ? v[k] : v[0], W, ex[k - kSmallRows]);
#pragma unroll
            for (unsigned k = 0; k < kSmallRows; k++) {
                s_ray[0][k * kSmallCells + t] = rv[k][0];
                s_ray[1][k * kSmallCells + t] = rv[k][1];
                s_ray[2][k * kSmallCells + t] = rv[k][2];
            }
        }
        __builtin_amdgcn_sched_barrier(0);                   // (the right bucket's registers are free before the left one's are taken)
        float r1[kNetLen][3];                                // the left bucket's rays, in push order
        {
            uint32_t v[kNetLen];
#pragma unroll
            for (unsigned k = 0; k < kNetLen; k++) v[k] = items[goff[0] + k] | (k < len[0] ? 0u : 0xFFFFFFFFu);
            sort16(v);
#pragma unroll
            for (unsigned k = 0; k < kNetLen; k++) load_ray(raysL, k < len[0] ? v[k] : v[0], W, r1[k]);
        }
#pragma unroll
        for (unsigned k1 = 0; k1 < kNetLen; k1++) {
            if (k1 < len[0]) {
                float nxt[3] = {s_ray[0][t], s_ray[1][t], s_ray[2][t]};
                for (unsigned k2 = 0; k2 < len[1]; k2++) {
                    const float ray2[3] = {nxt[0], nxt[1], nxt[2]};
                    const unsigned kn = k2 + 1 < len[1] ? k2 + 1 : k2;           // the next pair's ray is on its way during this one
                    const unsigned at = (kn < kSmallRows ? kn : kSmallRows - 1) * kSmallCells + t;
                    nxt[0] = s_ray[0][at]; nxt[1] = s_ray[1][at]; nxt[2] = s_ray[2][at];
                    if (kn >= kSmallRows) {
#pragma unroll
                        for (int c = 0; c < 3; c++) nxt[c] = kn == 12 ? ex[0][c] : kn == 13 ? ex[1][c] : kn == 14 ? ex[2][c] : ex[3][c];
                    }
                    float X[3];
                    if (line_line<X87>(pos[0], r1[k1], pos[1], ray2, X)) acc.add(cal, X);
                }
            }
            __builtin_amdgcn_sched_barrier(0);               // (rows scheduled into each other want hundreds of registers)
        }
        xyz_sum[3 * (size_t)b] = acc.sum[0]; xyz_sum[3 * (size_t)b + 1] = acc.sum[1]; xyz_sum[3 * (size_t)b + 2] = acc.sum[2];
        count[b] = (uint8_t)acc.num;
    }
}

// The general form of one cell: buckets of any size, everything from global memory, one lane.  Only cells with more items than
// the staged form holds come here (a projector pixel seen by thousands of camera pixels: a decode gone wrong).
template <bool X87>
__device__ __noinline__ void cell_general(const uint32_t *__restrict__ offs, uint32_t *__restrict__ items, const TriArgs &cal, unsigned nb,
                                          unsigned b, int W, const float *__restrict__ raysL, const float *__restrict__ raysR, CellSum &acc)
{
    const float posL[3] = {cal.pos[0][0], cal.pos[0][1], cal.pos[0][2]}, posR[3] = {cal.pos[1][0], cal.pos[1][1], cal.pos[1][2]};
    const unsigned l0 = offs[b], l1 = offs[b + 1];
    const unsigned r0 = offs[nb + b], r1 = offs[nb + b + 1];
    if (l1 - l0 > 1) sort_bucket(items, l0, l1);
    if (r1 - r0 > 1) sort_bucket(items, r0, r1);
    for (unsigned c1 = l0; c1 < l1; c1++) {
        float ray1[3];
        load_ray(raysL, items[c1], W, ray1);
        for (unsigned c2 = r0; c2 < r1; c2++) {
            float ray2[3], X[3];
            load_ray(raysR, items[c2], W, ray2);
            if (line_line<X87>(posL, ray1, posR, ray2, X)) acc.add(cal, X);
        }
    }
}

template <bool X87>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void ray_triangulate_staged_kernel(const uint32_t *__restrict__ offs, uint32_t *__restrict__ items,
                                                              TriArgs cal, unsigned nb, int W,
                                                              const float *__restrict__ raysL, const float *__restrict__ raysR,
                                                              const uint32_t *__restrict__ order, const uint32_t *__restrict__ hist,
                                                              float *__restrict__ xyz_sum, uint8_t *__restrict__ count)
{
    __shared__ float s_ray[2][3][kTriCap];                   // [camera][component][sorted place]; before: the raw items
    __shared__ uint16_t s_off[2][kTriCells + 2];             // cell -> first place of its bucket inside the run (+ the run's end)
    __shared__ uint8_t s_cell[2][kTriCap];                   // staged item -> its cell
    __shared__ unsigned s_scan[2][4], s_first[2];
    const float pos[2][3] = {{cal.pos[0][0], cal.pos[0][1], cal.pos[0][2]}, {cal.pos[1][0], cal.pos[1][1], cal.pos[1][2]}};
    const unsigned t = threadIdx.x;
    const unsigned listed = hist[kBins + 1];                 // the cells with a bucket beyond kNetLen items: the head of the list
    // (a fixed small grid: with none of them -- the usual frame -- the launch is over at once)
  for (unsigned blk = blockIdx.x; blk * kTriCells < listed; blk += gridDim.x) {
    const unsigned g = blk * kTriCells + t;
    const unsigned nc = listed - blk * kTriCells < kTriCells ? listed - blk * kTriCells : kTriCells;
    const bool mine = blk * kTriCells + t < listed;
    unsigned b = 0, goff[2] = {0, 0}, len[2] = {0, 0}, pre[2];
    if (mine) {
        b = order[g];
        goff[0] = offs[b]; len[0] = offs[b + 1] - goff[0];
        goff[1] = offs[(size_t)nb + b]; len[1] = offs[(size_t)nb + b + 1] - goff[1];
    }
    const auto plus = [](unsigned x, unsigned y) { return x + y; };
    pre[0] = wg_exclusive_scan<256>(len[0], 0u, plus, s_scan[0], (unsigned *)nullptr);
    pre[1] = wg_exclusive_scan<256>(len[1], 0u, plus, s_scan[1], (unsigned *)nullptr);
    CellSum acc;
    for (unsigned c0 = 0; c0 < nc;) {
        if (t == c0) { s_first[0] = pre[0]; s_first[1] = pre[1]; }
        __syncthreads();
        const unsigned lo[2] = {pre[0] - s_first[0], pre[1] - s_first[1]};   // (meaningful for t >= c0)
        const unsigned hi[2] = {lo[0] + len[0], lo[1] + len[1]};
        const bool fits = t >= c0 && t < nc && hi[0] <= kTriCap && hi[1] <= kTriCap;
        const unsigned run = (unsigned)__syncthreads_count(fits);    // (fits is monotone in t: the count is the run's length)
        if (run == 0) {                                      // one cell beyond the capacity: the general form, on its own lane
            if (t == c0) cell_general<X87>(offs, items, cal, nb, b, W, raysL, raysR, acc);
            c0 += 1;
            continue;
        }
        const unsigned c1 = c0 + run;
        const bool in_run = t >= c0 && t < c1;
        {
            if (in_run) {
    #pragma unroll
                for (int cam = 0; cam < 2; cam++) {
                    uint32_t *raw = reinterpret_cast<uint32_t *>(&s_ray[cam][0][0]);
                    s_off[cam][t] = (uint16_t)lo[cam];
                    if (t == c1 - 1) s_off[cam][c1] = (uint16_t)hi[cam];
                    for (unsigned k = 0; k < len[cam]; k++) {
                        raw[lo[cam] + k] = items[goff[cam] + k];
                        s_cell[cam][lo[cam] + k] = (uint8_t)t;
                    }
                }
            }
            __syncthreads();
            // one camera after the other: its raw items are read (ranks, table addresses), barrier, its rays take their place
    #pragma unroll 1
            for (int cam = 0; cam < 2; cam++) {
                const uint32_t *raw = reinterpret_cast<const uint32_t *>(&s_ray[cam][0][0]);
                const float *tab = cam ? raysR : raysL;
                const unsigned n = s_off[cam][c1];
                float rv[kTriPer][3];
                unsigned place[kTriPer];
    #pragma unroll
                for (unsigned m = 0; m < kTriPer; m++) {
                    const unsigned q = t + kTriCells * m;
                    place[m] = ~0u;
                    if (q < n) {
                        const uint32_t v = raw[q];
                        const unsigned c = s_cell[cam][q];
                        const unsigned s = s_off[cam][c], e = s_off[cam][c + 1];
                        unsigned rank = 0;
                        for (unsigned j = s; j < e; j++) rank += raw[j] < v ? 1u : 0u;   // (items of a bucket are distinct pixels)
                        place[m] = s + rank;
                        load_ray(tab, v, W, rv[m]);
                    }
                }
                __syncthreads();                                 // every raw item of this camera has been read
    #pragma unroll
                for (unsigned m = 0; m < kTriPer; m++)
                    if (place[m] != ~0u) {
                        s_ray[cam][0][place[m]] = rv[m][0];
                        s_ray[cam][1][place[m]] = rv[m][1];
                        s_ray[cam][2][place[m]] = rv[m][2];
                    }
            }
            __syncthreads();
        }
        if (in_run && hi[1] > lo[1])
            for (unsigned k1 = lo[0]; k1 < hi[0]; k1++) {
                const float ray1[3] = {s_ray[0][0][k1], s_ray[0][1][k1], s_ray[0][2][k1]};
                float nxt[3] = {s_ray[1][0][lo[1]], s_ray[1][1][lo[1]], s_ray[1][2][lo[1]]};
                for (unsigned k2 = lo[1]; k2 < hi[1]; k2++) {
                    const float ray2[3] = {nxt[0], nxt[1], nxt[2]};
                    const unsigned kn = k2 + 1 < hi[1] ? k2 + 1 : k2;            // the next pair's ray is on its way during this one
                    nxt[0] = s_ray[1][0][kn]; nxt[1] = s_ray[1][1][kn]; nxt[2] = s_ray[1][2][kn];
                    float X[3];
                    if (line_line<X87>(pos[0], ray1, pos[1], ray2, X)) acc.add(cal, X);
                }
            }
        c0 = c1;
        __syncthreads();                                     // (the next run overwrites the rays and s_first)
    }
    if (mine) {
        xyz_sum[3 * (size_t)b] = acc.sum[0]; xyz_sum[3 * (size_t)b + 1] = acc.sum[1]; xyz_sum[3 * (size_t)b + 2] = acc.sum[2];
        count[b] = (uint8_t)acc.num;
    }
    __syncthreads();                                         // (the next block of the list reuses everything)
  }
}

// order[cells], hist[kBins + 2], cursor[kBins]
size_t ray_list_words(size_t cells) { return cells + 2 * (size_t)kBins + 2; }
constexpr size_t kRayItemsPad = kNetLen;                     // the small kernel reads 16 items from every bucket start
size_t ray_items_words(size_t cam_pixels) { return 2 * cam_pixels + kRayItemsPad; }

// pass 2 of both cameras (items[offs[cell] + rank]) with the work list of launch_ray_triangulate made beside it (and the zeros of
// the cells without pairs written)
hipError_t launch_ray_scatter_list(const uint32_t *cellL, const uint32_t *rankL, const uint32_t *cellR, const uint32_t *rankR, int W, int H,
                                   const uint32_t *offs, uint32_t *items, int scan_w, int scan_h, uint32_t *list, float *xyz_sum,
                                   uint8_t *count, hipStream_t s)
{
    const size_t nb = (size_t)scan_w * scan_h;
    uint32_t *order = list, *hist = list + nb, *cursor = hist + kBins + 2;
    hipError_t e = hipMemsetAsync(hist, 0, (2 * (size_t)kBins + 2) * 4, s);   // hist, cursor
    if (e != hipSuccess) return e;
    const unsigned lb = (unsigned)((nb + kKeyCells - 1) / kKeyCells), sb = (unsigned)(((size_t)W + 255) / 256 * (size_t)H);
    SLR_LAUNCH(ray_scatter_list_kernel<0>, dim3(lb + sb), dim3(256), 0, s, cellL, rankL, W, offs, items, offs, (unsigned)nb, hist, cursor, order,
               xyz_sum, count, lb);
    SLR_LAUNCH(ray_scatter_list_kernel<1>, dim3(lb + sb), dim3(256), 0, s, cellR, rankR, W, offs + nb, items, offs, (unsigned)nb, hist, cursor, order,
               xyz_sum, count, lb);
    return hipGetLastError();
}

hipError_t launch_ray_triangulate(const uint32_t *offs, uint32_t *items, const DevCalib &cal, int scan_w, int scan_h,
                                  int W, const float *raysL, const float *raysR, const uint32_t *list, float *xyz_sum, uint8_t *count,
                                  hipStream_t s)
{
    const size_t nb = (size_t)scan_w * scan_h;
    TriArgs ta;
    for (int i = 0; i < 12; i++) ta.T[i] = (double)cal.T[i];
    for (int c = 0; c < 2; c++) { ta.pos[c][0] = ta.pos[c][1] = ta.pos[c][2] = 0.0f; cam2world(cal.cam[c], ta.pos[c]); }
    ta.has_T = cal.has_T;
    const uint32_t *order = list, *hist = list + nb;
    const unsigned sb = (unsigned)((nb + kSmallCells - 1) / kSmallCells);
    if (cal.eval_x87) {                                      // SLR_OPT_EVAL_MODEL = 1 (the ray tables were built under it as well)
        SLR_LAUNCH(ray_triangulate_small_kernel<true>, dim3(sb), dim3(kSmallCells), 0, s, offs, (const uint32_t *)items, ta, (unsigned)nb, W, raysL,
                   raysR, order, hist, xyz_sum, count);
        SLR_LAUNCH(ray_triangulate_staged_kernel<true>, dim3(256), dim3(kTriCells), 0, s, offs, items, ta, (unsigned)nb, W, raysL, raysR, order, hist,
                   xyz_sum, count);
        return hipGetLastError();
    }
    SLR_LAUNCH(ray_triangulate_small_kernel<false>, dim3(sb), dim3(kSmallCells), 0, s, offs, (const uint32_t *)items, ta, (unsigned)nb, W, raysL, raysR,
               order, hist, xyz_sum, count);
    SLR_LAUNCH(ray_triangulate_staged_kernel<false>, dim3(256), dim3(kTriCells), 0, s, offs, items, ta, (unsigned)nb, W, raysL, raysR, order, hist,
               xyz_sum, count);
    return hipGetLastError();
}

// ---- PointCloudImage adaptors --------------------------------------------------------------------------
// addPoint(i=row, j=col, p) against PointCloudImage(w=scan_w, h=scan_h): points[j][i] iff i<scan_w && j<scan_h
// (pointcloudimage.cpp:88).  Gather form: one thread per PointCloudImage cell.
__global__ __launch_bounds__(256) void pc_from_grid_kernel(const float *__restrict__ xyz, const uint8_t *__restrict__ has,
                                                           const uint8_t *__restrict__ color, int W, int H, int scan_w,
                                                           int scan_h, float *__restrict__ pc_sum,
                                                           uint8_t *__restrict__ pc_count, uint8_t *__restrict__ pc_color)
{
    const unsigned total = (unsigned)scan_w * (unsigned)scan_h;
    for (unsigned d = blockIdx.x * 256u + threadIdx.x; d < total; d += gridDim.x * 256u) {
        const unsigned j = d / scan_w, i = d - j * scan_w;   // cell (row=j_h, col=i_w)
        float x = 0, y = 0, z = 0;
        uint8_t c = 0, col = 0;
        if (i < (unsigned)H && j < (unsigned)W) {            // source pixel (row i, col j)
            const size_t o = (size_t)i * W + j;
            if (has[o]) {
                x = xyz[3 * o]; y = xyz[3 * o + 1]; z = xyz[3 * o + 2];
                c = 1;
                if (color) col = color[o];
            }
        }
        pc_sum[3 * (size_t)d] = x; pc_sum[3 * (size_t)d + 1] = y; pc_sum[3 * (size_t)d + 2] = z;
        pc_count[d] = c;
        if (pc_color) pc_color[d] = col;
    }
}

hipError_t launch_pc_from_grid(const float *xyz, const uint8_t *has, const uint8_t *color, int W, int H, int scan_w,
                               int scan_h, float *pc_sum, uint8_t *pc_count, uint8_t *pc_color, hipStream_t s)
{
    const size_t n = (size_t)scan_w * scan_h;
    const unsigned blocks = (unsigned)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
    SLR_LAUNCH(pc_from_grid_kernel, dim3(blocks), dim3(256), 0, s, xyz, has, color, W, H, scan_w, scan_h,
                       pc_sum, pc_count, pc_color);
    return hipGetLastError();
}

// getPoint: Vec3d(sum) / (float)num narrowed to Point3f (pointcloudimage.cpp:62)
__global__ __launch_bounds__(256) void pc_get_kernel(const float *__restrict__ pc_sum, const uint8_t *__restrict__ pc_count,
                                                     size_t n, float *__restrict__ out)
{
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256ull) {
        const unsigned c = pc_count[i];
        float x = 0, y = 0, z = 0;
        if (c) {
            const double dn = (double)(float)c;
            x = (float)((double)pc_sum[3 * i] / dn);
            y = (float)((double)pc_sum[3 * i + 1] / dn);
            z = (float)((double)pc_sum[3 * i + 2] / dn);
        }
        out[3 * i] = x; out[3 * i + 1] = y; out[3 * i + 2] = z;
    }
}

hipError_t launch_pc_get(const float *pc_sum, const uint8_t *pc_count, size_t n, float *out, hipStream_t s)
{
    const unsigned blocks = (unsigned)((n + 255) / 256 < 8192 ? ((n + 255) / 256 ? (n + 255) / 256 : 1) : 8192);
    SLR_LAUNCH(pc_get_kernel, dim3(blocks), dim3(256), 0, s, pc_sum, pc_count, n, out);
    return hipGetLastError();
}

}  // namespace slr
