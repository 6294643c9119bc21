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
// Buckets are numbered by their PointCloudImage cell  j*scan_w + i  (bucket ac = i*scan_h + j  <->  cell (j,i)),
// so the triangulation kernel reads its ranges and writes its sums fully coalesced.
#include "slr_device.hpp"

#include <hipcub/hipcub.hpp>
#include <math.h>

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
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, (const uint32_t *)nullptr, (uint32_t *)nullptr, (int)n);
    return bytes;
}

hipError_t launch_ray_scan(const uint32_t *cnt, uint32_t *offs, size_t n, void *temp, size_t temp_bytes, hipStream_t s)
{
    return hipcub::DeviceScan::ExclusiveSum(temp, temp_bytes, cnt, offs, (int)n, s);
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
__device__ __forceinline__ void cam2world(const DevCamera &c, float p[3])
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

__device__ __forceinline__ void pixel_ray(uint32_t item, const DevCamera &c, const float pos[3], float ray[3])
{
    float ux, uy, pt[3];
    undistort_point_ray((float)(item >> 16), (float)(item & 0xFFFFu), c, ux, uy);   // item = col<<16 | row
    pt[0] = (ux - c.ccx) / c.fcx;                  // utilities.cpp:51-53
    pt[1] = (uy - c.ccy) / c.fcy;
    pt[2] = 1.0f;
    cam2world(c, pt);
    ray[0] = pos[0] - pt[0]; ray[1] = pos[1] - pt[1]; ray[2] = pos[2] - pt[2];
    // utilities.cpp:21-25: sqrt(float) overload, max(0.000001, mag) in f64, divide by the f32 narrowing
    const double mag = (double)sqrtf(ray[0] * ray[0] + ray[1] * ray[1] + ray[2] * ray[2]);
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
    pixel_ray((col << 16) | row, cal.cam[0], posL, r);
    raysL[o] = r[0]; raysL[o + 1] = r[1]; raysL[o + 2] = r[2];
    pixel_ray((col << 16) | row, cal.cam[1], posR, r);
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

// utilities.cpp:399-425
__device__ __forceinline__ bool line_line(const float p1[3], const float v1[3], const float p2[3], const float v2[3],
                                          float out[3])
{
    const float v12[3] = {p1[0] - p2[0], p1[1] - p2[1], p1[2] - p2[2]};
    const float a = dot3(v1, v1), c = dot3(v2, v2), b = dot3(v1, v2);
    const float d = dot3(v12, v1), e = dot3(v12, v2);
    const float denom = a * c - b * b;
    if (fabsf(denom) < 0.1f) return false;
    const float s = (b / denom) * e - (c / denom) * d;
    const float t = -(b / denom) * d + (a / denom) * e;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float u = p1[k] + s * v1[k];
        const float w = p2[k] + t * v2[k];
        out[k] = 0.5f * (u + w);
    }
    return true;
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

// one thread per projector pixel (bucket == PointCloudImage cell j*scan_w + i)
__global__ __launch_bounds__(256) void ray_triangulate_kernel(const uint32_t *__restrict__ offs, uint32_t *__restrict__ items,
                                                              DevCalib cal, unsigned nb, int W,
                                                              const float *__restrict__ raysL, const float *__restrict__ raysR,
                                                              float *__restrict__ xyz_sum, uint8_t *__restrict__ count)
{
    // the right bucket's rays are read once per LEFT pixel of the bucket: the first kRayCap of them wait in LDS ([ray][thread]:
    // conflict-free), fetched once per bucket, instead of being gathered from the table (len_L x len_R) times
    constexpr unsigned kRayCap = 12;
    __shared__ float stage[kRayCap][3][256];
    float posL[3] = {0, 0, 0}, posR[3] = {0, 0, 0};
    cam2world(cal.cam[0], posL);                   // reconstruct.cpp:239-240
    cam2world(cal.cam[1], posR);
    for (unsigned b = blockIdx.x * 256u + threadIdx.x; b < nb; b += gridDim.x * 256u) {
        float sum[3] = {0.0f, 0.0f, 0.0f};
        unsigned num = 0;
        const unsigned l0 = offs[b], l1 = offs[b + 1];
        const unsigned r0 = offs[nb + b], r1 = offs[nb + b + 1];
        if (l1 > l0 && r1 > r0) {
            if (l1 - l0 > 1) sort_bucket(items, l0, l1);
            if (r1 - r0 > 1) sort_bucket(items, r0, r1);
            const unsigned nst = r1 - r0 < kRayCap ? r1 - r0 : kRayCap;
            for (unsigned k = 0; k < nst; k++) {
                float ry[3];
                load_ray(raysR, items[r0 + k], W, ry);
                stage[k][0][threadIdx.x] = ry[0]; stage[k][1][threadIdx.x] = ry[1]; stage[k][2][threadIdx.x] = ry[2];
            }
            for (unsigned c1 = l0; c1 < l1; c1++) {
                float ray1[3];
                load_ray(raysL, items[c1], W, ray1);
                for (unsigned c2 = r0; c2 < r1; c2++) {
                    float ray2[3], X[3];
                    if (c2 - r0 < kRayCap) {
                        ray2[0] = stage[c2 - r0][0][threadIdx.x]; ray2[1] = stage[c2 - r0][1][threadIdx.x]; ray2[2] = stage[c2 - r0][2][threadIdx.x];
                    } else load_ray(raysR, items[c2], W, ray2);
                    if (!line_line(posL, ray1, posR, ray2, X)) continue;
                    if (cal.has_T) {
                        float Y[3];
#pragma unroll
                        for (int r = 0; r < 3; r++) {
                            double s = 0;
                            s += (double)cal.T[r * 4 + 0] * (double)X[0];
                            s += (double)cal.T[r * 4 + 1] * (double)X[1];
                            s += (double)cal.T[r * 4 + 2] * (double)X[2];
                            s += (double)cal.T[r * 4 + 3] * 1.0;
                            Y[r] = (float)s;
                        }
                        X[0] = Y[0]; X[1] = Y[1]; X[2] = Y[2];
                    }
                    if (num == 0) { sum[0] = X[0]; sum[1] = X[1]; sum[2] = X[2]; num = 1; }   // setPoint
                    else {                                                                    // addPoint
                        sum[0] = X[0] + sum[0]; sum[1] = X[1] + sum[1]; sum[2] = X[2] + sum[2];
                        num = (num + 1) & 0xFFu;           // uchar counter wraps (pointcloudimage.cpp:95)
                    }
                }
            }
        }
        xyz_sum[3 * (size_t)b] = sum[0]; xyz_sum[3 * (size_t)b + 1] = sum[1]; xyz_sum[3 * (size_t)b + 2] = sum[2];
        count[b] = (uint8_t)num;
    }
}

hipError_t launch_ray_triangulate(const uint32_t *offs, uint32_t *items, const DevCalib &cal, int scan_w, int scan_h,
                                  int W, const float *raysL, const float *raysR, float *xyz_sum, uint8_t *count,
                                  hipStream_t s)
{
    const size_t nb = (size_t)scan_w * scan_h;
    const unsigned blocks = (unsigned)((nb + 255) / 256 < 16384 ? (nb + 255) / 256 : 16384);
    SLR_LAUNCH(ray_triangulate_kernel, dim3(blocks), dim3(256), 0, s, offs, items, cal, (unsigned)nb, W, raysL,
                       raysR, xyz_sum, count);
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
