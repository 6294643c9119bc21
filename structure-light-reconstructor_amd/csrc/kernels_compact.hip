// kernels_compact.hip -- ordered stream compaction: exclusive prefix index of a u8 flag image (row-major or in the
// column-major enumeration order the reference's MeshCreator walks a PointCloudImage in) and the compaction of the valid
// points of an XYZ grid.  gfx950 (MI355X) only.  Serves
//   * the mesh writer's vertex numbering (MeshCreator::exportPlyMesh / exportObjMesh, Duke/meshcreator.cpp:16-166: a vertex
//     per pixel with a point, numbered in i-outer / j-inner order) -- slr_prefix_index;
//   * the point-cloud assembly of the multi-GPU entry (only the valid points travel over xGMI) -- slr_compact_points.
// Three small kernels, no library scan: per-1024-element block counts (wave ballots), one workgroup scanning the block
// sums, and the index / scatter pass that redoes the ballots.  HBM-bound streaming: 1 B read + 4 B written per element.
#include "slr_device.hpp"

namespace slr {

constexpr int kScanBlock = 256, kScanPer = 4, kScanElems = kScanBlock * kScanPer;

// element e of the enumeration -> its position in the [h][w] row-major image
__device__ __forceinline__ size_t scan_pos(size_t e, int w, int h, int column_major)
{
    if (!column_major) return e;
    const size_t i = e / (size_t)h, j = e - i * (size_t)h;   // i: column (outer), j: row (inner)
    return j * (size_t)w + i;
}

// flags of the 4 consecutive elements of a thread, as a 4-bit mask; counts through wave ballots
__global__ __launch_bounds__(kScanBlock) void flag_count_kernel(const uint8_t *__restrict__ flags, size_t n, int w, int h,
                                                                int column_major, uint32_t *__restrict__ block_sums)
{
    __shared__ unsigned wsum[kScanBlock / 64];
    const size_t e0 = (size_t)blockIdx.x * kScanElems + (size_t)threadIdx.x * kScanPer;
    unsigned c = 0;
#pragma unroll
    for (int q = 0; q < kScanPer; q++)
        if (e0 + q < n && flags[scan_pos(e0 + q, w, h, column_major)]) c++;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) c += __shfl_xor(c, d);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) block_sums[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// exclusive scan of the block sums in place (one workgroup; nblocks is a few tens of thousands at most); total -> *total
__global__ __launch_bounds__(1024) void block_scan_kernel(uint32_t *__restrict__ block_sums, unsigned nblocks, uint32_t *__restrict__ total)
{
    __shared__ unsigned wtot[16];
    __shared__ unsigned carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (unsigned base = 0; base < nblocks; base += 1024) {
        const unsigned i = base + threadIdx.x;
        const unsigned v = i < nblocks ? block_sums[i] : 0u;
        unsigned incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned t = __shfl_up(incl, d);
            if (lane >= d) incl += t;
        }
        if (lane == 63) wtot[wv] = incl;
        __syncthreads();
        unsigned woff = 0;
        for (int k = 0; k < wv; k++) woff += wtot[k];
        const unsigned carry = carry_s;
        if (i < nblocks) block_sums[i] = carry + woff + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = carry + woff + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry_s;
}

// index[pos] = first + (number of flagged elements before e in enumeration order) for flagged elements, `none` otherwise;
// with xyz != null also out_xyz[k] = xyz[pos] and out_src[k] = pos for the k-th flagged element
__global__ __launch_bounds__(kScanBlock) void flag_index_kernel(const uint8_t *__restrict__ flags, size_t n, int w, int h, int column_major,
                                                                const uint32_t *__restrict__ block_excl, uint32_t first, uint32_t none,
                                                                uint32_t *__restrict__ index, const float *__restrict__ xyz,
                                                                float *__restrict__ out_xyz, uint32_t *__restrict__ out_src)
{
    __shared__ unsigned wsum[kScanBlock / 64];
    const size_t e0 = (size_t)blockIdx.x * kScanElems + (size_t)threadIdx.x * kScanPer;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    size_t pos[kScanPer];
    unsigned f[kScanPer], c = 0;
#pragma unroll
    for (int q = 0; q < kScanPer; q++) {
        pos[q] = e0 + q < n ? scan_pos(e0 + q, w, h, column_major) : 0;
        f[q] = e0 + q < n && flags[pos[q]] ? 1u : 0u;
        c += f[q];
    }
    unsigned incl = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned t = __shfl_up(incl, d);
        if (lane >= d) incl += t;
    }
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    unsigned k = block_excl[blockIdx.x] + incl - c;
    for (int u = 0; u < wv; u++) k += wsum[u];
#pragma unroll
    for (int q = 0; q < kScanPer; q++) {
        if (e0 + q >= n) break;
        if (index) index[pos[q]] = f[q] ? first + k : none;
        if (f[q]) {
            if (out_xyz) { out_xyz[3 * (size_t)k] = xyz[3 * pos[q]]; out_xyz[3 * (size_t)k + 1] = xyz[3 * pos[q] + 1]; out_xyz[3 * (size_t)k + 2] = xyz[3 * pos[q] + 2]; }
            if (out_src) out_src[k] = (uint32_t)pos[q];
            k++;
        }
    }
}

size_t flag_scan_temp_bytes(size_t n) { return ((n + kScanElems - 1) / kScanElems + 2) * sizeof(uint32_t); }

// temp: flag_scan_temp_bytes(n); the total lands in temp[nblocks] (device) -- the caller copies it out
hipError_t launch_flag_scan(const uint8_t *flags, size_t n, int w, int h, int column_major, uint32_t first, uint32_t none,
                            uint32_t *index, const float *xyz, float *out_xyz, uint32_t *out_src, void *temp,
                            uint32_t **total_dev, hipStream_t s)
{
    const unsigned nblocks = (unsigned)((n + kScanElems - 1) / kScanElems);
    uint32_t *sums = reinterpret_cast<uint32_t *>(temp);
    *total_dev = sums + nblocks;
    if (n == 0) return hipMemsetAsync(*total_dev, 0, sizeof(uint32_t), s);
    SLR_LAUNCH(flag_count_kernel, dim3(nblocks), dim3(kScanBlock), 0, s, flags, n, w, h, column_major, sums);
    SLR_LAUNCH(block_scan_kernel, dim3(1), dim3(1024), 0, s, sums, nblocks, *total_dev);
    SLR_LAUNCH(flag_index_kernel, dim3(nblocks), dim3(kScanBlock), 0, s, flags, n, w, h, column_major, (const uint32_t *)sums, first, none,
               index, xyz, out_xyz, out_src);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------
// One 64-bit checksum per frame of an assembled cloud (slr_cloud_checksums): position-weighted sums of the raw XYZ words and of
// the mask bytes, wrapping arithmetic.  Used to PROVE an exchange: every device checksums what it holds after the peer copies /
// the all-gather, the host compares the words.  Pure streaming (13 B per pixel read), one launch for all frames.
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cloud_checksum_kernel(const unsigned *__restrict__ xyz, const uint8_t *__restrict__ has,
                                                             size_t n_px, unsigned long long *__restrict__ out)
{
    const unsigned f = blockIdx.y;
    const unsigned *x = xyz + (size_t)f * n_px * 3;
    const uint8_t *h = has + (size_t)f * n_px;
    unsigned long long acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < n_px; i += (size_t)gridDim.x * 256u) {
        const unsigned long long w = (unsigned long long)(i % 65521u) + 1ull;
        acc += w * ((unsigned long long)x[3 * i] + 3ull * x[3 * i + 1] + 7ull * x[3 * i + 2]) + 1000003ull * w * h[i];
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) acc += __shfl_xor(acc, d);
    if ((threadIdx.x & 63) == 0) atomicAdd(out + f, acc);
}

hipError_t launch_cloud_checksums(const float *xyz, const uint8_t *has, int n_frames, size_t n_px, unsigned long long *d_out, hipStream_t s)
{
    hipError_t e = hipMemsetAsync(d_out, 0, sizeof(unsigned long long) * (size_t)n_frames, s);
    if (e != hipSuccess || n_frames == 0) return e;
    const unsigned bx = (unsigned)((n_px + 255) / 256 < 2048 ? (n_px + 255) / 256 : 2048);
    SLR_LAUNCH(cloud_checksum_kernel, dim3(bx ? bx : 1, (unsigned)n_frames), dim3(256), 0, s, reinterpret_cast<const unsigned *>(xyz), has,
               n_px, d_out);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------
// The box's streaming rate, measured the way MI355X_MICROARCH.md quotes it ("float4 copy": 6.29 TB/s): 16 bytes per lane,
// non-temporal both ways, ONE float4 per thread (round 6: the launch shape profiles/exp/r05/bw.txt found fastest on this part --
// 6.6 TB/s where a 2048-workgroup grid-stride loop copies at 5.2-5.5) -- and the same with the fused decode's read : write mix
// (stream_mix_kernel: `reads` 16-byte words read from `reads` streams per 16-byte word written; reads = 5 is the MF decode's
// 20 : 4 bytes per camera pixel).  bench.py reports the decode's rate against both figures next to the 8 TB/s peak (a library
// memcpy is NOT a ceiling: torch's copy_ ran at 5.1 TB/s where the fused decode's own unfused form streams 5.5).
// ------------------------------------------------------------------------------------------------------
typedef float f32x4_t __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void stream_copy_kernel(const f32x4_t *__restrict__ s4, f32x4_t *__restrict__ d4, size_t n16)
{
    const size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (i < n16) __builtin_nontemporal_store(__builtin_nontemporal_load(s4 + i), d4 + i);
}

template <int READS>
__global__ __launch_bounds__(256) void stream_mix_kernel(const f32x4_t *__restrict__ s4, f32x4_t *__restrict__ d4, size_t n16)
{
    const size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= n16) return;
    f32x4_t v[READS];
#pragma unroll
    for (int r = 0; r < READS; r++) v[r] = __builtin_nontemporal_load(s4 + (size_t)r * n16 + i);     // all loads in flight, then the sum
    f32x4_t a = v[0];
#pragma unroll
    for (int r = 1; r < READS; r++) a += v[r];
    __builtin_nontemporal_store(a, d4 + i);
}

hipError_t launch_stream_copy(const void *src, void *dst, size_t bytes, hipStream_t s)
{
    const size_t n16 = bytes / 16;
    const size_t grid = (n16 + 255) / 256;
    if (grid == 0) return hipSuccess;
    if (grid > 0x7fffffffu) return hipErrorInvalidValue;
    SLR_LAUNCH(stream_copy_kernel, dim3((unsigned)grid), dim3(256), 0, s, reinterpret_cast<const f32x4_t *>(src), reinterpret_cast<f32x4_t *>(dst), n16);
    return hipGetLastError();
}

hipError_t launch_stream_mix(const void *src, void *dst, size_t bytes_out, int reads, hipStream_t s)
{
    const size_t n16 = bytes_out / 16;
    const size_t grid = (n16 + 255) / 256;
    if (grid == 0) return hipSuccess;
    if (grid > 0x7fffffffu) return hipErrorInvalidValue;
    const f32x4_t *s4 = reinterpret_cast<const f32x4_t *>(src);
    f32x4_t *d4 = reinterpret_cast<f32x4_t *>(dst);
    switch (reads) {
        case 1: SLR_LAUNCH(stream_mix_kernel<1>, dim3((unsigned)grid), dim3(256), 0, s, s4, d4, n16); break;
        case 2: SLR_LAUNCH(stream_mix_kernel<2>, dim3((unsigned)grid), dim3(256), 0, s, s4, d4, n16); break;
        case 3: SLR_LAUNCH(stream_mix_kernel<3>, dim3((unsigned)grid), dim3(256), 0, s, s4, d4, n16); break;
        case 4: SLR_LAUNCH(stream_mix_kernel<4>, dim3((unsigned)grid), dim3(256), 0, s, s4, d4, n16); break;
        case 5: SLR_LAUNCH(stream_mix_kernel<5>, dim3((unsigned)grid), dim3(256), 0, s, s4, d4, n16); break;
        case 6: SLR_LAUNCH(stream_mix_kernel<6>, dim3((unsigned)grid), dim3(256), 0, s, s4, d4, n16); break;
        case 7: SLR_LAUNCH(stream_mix_kernel<7>, dim3((unsigned)grid), dim3(256), 0, s, s4, d4, n16); break;
        case 8: SLR_LAUNCH(stream_mix_kernel<8>, dim3((unsigned)grid), dim3(256), 0, s, s4, d4, n16); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace slr
