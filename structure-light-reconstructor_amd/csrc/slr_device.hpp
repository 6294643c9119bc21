// slr_device.hpp -- shared declarations between the HIP kernels and the C-ABI layer (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include <atomic>

#include "slr.h"

namespace slr {

// mfreconstruct.cpp:5 "float PI = 3.1416;" -- an f32 that is not pi.
constexpr float kPI = 3.1416f;
constexpr float kTwoPI = 2 * kPI;          // f32 expression, as in the reference (2*PI)
constexpr float kThreeHalfPI = 3 * kPI / 2;
constexpr float kHalfPI = kPI / 2;
constexpr int kDecodeLutWords = 512 + 10 * 256 + 1;   // 511 reciprocal/sign words + the wrapped-phase table (kernels_decode.hip)

// Per-device cache of a launch constant (resident workgroups of a kernel, CU count), filled on first use.  Host threads driving
// different contexts may race to fill a slot: they store the same value.  Devices beyond the table are looked up on every launch.
struct DevSlots {
    std::atomic<int> v[64];
    int get(int dev) const { return (unsigned)dev < 64u ? v[dev].load(std::memory_order_relaxed) : 0; }
    void put(int dev, int x) { if ((unsigned)dev < 64u) v[dev].store(x, std::memory_order_relaxed); }
};

struct MfPlanes { const uint8_t *p[SLR_MF_PLANES]; };
struct GrayPlanes { const uint8_t *p[SLR_MAX_GRAY_PLANES]; };

// calibration as the kernels consume it (by-value kernel argument)
struct DevCamera {
    double fx, fy, ifx, ify, cx, cy;   // utilities.cpp:68-73 (f32 values widened; ifx = 1./fx in f64)
    double k0, k1, k2, k3;             // utilities.cpp:62-65
    float fcx, fcy, ccx, ccy;          // f32 originals for pixelToImageSpace (utilities.cpp:51-52)
    float R[9], t[3];
};
struct DevCalib {
    DevCamera cam[2];
    double Q[16];
    float T[12];
    int has_T;
    int q_simple;                      // Q has cv::stereoRectify's zero/one pattern (kernels_match.hip reproject)
    int eval_x87;                      // SLR_OPT_EVAL_MODEL = 1: match predicate and disparity as the reference's x87 binary forms them
};

// kernel ids for the built-in HIP-event profiler (bench.py reads these)
enum KernelId {
    K_REMAP = 0, K_MF_DECODE, K_MF_RECT_DECODE, K_GRAY_DECODE, K_GRAY_RECT_DECODE,
    K_MF_MATCH, K_GE_MATCH, K_RAY_COUNT, K_RAY_SCAN, K_RAY_SCATTER, K_RAY_TRI, K_PC_FROM_GRID, K_PC_GET, K_UNDISTORT_TABLE, K_RAY_TABLE, K_MF_RECT_DECODE_PAIR, K_MFN_DECODE, K_GRAY_RECT_DECODE_PAIR, K_HYBRID_RECT_DECODE_PAIR, K_MFN_RECT_DECODE, K_COUNT
};

// Every kernel launch goes through SLR_LAUNCH.  When the C-ABI layer's profiler has armed a pair of events for the
// calling thread (ProfScope, slr_capi.hip) the launch is a hipExtLaunchKernelGGL that stamps them at the kernel's own start
// and end -- the same duration rocprofv3's kernel trace reports -- instead of two hipEventRecord around the launch, which
// also time the launch gap (4-5 % of a 0.2 ms kernel) and serialise the stream with barrier packets.
extern thread_local hipEvent_t tl_prof_start, tl_prof_stop;

// Test / tuning knobs of the launchers.  They are ctx fields (slr_set_option: SLR_OPT_DEBUG_RECT_RESIDENT, SLR_OPT_DEBUG_FLAGS) that the
// C-ABI layer publishes to the calling thread before every call -- never the process environment: a stray variable in a user's
// shell must not change which kernel form runs.  k4_stop (K4 phase ablation: the kernel returns early WITHOUT writing its
// outputs) only exists in builds with -DSLR_DEBUG_HOOKS (profiles/k4_stages.sh).
struct DebugKnobs {
    int rect_resident = 0;       // > 0: resident workgroups of the persistent fused decodes (tests: many tiles per workgroup)
    bool no_tiled_map = false;   // fused decode: read the caller's 6-byte map entries instead of the digest
    bool no_buffer_form = false; // fused decode: per-plane pointers instead of one buffer descriptor
    bool no_ge_lean = false;     // K5: the general kernel for every row
    bool no_decode_count = false;// GRAY_ONLY: separate decode and bucket-histogram kernels
    bool no_quad_sort = false;   // LDS-DMA fused decodes: every wave keeps the quads of its own block of the tile (read when maps are installed)
    bool no_promote = false;     // LDS-DMA fused decodes: straddling waves keep the three-row read mode (no promotion to per-pixel reads)
    bool gray_small_tiles = false;// fused Gray decode, LDS-tiled form: 64 x 4 tiles whatever the plane count (else: 42 planes and more)
    bool mfn_nt512 = false;      // config 5, LDS-DMA rectified decode: 512 threads x 2 pixels instead of 256 x 4
    bool mfn_ring4 = false;      // config 5, LDS-DMA rectified decode: a ring of 4 plane groups (two workgroups per CU) instead of 3
    int k4_stop = 0;
    bool poison_scratch = false; // SLR_OPT_DEBUG_POISON_SCRATCH
    bool eval_x87 = false;       // SLR_OPT_EVAL_MODEL = 1 (not a debug knob, but it travels the same way: per call, per thread)
};
extern thread_local DebugKnobs tl_debug;
#define SLR_LAUNCH(kernel, grid, block, lds, stream, ...)                                                            \
    do {                                                                                                             \
        if (::slr::tl_prof_start) {                                                                                  \
            hipExtLaunchKernelGGL(kernel, grid, block, (uint32_t)(lds), stream, ::slr::tl_prof_start, ::slr::tl_prof_stop, 0, \
                                  __VA_ARGS__);                                                                      \
            ::slr::tl_prof_start = nullptr;                                                                          \
        } else {                                                                                                     \
            hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                                       \
        }                                                                                                            \
    } while (0)

// ---- launchers (defined in the .hip files; all asynchronous on `s`) -----------------------------------
hipError_t launch_remap_u8(const uint8_t *src, int src_pitch, uint8_t *dst, int dst_pitch, int W, int H,
                           const int16_t *map_xy, const uint16_t *map_frac, hipStream_t s);

hipError_t launch_mf_decode(const MfPlanes &pl, int pitch, int W, int H, int black_thr,
                            const float *atan_lut, float *phase, uint8_t *valid,
                            const int16_t *map_xy, const uint16_t *map_frac /* null -> no rectify */,
                            const void *tile_boxes /* launch_tile_boxes output for this map, or null */,
                            int vec_hint /* 0 auto, 4/8/16 pixels per thread (tuning) */,
                            int rect_algo /* SLR_OPT_RECT_DECODE_ALGO, already resolved (0 is treated as 6) */, hipStream_t s);

// both cameras of a stereo frame in one launch (LDS-tiled fused form only; *done = false -> not applicable, nothing
// was launched)
hipError_t launch_mf_rect_decode_pair(const MfPlanes pl[2], int pitch, int W, int H, int black_thr, const float *atan_lut,
                                      float *const phase[2], uint8_t *const valid[2], const int16_t *const map_xy[2],
                                      const uint16_t *const map_frac[2], const void *const tile_boxes[2], int rect_algo,
                                      bool *done, hipStream_t s);

// cv::initUndistortRectifyMap (CV_16SC2 fixed-point maps) on the device
hipError_t launch_init_rectify_map(const double M[9], const double D[5], const double R[9], const double P[12], int W, int H,
                                   int16_t *map_xy, uint16_t *map_frac, hipStream_t s);

// per-tile source bounding boxes of a rectification map (64x16, 64x4, 64x8 destination tiles; int4 per tile), the
// pre-digested tiled copies of the map for the 64x8 and 128x8 fused decode, and (nofit_host, valid after the stream is
// synchronised) the number of tiles whose box does not fit the 64x8 / the 128x8 form
size_t     tile_boxes_bytes(int W, int H);
hipError_t launch_tile_boxes(const int16_t *map_xy, const uint16_t *map_frac, int W, int H, int4 *boxes, unsigned nofit_host[2],
                             hipStream_t s);

// LDS-DMA form of the fused rectify + multi-frequency decode (kernels_rectdma.hip).  shape (SLR_OPT_RECT_DMA_SHAPE): destination
// tile / threads 0 = 256x16 / 512, 1 = 256x8 / 512, 2 = 256x8 / 256, 3 = 128x16 / 512, 4 = 128x8 / 256, 5 = 256x4 / 256,
// 6 = 128x16 / 256.  launch_dma_tiles builds the per-tile source boxes + the map digest of one shape into
// `buf` (dma_tiles_bytes) and copies the number of tiles whose box does not fit the form to *nofit_host (valid after the
// stream is synchronised).  launch_mf_rect_decode_dma: one camera (n == 1) or both cameras of a frame (n == 2) in one
// launch; *done = false -> the stack layout / image width does not allow this form, nothing was launched.
// nofit_host[kDmaTileStats]: 0 = tiles no split makes fit (the gather fix-up rewrites them); 1..3 = quads of read class 0 / 1 / 2;
// 4..6 = (entry, wave) pairs whose wave-uniform read mode is 0 / 1 / 2; 7 = extra entries allocated for the parts of split tiles
// (may overshoot dma_extra_entries_capacity: clamp)
constexpr int kDmaTileStats = 8;
size_t     dma_tiles_bytes(int W, int H, int shape);
size_t     dma_tile_count_of(int W, int H, int shape);
unsigned   dma_extra_entries_capacity(int W, int H, int shape);
hipError_t launch_dma_tiles(const int16_t *map_xy, const uint16_t *map_frac, int W, int H, void *buf, int shape,
                            bool promote /* waves whose quads straddle source rows at several pixel positions read per pixel */,
                            bool sort_quads /* a tile's straddling quads handed to as few of its waves as hold them */,
                            unsigned *nofit_host, hipStream_t s);
// the tiles whose source box does not fit the form (launch_dma_tiles counted and listed them) are rewritten by a gather pass
// behind the main kernel: it needs the camera's original maps and the count
struct DmaFixup {
    const int16_t *map_xy[2];
    const uint16_t *map_frac[2];
    unsigned nofit[2];
    unsigned extras[2];          // entries of the tile tables behind the tiles themselves: the extra parts of split tiles
};
hipError_t launch_mf_rect_decode_dma(const MfPlanes *pl, int n, int pitch, int W, int H, int black_thr, const float *lut,
                                     float *const *phase, uint8_t *const *valid, const void *const *tiles, int shape, int depth,
                                     unsigned *sched /* dma_sched_bytes() of zeros, owned by the context: the tile tickets */,
                                     const DmaFixup *fix, bool *done, hipStream_t s, const int *fix_slot = nullptr);
constexpr int kDmaMaxJobs = 16;  // (frame, camera) jobs of one fused-decode launch: the two cameras of up to 8 frames
size_t     dma_sched_bytes();

hipError_t launch_gray_rect_decode_dma(const GrayPlanes *pl, int n, int ncol, int nrow, int pitch, int W, int H, int black_thr,
                                       int white_thr, int scan_w, int scan_h, int32_t *const *code_x, int32_t *const *code_y,
                                       uint8_t *const *valid, const void *const *tiles, int shape, int depth /* SLR_OPT_RECT_DMA_DEPTH: 2 = the
                                       counted-wait form (round 5), 1 = round 2's form */, unsigned *sched, const DmaFixup *fix,
                                       bool *done, hipStream_t s, const int *fix_slot = nullptr);
// BASELINE config 3: Gray code + multi-frequency phase in ONE pass over a hybrid stack (kernels_rectdma.hip)
hipError_t launch_hybrid_rect_decode_dma(const GrayPlanes *pl, int n, int ncol, int pitch, int W, int H, int black_thr, int white_thr,
                                         int scan_w, const float *lut, int32_t *const *code_x, float *const *phase,
                                         const void *const *tiles, int shape, unsigned *sched, const DmaFixup *fix, bool *done,
                                         hipStream_t s);
hipError_t launch_gray_decode(const GrayPlanes &pl, int n_col_bits, int n_row_bits, int pitch, int W, int H,
                              int black_thr, int white_thr, int scan_w, int scan_h,
                              int32_t *code_x, int32_t *code_y, uint8_t *valid,
                              const int16_t *map_xy, const uint16_t *map_frac, const void *tile_boxes, int rect_algo,
                              hipStream_t s);

hipError_t launch_mf_match(const float *phaseL, const uint8_t *validL, const float *phaseR,
                           const uint8_t *validR, int W, int H /* rows handed over */, int row0 /* image row of the
                           first one: the reprojection and the tables use absolute rows */, const DevCalib &cal,
                           float *xyz, uint8_t *has, int32_t *match_k, int algo,
                           const float *undL_xy /* [H][W][2] or null */, const float *undRx /* [H][W] or null */,
                           hipStream_t s, int nframes = 1 /* > 1: that many frames, frame_px pixels apart in every array (ask
                           mf_match_batches_frames first; no valid bytes, no match_k) */, size_t frame_px = 0,
                           int *defer = nullptr /* (H + 1) ints of device scratch: rows of 4097..8192 pixels whose index would hold an
                           overfull bin are listed there by the wide kernel and matched by the chunked kernel behind it; null: no such guard */);
bool mf_match_batches_frames(const float *phaseL, const float *phaseR, const float *xyz, const uint8_t *has, int W, const DevCalib &cal,
                             int algo, const float *undL_xy, const float *undRx, size_t frame_px);
// per-pixel Utilities::undistortPoints tables for a (calibration, W, H): left (x,y), right x
hipError_t launch_undistort_tables(const DevCalib &cal, int W, int H, float *undL_xy, float *undRx, hipStream_t s);

hipError_t launch_ge_match(const int32_t *codeL, const uint8_t *validL, const int32_t *codeR,
                           const uint8_t *validR, int W, int H, const DevCalib &cal,
                           const uint8_t *whiteL, const uint8_t *whiteR,
                           float *xyz, uint8_t *has, uint8_t *color, int32_t *match_k, int code_bound, hipStream_t s);

// build extension: generalised n_freq x n_step fp16 multi-frequency decode (kernels_mfn.hip)
hipError_t launch_mfn_decode(const uint16_t *const *planes, int n_freq, int n_step, int pitch, int W, int H, float black_thr,
                             float *phase, uint8_t *valid, hipStream_t s);
hipError_t launch_cloud_checksums(const float *xyz, const uint8_t *has, int n_frames, size_t n_px, unsigned long long *d_out, hipStream_t s);
hipError_t launch_stream_copy(const void *src, void *dst, size_t bytes /* multiple of 16, both 16-byte aligned */, hipStream_t s);
// `reads` (1..8) consecutive streams of bytes_out bytes each read from src per bytes_out written to dst (the fused decode's read : write mix)
hipError_t launch_stream_mix(const void *src, void *dst, size_t bytes_out, int reads, hipStream_t s);
hipError_t launch_mfn_rect_decode(const uint16_t *const *planes, int n_freq, int n_step, int pitch, int W, int H, float black_thr,
                                  const int16_t *map_xy, const uint16_t *map_frac, int row0, int rows, int src_row0, int src_rows,
                                  float *phase, uint8_t *valid, hipStream_t s);
hipError_t launch_map_source_rows(const int16_t *map_xy, int W, int H, int row0, int rows, int *d_out, hipStream_t s);

// GRAY_ONLY: counting sort of camera pixels by projector cell (both cameras share one histogram / offsets
// array: left cells [0,nb), right cells [nb,2nb), +1 pad), then one thread per cell
size_t     ray_scan_temp_bytes(size_t n);
hipError_t launch_ray_count(const int32_t *code_x, const int32_t *code_y, const uint8_t *valid, int W, int H,
                            int scan_w, int scan_h, uint32_t *cnt, uint32_t *cell_of, uint32_t *rank_of, hipStream_t s);
hipError_t launch_ray_scan(const uint32_t *cnt, uint32_t *offs, size_t n, void *temp, size_t temp_bytes, hipStream_t s);
hipError_t launch_line_line(size_t n, const float *p1, const float *p2, const float *v1, const float *v2, float *out, uint8_t *ok,
                            hipStream_t s);
// GRAY_ONLY: decode + pass 1 in one kernel (cell_of / rank_of straight from the planes), when the stack allows dword loads
bool ray_decode_count_applies(const GrayPlanes &pl, int nplanes, int n_row_bits, int pitch, int W, int H);
hipError_t launch_gray_decode_count(const GrayPlanes &pl, int n_col_bits, int n_row_bits, int pitch, int W, int H, int black_thr,
                                    int white_thr, int scan_w, int scan_h, uint32_t *cnt, uint32_t *cell_of, uint32_t *rank_of,
                                    hipStream_t s);
hipError_t launch_ray_scatter(const uint32_t *cell_of, const uint32_t *rank_of, int W, int H, const uint32_t *offs,
                              uint32_t *items, hipStream_t s);
// unit view rays of every camera pixel, [H][W][3] per camera (calibration constants, cached by the context)
hipError_t launch_ray_tables(const DevCalib &cal, int W, int H, float *raysL, float *raysR, hipStream_t s);
// `list`: ray_list_words(scan_w * scan_h) words of scratch (the cells ordered by bucket lengths), made by
// launch_ray_scatter_list beside the scatter (it also zeroes the cells without pairs) and read by launch_ray_triangulate
size_t ray_list_words(size_t cells);
size_t ray_items_words(size_t cam_pixels);             // both cameras' items + the padding K6 reads into
hipError_t launch_ray_scatter_list(const uint32_t *cellL, const uint32_t *rankL, const uint32_t *cellR, const uint32_t *rankR, int W, int H,
                                   const uint32_t *offs, uint32_t *items, int scan_w, int scan_h, uint32_t *list, float *xyz_sum,
                                   uint8_t *count, hipStream_t s);
hipError_t launch_ray_triangulate(const uint32_t *offs, uint32_t *items, const DevCalib &cal, int scan_w, int scan_h,
                                  int W, const float *raysL, const float *raysR, const uint32_t *list, float *xyz_sum, uint8_t *count,
                                  hipStream_t s);

// ordered prefix index / compaction of a u8 flag image (kernels_compact.hip): enumeration row-major or column-major over an
// [h][w] image; index (may be null) gets first + rank for flagged elements and `none` otherwise; with xyz the flagged points
// are compacted into out_xyz (+ their source positions into out_src).  *total_dev = device address of the flagged count.
size_t     flag_scan_temp_bytes(size_t n);
hipError_t launch_flag_scan(const uint8_t *flags, size_t n, int w, int h, int column_major, uint32_t first, uint32_t none,
                            uint32_t *index, const float *xyz, float *out_xyz, uint32_t *out_src, void *temp,
                            uint32_t **total_dev, hipStream_t s);

hipError_t launch_pc_from_grid(const float *xyz, const uint8_t *has, const uint8_t *color, int W, int H,
                               int scan_w, int scan_h, float *pc_sum, uint8_t *pc_count, uint8_t *pc_color,
                               hipStream_t s);
hipError_t launch_pc_get(const float *pc_sum, const uint8_t *pc_count, size_t n, float *out, hipStream_t s);

#if defined(__HIPCC__)
// Exclusive scan over a workgroup of BLOCK threads (the lean match kernels' index build, K6's bucket places): every wave scans itself with DPP lane
// shifts (row_shr 1 / 2 / 4 / 8, then row_bcast 15 and 31: six VALU steps, no LDS crossbar), the (at most 16) wave totals cross in
// LDS behind ONE barrier and every wave scans them for itself -- in place of hipcub::BlockScan, whose warp-scans form spends two
// barriers.  result = op(init, v[0], ..., v[tid - 1]) (init must be op's identity); *total = the reduction over the workgroup.
// wave_tot: LDS, BLOCK / 64 words, free for reuse after the caller's next barrier.
template <int CTRL, int ROW_MASK, typename T>
__device__ __forceinline__ T dpp_or(T identity, T v)       // the DPP source lane's v, or `identity` where there is none / the row is masked
{
    return (T)__builtin_amdgcn_update_dpp((int)identity, (int)v, CTRL, ROW_MASK, 0xF, false);
}
template <int BLOCK, typename T, typename Op>
__device__ __forceinline__ T wg_exclusive_scan(T v, T init, Op op, T *wave_tot, T *total)
{
    static_assert(sizeof(T) == 4, "32-bit DPP");
    constexpr int NW = BLOCK / 64;
    static_assert(NW >= 1 && NW <= 16, "the wave totals are scanned inside one DPP row");
    const int lane = (int)(threadIdx.x & 63u);
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    T inc = v;
    inc = op(dpp_or<0x111, 0xF>(init, inc), inc);           // row_shr:1
    inc = op(dpp_or<0x112, 0xF>(init, inc), inc);           // row_shr:2
    inc = op(dpp_or<0x114, 0xF>(init, inc), inc);           // row_shr:4
    inc = op(dpp_or<0x118, 0xF>(init, inc), inc);           // row_shr:8   -> inclusive inside each row of 16
    inc = op(dpp_or<0x142, 0xA>(init, inc), inc);           // row_bcast:15 into rows 1 and 3
    inc = op(dpp_or<0x143, 0xC>(init, inc), inc);           // row_bcast:31 into rows 2 and 3 -> inclusive over the wave
    if (lane == 63) wave_tot[wv] = inc;
    __syncthreads();
    T t = wave_tot[lane < NW ? lane : NW - 1];
    t = op(dpp_or<0x111, 0xF>(init, t), t);
    if (NW > 2) t = op(dpp_or<0x112, 0xF>(init, t), t);
    if (NW > 4) t = op(dpp_or<0x114, 0xF>(init, t), t);
    if (NW > 8) t = op(dpp_or<0x118, 0xF>(init, t), t);
    if (total) *total = (T)__builtin_amdgcn_readlane((int)t, NW - 1);
    const T base = wv > 0 ? (T)__builtin_amdgcn_readlane((int)t, wv > 0 ? wv - 1 : 0) : init;
    const T excl = dpp_or<0x138, 0xF>(init, inc);           // wave_shr:1: the wave's own exclusive prefix (lane 0: init)
    return op(base, excl);
}
#endif

}  // namespace slr
