/*
 * slr.h -- C ABI of libslr_hip.so: the MI355X (gfx950) drop-in for the dense per-pixel loops of
 * DrawZeroPoint/Structure-Light-Reconstructor ("Duke").
 *
 * The reference has NO plugin / FFI interface: the hot path is private member functions of
 * MFReconstruct and Reconstruct, called synchronously from MainWindow::startreconstruct
 * (Duke/mainwindow.cpp:562-652).  The boundary is therefore defined at the seam inside
 *     MFReconstruct::runReconstruction      Duke/mfreconstruct.cpp:160-187
 *     Reconstruct::runReconstruction_GE     Duke/reconstruct.cpp:271-307
 *     Reconstruct::runReconstruction        Duke/reconstruct.cpp:230-265
 * after cv::imread has produced the raw 8-bit planes (mfreconstruct.cpp:125, reconstruct.cpp:164) and
 * before MeshCreator consumes points3DProjView (mainwindow.cpp:632-637).  Every entry point below names
 * the reference function(s) it replaces.  INTEGRATION.md shows the ~40-line shim a maintainer adds to the
 * Qt application.
 *
 * Conventions
 *   - plain C, no exceptions cross the boundary; every call returns an slr_status (0 = ok, <0 error);
 *     slr_last_error(ctx) gives the message (reference: bool + QMessageBox, mfreconstruct.cpp:62-63).
 *   - images are row-major, 8-bit, `pitch` bytes between rows; all per-pixel outputs are dense [H][W].
 *   - `mem` says where EVERY data pointer of that call lives: SLR_MEM_HOST (library stages through its
 *     own device buffers and returns after the results are back on the host) or SLR_MEM_DEVICE (pointers
 *     are HIP device pointers on the ctx's GPU; work is enqueued on the ctx stream and NOT synchronised).
 *     Arrays of plane pointers (`planes`) are always host arrays; their elements follow `mem`.
 *   - one ctx per GPU, one HIP stream per ctx; calls on one ctx must be serialised by the caller
 *     (the reference is single-threaded; its hidden left/right flip-flop globals, reconstruct.cpp:4 and
 *     mfreconstruct.cpp:4, are replaced by the explicit `cam` argument: 0 = left, 1 = right).
 *   - there is NO CPU fallback: with no usable GPU slr_create fails with SLR_ERR_NO_DEVICE.
 */
#ifndef SLR_H
#define SLR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SLR_VERSION 100            /* 0.1.0 */
#define SLR_MF_PLANES 14           /* mfreconstruct.cpp:22 numberOfImgs: white, black, 3 freq x 4 steps */
#define SLR_MAX_GRAY_BITS 16       /* per axis; reference allows 44 planes total (graycodes.h:12) */
#define SLR_MAX_GRAY_PLANES (2 + 4 * SLR_MAX_GRAY_BITS)
#define SLR_MFN_MAX_FREQ 6         /* generalised multi-frequency decode (build extension, slr_mfn_decode) */
#define SLR_MFN_MAX_STEPS 16
#define SLR_MFN_MAX_PLANES (2 + SLR_MFN_MAX_FREQ * SLR_MFN_MAX_STEPS)

typedef enum slr_status {
    SLR_OK = 0,
    SLR_ERR_INVALID_ARG = -1,
    SLR_ERR_NO_DEVICE = -2,
    SLR_ERR_HIP = -3,
    SLR_ERR_NOT_CONFIGURED = -4,   /* calibration / rectify maps missing */
    SLR_ERR_UNSUPPORTED = -5,
    SLR_ERR_OOM = -6
} slr_status;

typedef enum slr_mem { SLR_MEM_HOST = 0, SLR_MEM_DEVICE = 1 } slr_mem;

/* VirtualCamera (Duke/virtualcamera.h:27-37): the fields the path reads.  All f32 because the reference
 * parses every calibration number through `float` (virtualcamera.cpp:82-84).  k[4] (k3) is carried but,
 * like the reference (utilities.cpp:66), ignored by the undistortion. */
typedef struct slr_camera {
    float fc[2];      /* cam_matrix (0,0),(1,1) */
    float cc[2];      /* cam_matrix (0,2),(1,2) */
    float k[5];       /* cam_distortion 5x1 */
    float R[9];       /* cam_rotation_matrix, row-major */
    float t[3];       /* cam_trans_vectror */
} slr_camera;

typedef struct slr_calib {
    slr_camera cam[2];   /* 0 = left, 1 = right */
    double Q[16];        /* stereoRect::Q, 4x4 row-major f64 (stereorect.cpp:41) */
    float T[12];         /* scan/transfer_mat<sn>.txt, 3x4 row-major (mfreconstruct.cpp:278-282) */
    int has_T;           /* scanSN > 0 */
} slr_calib;

typedef struct slr_ctx slr_ctx;

/* ---- lifecycle -------------------------------------------------------------------------------------- */
int          slr_version(void);
const char  *slr_status_string(int status);
int          slr_current_device(int *device_id);                  /* the calling thread's current HIP device (hipGetDevice) */
int          slr_create(int device_id, slr_ctx **out);           /* replaces `new MFReconstruct()/Reconstruct()` mainwindow.cpp:577-582 */
int          slr_destroy(slr_ctx *ctx);                          /* Reconstruct::~Reconstruct reconstruct.cpp:24-31 */
int          slr_set_stream(slr_ctx *ctx, void *hip_stream);     /* borrow a caller stream (NULL -> ctx-owned stream) */
int          slr_synchronize(slr_ctx *ctx);
const char  *slr_last_error(const slr_ctx *ctx);
/* tuning / test knobs.  SLR_OPT_MF_MATCH_ALGO: 0 = auto, 1 = linear LDS sweep (the literal form of
 * mfreconstruct.cpp:289-331), 2 = indexed exact form built by an LDS radix sort (distinct phases), 3 = indexed exact form
 * built by an LDS counting sort (rows wider than 4096 pixels are matched against the right row in chunks of 4096 columns, up to
 * 32768 pixels; wider rows take the sweep).  0 picks 3, in its lean variant (same index, fewer instructions) when the rows are
 * aligned, 513..1024 or 2049..4096 pixels wide and Q has cv::stereoRectify's pattern.  4 / 5 / 6 pin the lean variant's
 * shapes for rows of 2049..4096 pixels (0 picks 1024 threads x 4 pixels with the row's XYZ stored through LDS; 4 = the same
 * with per-thread stores, round 2; 5 / 6 = 512 threads x 8 pixels with two / three rows per CU: measured, not faster, `make FORMS=all` builds only)
 * and behave as 0 where the lean variant does not apply; 7 = the grouped launches of slr_reconstruct_mf_batch as a persistent kernel
 * with the next row's phases prefetched (round 5: measured slower, 105 against 77 us per frame; `make FORMS=all` builds only); 8 = 0's
 * lean kernel by name (its index dedups equal right phases through an LDS hash table); 9 = the same index WITHOUT the dedup, rows with
 * an overfull bin rebuilt with it inside the kernel (round 6: measured slower on the reference's phases, 90 against 81 us -- its
 * integer-quotient atan leaves ~1500 distinct values in a row of 4064 valid pixels; `make FORMS=all` builds only).  All give identical
 * results. */
#define SLR_OPT_MF_MATCH_ALGO 1
/* SLR_OPT_MF_DECODE_VEC: pixels per thread of the unfused K2 kernel: 0 = auto, 4, 8 or 16 (identical results) */
#define SLR_OPT_MF_DECODE_VEC 2
/* SLR_OPT_RECT_DECODE_ALGO: form of the fused rectify+decode kernel (identical results).  All LDS-tiled forms run
 * persistent workgroups with a register prefetch pipeline and fall back, per tile, to a direct gather when the tile's
 * source box does not fit their LDS budget.
 *   0 = auto (default): 7 when it applies, else 5, or 6 when the maps would make more pixels fall back in 5 than in 6
 *   1 = direct gather (no LDS tiles)
 *   2 = 64x16 tiles, two prefetch rounds
 *   3 = 64x8 tiles walking down tile columns with a 16-row sliding LDS window (no halo re-reads: 1.03x instead of
 *       1.13x the algorithmic HBM bytes, but ~3 % slower)
 *   4 = 128x8 tiles, 256 threads, two prefetch rounds
 *   5 = 128x8 tiles, 512 threads (waves 0-3 left half, 4-7 right half), one round, pre-digested 4-byte map entries
 *   6 = 64x8 tiles, 256 threads, one round, pre-digested 4-byte map entries
 *   7 = LDS-DMA form (round 2): 256- or 128-pixel-wide tiles (SLR_OPT_RECT_DMA_SHAPE) decoded in 7 phases of two
 *       planes whose source boxes go HBM -> LDS by buffer_load ... lds into a double / triple buffer.  Needs the 14 planes
 *       of a camera equally spaced in one allocation, 16-byte aligned rows, W % 16 == 0 and maps at least three quarters
 *       of whose tile boxes fit (the other tiles are rewritten by a gather pass behind the main kernel);
 *       auto falls back to 5 / 6 otherwise (also per call); an explicit 7 makes such calls fail with SLR_ERR_UNSUPPORTED.
 * The Gray fused decode only distinguishes 1 (gather), 2 (64x4 tiles) and everything else (64x8 tiles when they fit). */
#define SLR_OPT_RECT_DECODE_ALGO 3
/* SLR_OPT_ASYNC_HOST: 1 = calls with SLR_MEM_HOST buffers return once the H2D copies, the kernels and the D2H copies
 * are ENQUEUED on the ctx stream; outputs are valid (and inputs reusable) only after slr_synchronize(ctx).  The host
 * buffers should be pinned (hipHostMalloc / cudaHostRegister-style) for the copies to be truly asynchronous.  Two ctx on
 * one GPU, fed alternately, overlap the upload of frame i+1, the kernels of frame i and the download of frame i-1
 * (the pinned double-buffered loader of SURVEY 8f-1).  Default 0: host-buffer calls are synchronous. */
#define SLR_OPT_ASYNC_HOST 4
/* SLR_OPT_PROFILE_STRIDE: the built-in HIP-event profiler brackets every n-th launch of a kernel (default 1 = all) */
#define SLR_OPT_PROFILE_STRIDE 5
/* SLR_OPT_RECT_DMA_SHAPE: destination tile / workgroup size of form 7: 0 = 256x16 / 512 threads, 1 = 256x8 / 512, 2 = 256x8 / 256,
 * 3 = 128x16 / 512 (default), 4 = 128x8 / 256, 5 = 256x4 / 256, 6 = 128x16 / 256 (identical results; the tile tables of the installed maps
 * are rebuilt).  SLR_OPT_RECT_DMA_DEPTH: phases of LDS-DMA in flight ahead of
 * the decode, 1 (double buffer) or 2 (triple buffer, the default).  Under SLR_OPT_EVAL_MODEL = 1 the multi-frequency decode exists at
 * distance 2 only: the option is ignored there and slr_get_rectify_info reports the effective value (2). */
/* Forms that lost their measurements -- SLR_OPT_RECT_DECODE_ALGO 2, 3, 4, SLR_OPT_RECT_DMA_SHAPE 2, 4, 5, 6, SLR_OPT_MF_MATCH_ALGO 2, 5, 6, 7, 9
 * -- are compiled into the library with `make FORMS=all` only; a default build answers SLR_ERR_UNSUPPORTED to these values. */
#define SLR_OPT_RECT_DMA_SHAPE 6
#define SLR_OPT_RECT_DMA_DEPTH 7
/* Test knobs (results never change).  SLR_OPT_DEBUG_RECT_RESIDENT: n > 0 = run the persistent fused decodes on n workgroups
 * (many tiles per workgroup), 0 = as many as are resident.  SLR_OPT_DEBUG_FLAGS: bit 0 = fused decode reads the caller's map
 * entries instead of the digest, bit 1 = per-plane pointers instead of one buffer descriptor, bit 2 = the general Gray-code match
 * kernel for every row (instead of only the rows its lean form defers), bit 3 = slr_reconstruct_gray decodes into code
 * arrays and counts the buckets in a second kernel (instead of one fused kernel per camera), bit 4 = the LDS-tiled fused Gray
 * decode takes its 64 x 4 tiles (the form of stacks of 42 planes and more) whatever the plane count, bit 5 = the map digests of the
 * LDS-DMA fused decodes leave every wave the quads of its own block of a tile (instead of handing a tile's straddling quads to as
 * few waves as hold them; installed maps are digested again when the bit changes), bit 6 = straddling waves keep the three-row read
 * mode.  slr_mfn_rectify_decode (config 5): bit 0 = the per-pixel gather form, bit 1 = the register-staged tile form (as for the
 * u8 decodes: no digest / no buffer form), and two bits of its own: bit 7 = 512 threads x 2 pixels instead of 256 x 4 in the LDS-DMA
 * form, bit 8 = its ring of 4 plane groups (two workgroups per CU) instead of 3.  SLR_OPT_DEBUG_K4_STOP exists only
 * in -DSLR_DEBUG_HOOKS builds of the library (phase ablation of the match kernel; outputs are not written). */
#define SLR_OPT_DEBUG_RECT_RESIDENT 8
#define SLR_OPT_DEBUG_FLAGS 9
#define SLR_OPT_DEBUG_K4_STOP 10
/* SLR_OPT_HYBRID_ONE_PASS: how slr_hybrid_rectify_decode_pair / slr_reconstruct_hybrid_batch decode a hybrid stack (identical
 * results): 0 (default) = two launches over the one stack -- the fused Gray decode of both cameras on planes 0 .. 2 n + 1, then the
 * fused multi-frequency decode of both cameras on white, black and the fringes behind the Gray planes (measured faster: 349 vs
 * 385 us at 4096x3000); 1 = ONE kernel that walks all 2 n + 14 planes of a tile (one digest, one shadow mask). */
#define SLR_OPT_HYBRID_ONE_PASS 11
/* SLR_OPT_BATCH_STREAMS: 2 (default) = slr_reconstruct_batch in SLR_MODE_GRAY alternates its frames between two streams with a
 * scratch set each, the second one half a frame behind: a frame's ray-ray triangulation (arithmetic) runs beside the next
 * frame's decode, histogram and scatter (HBM streaming); 1 = one frame after the other on the context's stream.  Same results. */
#define SLR_OPT_BATCH_STREAMS 12
/* SLR_OPT_MF_BATCH_GROUP: frames slr_reconstruct_mf_batch hands to ONE match + triangulate launch (default 8; 1 = frame by frame as
 * in rounds 1-3).  The undistortion tables K4 reads are per calibration (12 of its 35 bytes per pixel): a launch over a group of
 * frames walks a row of all of them on the same XCD, so the tables cross HBM once per group.  The phases of a group live in the
 * context's scratch: 8 bytes per pixel and frame -- 0.8 GB at 4096x3000 with the default, 6.3 GB with the maximum of 64; the
 * scratch is kept until the context is destroyed.  Where one match launch cannot take a group (another match form, a Q that is
 * not cv::stereoRectify's, rows outside 2049..4096 pixels) the group shrinks to what one fused-decode launch serves, or to 1.
 * Identical results.
 * SLR_OPT_MF_BATCH_DECODE_GROUP: frames of such a group whose fused rectify + decode (both cameras each) share ONE launch of the
 * persistent LDS-DMA kernel (1..8, default 8; 1 = one launch per frame).  Identical results. */
#define SLR_OPT_MF_BATCH_GROUP 15
#define SLR_OPT_MF_BATCH_DECODE_GROUP 16
/* SLR_OPT_DEBUG_POISON_SCRATCH (tests): 1 = every scratch buffer the context hands to a call (phases, codes, buckets, staging --
 * not the cached calibration tables) is filled with 0x7B bytes first, behind a device synchronisation: an intermediate a kernel
 * fails to write cannot pass for the previous call's.  Slow; 0 (default) = off. */
#define SLR_OPT_DEBUG_POISON_SCRATCH 13
/* SLR_OPT_EVAL_MODEL: which evaluation of the reference's floating-point source text the multi-frequency path reproduces.
 *   0 (default) = strict IEEE: every f32 operation rounds to f32 (what oracle/slr_oracle.c restates; SSE2-style code).
 *   1 = x87: the reference's own binary is an MSVC2010 32-bit x87 build (Duke.pro, /fp:precise, 53-bit precision control) in which
 *       a compound float expression keeps 53 bits until it is stored -- P[count] = atan(...) + PI lands in a double unrounded,
 *       P123 and phase = P123/(2*PI)*255 round once, fabs(phiL - phiR) < 0.1 sees the exact difference, the disparity
 *       ulx - urx is stored to a double unrounded (mfreconstruct.cpp:246-268, :295, :299).  On BASELINE's scene the two models
 *       differ in the last place of 37 % of the phases and pick another first-match column for 2.2-2.4 % of the matched pixels
 *       (tests/x87_sensitivity.py); neither can be pinned without the MSVC2010 toolchain, so a host that must
 *       match the shipped Windows binary chooses here.  Since round 5 mode 1 runs the SAME kernels as mode 0 -- the LDS-DMA fused
 *       decode (single, pair and grouped launches, the hybrid stack's too) with the x87 variant of the wrapped-phase table and
 *       a three-instruction f64 quotient by 2*PI, the lean match kernel (grouped launches included) with the exact-difference
 *       predicate where it can differ at all (|phase| < 0.25: Sterbenz) and the unrounded disparity -- at 1-2 % more time per
 *       frame (bench.py --eval-model x87); bit for bit against oracle/slr_oracle_x87.c on whole 4096x3000 frames
 *       (tests/test_gpu_timed_path.py).  GRAY_ONLY (slr_ray_triangulate / slr_reconstruct_gray / slr_line_line_intersections):
 *       normalize, pixelToImageSpace and line_lineIntersection (utilities.cpp:19-28, 47-56, 399-425) keep their 53-bit
 *       intermediates under mode 1 as well (the unit-ray tables are rebuilt when the mode changes); GRAY_EPI is integer work up to
 *       the f64 Q reprojection and is the same under both.
 *   WHICH TO SET: a host that replaces the loop inside the reference's own Windows build (Duke.pro: MSVC2010, 32-bit, x87) and must
 *   reproduce that binary's clouds sets 1; a host built for x64 / SSE2 (any 64-bit MSVC, gcc or clang build of the same sources,
 *   where float expressions round per operation) sets 0, the default.  bench.py's line carries both (`models`). */
#define SLR_OPT_EVAL_MODEL 14
int          slr_set_option(slr_ctx *ctx, int option, int value);

/* ---- configuration ---------------------------------------------------------------------------------- */
/* replaces MFReconstruct::loadCameras (mfreconstruct.cpp:67-108) / Reconstruct::loadCameras
 * (reconstruct.cpp:99-148) + stereoRect::Q (stereorect.cpp:41): the host parses the text files, the
 * library keeps the numbers. */
int slr_set_calibration(slr_ctx *ctx, const slr_calib *calib);
/* replaces the map half of stereoRect::calParameters (stereorect.cpp:42-43): map_xy = CV_16SC2 [H][W][2]
 * (x,y), map_frac = CV_16UC1 [H][W]; copied into library-owned HBM. */
int slr_set_rectify_maps(slr_ctx *ctx, int cam, const int16_t *map_xy, const uint16_t *map_frac,
                         int W, int H, slr_mem mem);
/* the same maps computed ON THE DEVICE from the calibration: cv::initUndistortRectifyMap(M, D, R, P, size,
 * CV_16SC2, map1, map2) of stereoRect::calParameters (stereorect.cpp:42-43).  M 3x3, D = k1 k2 p1 p2 k3, R 3x3
 * (R1/R2 of cv::stereoRectify), P 3x4 (P1/P2), all row-major f64 host arrays.  Installs the result as camera `cam`'s
 * rectification maps (bit-identical to the host restatement, which the oracle checks). */
int slr_init_rectify_maps(slr_ctx *ctx, int cam, const double M[9], const double D[5], const double R[9],
                          const double P[12], int W, int H);
/* What the installed maps of `cam` mean for the fused rectify + decode kernels (they pick their form per call from this):
 *   mf_form          the SLR_OPT_RECT_DECODE_ALGO value the multi-frequency decode resolves to for these maps under the current
 *                    options (7 = LDS-DMA form; 5 / 6 = round-1 LDS tiles, which fall back per tile to a gather); -1: the option
 *                    asks for form 7 explicitly and these maps do not allow it (a decode call returns SLR_ERR_UNSUPPORTED)
 *   dma_tiles / dma_nofit_tiles   tiles of the LDS-DMA form's shape, and how many of them the form cannot hold even in parts:
 *                    a tile whose source box is larger than the LDS image (keystone corners) is decoded in wave-aligned halves,
 *                    quarters, ... (dma_extra_entries counts the extra parts); what no split makes fit is rewritten by a gather
 *                    pass behind the main kernel; beyond a quarter of the tiles auto does not use form 7 for these maps;
 *                    0xFFFFFFFF: no tables (W % 16 != 0)
 *   quads_by_class   4-pixel quads whose taps fit one 8-byte window of 2 source rows / of 3 rows / neither (fitting tiles)
 *   waves_by_mode    (tile, wave) pairs decoded in the two-row / three-row / per-pixel read mode (the three-row mode costs
 *                    ~27 % more instructions, the per-pixel one ~4x the LDS reads)
 *   lds_nofit_tiles  tiles that fit neither the 64x8 nor the 128x8 round-1 form (those pixels are gathered) */
typedef struct slr_rectify_info {
    int W, H, mf_form, dma_shape, dma_depth;
    unsigned dma_tiles, dma_nofit_tiles;
    unsigned quads_by_class[3], waves_by_mode[3];
    unsigned lds_nofit_tiles[2];
    unsigned dma_extra_entries;    /* extra parts of tiles that the LDS-DMA form decodes in wave-aligned halves / quarters / ... */
} slr_rectify_info;
int slr_get_rectify_info(slr_ctx *ctx, int cam, slr_rectify_info *out);
/* read back the maps currently installed for `cam` (host or device destination) */
int slr_get_rectify_maps(slr_ctx *ctx, int cam, int16_t *map_xy, uint16_t *map_frac, int W, int H, slr_mem mem);

/* ---- K1: stereoRect::doStereoRectify -> cv::remap INTER_LINEAR (stereorect.cpp:26-34) ----------------- */
int slr_remap_u8(slr_ctx *ctx, int cam, const uint8_t *src, int src_pitch,
                 uint8_t *dst, int dst_pitch, int W, int H, slr_mem mem);

/* ---- K2: MFReconstruct::computeShadows + decodePatterns + getPhase (mfreconstruct.cpp:190-269) -------
 * planes[14] already rectified.  phase [H][W] f32 (0 where mask==0), valid [H][W] u8. */
int slr_mf_decode(slr_ctx *ctx, const uint8_t *const planes[SLR_MF_PLANES], int pitch, int W, int H,
                  int black_thr, float *phase, uint8_t *valid, slr_mem mem);
/* fused K1+K2: loadCamImgs' 14x doStereoRectify (mfreconstruct.cpp:119-134) + the above, raw planes in */
int slr_mf_rectify_decode(slr_ctx *ctx, int cam, const uint8_t *const planes[SLR_MF_PLANES], int pitch,
                          int W, int H, int black_thr, float *phase, uint8_t *valid, slr_mem mem);

/* the same for BOTH cameras of a stereo frame -- one kernel launch when the LDS-tiled fused forms apply (what the whole-path
 * entries run): MFReconstruct::loadCamImgs + decodePatterns for camera 0 and 1 (mfreconstruct.cpp:119-134, 210-269).  validL and
 * validR both NULL: the flag travels inside the phase -- an invalid pixel's phase is a quiet NaN (0x7FC00000) instead of 0, which
 * slr_mf_triangulate* never matches and which lets them be called with NULL valid arrays too. */
int slr_mf_rectify_decode_pair(slr_ctx *ctx, const uint8_t *const planesL[SLR_MF_PLANES],
                               const uint8_t *const planesR[SLR_MF_PLANES], int pitch, int W, int H, int black_thr,
                               float *phaseL, uint8_t *validL, float *phaseR, uint8_t *validR, slr_mem mem);

/* ---- BUILD EXTENSION, no reference counterpart (the reference is hard-wired to 3 frequencies x 4 steps of u8,
 * mfreconstruct.cpp:21-22,237-242; BASELINE config 5; parity unpinned): n_freq x n_step phase-shift decode of fp16
 * planes with f32 accumulation.  planes[0] = white, planes[1] = black, planes[2 + f*n_step + k] = shift k (phase
 * offset 2*pi*k/n_step) of frequency f; elements are IEEE binary16 bit patterns, pitch in ELEMENTS.  Keeps the
 * reference's structure: shadow mask white - black > black_thr (mfreconstruct.cpp:190-207), one wrapped phase per
 * frequency (here atan2 of the n_step-point DFT bin, in [0, 2 pi)), the heterodyne cascade of neighbouring differences
 * with the "a > b ? a-b : a-b+2pi" rule (:265-267) down to one value, output scaled to 0..255 (:268).  valid = mask &&
 * every frequency has a modulation amplitude above half a grey level.  2 <= n_freq <= SLR_MFN_MAX_FREQ, 3 <= n_step <= SLR_MFN_MAX_STEPS. */
int slr_mfn_decode(slr_ctx *ctx, const uint16_t *const *planes, int n_freq, int n_step, int pitch, int W, int H,
                   float black_thr, float *phase, uint8_t *valid, slr_mem mem);
/* ... and THROUGH THE RECTIFICATION (BASELINE config 5 on the north_star path; build extension, parity unpinned): raw fp16
 * camera planes, camera `cam`'s maps applied with cv::remap's geometry as stereoRect::doStereoRectify uses it (stereorect.cpp:26-34:
 * CV_16SC2 + CV_16UC1 maps, 5-bit fractions, the four taps (sx, sy) .. (sx + 1, sy + 1), BORDER_CONSTANT 0) and f32 arithmetic --
 *     1024 x sample = (t00 w00 + t01 w01) + (t10 w10 + t11 w11),  w00 = (32 - fx)(32 - fy), w01 = fx (32 - fy), ...
 * the four products exact, summed in f32 row sy first, then row sy + 1 (v_dot2_f32_f16; OpenCV 2.4's remap has no fp16 mode, so
 * there is nothing to be bit-equal to) -- fused with the decode above: the rectified samples go into the DFT sums as f32.
 * ROW BANDS (one huge frame over several GPUs): the call decodes destination rows [row0, row0 + rows) of the H-row
 * image into phase / valid of [rows][W] elements.  The planes hold SOURCE rows [src_row0, src_row0 + src_rows) only: planes[i]
 * points at source row src_row0, pitch in elements; a tap outside that window reads 0.  slr_rectify_source_rows returns the window
 * a band needs (the rows its maps point into, clipped to the image); with a window at least that large a band's result equals
 * the same rows of the whole-frame call (row0 = 0, rows = H, src_row0 = 0, src_rows = H) bit for bit. */
int slr_mfn_rectify_decode(slr_ctx *ctx, int cam, const uint16_t *const *planes, int n_freq, int n_step, int pitch, int W, int H,
                           float black_thr, int row0, int rows, int src_row0, int src_rows, float *phase, uint8_t *valid,
                           slr_mem mem);
/* source rows [*src_row0, *src_row0 + *src_rows) that destination rows [row0, row0 + rows) of camera `cam` sample through the
 * installed maps (both bilinear rows, clipped to the image; 0 rows when the band samples nothing).  Synchronous. */
int slr_rectify_source_rows(slr_ctx *ctx, int cam, int row0, int rows, int *src_row0, int *src_rows);

/* ---- K3 / K3': Reconstruct::computeShadows + decodePatterns_GE/getProjPixel_GE (reconstruct.cpp:79-97,
 * 210-227,381-407) or decodePaterns/getProjPixel (:56-74,:325-370) + GrayCodes::grayToDec
 * (graycodes.cpp:116-128).  n_row_bits==0 -> GRAY_EPI (code_y may be NULL).  code = -1 where invalid. */
int slr_gray_decode(slr_ctx *ctx, const uint8_t *const *planes, int n_col_bits, int n_row_bits,
                    int pitch, int W, int H, int black_thr, int white_thr, int scan_w, int scan_h,
                    int32_t *code_x, int32_t *code_y, uint8_t *valid, slr_mem mem);
/* fused K1+K3 (GRAY_EPI rectifies every plane first, reconstruct.cpp:166-175) */
int slr_gray_rectify_decode(slr_ctx *ctx, int cam, const uint8_t *const *planes, int n_col_bits,
                            int n_row_bits, int pitch, int W, int H, int black_thr, int white_thr,
                            int scan_w, int scan_h, int32_t *code_x, int32_t *code_y, uint8_t *valid,
                            slr_mem mem);

/* ---- K4: MFReconstruct::triangulation (mfreconstruct.cpp:272-334) + Utilities::undistortPoints
 * (utilities.cpp:58-94).  Natural [H][W] output: xyz [H][W][3] f32 (0 where none), has [H][W] u8,
 * match_k [H][W] i32 (-1 where none; may be NULL).  Uses ctx calibration (cam[0], cam[1], Q, T). */
int slr_mf_triangulate(slr_ctx *ctx, const float *phaseL, const uint8_t *validL,
                       const float *phaseR, const uint8_t *validR, int W, int H,
                       float *xyz, uint8_t *has, int32_t *match_k, slr_mem mem);

/* the same for a band of `rows` image rows starting at row0 of an H-row image (arrays are [rows][W]): the match is
 * row-local, only the reprojection needs the absolute row.  This is how ONE frame shards over GPUs by row bands. */
int slr_mf_triangulate_rows(slr_ctx *ctx, const float *phaseL, const uint8_t *validL,
                            const float *phaseR, const uint8_t *validR, int W, int H, int row0, int rows,
                            float *xyz, uint8_t *has, int32_t *match_k, slr_mem mem);

/* ---- K5: Reconstruct::triangulation_ge (reconstruct.cpp:555-611).  whiteL/whiteR: rectified white planes
 * (pitch W) or NULL; color [H][W] u8 grey or NULL (haveColor, reconstruct.cpp:597-601).  Codes are Gray-decoded projector
 * columns in [0, 65535] (SLR_MAX_GRAY_BITS); a pixel whose code lies outside that range is treated as having no code. */
int slr_ge_triangulate(slr_ctx *ctx, const int32_t *codeL, const uint8_t *validL,
                       const int32_t *codeR, const uint8_t *validR, int W, int H,
                       const uint8_t *whiteL, const uint8_t *whiteR,
                       float *xyz, uint8_t *has, uint8_t *color, int32_t *match_k, slr_mem mem);

/* ---- K3' scatter + K6: decodePaterns' bucket scatter (reconstruct.cpp:61-72, reconstruct.h:94-97) +
 * Reconstruct::triangulation (reconstruct.cpp:417-481) + cam2WorldSpace (:310-322) + Utilities::
 * pixelToImageSpace/normalize/line_lineIntersection (utilities.cpp:19-28,47-56,399-425) +
 * PointCloudImage::addPoint accumulation (pointcloudimage.cpp:86-97).
 * xyz_sum [scan_h][scan_w][3] f32, count [scan_h][scan_w] u8. */
int slr_ray_triangulate(slr_ctx *ctx,
                        const int32_t *code_xL, const int32_t *code_yL, const uint8_t *validL,
                        const int32_t *code_xR, const int32_t *code_yR, const uint8_t *validR,
                        int W, int H, int scan_w, int scan_h,
                        float *xyz_sum, uint8_t *count, slr_mem mem);

/* ---- PointCloudImage contract (pointcloudimage.cpp:3-97) ---------------------------------------------
 * from_grid: what addPoint(i=row, j=col, p) against PointCloudImage(scan_w, scan_h) produces for the
 * GE/MF modes: transposed + cropped (Q11): pc[j][i] iff i<scan_w && j<scan_h.
 * pc_sum [scan_h][scan_w][3], pc_count [scan_h][scan_w], pc_color [scan_h][scan_w] (NULL ok). */
int slr_pointcloud_from_grid(slr_ctx *ctx, const float *xyz, const uint8_t *has, const uint8_t *color,
                             int W, int H, int scan_w, int scan_h,
                             float *pc_sum, uint8_t *pc_count, uint8_t *pc_color, slr_mem mem);
/* getPoint (pointcloudimage.cpp:56-67): out[n][3] = sum / (float)count, 0 where count==0 */
int slr_pointcloud_get(slr_ctx *ctx, const float *pc_sum, const uint8_t *pc_count, size_t n,
                       float *out, slr_mem mem);

/* Utilities::line_lineIntersection (utilities.cpp:399-425) over arrays: the line through p1[3] along v1[i][3] against the line
 * through p2[3] along v2[i][3]; out[i][3] = the midpoint of their closest points, ok[i] = 0 (and out = 0) where the reference
 * returns false (|a c - b b| < 0.1).  The function slr_ray_triangulate / slr_reconstruct_gray evaluate per pixel pair. */
int slr_line_line_intersections(slr_ctx *ctx, size_t n, const float *p1, const float *p2, const float *v1, const float *v2,
                                float *out, uint8_t *ok, slr_mem mem);

/* ---- whole-path drop-ins ----------------------------------------------------------------------------
 * One call per stereo frame = the body of run*Reconstruction between imread and MeshCreator.
 * rectify != 0: planes are RAW camera images and the ctx maps are applied (fused); 0: already rectified. */
int slr_reconstruct_mf(slr_ctx *ctx, const uint8_t *const planesL[SLR_MF_PLANES],
                       const uint8_t *const planesR[SLR_MF_PLANES], int pitch, int W, int H,
                       int black_thr, int rectify,
                       float *xyz, uint8_t *has, slr_mem mem);        /* MFReconstruct::runReconstruction */
int slr_reconstruct_ge(slr_ctx *ctx, const uint8_t *const *planesL, const uint8_t *const *planesR,
                       int n_col_bits, int pitch, int W, int H, int black_thr, int white_thr,
                       int scan_w, int rectify, int have_color,
                       float *xyz, uint8_t *has, uint8_t *color, slr_mem mem);  /* runReconstruction_GE */
int slr_reconstruct_gray(slr_ctx *ctx, const uint8_t *const *planesL, const uint8_t *const *planesR,
                         int n_col_bits, int n_row_bits, int pitch, int W, int H,
                         int black_thr, int white_thr, int scan_w, int scan_h,
                         float *xyz_sum, uint8_t *count, slr_mem mem);          /* runReconstruction */

/* ---- BASELINE config 3: Gray code + multi-frequency phase from ONE hybrid stack -----------------------------------------------
 * The reference's modes are exclusive (MainWindow::startreconstruct switches on codePatternUsed, mainwindow.h:94-95); a hybrid
 * scan projects both pattern sets and shares the white / black pair.  planes of a camera: 0 white, 1 black, 2 + 2c / 3 + 2c the
 * (pattern, inverse) pair of column bit c (MSB first, graycodes.cpp:55-114), then the 12 fringe planes 4f + s (3 frequencies x 4
 * steps, multifrequency.cpp:14-33): 2 + 2 n_col_bits + 12 planes.  Raw camera images; the ctx maps are applied (fused).
 * Each output is exactly what its own mode computes: code_x = Reconstruct::decodePatterns_GE / getProjPixel_GE
 * (reconstruct.cpp:79-97, 381-407; -1 where masked or erroneous), phase = MFReconstruct::decodePatterns / getPhase
 * (mfreconstruct.cpp:210-269; a quiet NaN where the pixel is invalid, see slr_mf_rectify_decode_pair).  When the LDS-DMA form
 * applies (equally spaced planes, W % 16 == 0, the default tile shape) the default is TWO launches over the one stack, each for both
 * cameras: the fused Gray decode on planes 0 .. 2 n + 1, then the fused multi-frequency decode on white, black and the fringes
 * behind the Gray planes; SLR_OPT_HYBRID_ONE_PASS = 1 selects ONE kernel over all planes of a tile instead (shadow mask, map digest
 * and source-box geometry shared; measured slower: DESIGN.md).  Otherwise two fused decodes per camera.  Same results every way. */
int slr_hybrid_rectify_decode_pair(slr_ctx *ctx, const uint8_t *const *planesL, const uint8_t *const *planesR, int n_col_bits,
                                   int pitch, int W, int H, int black_thr, int white_thr, int scan_w,
                                   int32_t *code_xL, float *phaseL, int32_t *code_xR, float *phaseR, slr_mem mem);
/* device-resident batch: stack [n_frames][2 cams][planes_per_cam >= 2 + 2 n_col_bits + 12][H][pitch]; per frame the hybrid decode
 * above, then MFReconstruct::triangulation on the two phase images (slr_mf_triangulate) -> xyz [n][H][W][3], has [n][H][W];
 * code_x: [n][2 cams][H][W] i32 or NULL (the codes are not needed by the caller). */
int slr_reconstruct_hybrid_batch(slr_ctx *ctx, int n_frames, const uint8_t *stack, int planes_per_cam, int n_col_bits, int pitch,
                                 int W, int H, int black_thr, int white_thr, int scan_w, float *xyz, uint8_t *has, int32_t *code_x);

/* slr_reconstruct_mf followed by slr_pointcloud_from_grid without the W x H XYZ grid ever leaving the device: what
 * MFReconstruct::runReconstruction hands to the application (mfreconstruct.cpp:160-187 -> points3DProjView). */
int slr_reconstruct_mf_cloud(slr_ctx *ctx, const uint8_t *const planesL[SLR_MF_PLANES], const uint8_t *const planesR[SLR_MF_PLANES],
                             int pitch, int W, int H, int black_thr, int rectify, int scan_w, int scan_h,
                             float *pc_sum, uint8_t *pc_count, slr_mem mem);

/* Device-resident batch fast path (frames of one GPU's shard).  stack = [n_frames][2 cams][14][H][pitch]
 * contiguous u8 in HBM; xyz = [n_frames][H][W][3], has = [n_frames][H][W].  Always SLR_MEM_DEVICE. */
int slr_reconstruct_mf_batch(slr_ctx *ctx, int n_frames, const uint8_t *stack, int pitch, int W, int H,
                             int black_thr, int rectify, float *xyz, uint8_t *has);

/* The same for every reconstruction mode (MainWindow::startreconstruct's switch on codePatternUsed, mainwindow.cpp:562-652;
 * mode ids as mainwindow.h:95): one GPU's shard of frames, device-resident, processed back to back on the ctx stream.
 *   stack            [n_frames][2 cams][planes_per_cam][H][pitch] contiguous u8 in HBM
 *   SLR_MODE_MF      planes_per_cam >= 14; xyz [n][H][W][3], has [n][H][W]
 *   SLR_MODE_GE      planes_per_cam >= 2 + 2 n_col_bits; xyz / has as above, color [n][H][W] or NULL (have_color = 0)
 *   SLR_MODE_GRAY    planes_per_cam >= 2 + 2 n_col_bits + 2 n_row_bits; xyz = PointCloudImage sums [n][scan_h][scan_w][3],
 *                    has = the u8 counts [n][scan_h][scan_w]
 * Every argument is validated before the first frame is enqueued, so a call either fails as a whole or only a HIP runtime
 * error (reported by status, frames before it already enqueued) can interrupt it. */
#define SLR_MODE_GRAY 0
#define SLR_MODE_GE 1
#define SLR_MODE_MF 2
typedef struct slr_batch_desc {
    int mode, n_frames, planes_per_cam;
    int pitch, W, H;
    int black_thr, white_thr;
    int n_col_bits, n_row_bits, scan_w, scan_h;   /* Gray modes */
    int rectify, have_color;
} slr_batch_desc;
int slr_reconstruct_batch(slr_ctx *ctx, const slr_batch_desc *desc, const uint8_t *stack, float *xyz, uint8_t *has,
                          uint8_t *color);

/* Several GPUs of one node from ONE process (what a Qt/C++ host can call; SURVEY 8e: frames shard across GPUs, no
 * data-path collective, the final point cloud is assembled over xGMI).  ctxs[k] lives on its own device (slr_create(dev, ..));
 * several ctx on one device are allowed (testing).  Frame f of the job is frame f / n_ctx of ctxs[f % n_ctx]'s shard:
 *   stacks[k]  device-resident on ctxs[k]'s device: [frames of k][2 cams][14][H][pitch] u8
 *   xyz[k], has[k]  ctxs[k]'s own results, [frames of k][H][W][3] / [frames of k][H][W], on its device
 *   gather_ctx >= 0: xyz_all [n_frames][H][W][3] and has_all [n_frames][H][W] on ctxs[gather_ctx]'s device receive every frame
 *   in job order by direct peer copies (hipMemcpyPeerAsync on the producing stream: one xGMI hop per source); -1: no assembly.
 * All devices run concurrently; the call returns when every stream is idle (the same calibration / maps / options must have
 * been installed on every ctx).  Every context's arguments are validated before any of them is given work, and after a HIP
 * runtime error the call returns only once every stream already started has drained (the caller may free its buffers).
 * MFReconstruct::runReconstruction x n_frames (mfreconstruct.cpp:160-187). */
int slr_reconstruct_mf_multi(slr_ctx *const *ctxs, int n_ctx, int n_frames, const uint8_t *const *stacks, int pitch, int W,
                             int H, int black_thr, int rectify, float *const *xyz, uint8_t *const *has, int gather_ctx,
                             float *xyz_all, uint8_t *has_all);

/* north_star's exchange step ("all-gather ... only to assemble the final point cloud") from one process: EVERY device ends with
 * the assembled cloud.  ctxs[k] computes its frames (f % n_ctx == k) straight into their slots of ITS OWN xyz_all[k] /
 * has_all[k] ([n_frames][H][W][3] / [n_frames][H][W] on its device); its stream then pushes each finished frame into the same
 * slot on every other device (hipMemcpyPeerAsync: n_ctx - 1 concurrent one-hop copies per frame over the point-to-point xGMI
 * mesh, no ring, no staging buffer).  Every argument of every context is validated before any work is enqueued; if a HIP
 * runtime error interrupts the call, it returns only after every stream already started has drained.
 *   *peer_direct (may be NULL) = 1: every destination was directly addressable from every source (same device or peer access),
 *                                0: the runtime stages at least one pair through the host (results are the same, only slower);
 *   require_peer != 0 turns that case into SLR_ERR_UNSUPPORTED (nothing enqueued). */
int slr_reconstruct_mf_allgather(slr_ctx *const *ctxs, int n_ctx, int n_frames, const uint8_t *const *stacks, int pitch, int W,
                                 int H, int black_thr, int rectify, float *const *xyz_all, uint8_t *const *has_all,
                                 int require_peer, int *peer_direct);

/* The same with the frame -> context assignment chosen by the caller, and the exchange step on its own.
 *   SLR_ASSIGN_CYCLIC   frame f -> ctxs[f % n_ctx], shard slot f / n_ctx (what slr_reconstruct_mf_multi / _allgather use);
 *   SLR_ASSIGN_BLOCKED  frame f -> ctxs[f / S], S = ceil(n_frames / n_ctx): a context's shard (stacks[k] = its frames in job
 *                       order) is ONE contiguous piece of the assembled arrays, computed in place by the batch entry in groups of
 *                       frames (one fused-decode and one match launch per group, SLR_OPT_MF_BATCH_GROUP) and pushed as one copy per
 *                       destination and group.  The faster of the two; config 4 (64 frames, 8 GPUs) is one group per GPU.
 * A source's pushes to its n_ctx - 1 peers run on per-destination streams (one point-to-point xGMI link each, side by side) behind
 * the group that produced them and beside the next group's kernels.
 * slr_allgather_clouds is the exchange alone: every ctxs[k] already holds ITS frames in their slots of its own xyz_all[k] /
 * has_all[k] (e.g. slr_reconstruct_mf_multi with gather_ctx = -1 and xyz[k] pointing at its shard's place in xyz_all[k]; any work
 * queued on ctxs[k]'s stream is waited for) and pushes them to every other device; it returns when every copy has landed.
 * No reference counterpart (the reference is single-threaded, single-device); north_star: "RCCL all-gather over xGMI
 * only to assemble the final point cloud" -- this is that step for a host that drives all GPUs from one process. */
#define SLR_ASSIGN_CYCLIC 0
#define SLR_ASSIGN_BLOCKED 1
int slr_reconstruct_mf_allgather_ex(slr_ctx *const *ctxs, int n_ctx, int n_frames, const uint8_t *const *stacks, int pitch, int W,
                                    int H, int black_thr, int rectify, float *const *xyz_all, uint8_t *const *has_all,
                                    int assignment, int require_peer, int *peer_direct);
int slr_allgather_clouds(slr_ctx *const *ctxs, int n_ctx, int n_frames, int W, int H, float *const *xyz_all,
                         uint8_t *const *has_all, int assignment, int require_peer, int *peer_direct);

/* Ordered prefix index of a u8 flag image [h][w] (nonzero = flagged), enumerated row-major (column_major = 0) or in the
 * column-outer / row-inner order in which MeshCreator numbers the vertices of a PointCloudImage (meshcreator.cpp:21-33,
 * 71-84): index[j][i] = first + (flagged elements before (i, j) in that order), or `none` where the flag is 0; *total = the
 * flagged count.  slr_compact_points: the flagged points of an XYZ array, in array order, into out_xyz (capacity n; out_src,
 * optional, gets their source positions) -- the sparse form of a point cloud for transport. */
int slr_prefix_index(slr_ctx *ctx, const uint8_t *flags, int w, int h, int column_major, uint32_t first, uint32_t none,
                     uint32_t *index, uint32_t *total, slr_mem mem);
int slr_compact_points(slr_ctx *ctx, const float *xyz, const uint8_t *has, size_t n, float *out_xyz, uint32_t *out_src,
                       uint32_t *count, slr_mem mem);

/* The proof of an exchange.  slr_cloud_checksums: one 64-bit word per frame of a device-resident cloud (xyz [n][H][W][3], has
 * [n][H][W] on ctx's device; position-weighted sums of the raw bits) -> out[n_frames] on the host.  slr_verify_assembled: every
 * context checksums the assembled arrays it holds after slr_reconstruct_mf_allgather (or any other all-gather) on ITS device and
 * the words are compared: *mismatches = (context, frame) pairs that differ from context 0's.  Synchronous. */
int slr_cloud_checksums(slr_ctx *ctx, int n_frames, int W, int H, const float *xyz, const uint8_t *has, uint64_t *out);
int slr_verify_assembled(slr_ctx *const *ctxs, int n_ctx, int n_frames, int W, int H, float *const *xyz_all,
                         uint8_t *const *has_all, int *mismatches);

/* page-locked host memory (the H2D / D2H copies of SLR_MEM_HOST calls are truly asynchronous only from / to it) */
int slr_host_alloc(void **ptr, size_t bytes);
int slr_host_free(void *ptr);

/* ---- measurement hooks (bench.py): HIP-event timing on the ctx stream ------------------------------ */
int slr_timer_begin(slr_ctx *ctx);                 /* records an event on the ctx stream */
int slr_timer_end(slr_ctx *ctx, float *ms);        /* records + synchronises, returns elapsed ms */
/* the box's streaming rate as MI355X_MICROARCH.md measures it: a float4 non-temporal copy (one word per thread) of `bytes` (a multiple of 16; both
 * device pointers 16-byte aligned) on the ctx stream, asynchronous -- bracket it with slr_timer_begin / _end */
int slr_stream_copy(slr_ctx *ctx, void *dst, const void *src, size_t bytes);
/* ... and with a kernel's read : write mix: every 16-byte word of dst is the sum of `reads` (1..8) words read from `reads`
 * consecutive streams of bytes_out bytes each (src holds reads * bytes_out bytes); reads = 5 is the fused MF decode's 20 bytes read
 * per 4 written.  Same conditions as slr_stream_copy (one 16-byte word per thread, non-temporal; bench.py's read-mix ceiling). */
int slr_stream_mix(slr_ctx *ctx, void *dst, const void *src, size_t bytes_out, int reads);
/* per-kernel profiler: when enabled every kernel launch is bracketed by two HIP events on the ctx stream */
int          slr_profile_enable(slr_ctx *ctx, int on);
int          slr_profile_reset(slr_ctx *ctx);
int          slr_profile_kernel_count(void);
const char  *slr_profile_kernel_name(int kernel_id);
int          slr_profile_get(slr_ctx *ctx, int kernel_id, double *total_ms, long *launches);   /* synchronises */

#ifdef __cplusplus
}
#endif
#endif /* SLR_H */
