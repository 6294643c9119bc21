// slr_capi.hip -- the C ABI of libslr_hip.so (include/slr.h): context, device memory staging, launch
// sequencing and the built-in HIP-event profiler.  Host-side C++ compiled by hipcc; every entry point is
// extern "C", takes plain pointers and sizes, and never lets an exception escape.
//
// There is deliberately no CPU implementation behind these symbols: if no gfx950-class device is usable,
// slr_create fails and nothing else can be called.
#include "slr_device.hpp"

#include <math.h>
#include <stdio.h>
#include <string.h>

#include <new>
#include <string>
#include <vector>

using namespace slr;

namespace {

const char *kKernelNames[K_COUNT] = {
    "slr_remap_u8", "slr_mf_decode", "slr_mf_rectify_decode", "slr_gray_decode", "slr_gray_rectify_decode",
    "slr_mf_match_triangulate", "slr_ge_match_triangulate", "slr_ray_count", "slr_ray_scan", "slr_ray_scatter",
    "slr_ray_triangulate",
    "slr_pc_from_grid", "slr_pc_get", "slr_undistort_table", "slr_ray_table", "slr_mf_rectify_decode_pair", "slr_mfn_decode",
    "slr_gray_rectify_decode_pair", "slr_hybrid_rectify_decode_pair", "slr_mfn_rectify_decode"};

// scratch slots (device buffers owned by the ctx, grown on demand, reused across calls)
enum Slot {
    S_STAGE0 = 0,            // host-mode staging: 16 generic slots
    S_PHASE_L = 16, S_VALID_L, S_PHASE_R, S_VALID_R,
    S_CODEX_L, S_CODEY_L, S_CODEX_R, S_CODEY_R,
    S_RAY_CELL, S_RAY_CELL2, S_RAY_RANK, S_RAY_CNT, S_RAY_OFFS, S_RAY_ITEMS, S_RAY_LIST, S_SCAN_TMP,
    S_XYZ, S_HAS, S_COLOR, S_UND_L, S_UND_R, S_RAYS_L, S_RAYS_R, S_FLAGSCAN, S_COMPACT, S_K4_DEFER,
    S_COUNT
};

struct ProfRec { hipEvent_t a, b; int id; int units = 1; };   // units: frames one launch served (the profile reports time per frame)

}  // namespace

namespace slr { thread_local hipEvent_t tl_prof_start = nullptr, tl_prof_stop = nullptr; thread_local DebugKnobs tl_debug; }

struct slr_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string err;
    bool has_calib = false;
    DevCalib cal;
    float *d_lut = nullptr;
    // tile tickets of the LDS-DMA fused decodes: zero between launches (the last workgroup of a launch clears them), which holds
    // because every fused decode of a context runs on c->stream, one launch after the other, and slr_set_stream drains the old stream
    // before it switches; a launch that faults leaves the context unusable anyway (every later call returns the HIP error)
    unsigned *d_sched = nullptr;
    int16_t *d_map_xy[2] = {nullptr, nullptr};
    uint16_t *d_map_frac[2] = {nullptr, nullptr};
    void *d_tile_box[2] = {nullptr, nullptr};   // per-tile source bounding boxes of the maps (launch_tile_boxes)
    unsigned tile_nofit[2][2] = {};             // per camera: tiles that do not fit the 64x8 / the 128x8 fused-decode form
    void *d_dma_tiles[2] = {nullptr, nullptr};  // boxes + map digest of the LDS-DMA fused decode (launch_dma_tiles), or null
    unsigned dma_stats[2][kDmaTileStats] = {};  // launch_dma_tiles' statistics; [0] = tiles whose box does not fit that form
    int dma_shape_built[2] = {-1, -1};          // SLR_OPT_RECT_DMA_SHAPE the tables were built for
    DebugKnobs debug;              // SLR_OPT_DEBUG_*
    int opt_dma_shape = 3, opt_dma_depth = 2;   // SLR_OPT_RECT_DMA_SHAPE / _DEPTH (128x16 tiles on 512 threads, two phases of DMA in flight: measured best)
    int map_w = 0, map_h = 0;
    int opt_mf_match_algo = 0;     // SLR_OPT_MF_MATCH_ALGO
    int opt_mf_decode_vec = 0;     // SLR_OPT_MF_DECODE_VEC
    int opt_rect_algo = 0;         // SLR_OPT_RECT_DECODE_ALGO
    int opt_async_host = 0;        // SLR_OPT_ASYNC_HOST
    int opt_hybrid_one_pass = 0;   // SLR_OPT_HYBRID_ONE_PASS
    int opt_batch_streams = 2;     // SLR_OPT_BATCH_STREAMS
    int opt_mf_batch_group = 8;    // SLR_OPT_MF_BATCH_GROUP
    int opt_mf_decode_group = 8;   // SLR_OPT_MF_BATCH_DECODE_GROUP
    bool und_valid = false;        // undistortion tables (S_UND_L/S_UND_R) match cal and und_w x und_h
    int und_w = 0, und_h = 0;
    bool rays_valid = false;       // unit-ray tables (S_RAYS_L/S_RAYS_R) match cal and rays_w x rays_h
    int rays_w = 0, rays_h = 0;
    void *scratch[2 * S_COUNT] = {};            // two sets: slr_reconstruct_batch pipelines GRAY_ONLY frames over two streams
    size_t scratch_cap[2 * S_COUNT] = {};
    int scratch_set = 0;                        // the set get_scratch hands out (calibration tables live in set 0 only)
    hipStream_t pipe = nullptr;                 // the second stream of that pipeline
    hipEvent_t ev_pipe[2] = {nullptr, nullptr};
    hipEvent_t mid_event = nullptr;             // recorded by core_ray between a GRAY_ONLY frame's two halves (the pipeline's skew)
    // one-process multi-GPU exchange: one push stream per DESTINATION context (the copies of one source to its n - 1 peers run
    // side by side, one per point-to-point xGMI link, instead of one after the other on the compute stream)
    std::vector<hipStream_t> push_streams;
    std::vector<hipEvent_t> push_done;
    hipEvent_t ev_computed = nullptr;           // recorded on `stream` when a group of frames is ready to be pushed
    // profiler
    bool profiling = false;
    std::vector<ProfRec> pending;
    std::vector<hipEvent_t> free_events;
    double prof_ms[K_COUNT] = {};
    long prof_n[K_COUNT] = {};
    long prof_seen[K_COUNT] = {};
    int opt_profile_stride = 1;    // SLR_OPT_PROFILE_STRIDE
    hipEvent_t t0 = nullptr, t1 = nullptr;
};

namespace {

int fail(slr_ctx *c, int code, const char *what, const char *detail = nullptr)
{
    if (c) {
        c->err = what;
        if (detail) { c->err += ": "; c->err += detail; }
    }
    return code;
}

// SLR_OPT_RECT_DECODE_ALGO resolved for the multi-frequency decode of cameras a..b: 0 (auto) takes the 128x8 / 512-thread
// form (5) unless more of these maps' pixels would fall back to the per-pixel gather there than in the 64x8 form (6)
int mf_rect_algo(const slr_ctx *c, int a, int b)
{
    if (c->opt_rect_algo != 0 && c->opt_rect_algo != 7) return c->opt_rect_algo;
    unsigned mid = 0, wide = 0;
    for (int cam = a; cam <= b; cam++) { mid += c->tile_nofit[cam][0]; wide += c->tile_nofit[cam][1]; }
    return 2u * wide <= mid ? 5 : 6;
}

#define SLR_HIP(c, expr)                                                                       \
    do {                                                                                       \
        hipError_t e__ = (expr);                                                               \
        if (e__ != hipSuccess) return fail((c), e__ == hipErrorOutOfMemory ? SLR_ERR_OOM : SLR_ERR_HIP, #expr, hipGetErrorString(e__)); \
    } while (0)

#define SLR_TRY(expr)                \
    do {                             \
        int s__ = (expr);            \
        if (s__ != SLR_OK) return s__; \
    } while (0)

// the LDS-DMA form (7) is used when it is asked for or auto is on, and the installed maps suit it: tables built for the
// current tile shape, and at most a quarter of the tiles with a source box the form does not hold (those are rewritten by its
// gather fix-up pass; beyond that round 1's forms, which gather per tile inside the kernel, are the better choice)
bool dma_form_wanted(const slr_ctx *c, int a, int b)
{
    if (c->opt_rect_algo != 0 && c->opt_rect_algo != 7) return false;
    for (int cam = a; cam <= b; cam++) {
        if (!c->d_dma_tiles[cam] || c->dma_shape_built[cam] != c->opt_dma_shape) return false;
        if (4ull * c->dma_stats[cam][0] > dma_tile_count_of(c->map_w, c->map_h, c->opt_dma_shape)) return false;
    }
    return true;
}

// what the fix-up pass of the LDS-DMA form needs for cameras a (and b): the original maps and the number of listed tiles
DmaFixup dma_fixup_of(const slr_ctx *c, int a, int b)
{
    DmaFixup f;
    const int cams[2] = {a, b};
    for (int k = 0; k < 2; k++) {
        f.map_xy[k] = c->d_map_xy[cams[k]]; f.map_frac[k] = c->d_map_frac[cams[k]]; f.nofit[k] = c->dma_stats[cams[k]][0];
        const unsigned cap = dma_extra_entries_capacity(c->map_w, c->map_h, c->opt_dma_shape), used = c->dma_stats[cams[k]][7];
        f.extras[k] = used < cap ? used : cap;
    }
    return f;
}

int use_device(slr_ctx *c)
{
    tl_debug = c->debug;                                     // every entry point passes here before it launches anything
    SLR_HIP(c, hipSetDevice(c->device));
    return SLR_OK;
}

int get_scratch(slr_ctx *c, int slot, size_t bytes, void **out)
{
    if (bytes == 0) bytes = 16;
    if (c->scratch_set && slot != S_RAYS_L && slot != S_RAYS_R && slot != S_UND_L && slot != S_UND_R) slot += S_COUNT;
    if (c->scratch_cap[slot] < bytes) {
        if (c->scratch[slot]) {
            SLR_HIP(c, hipStreamSynchronize(c->stream));
            SLR_HIP(c, hipFree(c->scratch[slot]));
            c->scratch[slot] = nullptr;
            c->scratch_cap[slot] = 0;
        }
        const size_t cap = (bytes + 255) & ~(size_t)255;
        SLR_HIP(c, hipMalloc(&c->scratch[slot], cap));
        c->scratch_cap[slot] = cap;
    }
    *out = c->scratch[slot];
    // tests (SLR_OPT_DEBUG_POISON_SCRATCH): what a call gets is never what the previous call left there.  The calibration tables
    // are cached across calls (und_valid / rays_valid) and keep their contents; the second scratch set is used on a second
    // stream, hence the device-wide synchronisation instead of stream order.
    if (c->debug.poison_scratch) {
        const int base = slot >= S_COUNT ? slot - S_COUNT : slot;
        if (base != S_RAYS_L && base != S_RAYS_R && base != S_UND_L && base != S_UND_R) {
            SLR_HIP(c, hipDeviceSynchronize());
            SLR_HIP(c, hipMemset(c->scratch[slot], 0x7B, c->scratch_cap[slot]));
            SLR_HIP(c, hipDeviceSynchronize());
        }
    }
    return SLR_OK;
}

// ---- per-call staging of host pointers ------------------------------------------------------------------
struct Stage {
    slr_ctx *c;
    slr_mem mem;
    int next = S_STAGE0;
    struct Out { void *host; void *dev; size_t bytes; };
    std::vector<Out> outs;

    Stage(slr_ctx *ctx, slr_mem m) : c(ctx), mem(m) {}

    int in(const void *p, size_t bytes, const void **dev)
    {
        if (!p || mem == SLR_MEM_DEVICE) { *dev = p; return SLR_OK; }
        if (next >= S_STAGE0 + 16) return fail(c, SLR_ERR_INVALID_ARG, "too many staged buffers");
        void *d;
        SLR_TRY(get_scratch(c, next++, bytes, &d));
        SLR_HIP(c, hipMemcpyAsync(d, p, bytes, hipMemcpyHostToDevice, c->stream));
        *dev = d;
        return SLR_OK;
    }
    int out(void *p, size_t bytes, void **dev)
    {
        if (!p || mem == SLR_MEM_DEVICE) { *dev = p; return SLR_OK; }
        if (next >= S_STAGE0 + 16) return fail(c, SLR_ERR_INVALID_ARG, "too many staged buffers");
        void *d;
        SLR_TRY(get_scratch(c, next++, bytes, &d));
        outs.push_back({p, d, bytes});
        *dev = d;
        return SLR_OK;
    }
    // N planes of H*pitch bytes each -> one contiguous device stack; fills dev plane pointers
    int planes(const uint8_t *const *pl, int n, int pitch, int H, const uint8_t **dev_ptrs)
    {
        if (mem == SLR_MEM_DEVICE) { for (int i = 0; i < n; i++) dev_ptrs[i] = pl[i]; return SLR_OK; }
        if (next >= S_STAGE0 + 16) return fail(c, SLR_ERR_INVALID_ARG, "too many staged buffers");
        const size_t plane = (size_t)pitch * H;
        void *d;
        SLR_TRY(get_scratch(c, next++, plane * n, &d));
        for (int i = 0; i < n; i++) {
            SLR_HIP(c, hipMemcpyAsync((uint8_t *)d + plane * i, pl[i], plane, hipMemcpyHostToDevice, c->stream));
            dev_ptrs[i] = (const uint8_t *)d + plane * i;
        }
        return SLR_OK;
    }
    int finish()
    {
        if (mem == SLR_MEM_DEVICE) return SLR_OK;
        for (auto &o : outs) SLR_HIP(c, hipMemcpyAsync(o.host, o.dev, o.bytes, hipMemcpyDeviceToHost, c->stream));
        if (!c->opt_async_host) SLR_HIP(c, hipStreamSynchronize(c->stream));   // SLR_OPT_ASYNC_HOST: the caller syncs
        return SLR_OK;
    }
};

// ---- profiler ---------------------------------------------------------------------------------------------
int prof_event(slr_ctx *c, hipEvent_t *e)
{
    if (!c->free_events.empty()) { *e = c->free_events.back(); c->free_events.pop_back(); return SLR_OK; }
    SLR_HIP(c, hipEventCreate(e));
    return SLR_OK;
}

// exact == true (scopes around ONE kernel launch): the events are armed for SLR_LAUNCH, which stamps them at the kernel's
// own start and end; otherwise (several launches, or a library call such as the rocprim scan) they bracket the scope
struct ProfScope {
    slr_ctx *c; int id; ProfRec r{}; bool on, exact;
    ProfScope(slr_ctx *ctx, int kid, bool exact_ = false) : c(ctx), id(kid), on(ctx->profiling), exact(exact_)
    {
        if (!on) return;
        // sampling: bracket only every opt_profile_stride-th launch of this kernel (two event records per launch cost
        // ~2.5 % of a 0.35 ms frame each)
        if (c->opt_profile_stride > 1 && (c->prof_seen[kid]++ % c->opt_profile_stride) != 0) { on = false; return; }
        if (prof_event(c, &r.a) != SLR_OK || prof_event(c, &r.b) != SLR_OK) { on = false; return; }
        r.id = id;
        if (exact) { tl_prof_start = r.a; tl_prof_stop = r.b; }
        else (void)hipEventRecord(r.a, c->stream);
    }
    ~ProfScope()
    {
        if (!on) return;
        if (exact) {
            if (tl_prof_start == r.a) {                  // no kernel was launched in this scope: no sample
                tl_prof_start = tl_prof_stop = nullptr;
                c->free_events.push_back(r.a); c->free_events.push_back(r.b);
                return;
            }
        } else {
            (void)hipEventRecord(r.b, c->stream);
        }
        c->pending.push_back(r);
    }
};

int prof_drain(slr_ctx *c)
{
    if (c->pending.empty()) return SLR_OK;
    SLR_HIP(c, hipStreamSynchronize(c->stream));
    for (auto &r : c->pending) {
        float ms = 0;
        SLR_HIP(c, hipEventElapsedTime(&ms, r.a, r.b));
        c->prof_ms[r.id] += ms;
        c->prof_n[r.id] += r.units;
        c->free_events.push_back(r.a);
        c->free_events.push_back(r.b);
    }
    c->pending.clear();
    return SLR_OK;
}

int check_dims(slr_ctx *c, int W, int H, int pitch)
{
    if (W <= 0 || H <= 0 || pitch < W) return fail(c, SLR_ERR_INVALID_ARG, "bad image size/pitch");
    if ((long long)W * H >= (1ll << 31)) return fail(c, SLR_ERR_UNSUPPORTED, "image too large");
    if (W > 65535 || H > 65535) return fail(c, SLR_ERR_UNSUPPORTED, "W,H must be < 65536");
    if ((long long)pitch * H >= (1ll << 31)) return fail(c, SLR_ERR_UNSUPPORTED, "pitch * H must be < 2^31 (32-bit plane offsets)");
    return SLR_OK;
}

int need_maps(slr_ctx *c, int cam, int W, int H)
{
    if (cam < 0 || cam > 1) return fail(c, SLR_ERR_INVALID_ARG, "cam must be 0 (left) or 1 (right)");
    if (!c->d_map_xy[cam]) return fail(c, SLR_ERR_NOT_CONFIGURED, "rectify maps not set (slr_set_rectify_maps)");
    if (c->map_w != W || c->map_h != H) return fail(c, SLR_ERR_INVALID_ARG, "image size differs from rectify map size");
    return SLR_OK;
}

int need_calib(slr_ctx *c)
{
    if (!c->has_calib) return fail(c, SLR_ERR_NOT_CONFIGURED, "calibration not set (slr_set_calibration)");
    return SLR_OK;
}

// ---- device-pointer cores (shared by the single-op entry points and the pipelines) ------------------------
int core_mf_decode(slr_ctx *c, int cam, bool rectify, const uint8_t *const *pl, int pitch, int W, int H,
                   int black_thr, float *phase, uint8_t *valid)
{
    MfPlanes mp;
    for (int i = 0; i < SLR_MF_PLANES; i++) mp.p[i] = pl[i];
    ProfScope ps(c, rectify ? K_MF_RECT_DECODE : K_MF_DECODE, true);
    if (rectify && dma_form_wanted(c, cam, cam)) {
        bool done = false;
        float *const ph[1] = {phase};
        uint8_t *const vd[1] = {valid};
        const void *const tl[1] = {c->d_dma_tiles[cam]};
        const DmaFixup fix = dma_fixup_of(c, cam, cam);
        SLR_HIP(c, launch_mf_rect_decode_dma(&mp, 1, pitch, W, H, black_thr, c->d_lut, ph, vd, tl, c->opt_dma_shape, c->opt_dma_depth,
                                             c->d_sched, &fix, &done, c->stream));
        if (done) return SLR_OK;
    }
    if (rectify && c->opt_rect_algo == 7)
        return fail(c, SLR_ERR_UNSUPPORTED, "SLR_OPT_RECT_DECODE_ALGO = 7 (LDS-DMA form) does not apply: it needs 14 equally spaced planes in one "
                                            "allocation, 16-byte aligned rows, W % 16 == 0 and maps whose tile boxes fit");
    SLR_HIP(c, launch_mf_decode(mp, pitch, W, H, black_thr, c->d_lut, phase, valid,
                                rectify ? c->d_map_xy[cam] : nullptr, rectify ? c->d_map_frac[cam] : nullptr,
                                rectify ? c->d_tile_box[cam] : nullptr, c->opt_mf_decode_vec, mf_rect_algo(c, cam, cam), c->stream));
    return SLR_OK;
}

// K4 with the per-(calibration,size) undistortion tables (built lazily, invalidated by slr_set_calibration)
int core_mf_match(slr_ctx *c, const float *phL, const uint8_t *vL, const float *phR, const uint8_t *vR, int W, int H,
                  float *xyz, uint8_t *has, int32_t *match_k, int row0 = 0, int rows = -1, int nframes = 1, size_t frame_px = 0,
                  bool *batched = nullptr /* nframes > 1: *batched = false and nothing launched when one launch cannot take them */)
{
    // H: height of the IMAGE (tables); the arrays hold `rows` rows starting at image row row0 (default: all of it)
    if (rows < 0) rows = H;
    const size_t n = (size_t)W * H;
    const float *undL = nullptr, *undR = nullptr;
    if (c->opt_mf_match_algo != 1 && W <= 4096 * 8) {     // the indexed forms (launch_mf_match) read the tables
        void *a, *b;
        if (c->scratch_cap[S_UND_L] < n * 8 || c->scratch_cap[S_UND_R] < n * 4) c->und_valid = false;   // will realloc
        SLR_TRY(get_scratch(c, S_UND_L, n * 8, &a));
        SLR_TRY(get_scratch(c, S_UND_R, n * 4, &b));
        if (!c->und_valid || c->und_w != W || c->und_h != H) {
            ProfScope ps(c, K_UNDISTORT_TABLE);
            SLR_HIP(c, launch_undistort_tables(c->cal, W, H, (float *)a, (float *)b, c->stream));
            c->und_valid = true; c->und_w = W; c->und_h = H;
        }
        undL = (const float *)a; undR = (const float *)b;
    }
    if (nframes > 1) {
        *batched = !vL && !vR && !match_k && row0 == 0 && rows == H &&
                   mf_match_batches_frames(phL, phR, xyz, has, W, c->cal, c->opt_mf_match_algo, undL, undR, frame_px);
        if (!*batched) return SLR_OK;
    }
    void *defer = nullptr;                                // rows of 4097..8192 pixels: the list of rows the wide kernel hands to the chunked one
    if (W > 4096 && W <= 8192) SLR_TRY(get_scratch(c, S_K4_DEFER, sizeof(int) * ((size_t)rows + 1), &defer));
    ProfScope ps(c, K_MF_MATCH, true);
    ps.r.units = nframes;
    SLR_HIP(c, launch_mf_match(phL, vL, phR, vR, W, rows, row0, c->cal, xyz, has, match_k, c->opt_mf_match_algo, undL, undR,
                               c->stream, nframes, frame_px, (int *)defer));
    return SLR_OK;
}

int core_gray_decode(slr_ctx *c, int cam, bool rectify, const uint8_t *const *pl, int ncol, int nrow, int pitch,
                     int W, int H, int black_thr, int white_thr, int scan_w, int scan_h, int32_t *cx, int32_t *cy,
                     uint8_t *valid)
{
    GrayPlanes gp;
    const int n = 2 + 2 * ncol + 2 * nrow;
    for (int i = 0; i < SLR_MAX_GRAY_PLANES; i++) gp.p[i] = i < n ? pl[i] : nullptr;
    ProfScope ps(c, rectify ? K_GRAY_RECT_DECODE : K_GRAY_DECODE, true);
    if (rectify && dma_form_wanted(c, cam, cam)) {          // the LDS-DMA form (kernels_rectdma.hip), when the maps and the stack allow
        bool done = false;
        int32_t *const xs[1] = {cx}, *const ys[1] = {cy};
        uint8_t *const vd[1] = {valid};
        const void *const tl[1] = {c->d_dma_tiles[cam]};
        const DmaFixup fix = dma_fixup_of(c, cam, cam);
        SLR_HIP(c, launch_gray_rect_decode_dma(&gp, 1, ncol, nrow, pitch, W, H, black_thr, white_thr, scan_w, scan_h, xs, ys, vd, tl,
                                               c->opt_dma_shape, c->opt_dma_depth, c->d_sched, &fix, &done, c->stream));
        if (done) return SLR_OK;
    }
    if (rectify && c->opt_rect_algo == 7)
        return fail(c, SLR_ERR_UNSUPPORTED, "SLR_OPT_RECT_DECODE_ALGO = 7 (LDS-DMA form) does not apply: it needs equally spaced planes in one "
                                            "allocation, 16-byte aligned rows, W % 16 == 0, at least two code bits, a 4-pixel tile shape "
                                            "(SLR_OPT_RECT_DMA_SHAPE 1, 3, 4 or 5) and maps whose tile boxes fit");
    SLR_HIP(c, launch_gray_decode(gp, ncol, nrow, pitch, W, H, black_thr, white_thr, scan_w, scan_h, cx, cy, valid,
                                  rectify ? c->d_map_xy[cam] : nullptr, rectify ? c->d_map_frac[cam] : nullptr,
                                  rectify ? c->d_tile_box[cam] : nullptr, c->opt_rect_algo, c->stream));
    return SLR_OK;
}

// GRAY_ONLY from the planes: the decode is fused into the bucket histogram (gray_decode_count_kernel) when the stack allows it
struct RayPlanes { const uint8_t *const *pl[2]; int ncol, nrow, pitch, black_thr, white_thr; };

int core_ray(slr_ctx *c, const int32_t *cxL, const int32_t *cyL, const uint8_t *vL, const int32_t *cxR,
             const int32_t *cyR, const uint8_t *vR, int W, int H, int scan_w, int scan_h, float *xyz_sum,
             uint8_t *count, const RayPlanes *planes = nullptr)
{
    const size_t n = (size_t)W * H;
    const unsigned long long nb = (unsigned long long)scan_w * scan_h;
    if (nb >= (1ull << 30)) return fail(c, SLR_ERR_UNSUPPORTED, "scan_w*scan_h too large");
    const size_t ncell = 2 * (size_t)nb + 1;                 // left cells, right cells, end sentinel
    void *cell, *rank, *cnt, *offs, *items, *tmp;
    const size_t tb = ray_scan_temp_bytes(ncell);
    SLR_TRY(get_scratch(c, S_RAY_CELL, n * 4, &cell));
    SLR_TRY(get_scratch(c, S_RAY_RANK, n * 8, &rank));       // [2][n]
    SLR_TRY(get_scratch(c, S_RAY_CNT, ncell * 4, &cnt));
    SLR_TRY(get_scratch(c, S_RAY_OFFS, ncell * 4, &offs));
    SLR_TRY(get_scratch(c, S_RAY_ITEMS, ray_items_words(n) * 4, &items));
    SLR_TRY(get_scratch(c, S_SCAN_TMP, tb, &tmp));
    void *list;
    SLR_TRY(get_scratch(c, S_RAY_LIST, ray_list_words((size_t)nb) * 4, &list));
    void *raysL, *raysR;
    if (c->scratch_cap[S_RAYS_L] < n * 12 || c->scratch_cap[S_RAYS_R] < n * 12) c->rays_valid = false;   // will realloc
    SLR_TRY(get_scratch(c, S_RAYS_L, n * 12, &raysL));
    SLR_TRY(get_scratch(c, S_RAYS_R, n * 12, &raysR));
    if (!c->rays_valid || c->rays_w != W || c->rays_h != H) {
        ProfScope ps(c, K_RAY_TABLE);
        SLR_HIP(c, launch_ray_tables(c->cal, W, H, (float *)raysL, (float *)raysR, c->stream));
        c->rays_valid = true; c->rays_w = W; c->rays_h = H;
    }
    uint32_t *cellL = (uint32_t *)cell, *rankL = (uint32_t *)rank, *rankR = rankL + n;
    SLR_HIP(c, hipMemsetAsync(cnt, 0, ncell * 4, c->stream));
    void *cell2;
    SLR_TRY(get_scratch(c, S_RAY_CELL2, n * 4, &cell2));
    uint32_t *cellR = (uint32_t *)cell2;
    if (planes) {
        const int np = 2 + 2 * planes->ncol + 2 * planes->nrow;
        for (int cam = 0; cam < 2; cam++) {
            GrayPlanes gp;
            for (int i = 0; i < SLR_MAX_GRAY_PLANES; i++) gp.p[i] = i < np ? planes->pl[cam][i] : nullptr;
            ProfScope ps(c, K_GRAY_DECODE, true);
            SLR_HIP(c, launch_gray_decode_count(gp, planes->ncol, planes->nrow, planes->pitch, W, H, planes->black_thr, planes->white_thr,
                                                scan_w, scan_h, (uint32_t *)cnt + (cam ? nb : 0), cam ? cellR : cellL, cam ? rankR : rankL,
                                                c->stream));
        }
    } else {
      ProfScope ps(c, K_RAY_COUNT);
      SLR_HIP(c, launch_ray_count(cxL, cyL, vL, W, H, scan_w, scan_h, (uint32_t *)cnt, cellL, rankL, c->stream));
      SLR_HIP(c, launch_ray_count(cxR, cyR, vR, W, H, scan_w, scan_h, (uint32_t *)cnt + nb, cellR, rankR, c->stream)); }
    { ProfScope ps(c, K_RAY_SCAN);
      SLR_HIP(c, launch_ray_scan((const uint32_t *)cnt, (uint32_t *)offs, ncell, tmp, tb, c->stream)); }
    { ProfScope ps(c, K_RAY_SCATTER);                    // (K6's work list is made inside the same two launches)
      SLR_HIP(c, launch_ray_scatter_list(cellL, rankL, cellR, rankR, W, H, (const uint32_t *)offs, (uint32_t *)items, scan_w, scan_h,
                                         (uint32_t *)list, xyz_sum, count, c->stream)); }
    if (c->mid_event) SLR_HIP(c, hipEventRecord(c->mid_event, c->stream));
    { ProfScope ps(c, K_RAY_TRI);
      SLR_HIP(c, launch_ray_triangulate((const uint32_t *)offs, (uint32_t *)items, c->cal, scan_w, scan_h, W,
                                        (const float *)raysL, (const float *)raysR, (const uint32_t *)list, xyz_sum, count, c->stream)); }
    return SLR_OK;
}

}  // namespace

// ==========================================================================================================
extern "C" {

int slr_version(void) { return SLR_VERSION; }

const char *slr_status_string(int s)
{
    switch (s) {
        case SLR_OK: return "ok";
        case SLR_ERR_INVALID_ARG: return "invalid argument";
        case SLR_ERR_NO_DEVICE: return "no usable HIP device";
        case SLR_ERR_HIP: return "HIP runtime error";
        case SLR_ERR_NOT_CONFIGURED: return "calibration or rectify maps not configured";
        case SLR_ERR_UNSUPPORTED: return "unsupported size or mode";
        case SLR_ERR_OOM: return "out of device memory";
        default: return "unknown status";
    }
}

int slr_current_device(int *device_id)
{
    if (!device_id) return SLR_ERR_INVALID_ARG;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return SLR_ERR_NO_DEVICE;
    return hipGetDevice(device_id) == hipSuccess ? SLR_OK : SLR_ERR_HIP;
}

// decode tables (kernels_decode.hip, wrapped_phase_q24): per difference d the 16-bit reciprocal of |d| with the
// table slot of sgn d in the top byte, and the wrapped phase itself -- atanf over the integer quotient (SURVEY
// Q1) plus the quadrant offset of mfreconstruct.cpp:246-261, evaluated by the host libm in the reference's f32
// arithmetic so the device never evaluates a transcendental (no libm-vs-ocml ULP drift) -- as 2^24-scaled integers.
// x87 (SLR_OPT_EVAL_MODEL = 1): the offsets are added the way the reference's MSVC2010 x87 / fp:precise binary adds them -- float +
// float on the 53-bit stack, stored to the double P[count] WITHOUT an f32 rounding (and 3*PI/2 without rounding 3*PI first):
// the sums are exact, still multiples of 2^-24 below 2^27, so the same integer table holds them.
static bool build_decode_lut(int lut[kDecodeLutWords], bool x87)
{
    memset(lut, 0, sizeof(int) * kDecodeLutWords);
    bool used[kDecodeLutWords] = {}, bad = false;
    const float PI = kPI;
    const float off[3][3] = {{PI, PI, PI},                  /* d < 0 : atan + PI             (:256-257) */
                             {PI / 2, 0.0f, 3 * PI / 2},    /* d == 0: n<0 PI/2, n==0 undefined, n>0 3PI/2 (:250-255) */
                             {0.0f, 0.0f, 2 * PI}};         /* d > 0 : n>0 atan + 2PI else atan (:258-261, :246-247) */
    const double off87[3][3] = {{(double)PI, (double)PI, (double)PI},
                                {(double)PI / 2.0, 0.0, 3.0 * (double)PI / 2.0},
                                {0.0, 0.0, 2.0 * (double)PI}};
    const int slot[3] = {2, 9, 6};
    for (int d = -255; d <= 255; d++) {
        const int sd = (d > 0) - (d < 0);
        const unsigned R = d == 0 ? 0u : 65536u / (unsigned)(d < 0 ? -d : d) + 1u;
        lut[d + 255] = (int)(R | (unsigned)slot[sd + 1] << 24);
    }
    for (int sd = -1; sd <= 1; sd++)
        for (int sn = -1; sn <= 1; sn++)
            for (int qa = 0; qa <= 255; qa++) {
                if ((sd == 0 || sn == 0) && qa != 0) continue;
                const int q = sd * sn * qa;                          /* the C quotient n / d */
                const int sidx = sn < 0 && sd != 0 ? ~qa : qa;        /* (n * R) >> 16, arithmetic */
                double scaled;
                if (x87) {
                    volatile float a = sd != 0 ? atanf((float)q) : 0.0f;      /* MSVC x86: (float)atan((double)q) -- the same 511 floats */
                    scaled = ((double)a + off87[sd + 1][sn + 1]) * 16777216.0;
                } else {
                    volatile float P = atanf((float)q) + off[sd + 1][sn + 1];
                    scaled = (double)P * 16777216.0;
                }
                const int w = 512 + ((slot[sd + 1] + sn) << 8) + sidx;
                if (scaled != (double)(int)scaled || w < 512 || w >= kDecodeLutWords || used[w]) { bad = true; continue; }
                lut[w] = (int)scaled;
                used[w] = true;
            }
    return !bad;
}

int slr_create(int device_id, slr_ctx **out)
{
    if (!out) return SLR_ERR_INVALID_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return SLR_ERR_NO_DEVICE;
    if (device_id < 0 || device_id >= n) return SLR_ERR_INVALID_ARG;
    slr_ctx *c = new (std::nothrow) slr_ctx();
    if (!c) return SLR_ERR_OOM;
    c->device = device_id;
    int st = SLR_OK;
    do {
        if (hipSetDevice(device_id) != hipSuccess) { st = SLR_ERR_NO_DEVICE; break; }
        if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { st = SLR_ERR_HIP; break; }
        c->own_stream = true;
        int lut[kDecodeLutWords];
        if (!build_decode_lut(lut, false)) { st = SLR_ERR_HIP; break; }   /* cannot happen: the table layout is checked once per context */
        if (hipMalloc((void **)&c->d_lut, sizeof(lut)) != hipSuccess) { st = SLR_ERR_OOM; break; }
        if (hipMemcpy(c->d_lut, lut, sizeof(lut), hipMemcpyHostToDevice) != hipSuccess) { st = SLR_ERR_HIP; break; }
        if (hipMalloc((void **)&c->d_sched, dma_sched_bytes()) != hipSuccess) { st = SLR_ERR_OOM; break; }
        if (hipMemset(c->d_sched, 0, dma_sched_bytes()) != hipSuccess) { st = SLR_ERR_HIP; break; }
        if (hipEventCreate(&c->t0) != hipSuccess || hipEventCreate(&c->t1) != hipSuccess) { st = SLR_ERR_HIP; break; }
    } while (0);
    if (st != SLR_OK) { slr_destroy(c); return st; }
    *out = c;
    return SLR_OK;
}

int slr_destroy(slr_ctx *c)
{
    if (!c) return SLR_OK;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (auto &r : c->pending) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    for (auto e : c->free_events) (void)hipEventDestroy(e);
    if (c->t0) (void)hipEventDestroy(c->t0);
    if (c->t1) (void)hipEventDestroy(c->t1);
    if (c->pipe) { (void)hipStreamSynchronize(c->pipe); (void)hipStreamDestroy(c->pipe); }
    for (int k = 0; k < 2; k++) if (c->ev_pipe[k]) (void)hipEventDestroy(c->ev_pipe[k]);
    for (auto ps : c->push_streams) { (void)hipStreamSynchronize(ps); (void)hipStreamDestroy(ps); }
    for (auto e : c->push_done) (void)hipEventDestroy(e);
    if (c->ev_computed) (void)hipEventDestroy(c->ev_computed);
    for (int i = 0; i < 2 * S_COUNT; i++) if (c->scratch[i]) (void)hipFree(c->scratch[i]);
    for (int k = 0; k < 2; k++) { if (c->d_map_xy[k]) (void)hipFree(c->d_map_xy[k]); if (c->d_map_frac[k]) (void)hipFree(c->d_map_frac[k]); if (c->d_tile_box[k]) (void)hipFree(c->d_tile_box[k]); if (c->d_dma_tiles[k]) (void)hipFree(c->d_dma_tiles[k]); }
    if (c->d_lut) (void)hipFree(c->d_lut);
    if (c->d_sched) (void)hipFree(c->d_sched);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return SLR_OK;
}

int slr_set_stream(slr_ctx *c, void *hip_stream)
{
    if (!c) return SLR_ERR_INVALID_ARG;
    SLR_TRY(use_device(c));
    SLR_HIP(c, hipStreamSynchronize(c->stream));
    if (hip_stream) {
        if (c->own_stream) { SLR_HIP(c, hipStreamDestroy(c->stream)); c->own_stream = false; }
        c->stream = (hipStream_t)hip_stream;
    } else if (!c->own_stream) {
        SLR_HIP(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        c->own_stream = true;
    }
    return SLR_OK;
}

int slr_synchronize(slr_ctx *c)
{
    if (!c) return SLR_ERR_INVALID_ARG;
    SLR_TRY(use_device(c));
    SLR_HIP(c, hipStreamSynchronize(c->stream));
    return SLR_OK;
}

const char *slr_last_error(const slr_ctx *c) { return c ? c->err.c_str() : "null context"; }

int slr_set_calibration(slr_ctx *c, const slr_calib *cal)
{
    if (!c || !cal) return fail(c, SLR_ERR_INVALID_ARG, "null argument");
    for (int k = 0; k < 2; k++)
        if (cal->cam[k].fc[0] == 0.0f || cal->cam[k].fc[1] == 0.0f) return fail(c, SLR_ERR_INVALID_ARG, "zero focal length");
    const DevCamera before[2] = {c->cal.cam[0], c->cal.cam[1]};
    const bool had = c->has_calib;
    for (int k = 0; k < 2; k++) {
        const slr_camera &s = cal->cam[k];
        DevCamera &d = c->cal.cam[k];
        d.fx = s.fc[0]; d.fy = s.fc[1];                 // utilities.cpp:68-73: f32 widened, ifx = 1./fx in f64
        d.ifx = 1. / d.fx; d.ify = 1. / d.fy;
        d.cx = s.cc[0]; d.cy = s.cc[1];
        d.k0 = s.k[0]; d.k1 = s.k[1]; d.k2 = s.k[2]; d.k3 = s.k[3];   // k[4] ignored (utilities.cpp:66)
        d.fcx = s.fc[0]; d.fcy = s.fc[1]; d.ccx = s.cc[0]; d.ccy = s.cc[1];
        memcpy(d.R, s.R, sizeof(d.R));
        memcpy(d.t, s.t, sizeof(d.t));
    }
    memcpy(c->cal.Q, cal->Q, sizeof(c->cal.Q));
    memcpy(c->cal.T, cal->T, sizeof(c->cal.T));
    c->cal.has_T = cal->has_T ? 1 : 0;
    {
        const double *Q = cal->Q;
        c->cal.q_simple = (Q[0] == 1.0 && Q[1] == 0.0 && Q[2] == 0.0 && Q[4] == 0.0 && Q[5] == 1.0 && Q[6] == 0.0 &&
                           Q[8] == 0.0 && Q[9] == 0.0 && Q[10] == 0.0 && Q[12] == 0.0 && Q[13] == 0.0) ? 1 : 0;
    }
    c->has_calib = true;
    // the per-pixel undistortion / ray tables depend on the cameras only: a new Q or transfer matrix (every scan of a series
    // brings its own, mfreconstruct.cpp:278-282) keeps them
    if (!had || memcmp(before, c->cal.cam, sizeof before) != 0) { c->und_valid = false; c->rays_valid = false; }
    return SLR_OK;
}

static int map_storage(slr_ctx *c, int cam, int W, int H)
{
    if (cam < 0 || cam > 1) return fail(c, SLR_ERR_INVALID_ARG, "cam must be 0 or 1");
    SLR_TRY(check_dims(c, W, H, W));
    SLR_TRY(use_device(c));
    SLR_HIP(c, hipStreamSynchronize(c->stream));
    if (c->map_w != W || c->map_h != H) {
        for (int k = 0; k < 2; k++) {
            if (c->d_map_xy[k]) { SLR_HIP(c, hipFree(c->d_map_xy[k])); c->d_map_xy[k] = nullptr; }
            if (c->d_map_frac[k]) { SLR_HIP(c, hipFree(c->d_map_frac[k])); c->d_map_frac[k] = nullptr; }
            if (c->d_tile_box[k]) { SLR_HIP(c, hipFree(c->d_tile_box[k])); c->d_tile_box[k] = nullptr; }
            if (c->d_dma_tiles[k]) { SLR_HIP(c, hipFree(c->d_dma_tiles[k])); c->d_dma_tiles[k] = nullptr; c->dma_shape_built[k] = -1; }
        }
        c->map_w = W; c->map_h = H;
    }
    const size_t n = (size_t)W * H;
    if (!c->d_map_xy[cam]) SLR_HIP(c, hipMalloc(&c->d_map_xy[cam], n * 4));
    if (!c->d_map_frac[cam]) SLR_HIP(c, hipMalloc(&c->d_map_frac[cam], n * 2));
    if (!c->d_tile_box[cam]) SLR_HIP(c, hipMalloc(&c->d_tile_box[cam], tile_boxes_bytes(W, H)));
    return SLR_OK;
}

// tile tables of the LDS-DMA fused decode for the maps of `cam` (enqueued; the caller synchronises the stream)
static int build_dma_tiles(slr_ctx *c, int cam)
{
    const int W = c->map_w, H = c->map_h;
    if (c->d_dma_tiles[cam] && c->dma_shape_built[cam] != c->opt_dma_shape) {
        SLR_HIP(c, hipStreamSynchronize(c->stream));
        SLR_HIP(c, hipFree(c->d_dma_tiles[cam]));
        c->d_dma_tiles[cam] = nullptr;
    }
    c->dma_shape_built[cam] = -1;
    if (W % 16 != 0) return SLR_OK;                         // the form needs whole 16-byte chunks per row
    if (!c->d_dma_tiles[cam]) SLR_HIP(c, hipMalloc(&c->d_dma_tiles[cam], dma_tiles_bytes(W, H, c->opt_dma_shape)));
    // first with the three-row read mode for every wave whose quads straddle source rows; when few waves do (at most 35 %: mild
    // maps), again with the heavily straddling ones promoted to per-pixel reads (launch_dma_tiles; the decision needs the count)
    SLR_HIP(c, launch_dma_tiles(c->d_map_xy[cam], c->d_map_frac[cam], W, H, c->d_dma_tiles[cam], c->opt_dma_shape, false,
                                !c->debug.no_quad_sort, c->dma_stats[cam], c->stream));
    SLR_HIP(c, hipStreamSynchronize(c->stream));
    const unsigned long long waves = (unsigned long long)c->dma_stats[cam][4] + c->dma_stats[cam][5] + c->dma_stats[cam][6];
    if (c->dma_stats[cam][5] > 0 && 100ull * c->dma_stats[cam][5] <= 35ull * waves && !c->debug.no_promote) {
        SLR_HIP(c, launch_dma_tiles(c->d_map_xy[cam], c->d_map_frac[cam], W, H, c->d_dma_tiles[cam], c->opt_dma_shape, true,
                                    !c->debug.no_quad_sort, c->dma_stats[cam], c->stream));
        SLR_HIP(c, hipStreamSynchronize(c->stream));
    }
    c->dma_shape_built[cam] = c->opt_dma_shape;
    return SLR_OK;
}

int slr_set_rectify_maps(slr_ctx *c, int cam, const int16_t *map_xy, const uint16_t *map_frac, int W, int H,
                         slr_mem mem)
{
    if (!c || !map_xy || !map_frac) return fail(c, SLR_ERR_INVALID_ARG, "null argument");
    SLR_TRY(map_storage(c, cam, W, H));
    const size_t n = (size_t)W * H;
    const hipMemcpyKind kind = mem == SLR_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
    SLR_HIP(c, hipMemcpyAsync(c->d_map_xy[cam], map_xy, n * 4, kind, c->stream));
    SLR_HIP(c, hipMemcpyAsync(c->d_map_frac[cam], map_frac, n * 2, kind, c->stream));
    SLR_HIP(c, launch_tile_boxes(c->d_map_xy[cam], c->d_map_frac[cam], W, H, (int4 *)c->d_tile_box[cam], c->tile_nofit[cam], c->stream));
    SLR_TRY(build_dma_tiles(c, cam));
    SLR_HIP(c, hipStreamSynchronize(c->stream));
    return SLR_OK;
}

int slr_init_rectify_maps(slr_ctx *c, int cam, const double M[9], const double D[5], const double R[9], const double P[12],
                          int W, int H)
{
    if (!c || !M || !D || !R || !P) return fail(c, SLR_ERR_INVALID_ARG, "null argument");
    SLR_TRY(map_storage(c, cam, W, H));
    SLR_HIP(c, launch_init_rectify_map(M, D, R, P, W, H, c->d_map_xy[cam], c->d_map_frac[cam], c->stream));
    SLR_HIP(c, launch_tile_boxes(c->d_map_xy[cam], c->d_map_frac[cam], W, H, (int4 *)c->d_tile_box[cam], c->tile_nofit[cam], c->stream));
    SLR_TRY(build_dma_tiles(c, cam));
    SLR_HIP(c, hipStreamSynchronize(c->stream));
    return SLR_OK;
}

int slr_get_rectify_info(slr_ctx *c, int cam, slr_rectify_info *out)
{
    if (!c || !out) return fail(c, SLR_ERR_INVALID_ARG, "null argument");
    if (cam < 0 || cam > 1) return fail(c, SLR_ERR_INVALID_ARG, "cam must be 0 (left) or 1 (right)");
    if (!c->d_map_xy[cam]) return fail(c, SLR_ERR_NOT_CONFIGURED, "rectify maps not set (slr_set_rectify_maps)");
    memset(out, 0, sizeof *out);
    out->W = c->map_w; out->H = c->map_h;
    const bool dma = dma_form_wanted(c, cam, cam);
    // an explicit 7 that these maps do not allow (no tables, or too many tiles that no split makes fit) makes the decode calls
    // fail with SLR_ERR_UNSUPPORTED: report that (-1), not the form auto would have fallen back to
    // (SLR_OPT_EVAL_MODEL = 1 has the LDS-DMA form and the per-pixel gather only: round 1's register-staged forms 2..6 are strict-model kernels)
    out->mf_form = dma ? 7 : c->opt_rect_algo == 7 ? -1 : c->debug.eval_x87 ? 1
                 : (c->opt_rect_algo != 0 ? c->opt_rect_algo : mf_rect_algo(c, cam, cam));
    out->dma_shape = c->opt_dma_shape;
    out->dma_depth = c->debug.eval_x87 ? 2 : c->opt_dma_depth;   // (the x87 form of the MF decode exists at the default distance only)
    if (c->d_dma_tiles[cam] && c->dma_shape_built[cam] == c->opt_dma_shape) {
        const unsigned *st = c->dma_stats[cam];
        out->dma_tiles = (unsigned)dma_tile_count_of(c->map_w, c->map_h, c->opt_dma_shape);
        out->dma_nofit_tiles = st[0];
        for (int k = 0; k < 3; k++) { out->quads_by_class[k] = st[1 + k]; out->waves_by_mode[k] = st[4 + k]; }
        const unsigned cap = dma_extra_entries_capacity(c->map_w, c->map_h, c->opt_dma_shape);
        out->dma_extra_entries = st[7] < cap ? st[7] : cap;
    } else {
        out->dma_nofit_tiles = 0xFFFFFFFFu;                 // no LDS-DMA tables for these maps (W % 16 != 0)
    }
    out->lds_nofit_tiles[0] = c->tile_nofit[cam][0]; out->lds_nofit_tiles[1] = c->tile_nofit[cam][1];
    return SLR_OK;
}

int slr_get_rectify_maps(slr_ctx *c, int cam, int16_t *map_xy, uint16_t *map_frac, int W, int H, slr_mem mem)
{
    if (!c || !map_xy || !map_frac) return fail(c, SLR_ERR_INVALID_ARG, "null argument");
    SLR_TRY(use_device(c));
    SLR_TRY(need_maps(c, cam, W, H));
    const size_t n = (size_t)W * H;
    const hipMemcpyKind kind = mem == SLR_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
    SLR_HIP(c, hipMemcpyAsync(map_xy, c->d_map_xy[cam], n * 4, kind, c->stream));
    SLR_HIP(c, hipMemcpyAsync(map_frac, c->d_map_frac[cam], n * 2, kind, c->stream));
    SLR_HIP(c, hipStreamSynchronize(c->stream));
    return SLR_OK;
}

int slr_remap_u8(slr_ctx *c, int cam, const uint8_t *src, int src_pitch, uint8_t *dst, int dst_pitch, int W, int H,
                 slr_mem mem)
{
    if (!c || !src || !dst) return fail(c, SLR_ERR_INVALID_ARG, "null argument");
    SLR_TRY(check_dims(c, W, H, src_pitch));
    if (dst_pitch < W) return fail(c, SLR_ERR_INVALID_ARG, "bad dst pitch");
    SLR_TRY(use_device(c));
    SLR_TRY(need_maps(c, cam, W, H));
    Stage st(c, mem);
    const void *ds; void *dd;
    SLR_TRY(st.in(src, (size_t)src_pitch * H, &ds));
    SLR_TRY(st.out(dst, (size_t)dst_pitch * H, &dd));
    { ProfScope ps(c, K_REMAP, true);
      SLR_HIP(c, launch_remap_u8((const uint8_t *)ds, src_pitch, (uint8_t *)dd, dst_pitch, W, H, c->d_map_xy[cam],
                                 c->d_map_frac[cam], c->stream)); }
    return st.finish();
}

static int mf_decode_entry(slr_ctx *c, int cam, bool rectify, const uint8_t *const *planes, int pitch, int W, int H,
                           int black_thr, float *phase, uint8_t *valid, slr_mem mem)
{
    if (!c || !planes || !phase || !valid) return fail(c, SLR_ERR_INVALID_ARG, "null argument");
    for (int i = 0; i < SLR_MF_PLANES; i++) if (!planes[i]) return fail(c, SLR_ERR_INVALID_ARG, "null plane");
    SLR_TRY(check_dims(c, W, H, pitch));
    SLR_TRY(use_device(c));
    if (rectify) SLR_TRY(need_maps(c, cam, W, H));
    Stage st(c, mem);
    const uint8_t *dp[SLR_MF_PLANES];
    void *dph, *dv;
    SLR_TRY(st.planes(planes, SLR_MF_PLANES, pitch, H, dp));
    SLR_TRY(st.out(phase, (size_t)W * H * 4, &dph));
    SLR_TRY(st.out(valid, (size_t)W * H, &dv));
    SLR_TRY(core_mf_decode(c, cam, rectify, dp, pitch, W, H, black_thr, (float *)dph, (uint8_t *)dv));
    return st.finish();
}

int slr_mf_decode(slr_ctx *c, const uint8_t *const planes[SLR_MF_PLANES], int pitch, int W, int H, int black_thr,
                  float *phase, uint8_t *valid, slr_mem mem)
{
    return mf_decode_entry(c, 0, false, planes, pitch, W, H, black_thr, phase, valid, mem);
}

int slr_mf_rectify_decode(slr_ctx *c, int cam, const uint8_t *const planes[SLR_MF_PLANES], int pitch, int W, int H,
                          int black_thr, float *phase, uint8_t *valid, slr_mem mem)
{
    return mf_decode_entry(c, cam, true, planes, pitch, W, H, black_thr, phase, valid, mem);
}

static int gray_decode_entry(slr_ctx *c, int cam, bool rectify, const uint8_t *const *planes, int ncol, int nrow,
                             int pitch, int W, int H, int black_thr, int white_thr, int scan_w, int scan_h,
                             int32_t *cx, int32_t *cy, uint8_t *valid, slr_mem mem)
{
    if (!c || !planes || !cx || !valid) return fail(c, SLR_ERR_INVALID_ARG, "null argument");
    if (ncol < 1 || ncol > SLR_MAX_GRAY_BITS || nrow < 0 || nrow > SLR_MAX_GRAY_BITS)
        return fail(c, SLR_ERR_INVALID_ARG, "bit counts out of range");
    if (nrow > 0 && !cy) return fail(c, SLR_ERR_INVALID_ARG, "code_y required when n_row_bits > 0");
    const int n = 2 + 2 * ncol + 2 * nrow;
    for (int i = 0; i < n; i++) if (!planes[i]) return fail(c, SLR_ERR_INVALID_ARG, "null plane");
    SLR_TRY(check_dims(c, W, H, pitch));
    SLR_TRY(use_device(c));
    if (rectify) SLR_TRY(need_maps(c, cam, W, H));
    Stage st(c, mem);
    const uint8_t *dp[SLR_MAX_GRAY_PLANES];
    void *dx, *dy, *dv;
    SLR_TRY(st.planes(planes, n, pitch, H, dp));
    SLR_TRY(st.out(cx, (size_t)W * H * 4, &dx));
    SLR_TRY(st.out(cy, (size_t)W * H * 4, &dy));
    SLR_TRY(st.out(valid, (size_t)W * H, &dv));
    SLR_TRY(core_gray_decode(c, cam, rectify, dp, ncol, nrow, pitch, W, H, black_thr, white_thr, scan_w, scan_h,
                             (int32_t *)dx, (int32_t *)dy, (uint8_t *)dv));
    return st.finish();
}

int slr_mfn_decode(slr_ctx *c, const uint16_t *const *planes, int n_freq, int n_step, int pitch, int W, int H,
                   float black_thr, float *phase, uint8_t *valid, slr_mem mem)
{
    if (!c || !planes || !phase || !valid) return fail(c, SLR_ERR_INVALID_ARG, "null argument");
    if (n_freq < 2 || n_freq > SLR_MFN_MAX_FREQ || n_step < 3 || n_step > SLR_MFN_MAX_STEPS)
        return fail(c, SLR_ERR_INVALID_ARG, "n_freq must be 2..6 and n_step 3..16");
    const int np = 2 + n_freq * n_step;
    for (int i = 0; i < np; i++) if (!planes[i]) return fail(c, SLR_ERR_INVALID_ARG, "null plane");
    SLR_TRY(check_dims(c, W, H, pitch));
    SLR_TRY(use_device(c));
    const uint16_t *dp[SLR_MFN_MAX_PLANES];
    void *dph, *dv;
    if (mem == SLR_MEM_DEVICE) {
        for (int i = 0; i < np; i++) dp[i] = planes[i];
        dph = phase; dv = valid;
    } else {                                             // one contiguous staging buffer for the whole stack
        const size_t plane = (size_t)pitch * H * 2, n = (size_t)W * H;
        void *d;
        SLR_TRY(get_scratch(c, S_STAGE0, plane * np, &d));
        for (int i = 0; i < np; i++) {
            SLR_HIP(c, hipMemcpyAsync((uint8_t *)d + plane * i, planes[i], plane, hipMemcpyHostToDevice, c->stream));
            dp[i] = (const uint16_t *)((uint8_t *)d + plane * i);
        }
        SLR_TRY(get_scratch(c, S_STAGE0 + 1, n * 4, &dph));
        SLR_TRY(get_scratch(c, S_STAGE0 + 2, n, &dv));
    }
    { ProfScope ps(c, K_MFN_DECODE, true);
      SLR_HIP(c, launch_mfn_decode(dp, n_freq, n_step, pitch, W, H, black_thr, (float *)dph, (uint8_t *)dv, c->stream)); }
    if (mem != SLR_MEM_DEVICE) {
        const size_t n = (size_t)W * H;
        SLR_HIP(c, hipMemcpyAsync(phase, dph, n * 4, hipMemcpyDeviceToHost, c->stream));
        SLR_HIP(c, hipMemcpyAsync(valid, dv, n, hipMemcpyDeviceToHost, c->stream));
        SLR_HIP(c, hipStreamSynchronize(c->stream));
    }
    return SLR_OK;
}

int slr_rectify_source_rows(slr_ctx *c, int cam, int row0, int rows, int *src_row0, int *src_rows)
{
    if (!c || !src_row0 || !src_rows) return fail(c, SLR_ERR_INVALID_ARG, "null argument");
    if (cam < 0 || cam > 1) return fail(c, SLR_ERR_INVALID_ARG, "cam must be 0 (left) or 1 (right)");
    if (!c->d_map_xy[cam]) return fail(c, SLR_ERR_NOT_CONFIGURED, "rectify maps not set (slr_set_rectify_maps)");
    if (row0 < 0 || rows < 0 || row0 + rows > c->map_h) return fail(c, SLR_ERR_INVALID_ARG, "row band outside the image");
    SLR_TRY(use_device(c));
    *src_row0 = 0; *src_rows = 0;
    if (rows == 0) return SLR_OK;
    void *d;
    SLR_TRY(get_scratch(c, S_STAGE0 + 3, 2 * sizeof(int), &d));
    SLR_HIP(c, launch_map_source_rows(c->d_map_xy[cam], c->map_w, c->map_h, row0, rows, (int *)d, c->stream));
    int r[2];
    SLR_HIP(c, hipMemcpyAsync(r, d, sizeof r, hipMemcpyDeviceToHost, c->stream));
    SLR_HIP(c, hipStreamSynchronize(c->stream));
    if (r[0] > r[1]) return SLR_OK;                      // the band samples nothing
    const int lo = r[0] < 0 ? 0 : r[0], hi = r[1] > c->map_h - 1 ? c->map_h - 1 : r[1];
    *src_row0 = lo; *src_rows = hi - lo + 1;
    return SLR_OK;
}

int slr_mfn_rectify_decode(slr_ctx *c, int cam, const uint16_t *const *planes, int n_freq, int n_step, int pitch, int W, int H,
                           float black_thr, int row0, int rows, int src_row0, int src_rows, float *phase, uint8_t *valid, slr_mem mem)
{
    if (!c || !planes || !phase || !valid) return fail(c, SLR_ERR_INVALID_ARG, "null argument");
    if (cam < 0 || cam > 1) return fail(c, SLR_ERR_INVALID_ARG, "cam must be 0 (left) or 1 (right)");
    if (n_freq < 2 || n_freq > SLR_MFN_MAX_FREQ || n_step < 3 || n_step > SLR_MFN_MAX_STEPS)
        return fail(c, SLR_ERR_INVALID_ARG, "n_freq must be 2..6 and n_step 3..16");
    const int np = 2 + n_freq * n_step;
    for (int i = 0; i < np; i++) if (!planes[i]) return fail(c, SLR_ERR_INVALID_ARG, "null plane");
    SLR_TRY(check_dims(c, W, H, pitch));
    if (row0 < 0 || rows < 0 || row0 + rows > H) return fail(c, SLR_ERR_INVALID_ARG, "destination row band outside the image");
    if (src_row0 < 0 || src_rows < 0 || src_row0 + src_rows > H) return fail(c, SLR_ERR_INVALID_ARG, "source row window outside the image");
    if ((size_t)pitch * (size_t)(src_rows > 0 ? src_rows : 1) >= (1ull << 31)) return fail(c, SLR_ERR_UNSUPPORTED, "plane window beyond 2^31 elements");
    SLR_TRY(use_device(c));
    SLR_TRY(need_maps(c, cam, W, H));
    if (rows == 0) return SLR_OK;
    const uint16_t *dp[SLR_MFN_MAX_PLANES];
    void *dph, *dv;
    const size_t n = (size_t)W * rows;
    if (mem == SLR_MEM_DEVICE) {
        for (int i = 0; i < np; i++) dp[i] = planes[i];
        dph = phase; dv = valid;
    } else {                                             // one contiguous staging buffer for the window of every plane
        const size_t plane = (size_t)pitch * (src_rows > 0 ? src_rows : 1) * 2;
        void *d;
        SLR_TRY(get_scratch(c, S_STAGE0, plane * np, &d));
        for (int i = 0; i < np; i++) {
            if (src_rows > 0) SLR_HIP(c, hipMemcpyAsync((uint8_t *)d + plane * i, planes[i], plane, hipMemcpyHostToDevice, c->stream));
            dp[i] = (const uint16_t *)((uint8_t *)d + plane * i);
        }
        SLR_TRY(get_scratch(c, S_STAGE0 + 1, n * 4, &dph));
        SLR_TRY(get_scratch(c, S_STAGE0 + 2, n, &dv));
    }
    { ProfScope ps(c, K_MFN_RECT_DECODE);                   // (not "exact": the LDS-DMA form is two launches, the decode and its fix-up pass)
      SLR_HIP(c, launch_mfn_rect_decode(dp, n_freq, n_step, pitch, W, H, black_thr, c->d_map_xy[cam], c->d_map_frac[cam], row0, rows,
                                        src_row0, src_rows, (float *)dph, (uint8_t *)dv, c->stream)); }
    if (mem != SLR_MEM_DEVICE) {
        SLR_HIP(c, hipMemcpyAsync(phase, dph, n * 4, hipMemcpyDeviceToHost, c->stream));
        SLR_HIP(c, hipMemcpyAsync(valid, dv, n, hipMemcpyDeviceToHost, c->stream));
        SLR_HIP(c, hipStreamSynchronize(c->stream));
    }
    return SLR_OK;
}

int slr_gray_decode(slr_ctx *c, const uint8_t *const *planes, int ncol, int nrow, int pitch, int W, int H,
                    int black_thr, int white_thr, int scan_w, int scan_h, int32_t *cx, int32_t *cy, uint8_t *valid,
                    slr_mem mem)
{
    return gray_decode_entry(c, 0, false, planes, ncol, nrow, pitch, W, H, black_thr, white_thr, scan_w, scan_h, cx,
                             cy, valid, mem);
}

int slr_gray_rectify_decode(slr_ctx *c, int cam, const uint8_t *const *planes, int ncol, int nrow, int pitch, int W,
                            int H, int black_thr, int white_thr, int scan_w, int scan_h, int32_t *cx, int32_t *cy,
                            uint8_t *valid, slr_mem mem)
{
    return gray_decode_entry(c, cam, true, planes, ncol, nrow, pitch, W, H, black_thr, white_thr, scan_w, scan_h, cx,
                             cy, valid, mem);
}

int slr_mf_triangulate(slr_ctx *c, const float *phaseL, const uint8_t *validL, const float *phaseR,
                       const uint8_t *validR, int W, int H, float *xyz, uint8_t *has, int32_t *match_k, slr_mem mem)
{
    if (!c || !phaseL || !validL || !phaseR || !validR || !xyz || !has) return fail(c, SLR_ERR_INVALID_ARG, "null argument");
    SLR_TRY(check_dims(c, W, H, W));
    if (W > 32768) return fail(c, SLR_ERR_UNSUPPORTED, "W > 32768 does not fit the LDS row");
    SLR_TRY(use_device(c));
    SLR_TRY(need_calib(c));
    Stage st(c, mem);
    const size_t n = (size_t)W * H;
    const void *pl, *vl, *pr, *vr; void *dx, *dh, *dk;
    SLR_TRY(st.in(phaseL, n * 4, &pl)); SLR_TRY(st.in(validL, n, &vl));
    SLR_TRY(st.in(phaseR, n * 4, &pr)); SLR_TRY(st.in(validR, n, &vr));
    SLR_TRY(st.out(xyz, n * 12, &dx)); SLR_TRY(st.out(has, n, &dh)); SLR_TRY(st.out(match_k, n * 4, &dk));
    SLR_TRY(core_mf_match(c, (const float *)pl, (const uint8_t *)vl, (const float *)pr, (const uint8_t *)vr, W, H,
                          (float *)dx, (uint8_t *)dh, (int32_t *)dk));
    return st.finish();
}

int slr_mf_triangulate_rows(slr_ctx *c, const float *phaseL, const uint8_t *validL, const float *phaseR,
                            const uint8_t *validR, int W, int H, int row0, int rows, float *xyz, uint8_t *has,
                            int32_t *match_k, slr_mem mem)
{
    if (!c || !phaseL || !validL || !phaseR || !validR || !xyz || !has) return fail(c, SLR_ERR_INVALID_ARG, "null argument");
    SLR_TRY(check_dims(c, W, H, W));
    if (row0 < 0 || rows < 0 || row0 + rows > H) return fail(c, SLR_ERR_INVALID_ARG, "row band outside the image");
    if (W > 32768) return fail(c, SLR_ERR_UNSUPPORTED, "W > 32768 does not fit the LDS row");
    if (rows == 0) return SLR_OK;
    SLR_TRY(use_device(c));
    SLR_TRY(need_calib(c));
    Stage st(c, mem);
    const size_t n = (size_t)W * rows;
    const void *pl, *vl, *pr, *vr; void *dx, *dh, *dk;
    SLR_TRY(st.in(phaseL, n * 4, &pl)); SLR_TRY(st.in(validL, n, &vl));
    SLR_TRY(st.in(phaseR, n * 4, &pr)); SLR_TRY(st.in(validR, n, &vr));
    SLR_TRY(st.out(xyz, n * 12, &dx)); SLR_TRY(st.out(has, n, &dh)); SLR_TRY(st.out(match_k, n * 4, &dk));
    SLR_TRY(core_mf_match(c, (const float *)pl, (const uint8_t *)vl, (const float *)pr, (const uint8_t *)vr, W, H,
                          (float *)dx, (uint8_t *)dh, (int32_t *)dk, row0, rows));
    return st.finish();
}

int slr_ge_triangulate(slr_ctx *c, const int32_t *codeL, const uint8_t *validL, const int32_t *codeR,
                       const uint8_t *validR, int W, int H, const uint8_t *whiteL, const uint8_t *whiteR, float *xyz,
                       uint8_t *has, uint8_t *color, int32_t *match_k, slr_mem mem)
{
    if (!c || !codeL || !validL || !codeR || !validR || !xyz || !has) return fail(c, SLR_ERR_INVALID_ARG, "null argument");
    if (color && (!whiteL || !whiteR)) return fail(c, SLR_ERR_INVALID_ARG, "color output needs both white planes");
    SLR_TRY(check_dims(c, W, H, W));
    if (W > 16384) return fail(c, SLR_ERR_UNSUPPORTED, "W > 16384 does not fit the LDS row (Gray-code match)");
    SLR_TRY(use_device(c));
    SLR_TRY(need_calib(c));
    Stage st(c, mem);
    const size_t n = (size_t)W * H;
    const void *cl, *vl, *cr, *vr, *wl, *wr; void *dx, *dh, *dc, *dk;
    SLR_TRY(st.in(codeL, n * 4, &cl)); SLR_TRY(st.in(validL, n, &vl));
    SLR_TRY(st.in(codeR, n * 4, &cr)); SLR_TRY(st.in(validR, n, &vr));
    SLR_TRY(st.in(whiteL, n, &wl)); SLR_TRY(st.in(whiteR, n, &wr));
    SLR_TRY(st.out(xyz, n * 12, &dx)); SLR_TRY(st.out(has, n, &dh));
    SLR_TRY(st.out(color, n, &dc)); SLR_TRY(st.out(match_k, n * 4, &dk));
    { ProfScope ps(c, K_GE_MATCH, true);
      SLR_HIP(c, launch_ge_match((const int32_t *)cl, (const uint8_t *)vl, (const int32_t *)cr, (const uint8_t *)vr, W, H,
                                 c->cal, (const uint8_t *)wl, (const uint8_t *)wr, (float *)dx, (uint8_t *)dh,
                                 (uint8_t *)dc, (int32_t *)dk, 0, c->stream)); }
    return st.finish();
}

int slr_ray_triangulate(slr_ctx *c, const int32_t *cxL, const int32_t *cyL, const uint8_t *vL, const int32_t *cxR,
                        const int32_t *cyR, const uint8_t *vR, int W, int H, int scan_w, int scan_h, float *xyz_sum,
                        uint8_t *count, slr_mem mem)
{
    if (!c || !cxL || !cyL || !vL || !cxR || !cyR || !vR || !xyz_sum || !count) return fail(c, SLR_ERR_INVALID_ARG, "null argument");
    SLR_TRY(check_dims(c, W, H, W));
    if (scan_w <= 0 || scan_h <= 0) return fail(c, SLR_ERR_INVALID_ARG, "bad scan size");
    SLR_TRY(use_device(c));
    SLR_TRY(need_calib(c));
    Stage st(c, mem);
    const size_t n = (size_t)W * H, nb = (size_t)scan_w * scan_h;
    const void *a, *b, *d, *e, *f, *g; void *dx, *dc;
    SLR_TRY(st.in(cxL, n * 4, &a)); SLR_TRY(st.in(cyL, n * 4, &b)); SLR_TRY(st.in(vL, n, &d));
    SLR_TRY(st.in(cxR, n * 4, &e)); SLR_TRY(st.in(cyR, n * 4, &f)); SLR_TRY(st.in(vR, n, &g));
    SLR_TRY(st.out(xyz_sum, nb * 12, &dx)); SLR_TRY(st.out(count, nb, &dc));
    SLR_TRY(core_ray(c, (const int32_t *)a, (const int32_t *)b, (const uint8_t *)d, (const int32_t *)e,
                     (const int32_t *)f, (const uint8_t *)g, W, H, scan_w, scan_h, (float *)dx, (uint8_t *)dc));
    return st.finish();
}

int slr_pointcloud_from_grid(slr_ctx *c, const float *xyz, const uint8_t *has, const uint8_t *color, int W, int H,
                             int scan_w, int scan_h, float *pc_sum, uint8_t *pc_count, uint8_t *pc_color, slr_mem mem)
{
    if (!c || !xyz || !has || !pc_sum || !pc_count) return fail(c, SLR_ERR_INVALID_ARG, "null argument");
    SLR_TRY(check_dims(c, W, H, W));
    if (scan_w <= 0 || scan_h <= 0) return fail(c, SLR_ERR_INVALID_ARG, "bad scan size");
    SLR_TRY(use_device(c));
    Stage st(c, mem);
    const size_t n = (size_t)W * H, nb = (size_t)scan_w * scan_h;
    const void *a, *b, *d; void *ps_, *pc_, *pk_;
    SLR_TRY(st.in(xyz, n * 12, &a)); SLR_TRY(st.in(has, n, &b)); SLR_TRY(st.in(color, n, &d));
    SLR_TRY(st.out(pc_sum, nb * 12, &ps_)); SLR_TRY(st.out(pc_count, nb, &pc_)); SLR_TRY(st.out(pc_color, nb, &pk_));
    { ProfScope ps(c, K_PC_FROM_GRID);
      SLR_HIP(c, launch_pc_from_grid((const float *)a, (const uint8_t *)b, (const uint8_t *)d, W, H, scan_w, scan_h,
                                     (float *)ps_, (uint8_t *)pc_, (uint8_t *)pk_, c->stream)); }
    return st.finish();
}

int slr_pointcloud_get(slr_ctx *c, const float *pc_sum, const uint8_t *pc_count, size_t n, float *out, slr_mem mem)
{
    if (!c || !pc_sum || !pc_count || !out) return fail(c, SLR_ERR_INVALID_ARG, "null argument");
    if (n == 0) return SLR_OK;
    SLR_TRY(use_device(c));
    Stage st(c, mem);
    const void *a, *b; void *o;
    SLR_TRY(st.in(pc_sum, n * 12, &a)); SLR_TRY(st.in(pc_count, n, &b)); SLR_TRY(st.out(out, n * 12, &o));
    { ProfScope ps(c, K_PC_GET);
      SLR_HIP(c, launch_pc_get((const float *)a, (const uint8_t *)b, n, (float *)o, c->stream)); }
    return st.finish();
}

int slr_line_line_intersections(slr_ctx *c, size_t n, const float *p1, const float *p2, const float *v1, const float *v2, float *out,
                                uint8_t *ok, slr_mem mem)
{
    if (!c || !p1 || !p2 || !v1 || !v2 || !out || !ok) return fail(c, SLR_ERR_INVALID_ARG, "null argument");
    if (n == 0) return SLR_OK;
    SLR_TRY(use_device(c));
    Stage st(c, mem);
    const void *a, *b, *d, *e; void *o, *k;
    SLR_TRY(st.in(p1, 12, &a)); SLR_TRY(st.in(p2, 12, &b)); SLR_TRY(st.in(v1, n * 12, &d)); SLR_TRY(st.in(v2, n * 12, &e));
    SLR_TRY(st.out(out, n * 12, &o)); SLR_TRY(st.out(ok, n, &k));
    SLR_HIP(c, launch_line_line(n, (const float *)a, (const float *)b, (const float *)d, (const float *)e, (float *)o, (uint8_t *)k, c->stream));
    return st.finish();
}

// ---- whole-path drop-ins ----------------------------------------------------------------------------------
// both cameras' (rectifying) decode of one stereo frame: ONE launch when an LDS-tiled fused form applies, else one per camera.
// vL / vR null: the valid flag travels INSIDE the phase (invalid pixels carry a NaN, which K4 never matches).
static int decode_pair_dev(slr_ctx *c, const uint8_t *const *pL, const uint8_t *const *pR, int pitch, int W, int H, int black_thr,
                           int rectify, float *phL, uint8_t *vL, float *phR, uint8_t *vR)
{
    bool paired = false;
    if (rectify) {                                       // both cameras in one launch when the LDS-tiled form applies
        MfPlanes mp[2];
        for (int i = 0; i < SLR_MF_PLANES; i++) { mp[0].p[i] = pL[i]; mp[1].p[i] = pR[i]; }
        float *const ph[2] = {phL, phR};
        uint8_t *const vd[2] = {vL, vR};
        const int16_t *const mxy[2] = {c->d_map_xy[0], c->d_map_xy[1]};
        const uint16_t *const mfr[2] = {c->d_map_frac[0], c->d_map_frac[1]};
        const void *const box[2] = {c->d_tile_box[0], c->d_tile_box[1]};
        ProfScope ps(c, K_MF_RECT_DECODE_PAIR, true);
        if (dma_form_wanted(c, 0, 1)) {
            const void *const tl[2] = {c->d_dma_tiles[0], c->d_dma_tiles[1]};
            const DmaFixup fix = dma_fixup_of(c, 0, 1);
            SLR_HIP(c, launch_mf_rect_decode_dma(mp, 2, pitch, W, H, black_thr, c->d_lut, ph, vd, tl, c->opt_dma_shape, c->opt_dma_depth,
                                                 c->d_sched, &fix, &paired, c->stream));
        }
        if (!paired && c->opt_rect_algo != 7)
        SLR_HIP(c, launch_mf_rect_decode_pair(mp, pitch, W, H, black_thr, c->d_lut, ph, vd, mxy, mfr, box, mf_rect_algo(c, 0, 1),
                                              &paired, c->stream));
    }
    if (!paired) {
        SLR_TRY(core_mf_decode(c, 0, rectify != 0, pL, pitch, W, H, black_thr, phL, vL));
        SLR_TRY(core_mf_decode(c, 1, rectify != 0, pR, pitch, W, H, black_thr, phR, vR));
    }
    return SLR_OK;
}

// the fused decode of both cameras of g frames of a batch in ONE launch (LDS-DMA form only): job 2 f + cam; the workgroups of the
// frames walk a camera's tiles at the same pace, so its map digest and boxes (4 of the decode's 23 bytes per camera pixel) cross HBM
// once per group.  *grouped = false: nothing launched (the caller decodes frame by frame).  Phases of frame f at phL / phR + f * n.
static int decode_group_dev(slr_ctx *c, int g, const uint8_t *stack, size_t plane, int pitch, int W, int H, int black_thr,
                            float *phL, float *phR, bool *grouped, const MfPlanes *sets = nullptr /* [2 g]: job 2 f + cam (else: from `stack`) */)
{
    *grouped = false;
    if (g < 2 || 2 * g > kDmaMaxJobs || !dma_form_wanted(c, 0, 1)) return SLR_OK;
    const size_t n = (size_t)W * H;
    MfPlanes mp[kDmaMaxJobs];
    float *ph[kDmaMaxJobs];
    uint8_t *vd[kDmaMaxJobs];
    const void *tl[kDmaMaxJobs];
    int slot[kDmaMaxJobs];
    for (int f = 0; f < g; f++)
        for (int cam = 0; cam < 2; cam++) {
            if (sets) mp[2 * f + cam] = sets[2 * f + cam];
            else {
                const uint8_t *base = stack + (size_t)f * 2 * SLR_MF_PLANES * plane + (size_t)cam * SLR_MF_PLANES * plane;
                for (int i = 0; i < SLR_MF_PLANES; i++) mp[2 * f + cam].p[i] = base + plane * i;
            }
            ph[2 * f + cam] = (cam ? phR : phL) + (size_t)f * n;
            vd[2 * f + cam] = nullptr;
            tl[2 * f + cam] = c->d_dma_tiles[cam];
            slot[2 * f + cam] = cam;
        }
    const DmaFixup fix = dma_fixup_of(c, 0, 1);
    ProfScope ps(c, K_MF_RECT_DECODE_PAIR, true);
    ps.r.units = g;
    SLR_HIP(c, launch_mf_rect_decode_dma(mp, 2 * g, pitch, W, H, black_thr, c->d_lut, ph, vd, tl, c->opt_dma_shape, c->opt_dma_depth,
                                         c->d_sched, &fix, grouped, c->stream, slot));
    return SLR_OK;
}

static int reconstruct_mf_dev(slr_ctx *c, const uint8_t *const *pL, const uint8_t *const *pR, int pitch, int W, int H,
                              int black_thr, int rectify, float *xyz, uint8_t *has)
{
    const size_t n = (size_t)W * H;
    // the phase images between the decode and K4 are internal: the valid flag travels INSIDE the phase (invalid pixels
    // carry a NaN, which K4 never matches -- the decode kernels do that when their valid pointer is null), so neither
    // side touches separate valid bytes (the decode's 64-byte partial-line stores, 2 B per stereo pixel each way)
    void *phL, *phR;
    SLR_TRY(get_scratch(c, S_PHASE_L, n * 4, &phL));
    SLR_TRY(get_scratch(c, S_PHASE_R, n * 4, &phR));
    SLR_TRY(decode_pair_dev(c, pL, pR, pitch, W, H, black_thr, rectify, (float *)phL, nullptr, (float *)phR, nullptr));
    return core_mf_match(c, (const float *)phL, nullptr, (const float *)phR, nullptr, W, H, xyz, has, nullptr);
}

// Frames per group of the batch entries.  The phase scratch of a group costs 8 bytes per pixel and frame (0.8 GB at 4096x3000 with
// the default 8), so a group is only as large as it pays: SLR_OPT_MF_BATCH_GROUP frames when ONE match launch can take them (the
// lean K4: stereoRectify's Q, rows of 2049..4096 pixels, W % 4 == 0 -- the pointer-independent part of mf_match_batches_frames),
// else no more than one fused-decode launch serves (SLR_OPT_MF_BATCH_DECODE_GROUP, when the LDS-DMA form applies), else 1.
static int mf_batch_group_of(const slr_ctx *c, int W, int rectify, int H = 0, const float *xyz = nullptr, const uint8_t *has = nullptr)
{
    // (the pointer-dependent half of mf_match_batches_frames too, when the caller's outputs are known: a group whose match launch
    //  would be refused must not cost a group's phase scratch -- 0.8 GB at the default group, kept for the context's lifetime)
    const bool outs_ok = !xyz || ((uintptr_t)xyz % 16 == 0 && (uintptr_t)has % 4 == 0 && ((size_t)W * (size_t)H) % 4 == 0);
    const bool match_groups = (c->opt_mf_match_algo == 0 || c->opt_mf_match_algo == 4 || c->opt_mf_match_algo == 7 || c->opt_mf_match_algo == 8 || c->opt_mf_match_algo == 9) && c->cal.q_simple && W > 2048 && W <= 4096 && W % 4 == 0 && outs_ok;
    if (match_groups) return c->opt_mf_batch_group;
    const int dg = rectify && dma_form_wanted(c, 0, 1) ? c->opt_mf_decode_group : 1;
    return dg < c->opt_mf_batch_group ? dg : c->opt_mf_batch_group;
}

// the checks slr_reconstruct_mf_batch applies to its arguments (also run for EVERY context of a multi-GPU call before any of
// them is given work)
static int mf_batch_check(slr_ctx *c, int pitch, int W, int H, int rectify)
{
    SLR_TRY(check_dims(c, W, H, pitch));
    if (W > 32768) return fail(c, SLR_ERR_UNSUPPORTED, "W > 32768 does not fit the LDS row");
    SLR_TRY(use_device(c));
    SLR_TRY(need_calib(c));
    if (rectify) { SLR_TRY(need_maps(c, 0, W, H)); SLR_TRY(need_maps(c, 1, W, H)); }
    return SLR_OK;
}

// fused K1+K2 of BOTH cameras of a stereo frame (one launch when the LDS-tiled forms apply): loadCamImgs' 2 x 14 doStereoRectify
// + decodePatterns of both cameras (mfreconstruct.cpp:119-134, 190-269)
int slr_mf_rectify_decode_pair(slr_ctx *c, const uint8_t *const planesL[SLR_MF_PLANES], const uint8_t *const planesR[SLR_MF_PLANES],
                               int pitch, int W, int H, int black_thr, float *phaseL, uint8_t *validL, float *phaseR, uint8_t *validR,
                               slr_mem mem)
{
    if (!c || !planesL || !planesR || !phaseL || !phaseR) return fail(c, SLR_ERR_INVALID_ARG, "null argument");
    if ((validL == nullptr) != (validR == nullptr)) return fail(c, SLR_ERR_INVALID_ARG, "validL and validR must both be given or both be NULL");
    for (int i = 0; i < SLR_MF_PLANES; i++) if (!planesL[i] || !planesR[i]) return fail(c, SLR_ERR_INVALID_ARG, "null plane");
    SLR_TRY(check_dims(c, W, H, pitch));
    SLR_TRY(use_device(c));
    SLR_TRY(need_maps(c, 0, W, H)); SLR_TRY(need_maps(c, 1, W, H));
    Stage st(c, mem);
    const uint8_t *dl[SLR_MF_PLANES], *dr[SLR_MF_PLANES];
    void *pl, *vl, *pr, *vr;
    const size_t n = (size_t)W * H;
    SLR_TRY(st.planes(planesL, SLR_MF_PLANES, pitch, H, dl));
    SLR_TRY(st.planes(planesR, SLR_MF_PLANES, pitch, H, dr));
    SLR_TRY(st.out(phaseL, n * 4, &pl)); SLR_TRY(st.out(validL, n, &vl));
    SLR_TRY(st.out(phaseR, n * 4, &pr)); SLR_TRY(st.out(validR, n, &vr));
    SLR_TRY(decode_pair_dev(c, dl, dr, pitch, W, H, black_thr, 1, (float *)pl, (uint8_t *)vl, (float *)pr, (uint8_t *)vr));
    return st.finish();
}

int slr_reconstruct_mf(slr_ctx *c, const uint8_t *const planesL[SLR_MF_PLANES], const uint8_t *const planesR[SLR_MF_PLANES],
                       int pitch, int W, int H, int black_thr, int rectify, float *xyz, uint8_t *has, slr_mem mem)
{
    if (!c || !planesL || !planesR || !xyz || !has) return fail(c, SLR_ERR_INVALID_ARG, "null argument");
    for (int i = 0; i < SLR_MF_PLANES; i++) if (!planesL[i] || !planesR[i]) return fail(c, SLR_ERR_INVALID_ARG, "null plane");
    SLR_TRY(check_dims(c, W, H, pitch));
    if (W > 32768) return fail(c, SLR_ERR_UNSUPPORTED, "W > 32768 does not fit the LDS row");
    SLR_TRY(use_device(c));
    SLR_TRY(need_calib(c));
    if (rectify) { SLR_TRY(need_maps(c, 0, W, H)); SLR_TRY(need_maps(c, 1, W, H)); }
    Stage st(c, mem);
    const uint8_t *dl[SLR_MF_PLANES], *dr[SLR_MF_PLANES];
    void *dx, *dh;
    SLR_TRY(st.planes(planesL, SLR_MF_PLANES, pitch, H, dl));
    SLR_TRY(st.planes(planesR, SLR_MF_PLANES, pitch, H, dr));
    SLR_TRY(st.out(xyz, (size_t)W * H * 12, &dx));
    SLR_TRY(st.out(has, (size_t)W * H, &dh));
    SLR_TRY(reconstruct_mf_dev(c, dl, dr, pitch, W, H, black_thr, rectify, (float *)dx, (uint8_t *)dh));
    return st.finish();
}

// MFReconstruct::runReconstruction as the application consumes it: the PointCloudImage of the scan (transposed + cropped grid,
// Q11).  The full-resolution XYZ grid stays on the device: a host caller uploads 2 x 14 planes and downloads scan_w x scan_h x 13
// bytes instead of W x H x 13.
int slr_reconstruct_mf_cloud(slr_ctx *c, const uint8_t *const planesL[SLR_MF_PLANES], const uint8_t *const planesR[SLR_MF_PLANES],
                             int pitch, int W, int H, int black_thr, int rectify, int scan_w, int scan_h, float *pc_sum,
                             uint8_t *pc_count, slr_mem mem)
{
    if (!c || !planesL || !planesR || !pc_sum || !pc_count) return fail(c, SLR_ERR_INVALID_ARG, "null argument");
    for (int i = 0; i < SLR_MF_PLANES; i++) if (!planesL[i] || !planesR[i]) return fail(c, SLR_ERR_INVALID_ARG, "null plane");
    SLR_TRY(check_dims(c, W, H, pitch));
    if (W > 32768) return fail(c, SLR_ERR_UNSUPPORTED, "W > 32768 does not fit the LDS row");
    if (scan_w <= 0 || scan_h <= 0) return fail(c, SLR_ERR_INVALID_ARG, "bad scan size");
    SLR_TRY(use_device(c));
    SLR_TRY(need_calib(c));
    if (rectify) { SLR_TRY(need_maps(c, 0, W, H)); SLR_TRY(need_maps(c, 1, W, H)); }
    Stage st(c, mem);
    const size_t n = (size_t)W * H, nb = (size_t)scan_w * scan_h;
    const uint8_t *dl[SLR_MF_PLANES], *dr[SLR_MF_PLANES];
    void *dx, *dh, *ps_, *pc_;
    SLR_TRY(st.planes(planesL, SLR_MF_PLANES, pitch, H, dl));
    SLR_TRY(st.planes(planesR, SLR_MF_PLANES, pitch, H, dr));
    SLR_TRY(get_scratch(c, S_XYZ, n * 12, &dx));
    SLR_TRY(get_scratch(c, S_HAS, n, &dh));
    SLR_TRY(st.out(pc_sum, nb * 12, &ps_)); SLR_TRY(st.out(pc_count, nb, &pc_));
    SLR_TRY(reconstruct_mf_dev(c, dl, dr, pitch, W, H, black_thr, rectify, (float *)dx, (uint8_t *)dh));
    { ProfScope ps(c, K_PC_FROM_GRID);
      SLR_HIP(c, launch_pc_from_grid((const float *)dx, (const uint8_t *)dh, nullptr, W, H, scan_w, scan_h, (float *)ps_,
                                     (uint8_t *)pc_, nullptr, c->stream)); }
    return st.finish();
}

int slr_reconstruct_mf_batch(slr_ctx *c, int n_frames, const uint8_t *stack, int pitch, int W, int H, int black_thr,
                             int rectify, float *xyz, uint8_t *has)
{
    if (!c || !stack || !xyz || !has || n_frames < 0) return fail(c, SLR_ERR_INVALID_ARG, "bad argument");
    SLR_TRY(mf_batch_check(c, pitch, W, H, rectify));
    const size_t plane = (size_t)pitch * H, n = (size_t)W * H;
    // K4 reads 12 bytes of per-CALIBRATION undistortion tables per pixel (a third of its traffic): the frames of a batch are decoded
    // into a phase scratch of `group` frames and matched by ONE launch whose workgroups take a row of all those frames one after the
    // other on the same XCD -- the tables come from HBM once per group (SLR_OPT_MF_BATCH_GROUP, default 8; 1 = frame by frame)
    const int gmax = mf_batch_group_of(c, W, rectify, H, xyz, has);
    for (int f0 = 0; f0 < n_frames;) {
        const int g = n_frames - f0 < gmax ? n_frames - f0 : gmax;
        void *phL = nullptr, *phR = nullptr;
        if (g > 1) {
            SLR_TRY(get_scratch(c, S_PHASE_L, (size_t)g * n * 4, &phL));
            SLR_TRY(get_scratch(c, S_PHASE_R, (size_t)g * n * 4, &phR));
        }
        bool batched = false;
        for (int j = 0; j < g && g > 1;) {                  // the fused decodes in sub-groups of SLR_OPT_MF_BATCH_DECODE_GROUP frames per launch
            const int dg = g - j < c->opt_mf_decode_group ? g - j : c->opt_mf_decode_group;
            bool decoded = false;
            if (dg > 1 && rectify)
                SLR_TRY(decode_group_dev(c, dg, stack + (size_t)(f0 + j) * 2 * SLR_MF_PLANES * plane, plane, pitch, W, H, black_thr,
                                         (float *)phL + (size_t)j * n, (float *)phR + (size_t)j * n, &decoded));
            if (decoded) { j += dg; continue; }
            const uint8_t *pl[SLR_MF_PLANES], *pr[SLR_MF_PLANES];
            const uint8_t *base = stack + (size_t)(f0 + j) * 2 * SLR_MF_PLANES * plane;
            for (int i = 0; i < SLR_MF_PLANES; i++) { pl[i] = base + plane * i; pr[i] = base + plane * (SLR_MF_PLANES + i); }
            SLR_TRY(decode_pair_dev(c, pl, pr, pitch, W, H, black_thr, rectify, (float *)phL + (size_t)j * n, nullptr,
                                    (float *)phR + (size_t)j * n, nullptr));
            j++;
        }
        if (g > 1)
            SLR_TRY(core_mf_match(c, (const float *)phL, nullptr, (const float *)phR, nullptr, W, H, xyz + (size_t)f0 * n * 3,
                                  has + (size_t)f0 * n, nullptr, 0, -1, g, n, &batched));
        for (int j = 0; j < g && !batched; j++) {           // one launch cannot take the group (rows, alignment, match form): frame by frame
            if (g > 1) {
                SLR_TRY(core_mf_match(c, (const float *)phL + (size_t)j * n, nullptr, (const float *)phR + (size_t)j * n, nullptr, W, H,
                                      xyz + (size_t)(f0 + j) * n * 3, has + (size_t)(f0 + j) * n, nullptr));
                continue;
            }
            const uint8_t *pl[SLR_MF_PLANES], *pr[SLR_MF_PLANES];
            const uint8_t *base = stack + (size_t)(f0 + j) * 2 * SLR_MF_PLANES * plane;
            for (int i = 0; i < SLR_MF_PLANES; i++) { pl[i] = base + plane * i; pr[i] = base + plane * (SLR_MF_PLANES + i); }
            SLR_TRY(reconstruct_mf_dev(c, pl, pr, pitch, W, H, black_thr, rectify, xyz + (size_t)(f0 + j) * n * 3, has + (size_t)(f0 + j) * n));
        }
        f0 += g;
    }
    return SLR_OK;
}

// ---- BASELINE config 3: Gray code + multi-frequency phase from one hybrid stack ----------------------------------------------------
// planes of a camera: white, black, 2 * ncol Gray planes (pattern, inverse per column bit, MSB first), 12 fringe planes (3 x 4).
// One pass (one launch for both cameras) when the LDS-DMA form applies; else the two fused decodes of each camera.
static int hybrid_pair_dev(slr_ctx *c, const uint8_t *const *pL, const uint8_t *const *pR, int ncol, int pitch, int W, int H,
                           int black_thr, int white_thr, int scan_w, int32_t *cxL, float *phL, int32_t *cxR, float *phR,
                           bool gray_part_only = false /* two-launch mode: the caller decodes the fringes (a group of frames in one launch) */)
{
    const int np = 2 + 2 * ncol + 12;
    bool done = false;
    if (!c->opt_hybrid_one_pass) {
        // two launches over the one stack (the default, measured faster than the one-pass kernel): the Gray planes of both
        // cameras, then white + black + the fringes behind the Gray planes of both cameras (the LDS-DMA forms when they apply --
        // the multi-frequency one takes the gap between black and the first fringe as a constant plane skip)
        bool paired = false;
        if (dma_form_wanted(c, 0, 1)) {
            GrayPlanes gp[2];
            for (int i = 0; i < SLR_MAX_GRAY_PLANES; i++) { gp[0].p[i] = i < 2 + 2 * ncol ? pL[i] : nullptr; gp[1].p[i] = i < 2 + 2 * ncol ? pR[i] : nullptr; }
            int32_t *const xs[2] = {cxL, cxR}, *const ys[2] = {nullptr, nullptr};
            uint8_t *const vd[2] = {nullptr, nullptr};
            const void *const tl[2] = {c->d_dma_tiles[0], c->d_dma_tiles[1]};
            const DmaFixup fix = dma_fixup_of(c, 0, 1);
            ProfScope ps(c, K_GRAY_RECT_DECODE_PAIR, true);
            SLR_HIP(c, launch_gray_rect_decode_dma(gp, 2, ncol, 0, pitch, W, H, black_thr, white_thr, scan_w, 0, xs, ys, vd, tl,
                                                   c->opt_dma_shape, c->opt_dma_depth, c->d_sched, &fix, &paired, c->stream));
        }
        if (!paired) {
            SLR_TRY(core_gray_decode(c, 0, true, pL, ncol, 0, pitch, W, H, black_thr, white_thr, scan_w, 0, cxL, nullptr, nullptr));
            SLR_TRY(core_gray_decode(c, 1, true, pR, ncol, 0, pitch, W, H, black_thr, white_thr, scan_w, 0, cxR, nullptr, nullptr));
        }
        if (gray_part_only) return SLR_OK;
        const uint8_t *ml[SLR_MF_PLANES], *mr[SLR_MF_PLANES];
        ml[0] = pL[0]; ml[1] = pL[1]; mr[0] = pR[0]; mr[1] = pR[1];
        for (int k = 0; k < 12; k++) { ml[2 + k] = pL[2 + 2 * ncol + k]; mr[2 + k] = pR[2 + 2 * ncol + k]; }
        return decode_pair_dev(c, ml, mr, pitch, W, H, black_thr, 1, phL, nullptr, phR, nullptr);
    }
    if (dma_form_wanted(c, 0, 1)) {
        GrayPlanes gp[2];
        for (int i = 0; i < SLR_MAX_GRAY_PLANES; i++) { gp[0].p[i] = i < np ? pL[i] : nullptr; gp[1].p[i] = i < np ? pR[i] : nullptr; }
        int32_t *const xs[2] = {cxL, cxR};
        float *const ps[2] = {phL, phR};
        const void *const tl[2] = {c->d_dma_tiles[0], c->d_dma_tiles[1]};
        const DmaFixup fix = dma_fixup_of(c, 0, 1);
        ProfScope ps_(c, K_HYBRID_RECT_DECODE_PAIR, true);
        SLR_HIP(c, launch_hybrid_rect_decode_dma(gp, 2, ncol, pitch, W, H, black_thr, white_thr, scan_w, c->d_lut, xs, ps, tl,
                                                 c->opt_dma_shape, c->d_sched, &fix, &done, c->stream));
    }
    if (done) return SLR_OK;
    if (c->opt_rect_algo == 7)
        return fail(c, SLR_ERR_UNSUPPORTED, "SLR_OPT_RECT_DECODE_ALGO = 7 (LDS-DMA form) does not apply to this hybrid stack / these maps");
    for (int cam = 0; cam < 2; cam++) {                      // two passes per camera: the Gray planes, then white + black + fringes
        const uint8_t *const *p = cam == 0 ? pL : pR;
        const uint8_t *mf[SLR_MF_PLANES];
        mf[0] = p[0]; mf[1] = p[1];
        for (int k = 0; k < 12; k++) mf[2 + k] = p[2 + 2 * ncol + k];
        SLR_TRY(core_gray_decode(c, cam, true, p, ncol, 0, pitch, W, H, black_thr, white_thr, scan_w, 0, cam == 0 ? cxL : cxR, nullptr, nullptr));
        SLR_TRY(core_mf_decode(c, cam, true, mf, pitch, W, H, black_thr, cam == 0 ? phL : phR, nullptr));
    }
    return SLR_OK;
}

static int hybrid_check(slr_ctx *c, int ncol, int pitch, int W, int H, int scan_w)
{
    if (ncol < 1 || ncol > SLR_MAX_GRAY_BITS || scan_w <= 0) return fail(c, SLR_ERR_INVALID_ARG, "bit count / scan size out of range");
    SLR_TRY(check_dims(c, W, H, pitch));
    SLR_TRY(use_device(c));
    SLR_TRY(need_maps(c, 0, W, H)); SLR_TRY(need_maps(c, 1, W, H));
    return SLR_OK;
}

int slr_hybrid_rectify_decode_pair(slr_ctx *c, const uint8_t *const *planesL, const uint8_t *const *planesR, int ncol, int pitch, int W,
                                   int H, int black_thr, int white_thr, int scan_w, int32_t *code_xL, float *phaseL, int32_t *code_xR,
                                   float *phaseR, slr_mem mem)
{
    if (!c || !planesL || !planesR || !code_xL || !phaseL || !code_xR || !phaseR) return fail(c, SLR_ERR_INVALID_ARG, "null argument");
    SLR_TRY(hybrid_check(c, ncol, pitch, W, H, scan_w));
    const int np = 2 + 2 * ncol + 12;
    for (int i = 0; i < np; i++) if (!planesL[i] || !planesR[i]) return fail(c, SLR_ERR_INVALID_ARG, "null plane");
    Stage st(c, mem);
    const uint8_t *dl[SLR_MAX_GRAY_PLANES], *dr[SLR_MAX_GRAY_PLANES];
    void *xl, *pl, *xr, *pr;
    const size_t n = (size_t)W * H;
    SLR_TRY(st.planes(planesL, np, pitch, H, dl));
    SLR_TRY(st.planes(planesR, np, pitch, H, dr));
    SLR_TRY(st.out(code_xL, n * 4, &xl)); SLR_TRY(st.out(phaseL, n * 4, &pl));
    SLR_TRY(st.out(code_xR, n * 4, &xr)); SLR_TRY(st.out(phaseR, n * 4, &pr));
    SLR_TRY(hybrid_pair_dev(c, dl, dr, ncol, pitch, W, H, black_thr, white_thr, scan_w, (int32_t *)xl, (float *)pl, (int32_t *)xr, (float *)pr));
    return st.finish();
}

int slr_reconstruct_hybrid_batch(slr_ctx *c, int n_frames, const uint8_t *stack, int planes_per_cam, int ncol, int pitch, int W, int H,
                                 int black_thr, int white_thr, int scan_w, float *xyz, uint8_t *has, int32_t *code_x)
{
    if (!c || !stack || !xyz || !has || n_frames < 0) return fail(c, SLR_ERR_INVALID_ARG, "bad argument");
    SLR_TRY(hybrid_check(c, ncol, pitch, W, H, scan_w));
    const int np = 2 + 2 * ncol + 12;
    if (planes_per_cam < np || planes_per_cam > SLR_MAX_GRAY_PLANES) return fail(c, SLR_ERR_INVALID_ARG, "planes_per_cam does not hold the hybrid stack");
    if (W > 32768) return fail(c, SLR_ERR_UNSUPPORTED, "W > 32768 does not fit the LDS row");
    SLR_TRY(need_calib(c));
    const size_t plane = (size_t)pitch * H, n = (size_t)W * H;
    void *phL, *phR, *cxs = nullptr;
    if (!code_x) SLR_TRY(get_scratch(c, S_CODEX_L, n * 8, &cxs));      // nobody wants the codes: one scratch pair for every frame
    // as slr_reconstruct_mf_batch: the phases of a group of frames, ONE match launch per group (the undistortion tables once per group)
    const int gmax = mf_batch_group_of(c, W, 1, H, xyz, has);
    for (int f0 = 0; f0 < n_frames;) {
        const int g = n_frames - f0 < gmax ? n_frames - f0 : gmax;
        SLR_TRY(get_scratch(c, S_PHASE_L, (size_t)g * n * 4, &phL));
        SLR_TRY(get_scratch(c, S_PHASE_R, (size_t)g * n * 4, &phR));
        // two-launch mode: the fringe decodes (white, black, 12 fringes behind the Gray planes) of up to 8 frames in ONE launch of the
        // persistent kernel, as in the multi-frequency batch; the Gray decodes stay one launch per frame (grouping them buys nothing)
        const bool group_mf = !c->opt_hybrid_one_pass && g > 1 && g <= c->opt_mf_decode_group && 2 * g <= kDmaMaxJobs && dma_form_wanted(c, 0, 1);
        MfPlanes sets[kDmaMaxJobs];
        for (int j = 0; j < g; j++) {
            const int f = f0 + j;
            const uint8_t *pl[SLR_MAX_GRAY_PLANES], *pr[SLR_MAX_GRAY_PLANES];
            const uint8_t *base = stack + (size_t)f * 2 * planes_per_cam * plane;
            for (int i = 0; i < np; i++) { pl[i] = base + plane * i; pr[i] = base + plane * (planes_per_cam + i); }
            int32_t *cx = code_x ? code_x + (size_t)f * 2 * n : (int32_t *)cxs;
            SLR_TRY(hybrid_pair_dev(c, pl, pr, ncol, pitch, W, H, black_thr, white_thr, scan_w, cx, (float *)phL + (size_t)j * n, cx + n,
                                    (float *)phR + (size_t)j * n, group_mf));
            if (group_mf && 2 * j + 1 < kDmaMaxJobs) {
                sets[2 * j].p[0] = pl[0]; sets[2 * j].p[1] = pl[1]; sets[2 * j + 1].p[0] = pr[0]; sets[2 * j + 1].p[1] = pr[1];
                for (int k = 0; k < 12; k++) { sets[2 * j].p[2 + k] = pl[2 + 2 * ncol + k]; sets[2 * j + 1].p[2 + k] = pr[2 + 2 * ncol + k]; }
            }
        }
        if (group_mf) {
            bool decoded = false;
            SLR_TRY(decode_group_dev(c, g, nullptr, plane, pitch, W, H, black_thr, (float *)phL, (float *)phR, &decoded, sets));
            for (int j = 0; j < g && !decoded; j++) {       // (the form does not take this stack after all: frame by frame)
                const uint8_t *ml[SLR_MF_PLANES], *mr[SLR_MF_PLANES];
                for (int k = 0; k < SLR_MF_PLANES; k++) { ml[k] = sets[2 * j].p[k]; mr[k] = sets[2 * j + 1].p[k]; }
                SLR_TRY(decode_pair_dev(c, ml, mr, pitch, W, H, black_thr, 1, (float *)phL + (size_t)j * n, nullptr, (float *)phR + (size_t)j * n,
                                        nullptr));
            }
        }
        bool batched = false;
        if (g > 1)
            SLR_TRY(core_mf_match(c, (const float *)phL, nullptr, (const float *)phR, nullptr, W, H, xyz + (size_t)f0 * n * 3,
                                  has + (size_t)f0 * n, nullptr, 0, -1, g, n, &batched));
        for (int j = 0; j < g && !batched; j++)
            SLR_TRY(core_mf_match(c, (const float *)phL + (size_t)j * n, nullptr, (const float *)phR + (size_t)j * n, nullptr, W, H,
                                  xyz + (size_t)(f0 + j) * n * 3, has + (size_t)(f0 + j) * n, nullptr));
        f0 += g;
    }
    return SLR_OK;
}

int slr_reconstruct_ge(slr_ctx *c, const uint8_t *const *planesL, const uint8_t *const *planesR, int ncol, int pitch,
                       int W, int H, int black_thr, int white_thr, int scan_w, int rectify, int have_color, float *xyz,
                       uint8_t *has, uint8_t *color, slr_mem mem)
{
    if (!c || !planesL || !planesR || !xyz || !has) return fail(c, SLR_ERR_INVALID_ARG, "null argument");
    if (ncol < 1 || ncol > SLR_MAX_GRAY_BITS) return fail(c, SLR_ERR_INVALID_ARG, "bit count out of range");
    if (have_color && !color) return fail(c, SLR_ERR_INVALID_ARG, "color buffer required");
    const int np = 2 + 2 * ncol;
    for (int i = 0; i < np; i++) if (!planesL[i] || !planesR[i]) return fail(c, SLR_ERR_INVALID_ARG, "null plane");
    SLR_TRY(check_dims(c, W, H, pitch));
    if (W > 16384) return fail(c, SLR_ERR_UNSUPPORTED, "W > 16384 does not fit the LDS row (Gray-code match)");
    SLR_TRY(use_device(c));
    SLR_TRY(need_calib(c));
    if (rectify) { SLR_TRY(need_maps(c, 0, W, H)); SLR_TRY(need_maps(c, 1, W, H)); }
    Stage st(c, mem);
    const size_t n = (size_t)W * H;
    const uint8_t *dl[SLR_MAX_GRAY_PLANES], *dr[SLR_MAX_GRAY_PLANES];
    // the code images between the decode and K5 are internal: an invalid pixel holds code -1, no separate valid bytes
    void *dx, *dh, *dc, *cxl, *vl = nullptr, *cxr, *vr = nullptr;
    SLR_TRY(st.planes(planesL, np, pitch, H, dl));
    SLR_TRY(st.planes(planesR, np, pitch, H, dr));
    SLR_TRY(st.out(xyz, n * 12, &dx)); SLR_TRY(st.out(has, n, &dh));
    SLR_TRY(st.out(have_color ? color : nullptr, n, &dc));
    SLR_TRY(get_scratch(c, S_CODEX_L, n * 4, &cxl));
    SLR_TRY(get_scratch(c, S_CODEX_R, n * 4, &cxr));
    bool paired = false;
    if (rectify && dma_form_wanted(c, 0, 1)) {               // both cameras' fused decodes in one launch (LDS-DMA form)
        GrayPlanes gp[2];
        for (int i = 0; i < SLR_MAX_GRAY_PLANES; i++) { gp[0].p[i] = i < np ? dl[i] : nullptr; gp[1].p[i] = i < np ? dr[i] : nullptr; }
        int32_t *const xs[2] = {(int32_t *)cxl, (int32_t *)cxr}, *const ys[2] = {nullptr, nullptr};
        uint8_t *const vd[2] = {(uint8_t *)vl, (uint8_t *)vr};
        const void *const tl[2] = {c->d_dma_tiles[0], c->d_dma_tiles[1]};
        ProfScope ps(c, K_GRAY_RECT_DECODE_PAIR, true);
        const DmaFixup fix = dma_fixup_of(c, 0, 1);
        SLR_HIP(c, launch_gray_rect_decode_dma(gp, 2, ncol, 0, pitch, W, H, black_thr, white_thr, scan_w, 0, xs, ys, vd, tl,
                                               c->opt_dma_shape, c->opt_dma_depth, c->d_sched, &fix, &paired, c->stream));
    }
    if (!paired) {
        SLR_TRY(core_gray_decode(c, 0, rectify != 0, dl, ncol, 0, pitch, W, H, black_thr, white_thr, scan_w, 0,
                                 (int32_t *)cxl, nullptr, (uint8_t *)vl));
        SLR_TRY(core_gray_decode(c, 1, rectify != 0, dr, ncol, 0, pitch, W, H, black_thr, white_thr, scan_w, 0,
                                 (int32_t *)cxr, nullptr, (uint8_t *)vr));
    }
    // colour uses the RECTIFIED white images (reconstruct.cpp:193 color = camImgs[0])
    const uint8_t *wl = nullptr, *wr = nullptr;
    if (have_color) {
        if (rectify) {
            void *a, *b;
            SLR_TRY(get_scratch(c, S_COLOR, n * 2, &a));
            b = (uint8_t *)a + n;
            { ProfScope ps(c, K_REMAP, true);
              SLR_HIP(c, launch_remap_u8(dl[0], pitch, (uint8_t *)a, W, W, H, c->d_map_xy[0], c->d_map_frac[0], c->stream)); }
            { ProfScope ps(c, K_REMAP, true);
              SLR_HIP(c, launch_remap_u8(dr[0], pitch, (uint8_t *)b, W, W, H, c->d_map_xy[1], c->d_map_frac[1], c->stream)); }
            wl = (const uint8_t *)a; wr = (const uint8_t *)b;
        } else {
            if (pitch != W) return fail(c, SLR_ERR_UNSUPPORTED, "have_color without rectify needs pitch == W");
            wl = dl[0]; wr = dr[0];
        }
    }
    { ProfScope ps(c, K_GE_MATCH, true);
      SLR_HIP(c, launch_ge_match((const int32_t *)cxl, (const uint8_t *)vl, (const int32_t *)cxr, (const uint8_t *)vr, W, H,
                                 c->cal, wl, wr, (float *)dx, (uint8_t *)dh, (uint8_t *)dc, nullptr,
                                 ncol < 31 ? 1 << ncol : 0, c->stream)); }   // (the decode's codes have ncol bits)
    return st.finish();
}

int slr_reconstruct_gray(slr_ctx *c, const uint8_t *const *planesL, const uint8_t *const *planesR, int ncol, int nrow,
                         int pitch, int W, int H, int black_thr, int white_thr, int scan_w, int scan_h, float *xyz_sum,
                         uint8_t *count, slr_mem mem)
{
    if (!c || !planesL || !planesR || !xyz_sum || !count) return fail(c, SLR_ERR_INVALID_ARG, "null argument");
    if (ncol < 1 || ncol > SLR_MAX_GRAY_BITS || nrow < 1 || nrow > SLR_MAX_GRAY_BITS)
        return fail(c, SLR_ERR_INVALID_ARG, "bit counts out of range");
    if (scan_w <= 0 || scan_h <= 0) return fail(c, SLR_ERR_INVALID_ARG, "bad scan size");
    const int np = 2 + 2 * ncol + 2 * nrow;
    for (int i = 0; i < np; i++) if (!planesL[i] || !planesR[i]) return fail(c, SLR_ERR_INVALID_ARG, "null plane");
    SLR_TRY(check_dims(c, W, H, pitch));
    SLR_TRY(use_device(c));
    SLR_TRY(need_calib(c));
    Stage st(c, mem);
    const size_t n = (size_t)W * H, nb = (size_t)scan_w * scan_h;
    const uint8_t *dl[SLR_MAX_GRAY_PLANES], *dr[SLR_MAX_GRAY_PLANES];
    void *dx, *dc, *cxl, *cyl, *vl, *cxr, *cyr, *vr;
    SLR_TRY(st.planes(planesL, np, pitch, H, dl));
    SLR_TRY(st.planes(planesR, np, pitch, H, dr));
    SLR_TRY(st.out(xyz_sum, nb * 12, &dx)); SLR_TRY(st.out(count, nb, &dc));
    {   // the decode inside the bucket histogram: the codes never reach HBM
        GrayPlanes gl, gr;
        for (int i = 0; i < SLR_MAX_GRAY_PLANES; i++) { gl.p[i] = i < np ? dl[i] : nullptr; gr.p[i] = i < np ? dr[i] : nullptr; }
        if (!tl_debug.no_decode_count && ray_decode_count_applies(gl, np, nrow, pitch, W, H) && ray_decode_count_applies(gr, np, nrow, pitch, W, H)) {
            const RayPlanes rp = {{dl, dr}, ncol, nrow, pitch, black_thr, white_thr};
            SLR_TRY(core_ray(c, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, W, H, scan_w, scan_h, (float *)dx, (uint8_t *)dc, &rp));
            return st.finish();
        }
    }
    SLR_TRY(get_scratch(c, S_CODEX_L, n * 4, &cxl)); SLR_TRY(get_scratch(c, S_CODEY_L, n * 4, &cyl));
    SLR_TRY(get_scratch(c, S_VALID_L, n, &vl));
    SLR_TRY(get_scratch(c, S_CODEX_R, n * 4, &cxr)); SLR_TRY(get_scratch(c, S_CODEY_R, n * 4, &cyr));
    SLR_TRY(get_scratch(c, S_VALID_R, n, &vr));
    SLR_TRY(core_gray_decode(c, 0, false, dl, ncol, nrow, pitch, W, H, black_thr, white_thr, scan_w, scan_h,
                             (int32_t *)cxl, (int32_t *)cyl, (uint8_t *)vl));
    SLR_TRY(core_gray_decode(c, 1, false, dr, ncol, nrow, pitch, W, H, black_thr, white_thr, scan_w, scan_h,
                             (int32_t *)cxr, (int32_t *)cyr, (uint8_t *)vr));
    SLR_TRY(core_ray(c, (const int32_t *)cxl, (const int32_t *)cyl, (const uint8_t *)vl, (const int32_t *)cxr,
                     (const int32_t *)cyr, (const uint8_t *)vr, W, H, scan_w, scan_h, (float *)dx, (uint8_t *)dc));
    return st.finish();
}

// ---- ordered prefix index / compaction (mesh vertex numbering, sparse point-cloud assembly) --------------------------------
int slr_prefix_index(slr_ctx *c, const uint8_t *flags, int w, int h, int column_major, uint32_t first, uint32_t none,
                     uint32_t *index, uint32_t *total, slr_mem mem)
{
    if (!c || !flags || !index || !total || w <= 0 || h <= 0) return fail(c, SLR_ERR_INVALID_ARG, "bad argument");
    const size_t n = (size_t)w * h;
    if (n >= (1ull << 31)) return fail(c, SLR_ERR_UNSUPPORTED, "image too large");
    SLR_TRY(use_device(c));
    Stage st(c, mem);
    const void *df; void *di, *tmp;
    SLR_TRY(st.in(flags, n, &df));
    SLR_TRY(st.out(index, n * 4, &di));
    SLR_TRY(get_scratch(c, S_FLAGSCAN, flag_scan_temp_bytes(n), &tmp));
    uint32_t *tot_dev = nullptr;
    SLR_HIP(c, launch_flag_scan((const uint8_t *)df, n, w, h, column_major != 0, first, none, (uint32_t *)di, nullptr, nullptr, nullptr,
                                tmp, &tot_dev, c->stream));
    SLR_HIP(c, hipMemcpyAsync(total, tot_dev, sizeof(uint32_t), mem == SLR_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, c->stream));
    return st.finish();
}

int slr_compact_points(slr_ctx *c, const float *xyz, const uint8_t *has, size_t n, float *out_xyz, uint32_t *out_src,
                       uint32_t *count, slr_mem mem)
{
    if (!c || !xyz || !has || !out_xyz || !count) return fail(c, SLR_ERR_INVALID_ARG, "bad argument");
    if (n >= (1ull << 31)) return fail(c, SLR_ERR_UNSUPPORTED, "too many points");
    SLR_TRY(use_device(c));
    Stage st(c, mem);
    const void *dx, *dh; void *ox, *os, *tmp;
    SLR_TRY(st.in(xyz, n * 12, &dx)); SLR_TRY(st.in(has, n, &dh));
    SLR_TRY(st.out(out_xyz, n * 12, &ox)); SLR_TRY(st.out(out_src, n * 4, &os));
    SLR_TRY(get_scratch(c, S_FLAGSCAN, flag_scan_temp_bytes(n), &tmp));
    uint32_t *tot_dev = nullptr;
    SLR_HIP(c, launch_flag_scan((const uint8_t *)dh, n, (int)(n > 0 ? n : 1), 1, 0, 0u, 0u, nullptr, (const float *)dx, (float *)ox,
                                (uint32_t *)os, tmp, &tot_dev, c->stream));
    SLR_HIP(c, hipMemcpyAsync(count, tot_dev, sizeof(uint32_t), mem == SLR_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, c->stream));
    return st.finish();
}

// ---- several GPUs in one process -------------------------------------------------------------------------------------------------
// Frame f of the job runs on ctxs[f % n_ctx] (its shard index is f / n_ctx); every ctx works through its shard on its own
// stream, so the devices run concurrently; afterwards the cloud is assembled by direct peer copies on the PRODUCING streams
// (one hop over xGMI per source-destination pair; a frame's XYZ + mask go straight to their slot of the assembled [n_frames]
// arrays): on one device (slr_reconstruct_mf_multi, a gather) or on every device (slr_reconstruct_mf_allgather).
namespace {

int multi_share(int n_frames, int n_ctx, int k) { return (n_frames - k + n_ctx - 1) / n_ctx; }   // frames k, k + n_ctx, ...

// EVERYTHING that can be refused is checked here, for every context, before any of them is given work: a call fails as a whole
// or only a HIP runtime error can interrupt it (and then multi_drain waits for what was already started)
int multi_check(slr_ctx *const *ctxs, int n_ctx, int n_frames, const uint8_t *const *stacks, int pitch, int W, int H, int rectify)
{
    slr_ctx *c0 = ctxs[0];
    for (int k = 0; k < n_ctx; k++) {
        if (!ctxs[k]) return fail(c0, SLR_ERR_INVALID_ARG, "null context");
        if (multi_share(n_frames, n_ctx, k) > 0 && !stacks[k]) return fail(c0, SLR_ERR_INVALID_ARG, "null shard");
        const int st = mf_batch_check(ctxs[k], pitch, W, H, rectify);
        if (st != SLR_OK) { if (ctxs[k] != c0) fail(c0, st, "a context refused the job", slr_last_error(ctxs[k])); return st; }
    }
    return SLR_OK;
}

// after a failure in the middle of a multi-GPU call: nothing may still be running (and writing the caller's buffers) when
// the error is returned
void multi_drain(slr_ctx *const *ctxs, int n_ctx)
{
    for (int k = 0; k < n_ctx; k++) {
        if (hipSetDevice(ctxs[k]->device) == hipSuccess) (void)hipStreamSynchronize(ctxs[k]->stream);
        (void)hipGetLastError();
    }
}

#define SLR_MULTI_HIP(c, expr)                                                                                     \
    do {                                                                                                           \
        hipError_t e__ = (expr);                                                                                   \
        if (e__ != hipSuccess) {                                                                                   \
            const int st__ = fail((c), e__ == hipErrorOutOfMemory ? SLR_ERR_OOM : SLR_ERR_HIP, #expr, hipGetErrorString(e__)); \
            if ((c) != ctxs[0]) fail(ctxs[0], st__, "a context failed", slr_last_error(c));                        \
            multi_drain(ctxs, n_ctx);                                                                              \
            return st__;                                                                                           \
        }                                                                                                          \
    } while (0)

// is dst's memory directly addressable from src's device (same device, or peer access enabled now / before)?
int peer_direct(slr_ctx *src, slr_ctx *dst, bool *direct)
{
    *direct = true;
    if (src->device == dst->device) return SLR_OK;
    int can = 0;
    SLR_HIP(src, hipDeviceCanAccessPeer(&can, src->device, dst->device));
    if (!can) { *direct = false; return SLR_OK; }
    SLR_HIP(src, hipSetDevice(src->device));
    const hipError_t e = hipDeviceEnablePeerAccess(dst->device, 0);
    (void)hipGetLastError();
    if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) *direct = false;
    return SLR_OK;
}

}  // namespace

int slr_reconstruct_mf_multi(slr_ctx *const *ctxs, int n_ctx, int n_frames, const uint8_t *const *stacks, int pitch, int W, int H,
                             int black_thr, int rectify, float *const *xyz, uint8_t *const *has, int gather_ctx,
                             float *xyz_all, uint8_t *has_all)
{
    if (!ctxs || n_ctx < 1 || !ctxs[0]) return SLR_ERR_INVALID_ARG;
    slr_ctx *c0 = ctxs[0];
    if (!stacks || !xyz || !has || n_frames < 0) return fail(c0, SLR_ERR_INVALID_ARG, "bad argument");
    if (gather_ctx >= n_ctx || (gather_ctx >= 0 && (!xyz_all || !has_all))) return fail(c0, SLR_ERR_INVALID_ARG, "gather target");
    SLR_TRY(multi_check(ctxs, n_ctx, n_frames, stacks, pitch, W, H, rectify));
    for (int k = 0; k < n_ctx; k++)
        if (multi_share(n_frames, n_ctx, k) > 0 && (!xyz[k] || !has[k])) return fail(c0, SLR_ERR_INVALID_ARG, "null shard");
    if (gather_ctx >= 0)                                        // peer access (idempotent); without it the runtime stages the copies
        for (int k = 0; k < n_ctx; k++) { bool d; SLR_TRY(peer_direct(ctxs[k], ctxs[gather_ctx], &d)); }
    const size_t n = (size_t)W * H;
    // 1. every device starts on its shard (asynchronous on its own stream)
    for (int k = 0; k < n_ctx; k++) {
        const int share = multi_share(n_frames, n_ctx, k);
        if (share <= 0) continue;
        const int st = slr_reconstruct_mf_batch(ctxs[k], share, stacks[k], pitch, W, H, black_thr, rectify, xyz[k], has[k]);
        if (st != SLR_OK) {                                     // (only a HIP runtime error can get here: see multi_check)
            if (ctxs[k] != c0) fail(c0, st, "shard failed", slr_last_error(ctxs[k]));
            multi_drain(ctxs, n_ctx);
            return st;
        }
    }
    // 2. assembly: each source stream pushes its frames to the target device as soon as they are done
    if (gather_ctx >= 0) {
        slr_ctx *g = ctxs[gather_ctx];
        for (int k = 0; k < n_ctx; k++) {
            const int share = multi_share(n_frames, n_ctx, k);
            slr_ctx *c = ctxs[k];
            SLR_MULTI_HIP(c, hipSetDevice(c->device));
            for (int j = 0; j < share; j++) {
                const size_t f = (size_t)j * n_ctx + k;
                SLR_MULTI_HIP(c, hipMemcpyPeerAsync(xyz_all + f * n * 3, g->device, xyz[k] + (size_t)j * n * 3, c->device, n * 12, c->stream));
                SLR_MULTI_HIP(c, hipMemcpyPeerAsync(has_all + f * n, g->device, has[k] + (size_t)j * n, c->device, n, c->stream));
            }
        }
    }
    // 3. the call returns when every device is done (the assembled cloud is complete)
    for (int k = 0; k < n_ctx; k++) {
        SLR_MULTI_HIP(ctxs[k], hipSetDevice(ctxs[k]->device));
        SLR_MULTI_HIP(ctxs[k], hipStreamSynchronize(ctxs[k]->stream));
    }
    return SLR_OK;
}

// ---- north_star's exchange step from one process: EVERY device ends with the assembled cloud -------------------------------------
// Which context owns which frames of the job: SLR_ASSIGN_CYCLIC frame f -> ctxs[f % n] (shard slot f / n), SLR_ASSIGN_BLOCKED
// frame f -> ctxs[f / S] with S = ceil(n_frames / n): a context's shard is then ONE contiguous piece of the assembled arrays, the
// batch entry fills it in place in groups of frames (one fused-decode and one match launch per group) and a push is one copy per
// destination and group instead of one per frame.
namespace {

struct Share { int first, count, step; };                   // frames first, first + step, ... (count of them)
Share share_of(int n_frames, int n_ctx, int k, int assignment)
{
    if (assignment == SLR_ASSIGN_BLOCKED) {
        const int S = (n_frames + n_ctx - 1) / n_ctx;
        const int a = k * S < n_frames ? k * S : n_frames, b = (k + 1) * S < n_frames ? (k + 1) * S : n_frames;
        return {a, b - a, 1};
    }
    return {k, multi_share(n_frames, n_ctx, k), n_ctx};
}

// the push streams (one per destination) and events of context c on its device; idempotent
int push_streams_of(slr_ctx *c, int n_dst)
{
    SLR_HIP(c, hipSetDevice(c->device));
    if (!c->ev_computed) SLR_HIP(c, hipEventCreateWithFlags(&c->ev_computed, hipEventDisableTiming));
    while ((int)c->push_streams.size() < n_dst) {
        hipStream_t s = nullptr;
        SLR_HIP(c, hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        c->push_streams.push_back(s);
    }
    while ((int)c->push_done.size() < n_dst) {
        hipEvent_t e = nullptr;
        SLR_HIP(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        c->push_done.push_back(e);
    }
    return SLR_OK;
}

int peer_matrix(slr_ctx *const *ctxs, int n_ctx, int require_peer, int *peer_direct_out)
{
    bool all_direct = true;
    for (int k = 0; k < n_ctx; k++)
        for (int j = 0; j < n_ctx; j++) {
            bool d = true;
            if (j != k) SLR_TRY(peer_direct(ctxs[k], ctxs[j], &d));
            all_direct = all_direct && d;
        }
    if (peer_direct_out) *peer_direct_out = all_direct ? 1 : 0;
    if (!all_direct && require_peer) return fail(ctxs[0], SLR_ERR_UNSUPPORTED, "peer access between two of the devices is not available");
    return SLR_OK;
}

// frames [f0, f0 + g) (consecutive, or g == 1) of source k -> the same slots of every other context's arrays, behind whatever
// c->stream holds now; destination d's copies go to c's push stream d (n - 1 one-hop copies side by side)
int push_frames(slr_ctx *const *ctxs, int n_ctx, int k, size_t f0, int g, size_t n, float *const *xyz_all, uint8_t *const *has_all)
{
    slr_ctx *c = ctxs[k];
    SLR_MULTI_HIP(c, hipSetDevice(c->device));
    SLR_MULTI_HIP(c, hipEventRecord(c->ev_computed, c->stream));
    for (int d = 0; d < n_ctx; d++) {
        if (d == k || (xyz_all[d] == xyz_all[k] && ctxs[d]->device == c->device)) continue;
        hipStream_t ps = c->push_streams[(size_t)d];
        SLR_MULTI_HIP(c, hipStreamWaitEvent(ps, c->ev_computed, 0));
        SLR_MULTI_HIP(c, hipMemcpyPeerAsync(xyz_all[d] + f0 * n * 3, ctxs[d]->device, xyz_all[k] + f0 * n * 3, c->device, (size_t)g * n * 12, ps));
        SLR_MULTI_HIP(c, hipMemcpyPeerAsync(has_all[d] + f0 * n, ctxs[d]->device, has_all[k] + f0 * n, c->device, (size_t)g * n, ps));
    }
    return SLR_OK;
}

// every context's stream waits for its own pushes, then the call waits for every stream: the assembled clouds are complete
int finish_exchange(slr_ctx *const *ctxs, int n_ctx)
{
    for (int k = 0; k < n_ctx; k++) {
        slr_ctx *c = ctxs[k];
        SLR_MULTI_HIP(c, hipSetDevice(c->device));
        for (int d = 0; d < n_ctx && d < (int)c->push_streams.size(); d++) {
            if (d == k) continue;
            SLR_MULTI_HIP(c, hipEventRecord(c->push_done[(size_t)d], c->push_streams[(size_t)d]));
            SLR_MULTI_HIP(c, hipStreamWaitEvent(c->stream, c->push_done[(size_t)d], 0));
        }
    }
    for (int k = 0; k < n_ctx; k++) {
        SLR_MULTI_HIP(ctxs[k], hipSetDevice(ctxs[k]->device));
        SLR_MULTI_HIP(ctxs[k], hipStreamSynchronize(ctxs[k]->stream));
    }
    return SLR_OK;
}

int drain_pushes_then(slr_ctx *const *ctxs, int n_ctx, int st)     // an error in the middle: nothing may still be copying
{
    for (int k = 0; k < n_ctx; k++) {
        if (hipSetDevice(ctxs[k]->device) != hipSuccess) continue;
        for (auto ps : ctxs[k]->push_streams) (void)hipStreamSynchronize(ps);
    }
    multi_drain(ctxs, n_ctx);
    return st;
}

}  // namespace

// The exchange alone: every context already holds ITS frames in their slots of its own assembled arrays (e.g. written there by
// slr_reconstruct_mf_multi with xyz[k] pointing into xyz_all[k], or by any other producer on ctxs[k]'s stream) and pushes them
// to every other device.
int slr_allgather_clouds(slr_ctx *const *ctxs, int n_ctx, int n_frames, int W, int H, float *const *xyz_all, uint8_t *const *has_all,
                         int assignment, int require_peer, int *peer_direct_out)
{
    if (!ctxs || n_ctx < 1 || !ctxs[0]) return SLR_ERR_INVALID_ARG;
    slr_ctx *c0 = ctxs[0];
    if (!xyz_all || !has_all || n_frames < 0) return fail(c0, SLR_ERR_INVALID_ARG, "bad argument");
    if (assignment != SLR_ASSIGN_CYCLIC && assignment != SLR_ASSIGN_BLOCKED) return fail(c0, SLR_ERR_INVALID_ARG, "assignment must be SLR_ASSIGN_CYCLIC or SLR_ASSIGN_BLOCKED");
    for (int k = 0; k < n_ctx; k++) {
        if (!ctxs[k] || !xyz_all[k] || !has_all[k]) return fail(c0, SLR_ERR_INVALID_ARG, "null context or array");
        SLR_TRY(check_dims(ctxs[k], W, H, W));
    }
    SLR_TRY(peer_matrix(ctxs, n_ctx, require_peer, peer_direct_out));
    for (int k = 0; k < n_ctx; k++) {
        const int st = push_streams_of(ctxs[k], n_ctx);
        if (st != SLR_OK) { if (ctxs[k] != c0) fail(c0, st, "a context failed", slr_last_error(ctxs[k])); return st; }
    }
    const size_t n = (size_t)W * H;
    for (int k = 0; k < n_ctx; k++) {
        const Share sh = share_of(n_frames, n_ctx, k, assignment);
        if (sh.count <= 0) continue;
        if (sh.step == 1) { const int st = push_frames(ctxs, n_ctx, k, (size_t)sh.first, sh.count, n, xyz_all, has_all); if (st != SLR_OK) return drain_pushes_then(ctxs, n_ctx, st); }
        else
            for (int j = 0; j < sh.count; j++) {
                const int st = push_frames(ctxs, n_ctx, k, (size_t)sh.first + (size_t)j * sh.step, 1, n, xyz_all, has_all);
                if (st != SLR_OK) return drain_pushes_then(ctxs, n_ctx, st);
            }
    }
    const int st = finish_exchange(ctxs, n_ctx);
    return st == SLR_OK ? SLR_OK : drain_pushes_then(ctxs, n_ctx, st);
}

// Compute + exchange.  ctxs[k] computes its frames straight into their slots of its own xyz_all[k] / has_all[k] ([n_frames][H][W][3]
// / [n_frames][H][W] on its device) and pushes each finished frame (cyclic) or group of frames (blocked) into the same slots on
// every other device -- n_ctx - 1 concurrent one-hop copies over the point-to-point xGMI mesh on per-destination streams, no
// staging buffer and no ring, overlapping the next group's kernels.  *peer_direct (may be NULL) = 1 when every destination was
// directly addressable from every source (same device or peer access), 0 when the runtime had to stage at least one pair through
// the host; require_peer != 0 turns that case into SLR_ERR_UNSUPPORTED before any work is enqueued.
int slr_reconstruct_mf_allgather_ex(slr_ctx *const *ctxs, int n_ctx, int n_frames, const uint8_t *const *stacks, int pitch, int W, int H,
                                    int black_thr, int rectify, float *const *xyz_all, uint8_t *const *has_all, int assignment,
                                    int require_peer, int *peer_direct_out)
{
    if (!ctxs || n_ctx < 1 || !ctxs[0]) return SLR_ERR_INVALID_ARG;
    slr_ctx *c0 = ctxs[0];
    if (!stacks || !xyz_all || !has_all || n_frames < 0) return fail(c0, SLR_ERR_INVALID_ARG, "bad argument");
    if (assignment != SLR_ASSIGN_CYCLIC && assignment != SLR_ASSIGN_BLOCKED) return fail(c0, SLR_ERR_INVALID_ARG, "assignment must be SLR_ASSIGN_CYCLIC or SLR_ASSIGN_BLOCKED");
    for (int k = 0; k < n_ctx; k++) {                        // (multi_check's shard test, for either assignment)
        if (!ctxs[k]) return fail(c0, SLR_ERR_INVALID_ARG, "null context");
        if (share_of(n_frames, n_ctx, k, assignment).count > 0 && !stacks[k]) return fail(c0, SLR_ERR_INVALID_ARG, "null shard");
        const int st = mf_batch_check(ctxs[k], pitch, W, H, rectify);
        if (st != SLR_OK) { if (ctxs[k] != c0) fail(c0, st, "a context refused the job", slr_last_error(ctxs[k])); return st; }
    }
    for (int k = 0; k < n_ctx; k++) if (!xyz_all[k] || !has_all[k]) return fail(c0, SLR_ERR_INVALID_ARG, "null destination");
    SLR_TRY(peer_matrix(ctxs, n_ctx, require_peer, peer_direct_out));
    for (int k = 0; k < n_ctx; k++) {
        const int st = push_streams_of(ctxs[k], n_ctx);
        if (st != SLR_OK) { if (ctxs[k] != c0) fail(c0, st, "a context failed", slr_last_error(ctxs[k])); return st; }
    }
    const size_t n = (size_t)W * H, frame = (size_t)2 * SLR_MF_PLANES * pitch * H;
    // group by group, round robin over the contexts: every device gets its first group before any gets its second
    int gmax = 1, rounds = 0;
    for (int k = 0; k < n_ctx; k++) {
        const Share sh = share_of(n_frames, n_ctx, k, assignment);
        const int g = sh.step == 1 ? mf_batch_group_of(ctxs[k], W, rectify) : 1;
        gmax = g > gmax ? g : gmax;
    }
    for (int k = 0; k < n_ctx; k++) { const int cnt = share_of(n_frames, n_ctx, k, assignment).count; rounds = (cnt + gmax - 1) / gmax > rounds ? (cnt + gmax - 1) / gmax : rounds; }
    for (int r = 0; r < rounds; r++)
        for (int k = 0; k < n_ctx; k++) {
            const Share sh = share_of(n_frames, n_ctx, k, assignment);
            const int j0 = r * gmax;
            if (j0 >= sh.count) continue;
            const int g = sh.count - j0 < gmax ? sh.count - j0 : gmax;
            slr_ctx *c = ctxs[k];
            const size_t f0 = (size_t)sh.first + (size_t)j0 * sh.step;
            // frames f0 .. of the job = frames j0 .. of this shard, computed in place
            int st = slr_reconstruct_mf_batch(c, g, stacks[k] + (size_t)j0 * frame, pitch, W, H, black_thr, rectify, xyz_all[k] + f0 * n * 3, has_all[k] + f0 * n);
            if (st != SLR_OK) { if (c != c0) fail(c0, st, "shard failed", slr_last_error(c)); return drain_pushes_then(ctxs, n_ctx, st); }
            st = push_frames(ctxs, n_ctx, k, f0, g, n, xyz_all, has_all);
            if (st != SLR_OK) return drain_pushes_then(ctxs, n_ctx, st);
        }
    const int st = finish_exchange(ctxs, n_ctx);
    return st == SLR_OK ? SLR_OK : drain_pushes_then(ctxs, n_ctx, st);
}

int slr_reconstruct_mf_allgather(slr_ctx *const *ctxs, int n_ctx, int n_frames, const uint8_t *const *stacks, int pitch, int W, int H,
                                 int black_thr, int rectify, float *const *xyz_all, uint8_t *const *has_all, int require_peer,
                                 int *peer_direct_out)
{
    return slr_reconstruct_mf_allgather_ex(ctxs, n_ctx, n_frames, stacks, pitch, W, H, black_thr, rectify, xyz_all, has_all,
                                           SLR_ASSIGN_CYCLIC, require_peer, peer_direct_out);
}

// One checksum per frame of a device-resident cloud; and the proof of an exchange: every context checksums the assembled arrays
// on ITS device, the words are compared on the host.
int slr_cloud_checksums(slr_ctx *c, int n_frames, int W, int H, const float *xyz, const uint8_t *has, uint64_t *out)
{
    if (!c || !xyz || !has || !out || n_frames < 0) return fail(c, SLR_ERR_INVALID_ARG, "bad argument");
    SLR_TRY(check_dims(c, W, H, W));
    SLR_TRY(use_device(c));
    if (n_frames == 0) return SLR_OK;
    void *d;
    SLR_TRY(get_scratch(c, S_STAGE0 + 4, sizeof(unsigned long long) * (size_t)n_frames, &d));
    SLR_HIP(c, launch_cloud_checksums(xyz, has, n_frames, (size_t)W * H, (unsigned long long *)d, c->stream));
    SLR_HIP(c, hipMemcpyAsync(out, d, sizeof(uint64_t) * (size_t)n_frames, hipMemcpyDeviceToHost, c->stream));
    SLR_HIP(c, hipStreamSynchronize(c->stream));
    return SLR_OK;
}

int slr_verify_assembled(slr_ctx *const *ctxs, int n_ctx, int n_frames, int W, int H, float *const *xyz_all, uint8_t *const *has_all,
                         int *mismatches)
{
    if (!ctxs || n_ctx < 1 || !ctxs[0]) return SLR_ERR_INVALID_ARG;
    slr_ctx *c0 = ctxs[0];
    if (!xyz_all || !has_all || !mismatches || n_frames < 0) return fail(c0, SLR_ERR_INVALID_ARG, "bad argument");
    for (int k = 0; k < n_ctx; k++) if (!ctxs[k] || !xyz_all[k] || !has_all[k]) return fail(c0, SLR_ERR_INVALID_ARG, "null context or array");
    *mismatches = 0;
    std::vector<uint64_t> ref((size_t)n_frames), cur((size_t)n_frames);
    for (int k = 0; k < n_ctx; k++) {
        const int st = slr_cloud_checksums(ctxs[k], n_frames, W, H, xyz_all[k], has_all[k], k == 0 ? ref.data() : cur.data());
        if (st != SLR_OK) { if (ctxs[k] != c0) fail(c0, st, "checksum failed", slr_last_error(ctxs[k])); return st; }
        if (k > 0)
            for (int f = 0; f < n_frames; f++) if (cur[(size_t)f] != ref[(size_t)f]) ++*mismatches;
    }
    return SLR_OK;
}

// pinned host memory for callers that feed SLR_MEM_HOST buffers (PNG decoders, camera SDK ring buffers): the H2D / D2H
// copies of the host-buffer entry points are only asynchronous (SLR_OPT_ASYNC_HOST) from page-locked memory
int slr_host_alloc(void **p, size_t bytes)
{
    if (!p) return SLR_ERR_INVALID_ARG;
    *p = nullptr;
    const hipError_t e = hipHostMalloc(p, bytes ? bytes : 1, hipHostMallocDefault);
    return e == hipSuccess ? SLR_OK : (e == hipErrorOutOfMemory ? SLR_ERR_OOM : SLR_ERR_HIP);
}
int slr_host_free(void *p)
{
    if (!p) return SLR_OK;
    return hipHostFree(p) == hipSuccess ? SLR_OK : SLR_ERR_HIP;
}

int slr_reconstruct_batch(slr_ctx *c, const slr_batch_desc *d, const uint8_t *stack, float *xyz, uint8_t *has, uint8_t *color)
{
    if (!c || !d || !stack || !xyz || !has || d->n_frames < 0) return fail(c, SLR_ERR_INVALID_ARG, "bad argument");
    int need = 0;
    switch (d->mode) {
        case SLR_MODE_MF:   need = SLR_MF_PLANES; break;
        case SLR_MODE_GE:   need = 2 + 2 * d->n_col_bits; break;
        case SLR_MODE_GRAY: need = 2 + 2 * d->n_col_bits + 2 * d->n_row_bits; break;
        default: return fail(c, SLR_ERR_INVALID_ARG, "mode must be SLR_MODE_GRAY, SLR_MODE_GE or SLR_MODE_MF");
    }
    if (d->mode != SLR_MODE_MF && (d->n_col_bits < 1 || d->n_col_bits > SLR_MAX_GRAY_BITS || d->scan_w <= 0))
        return fail(c, SLR_ERR_INVALID_ARG, "bit counts / scan size out of range");
    if (d->mode == SLR_MODE_GRAY && (d->n_row_bits < 1 || d->n_row_bits > SLR_MAX_GRAY_BITS || d->scan_h <= 0))
        return fail(c, SLR_ERR_INVALID_ARG, "bit counts / scan size out of range");
    if (d->planes_per_cam < need || d->planes_per_cam > SLR_MAX_GRAY_PLANES) return fail(c, SLR_ERR_INVALID_ARG, "planes_per_cam does not hold the mode's stack");
    if (d->mode == SLR_MODE_GE && d->have_color && !color) return fail(c, SLR_ERR_INVALID_ARG, "color buffer required");
    SLR_TRY(check_dims(c, d->W, d->H, d->pitch));
    if (d->mode == SLR_MODE_MF && d->W > 32768) return fail(c, SLR_ERR_UNSUPPORTED, "W > 32768 does not fit the LDS row");
    if (d->mode == SLR_MODE_GE && d->W > 16384) return fail(c, SLR_ERR_UNSUPPORTED, "W > 16384 does not fit the LDS row (Gray-code match)");
    SLR_TRY(use_device(c));
    SLR_TRY(need_calib(c));
    const bool rect = d->rectify != 0 && d->mode != SLR_MODE_GRAY;          // GRAY_ONLY never rectifies (reconstruct.cpp:230-265)
    if (rect) { SLR_TRY(need_maps(c, 0, d->W, d->H)); SLR_TRY(need_maps(c, 1, d->W, d->H)); }
    if (d->mode == SLR_MODE_GE && d->have_color && !rect && d->pitch != d->W)
        return fail(c, SLR_ERR_UNSUPPORTED, "have_color without rectify needs pitch == W");
    if (d->mode == SLR_MODE_MF && d->planes_per_cam == SLR_MF_PLANES)         // the grouped launches of the multi-frequency batch entry
        return slr_reconstruct_mf_batch(c, d->n_frames, stack, d->pitch, d->W, d->H, d->black_thr, rect ? 1 : 0, xyz, has);
    const size_t plane = (size_t)d->pitch * d->H, n = (size_t)d->W * d->H, cells = (size_t)(d->scan_w > 0 ? d->scan_w : 0) * (d->scan_h > 0 ? d->scan_h : 0);
    for (int f = 0; f < d->n_frames; f++) {
        const uint8_t *pl[SLR_MAX_GRAY_PLANES], *pr[SLR_MAX_GRAY_PLANES];
        const uint8_t *base = stack + (size_t)f * 2 * d->planes_per_cam * plane;
        for (int i = 0; i < need; i++) { pl[i] = base + plane * i; pr[i] = base + plane * (d->planes_per_cam + i); }
        if (d->mode == SLR_MODE_MF)
            SLR_TRY(reconstruct_mf_dev(c, pl, pr, d->pitch, d->W, d->H, d->black_thr, rect, xyz + (size_t)f * n * 3, has + (size_t)f * n));
        else if (d->mode == SLR_MODE_GE)
            SLR_TRY(slr_reconstruct_ge(c, pl, pr, d->n_col_bits, d->pitch, d->W, d->H, d->black_thr, d->white_thr, d->scan_w, rect,
                                       d->have_color, xyz + (size_t)f * n * 3, has + (size_t)f * n,
                                       d->have_color ? color + (size_t)f * n : nullptr, SLR_MEM_DEVICE));
        else {
            // GRAY_ONLY frames alternate between two streams, each with its own scratch set: a frame's first half (decode +
            // bucket histogram, scan, scatter: ~430 us of HBM streaming) and its second half (K6: ~350 us of VALU work on data
            // that is mostly in LDS and registers) use different units, so one frame's second half runs beside the next
            // frame's first.  The second stream starts when frame 0 reaches its K6 (that is the skew which keeps the two
            // streams in opposite halves; it is also behind the ray tables and whatever the caller queued before the batch),
            // and the context's stream waits for it at the end.
            const bool piped = d->n_frames > 1 && c->opt_batch_streams > 1;
            const int odd = piped ? f & 1 : 0;
            if (piped && !c->pipe) {
                // all three or none: a context with a second stream but a null event would be broken for every later batch
                hipStream_t ps = nullptr;
                hipEvent_t pe[2] = {nullptr, nullptr};
                hipError_t pe_err = hipStreamCreateWithFlags(&ps, hipStreamNonBlocking);
                for (int k = 0; k < 2 && pe_err == hipSuccess; k++) pe_err = hipEventCreateWithFlags(&pe[k], hipEventDisableTiming);
                if (pe_err != hipSuccess) {
                    for (int k = 0; k < 2; k++) if (pe[k]) (void)hipEventDestroy(pe[k]);
                    if (ps) (void)hipStreamDestroy(ps);
                    SLR_HIP(c, pe_err);
                }
                c->pipe = ps; c->ev_pipe[0] = pe[0]; c->ev_pipe[1] = pe[1];
            }
            if (f == 0 && piped) c->mid_event = c->ev_pipe[0];
            if (f == 1 && piped) SLR_HIP(c, hipStreamWaitEvent(c->pipe, c->ev_pipe[0], 0));
            hipStream_t own = c->stream;
            if (odd) { c->stream = c->pipe; c->scratch_set = 1; }
            const int st = slr_reconstruct_gray(c, pl, pr, d->n_col_bits, d->n_row_bits, d->pitch, d->W, d->H, d->black_thr, d->white_thr,
                                                d->scan_w, d->scan_h, xyz + (size_t)f * cells * 3, has + (size_t)f * cells, SLR_MEM_DEVICE);
            c->stream = own; c->scratch_set = 0; c->mid_event = nullptr;
            if (st != SLR_OK) { if (c->pipe) (void)hipStreamSynchronize(c->pipe); return st; }
        }
    }
    if (d->mode == SLR_MODE_GRAY && d->n_frames > 1 && c->opt_batch_streams > 1) {
        SLR_HIP(c, hipEventRecord(c->ev_pipe[1], c->pipe));
        SLR_HIP(c, hipStreamWaitEvent(c->stream, c->ev_pipe[1], 0));
    }
    return SLR_OK;
}

int slr_set_option(slr_ctx *c, int option, int value)
{
    if (!c) return SLR_ERR_INVALID_ARG;
    switch (option) {
        case SLR_OPT_MF_MATCH_ALGO:
            if (value < 0 || value > 9) return fail(c, SLR_ERR_INVALID_ARG, "SLR_OPT_MF_MATCH_ALGO must be 0..9");
#ifndef SLR_ALL_FORMS
            if (value == 2) return fail(c, SLR_ERR_UNSUPPORTED, "SLR_OPT_MF_MATCH_ALGO = 2 (sorted form) is compiled with -DSLR_ALL_FORMS only");
            if (value == 5 || value == 6) return fail(c, SLR_ERR_UNSUPPORTED, "SLR_OPT_MF_MATCH_ALGO = 5 / 6 (512 x 8 shapes) are compiled with -DSLR_ALL_FORMS only");
            if (value == 7) return fail(c, SLR_ERR_UNSUPPORTED, "SLR_OPT_MF_MATCH_ALGO = 7 (persistent grouped K4) is compiled with -DSLR_ALL_FORMS only");
            if (value == 9) return fail(c, SLR_ERR_UNSUPPORTED, "SLR_OPT_MF_MATCH_ALGO = 9 (lean K4 without the hash dedup) is compiled with -DSLR_ALL_FORMS only");
#endif
            c->opt_mf_match_algo = value;
            return SLR_OK;
        case SLR_OPT_MF_DECODE_VEC:
            if (value != 0 && value != 4 && value != 8 && value != 16)
                return fail(c, SLR_ERR_INVALID_ARG, "SLR_OPT_MF_DECODE_VEC must be 0, 4, 8 or 16");
            c->opt_mf_decode_vec = value;
            return SLR_OK;
        case SLR_OPT_RECT_DECODE_ALGO:
            if (value < 0 || value > 7) return fail(c, SLR_ERR_INVALID_ARG, "SLR_OPT_RECT_DECODE_ALGO must be 0..7");
#ifndef SLR_ALL_FORMS
            if (value >= 2 && value <= 4) return fail(c, SLR_ERR_UNSUPPORTED, "SLR_OPT_RECT_DECODE_ALGO = 2, 3, 4 are compiled with -DSLR_ALL_FORMS only");
#endif
            c->opt_rect_algo = value;
            return SLR_OK;
        case SLR_OPT_RECT_DMA_SHAPE:
            if (value < 0 || value > 6) return fail(c, SLR_ERR_INVALID_ARG, "SLR_OPT_RECT_DMA_SHAPE must be 0..6");
#ifndef SLR_ALL_FORMS
            if (value == 2 || value >= 4) return fail(c, SLR_ERR_UNSUPPORTED, "SLR_OPT_RECT_DMA_SHAPE = 2, 4, 5, 6 are compiled with -DSLR_ALL_FORMS only");
#endif
            if (value != c->opt_dma_shape) {
                c->opt_dma_shape = value;
                SLR_TRY(use_device(c));
                for (int cam = 0; cam < 2; cam++)
                    if (c->d_map_xy[cam]) SLR_TRY(build_dma_tiles(c, cam));
                SLR_HIP(c, hipStreamSynchronize(c->stream));
            }
            return SLR_OK;
        case SLR_OPT_DEBUG_RECT_RESIDENT:
            if (value < 0) return fail(c, SLR_ERR_INVALID_ARG, "SLR_OPT_DEBUG_RECT_RESIDENT must be >= 0");
            c->debug.rect_resident = value;
            return SLR_OK;
        case SLR_OPT_DEBUG_FLAGS:
            if (value < 0 || value > 511) return fail(c, SLR_ERR_INVALID_ARG, "SLR_OPT_DEBUG_FLAGS must be 0..511");
            c->debug.mfn_nt512 = (value & 128) != 0;
            c->debug.mfn_ring4 = (value & 256) != 0;
            c->debug.no_tiled_map = (value & 1) != 0;
            c->debug.no_buffer_form = (value & 2) != 0;
            c->debug.no_ge_lean = (value & 4) != 0;
            c->debug.no_decode_count = (value & 8) != 0;
            c->debug.gray_small_tiles = (value & 16) != 0;
            if (c->debug.no_quad_sort != ((value & 32) != 0) || c->debug.no_promote != ((value & 64) != 0)) {   // the map digests of the LDS-DMA forms are built either way
                c->debug.no_quad_sort = (value & 32) != 0;
                c->debug.no_promote = (value & 64) != 0;
                SLR_TRY(use_device(c));                              // (the rebuild launches kernels: on the context's device)
                for (int cam = 0; cam < 2; cam++)
                    if (c->d_map_xy[cam] && c->d_dma_tiles[cam]) SLR_TRY(build_dma_tiles(c, cam));
                SLR_HIP(c, hipStreamSynchronize(c->stream));
            }
            return SLR_OK;
#ifdef SLR_DEBUG_HOOKS
        case SLR_OPT_DEBUG_K4_STOP:
            c->debug.k4_stop = value;
            return SLR_OK;
#endif
        case SLR_OPT_RECT_DMA_DEPTH:
            if (value < 1 || value > 2) return fail(c, SLR_ERR_INVALID_ARG, "SLR_OPT_RECT_DMA_DEPTH must be 1 or 2");
            c->opt_dma_depth = value;
            return SLR_OK;
        case SLR_OPT_MF_BATCH_GROUP:
            if (value < 1 || value > 64) return fail(c, SLR_ERR_INVALID_ARG, "SLR_OPT_MF_BATCH_GROUP must be 1..64");
            c->opt_mf_batch_group = value;
            return SLR_OK;
        case SLR_OPT_MF_BATCH_DECODE_GROUP:
            if (value < 1 || value > kDmaMaxJobs / 2) return fail(c, SLR_ERR_INVALID_ARG, "SLR_OPT_MF_BATCH_DECODE_GROUP must be 1..8");
            c->opt_mf_decode_group = value;
            return SLR_OK;
        case SLR_OPT_BATCH_STREAMS:
            if (value < 1 || value > 2) return fail(c, SLR_ERR_INVALID_ARG, "SLR_OPT_BATCH_STREAMS must be 1 or 2");
            c->opt_batch_streams = value;
            return SLR_OK;
        case SLR_OPT_DEBUG_POISON_SCRATCH:
            if (value < 0 || value > 1) return fail(c, SLR_ERR_INVALID_ARG, "SLR_OPT_DEBUG_POISON_SCRATCH must be 0 or 1");
            c->debug.poison_scratch = value != 0;
            return SLR_OK;
        case SLR_OPT_EVAL_MODEL: {
            if (value < 0 || value > 1) return fail(c, SLR_ERR_INVALID_ARG, "SLR_OPT_EVAL_MODEL must be 0 (strict IEEE) or 1 (x87)");
            if ((value != 0) != c->debug.eval_x87) {
                int lut[kDecodeLutWords];
                if (!build_decode_lut(lut, value != 0)) return fail(c, SLR_ERR_HIP, "decode table layout");
                SLR_TRY(use_device(c));
                SLR_HIP(c, hipDeviceSynchronize());                  // no launch (on the context's stream or its pipeline stream) may still be reading the old tables
                SLR_HIP(c, hipMemcpy(c->d_lut, lut, sizeof(lut), hipMemcpyHostToDevice));
                c->debug.eval_x87 = value != 0;
                c->cal.eval_x87 = value;
                c->rays_valid = false;                               // GRAY_ONLY's unit-ray tables depend on the model (pixel_ray, kernels_ray.hip)
            }
            return SLR_OK;
        }
        case SLR_OPT_HYBRID_ONE_PASS:
            if (value < 0 || value > 1) return fail(c, SLR_ERR_INVALID_ARG, "SLR_OPT_HYBRID_ONE_PASS must be 0 or 1");
            c->opt_hybrid_one_pass = value;
            return SLR_OK;
        case SLR_OPT_PROFILE_STRIDE:
            if (value < 1) return fail(c, SLR_ERR_INVALID_ARG, "SLR_OPT_PROFILE_STRIDE must be >= 1");
            c->opt_profile_stride = value;
            return SLR_OK;
        case SLR_OPT_ASYNC_HOST:
            if (value < 0 || value > 1) return fail(c, SLR_ERR_INVALID_ARG, "SLR_OPT_ASYNC_HOST must be 0 or 1");
            c->opt_async_host = value;
            return SLR_OK;
        default:
            return fail(c, SLR_ERR_INVALID_ARG, "unknown option");
    }
}

// ---- measurement hooks --------------------------------------------------------------------------------------
int slr_timer_begin(slr_ctx *c)
{
    if (!c) return SLR_ERR_INVALID_ARG;
    SLR_TRY(use_device(c));
    SLR_HIP(c, hipEventRecord(c->t0, c->stream));
    return SLR_OK;
}

int slr_timer_end(slr_ctx *c, float *ms)
{
    if (!c || !ms) return SLR_ERR_INVALID_ARG;
    SLR_TRY(use_device(c));
    SLR_HIP(c, hipEventRecord(c->t1, c->stream));
    SLR_HIP(c, hipEventSynchronize(c->t1));
    SLR_HIP(c, hipEventElapsedTime(ms, c->t0, c->t1));
    return SLR_OK;
}

int slr_stream_copy(slr_ctx *c, void *dst, const void *src, size_t bytes)
{
    if (!c || !dst || !src) return fail(c, SLR_ERR_INVALID_ARG, "null argument");
    if (bytes % 16 != 0 || (uintptr_t)dst % 16 != 0 || (uintptr_t)src % 16 != 0) return fail(c, SLR_ERR_INVALID_ARG, "slr_stream_copy: 16-byte granularity");
    SLR_TRY(use_device(c));
    SLR_HIP(c, launch_stream_copy(src, dst, bytes, c->stream));
    return SLR_OK;
}

int slr_stream_mix(slr_ctx *c, void *dst, const void *src, size_t bytes_out, int reads)
{
    if (!c || !dst || !src) return fail(c, SLR_ERR_INVALID_ARG, "null argument");
    if (reads < 1 || reads > 8) return fail(c, SLR_ERR_INVALID_ARG, "slr_stream_mix: reads must be 1..8");
    if (bytes_out % 16 != 0 || (uintptr_t)dst % 16 != 0 || (uintptr_t)src % 16 != 0) return fail(c, SLR_ERR_INVALID_ARG, "slr_stream_mix: 16-byte granularity");
    SLR_TRY(use_device(c));
    SLR_HIP(c, launch_stream_mix(src, dst, bytes_out, reads, c->stream));
    return SLR_OK;
}

int slr_profile_enable(slr_ctx *c, int on)
{
    if (!c) return SLR_ERR_INVALID_ARG;
    SLR_TRY(use_device(c));
    SLR_TRY(prof_drain(c));
    c->profiling = on != 0;
    return SLR_OK;
}

int slr_profile_reset(slr_ctx *c)
{
    if (!c) return SLR_ERR_INVALID_ARG;
    SLR_TRY(use_device(c));
    SLR_TRY(prof_drain(c));
    for (int i = 0; i < K_COUNT; i++) { c->prof_ms[i] = 0; c->prof_n[i] = 0; c->prof_seen[i] = 0; }
    return SLR_OK;
}

int slr_profile_kernel_count(void) { return K_COUNT; }

const char *slr_profile_kernel_name(int id) { return (id >= 0 && id < K_COUNT) ? kKernelNames[id] : ""; }

int slr_profile_get(slr_ctx *c, int id, double *total_ms, long *launches)
{
    if (!c || id < 0 || id >= K_COUNT) return SLR_ERR_INVALID_ARG;
    SLR_TRY(use_device(c));
    SLR_TRY(prof_drain(c));
    if (total_ms) *total_ms = c->prof_ms[id];
    if (launches) *launches = c->prof_n[id];
    return SLR_OK;
}

}  // extern "C"
