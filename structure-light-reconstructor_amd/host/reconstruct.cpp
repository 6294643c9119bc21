// reconstruct.cpp -- Reconstruct / MFReconstruct of the host mirror (Duke/reconstruct.cpp:99-307, Duke/mfreconstruct.cpp:67-187:
// the public call sequence of the reference; every per-pixel loop is ONE call into libslr_hip.so) and the scan-directory reader
// in front of them (reconstruct.cpp:158-164, mfreconstruct.cpp:119-125).
//
// Loader.  PNG inflate runs at ~300 MB/s per core: ~40 ms per 4096x3000 plane, 1.2 s for the 28 images of one
// multi-frequency scan when done one file after the other as the reference does -- three orders of magnitude more than the GPU
// path.  Here the files of a scan are decoded by a pool of host threads (one file per task) STRAIGHT INTO page-locked memory
// (slr_host_alloc), plane after plane in the layout the C ABI stages from, and a series of scans is pipelined: two contexts with
// SLR_OPT_ASYNC_HOST alternate on the GPU side (upload / reconstruct / download of scan i while scan i + 1 is enqueued), and ONE
// pool of as many threads as the process may run (cpu_budget(): affinity mask and cgroup quota, not the machine's core count)
// drains the files of the next two or three scans in order into page-locked input slots.  The steady state costs
// max(28 files x inflate / threads, upload + kernels + download) per scan: with a 16-CPU quota that is the inflate, ~72 ms.
#include "duke.hpp"

#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <fstream>
#include <mutex>
#include <sstream>
#include <thread>

#include <sched.h>

namespace duke {

void warn(const std::string &title, const std::string &msg);
bool load_camera_files(VirtualCamera &cam, const std::string &camFolder, const std::string &projectCalib, bool withHomographies);

namespace {

// page-locked buffer that frees itself
struct Pinned {
    void *p = nullptr;
    size_t bytes = 0;
    Pinned() {}
    Pinned(const Pinned &) = delete;
    Pinned &operator=(const Pinned &) = delete;
    ~Pinned() { if (p) slr_host_free(p); }
    bool ensure(size_t n) { if (n <= bytes) return true; if (p) { slr_host_free(p); p = nullptr; bytes = 0; } if (slr_host_alloc(&p, n) != SLR_OK) return false; bytes = n; return true; }
    uint8_t *u8() const { return (uint8_t *)p; }
};

// CPUs this process can actually run on: the affinity mask and -- for work that goes on for longer than a scheduler period
// (`sustained`: a series) -- the cgroup CPU quota (v2 cpu.max, v1 cfs_quota_us), not the core count of the machine: 84 inflate
// threads under a 16-CPU quota are throttled and finish LATER than 16 are.  One scan's files fit inside one period's quota,
// and a burst of 2 n threads does finish them sooner (measured: 70 ms vs 114 ms with 16), so one-shot loads ignore the quota.
unsigned cpu_budget(bool sustained)
{
    unsigned nt = std::thread::hardware_concurrency();
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) { const int c = CPU_COUNT(&set); if (c > 0 && (unsigned)c < nt) nt = (unsigned)c; }
    long long quota = -1, period = 100000;
    if (sustained) {
        std::ifstream f("/sys/fs/cgroup/cpu.max");
        std::string q;
        if (f >> q >> period) { if (q != "max") quota = atoll(q.c_str()); }
        else {
            std::ifstream fq("/sys/fs/cgroup/cpu/cpu.cfs_quota_us"), fp("/sys/fs/cgroup/cpu/cpu.cfs_period_us");
            if (!(fq >> quota) || !(fp >> period)) quota = -1;
        }
    }
    if (quota > 0 && period > 0) { const unsigned c = (unsigned)((quota + period - 1) / period); if (c >= 1 && c < nt) nt = c; }
    return nt < 1 ? 1 : nt;
}

unsigned loader_threads(int n, bool sustained = false)
{
    unsigned nt = cpu_budget(sustained);
    if (const char *e = getenv("SLR_LOADER_THREADS")) nt = (unsigned)atoi(e);
    if (nt < 1) nt = 1;
    if (nt > (unsigned)n) nt = (unsigned)(n > 0 ? n : 1);
    return nt;
}

bool imread_scan_file(const std::string &stem, const std::string &suffix, int w, int h, uint8_t *out, std::string &err)
{
    std::string e1, e2;
    if (imread_gray_into(stem + suffix, w, h, out, e1)) return true;
    if (suffix != ".pgm" && imread_gray_into(stem + ".pgm", w, h, out, e2)) return true;
    err = e1.compare(0, 11, "cannot open") == 0 ? "Scan Images not found! (" + stem + suffix + ")" : e1;
    return false;
}

// n images <folder[cam]><prefix[cam]><i><suffix> per camera (".pgm" is tried when the configured suffix is missing) of w x h
// pixels into dst + (cam * n + i) * w * h: ONE pool over the 2 n files.  The first failing file in (camera, index) order is
// reported, like the reference's sequential loops would.
bool load_stacks_into(const std::string folder[2], const std::string prefix[2], const std::string &suffix, int n, int w, int h,
                      uint8_t *dst, std::string &err)
{
    if (n <= 0) return true;
    const int total = 2 * n;
    std::vector<std::string> errs((size_t)total);
    std::atomic<int> next(0);
    auto work = [&]() {
        for (int t = next.fetch_add(1); t < total; t = next.fetch_add(1)) {
            try {
                const int cam = t / n, i = t - cam * n;
                std::ostringstream p;
                p << folder[cam] << prefix[cam] << i;
                imread_scan_file(p.str(), suffix, w, h, dst + (size_t)t * w * h, errs[(size_t)t]);
            } catch (const std::exception &ex) {             // bad_alloc and the like must not leave a worker thread
                errs[(size_t)t] = std::string("image decoder: ") + ex.what();
            } catch (...) {
                errs[(size_t)t] = "image decoder: unknown exception";
            }
        }
    };
    std::vector<std::thread> pool;
    const unsigned nt = loader_threads(total);
    try {
        for (unsigned t = 1; t < nt; t++) pool.emplace_back(work);
    } catch (...) { /* fewer threads than hoped: the ones that exist (and this one) drain the queue */ }
    work();
    for (auto &t : pool) t.join();
    for (int t = 0; t < total; t++)
        if (!errs[(size_t)t].empty()) { err = errs[(size_t)t]; warn("Load Images", err); return false; }
    return true;
}

// The decoder of a series: `threads` workers drain ONE queue of files in the order the scans were announced, so the files of
// scan i + 2 start the moment a worker runs out of scan i + 1's (no idle tail per scan) and never more threads run than the
// process has CPUs for.  A ScanLoad is one scan's 2 n files into one page-locked buffer (pinned by the first worker to arrive).
struct ScanLoad {
    Pinned *buf = nullptr;
    std::string folder[2], prefix[2], suffix;
    int n = 0, w = 0, h = 0, sn = -1;
    std::mutex m;
    std::condition_variable cv;
    int left = 0;                                            // files not decoded yet
    int pinned = 0;                                          // 0 not tried, 1 ok, -1 failed
    std::vector<std::string> errs;
    double t_start = 0, t_end = 0;
    void arm(int files) { std::lock_guard<std::mutex> g(m); left = files; pinned = 0; errs.assign((size_t)files, std::string()); }
    void wait() { std::unique_lock<std::mutex> g(m); cv.wait(g, [this]() { return left == 0; }); }
    bool ok(std::string &err)                                // after wait(): the first failing file in (camera, index) order
    {
        for (auto &e : errs) if (!e.empty()) { err = e; warn("Load Images", err); return false; }
        return true;
    }
};

class FilePool {
public:
    explicit FilePool(unsigned threads)
    {
        try { for (unsigned t = 0; t < threads; t++) th_.emplace_back([this]() { run(); }); }
        catch (...) { /* fewer workers than hoped; with none, submit() decodes inline */ }
    }
    ~FilePool()
    {
        { std::lock_guard<std::mutex> g(m_); stop_ = true; q_.clear(); }   // (nobody waits for a scan once the series is over)
        cv_.notify_all();
        for (auto &t : th_) t.join();
    }
    void submit(ScanLoad *l)
    {
        const int files = 2 * l->n;
        l->arm(files);
        if (th_.empty()) { for (int t = 0; t < files; t++) one(l, t); return; }
        { std::lock_guard<std::mutex> g(m_); for (int t = 0; t < files; t++) q_.push_back(Job{l, t}); }
        cv_.notify_all();
    }
private:
    struct Job { ScanLoad *l; int t; };
    static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    static void one(ScanLoad *l, int t)
    {
        std::string err;
        try {
            const size_t plane = (size_t)l->w * l->h;
            {
                std::lock_guard<std::mutex> g(l->m);         // the first file of a scan pins the slot; the others wait here for it
                if (l->pinned == 0) { l->t_start = now(); l->pinned = l->buf->ensure(2 * (size_t)l->n * plane) ? 1 : -1; }
                if (l->pinned < 0) err = "out of page-locked memory";
            }
            if (err.empty()) {
                const int cam = t / l->n, i = t - cam * l->n;
                std::ostringstream p;
                p << l->folder[cam] << l->prefix[cam] << i;
                imread_scan_file(p.str(), l->suffix, l->w, l->h, l->buf->u8() + (size_t)t * plane, err);
            }
        } catch (const std::exception &ex) { err = std::string("image decoder: ") + ex.what(); }
        catch (...) { err = "image decoder: unknown exception"; }
        std::lock_guard<std::mutex> g(l->m);
        l->errs[(size_t)t].swap(err);
        if (--l->left == 0) { l->t_end = now(); l->cv.notify_all(); }
    }
    void run()
    {
        for (;;) {
            Job j;
            {
                std::unique_lock<std::mutex> g(m_);
                cv_.wait(g, [this]() { return stop_ || !q_.empty(); });
                if (q_.empty()) return;
                j = q_.front(); q_.pop_front();
            }
            one(j.l, j.t);
        }
    }
    std::mutex m_;
    std::condition_variable cv_;
    std::deque<Job> q_;
    bool stop_ = false;
    std::vector<std::thread> th_;
};

bool ensure_ctx(slr_ctx *&ctx, std::string &err, int device = 0)
{
    if (ctx) return true;
    const int st = slr_create(device, &ctx);
    if (st != SLR_OK) { err = std::string("no GPU context: ") + slr_status_string(st); warn("Reconstruct", err); return false; }
    return true;
}

bool load_transfer(const std::string &savePath, int scanSN, slr_calib &cal)
{
    cal.has_T = 0;
    if (scanSN <= 0) return true;
    VirtualCamera tmp;
    Matf m;                                                  // scan/transfer_mat<sn>.txt, 3x4 (mfreconstruct.cpp:278-282)
    std::ostringstream p;
    p << savePath << "/scan/transfer_mat" << scanSN << ".txt";
    if (tmp.loadMatrix(m, 3, 4, p.str()) < 0) return false;
    memcpy(cal.T, m.v.data(), sizeof(float) * 12);
    cal.has_T = 1;
    return true;
}

std::string scan_prefix(int sn, char side)
{
    std::ostringstream s;
    s << sn << "/" << side;
    return s.str();
}

}  // namespace

// ---- Reconstruct (Gray-code modes) -------------------------------------------------------------------------------------------
Reconstruct::Reconstruct(bool useEpi) : EPI(useEpi)
{
    cameras = new VirtualCamera[2];
    calibFolder = new std::string[2];
    points3DProjView = nullptr;
}
Reconstruct::~Reconstruct()
{
    delete points3DProjView;
    delete sr;
    delete[] cameras;
    delete[] calibFolder;
    if (ctx) slr_destroy(ctx);
}
void Reconstruct::setCalibPath(const std::string &folder, int cam_no) { calibFolder[cam_no] = folder; pathSet = true; }

void Reconstruct::getParameters(int scanw, int scanh, int camw, int camh, bool autocontrast, bool havecolor,
                                const std::string &savePath)
{
    scan_w = scanw; scan_h = scanh; cameraWidth = camw; cameraHeight = camh;
    autoContrast_ = autocontrast; haveColor = havecolor; savePath_ = savePath;
    // the reference stores autoContrast and never reads it on these paths (reconstruct.cpp:38-54); say so instead of ignoring it
    if (autocontrast) warn("Reconstruct", "autocontrast has no effect on the reconstruction (as in the reference)");
    if (EPI) { delete sr; sr = new stereoRect(savePath_, camw, camh); sr->getParameters(); }
    scanFolder[0] = savePath + "/scan/left/";  imgPrefix[0] = scan_prefix(scanSN, 'L');
    scanFolder[1] = savePath + "/scan/right/"; imgPrefix[1] = scan_prefix(scanSN, 'R');
}

bool Reconstruct::loadCameras()
{
    for (int i = 0; i < 2; i++) {
        if (!load_camera_files(cameras[i], calibFolder[i], savePath_ + "/calib/", true)) return false;
        cameras[i].height = 0; cameras[i].width = 0;
    }
    return true;
}

bool Reconstruct::fillCalib(slr_calib &cal)
{
    memset(&cal, 0, sizeof cal);
    cameras[0].fill(cal.cam[0]);
    cameras[1].fill(cal.cam[1]);
    if (EPI && sr && !sr->Q.empty()) memcpy(cal.Q, sr->Q.v.data(), sizeof(double) * 16);
    else { cal.Q[0] = cal.Q[5] = cal.Q[10] = cal.Q[15] = 1; }
    return load_transfer(savePath_, scanSN, cal);
}

// both cameras' stacks into one pinned buffer: [cam][numberOfImgs][H][W]
static bool load_pair(const std::string folder[2], const std::string prefix[2], const std::string &suffix, int n, int w, int h,
                      Pinned &buf, std::string &err)
{
    const size_t plane = (size_t)w * h;
    if (!buf.ensure(2 * (size_t)n * plane)) { err = "out of page-locked memory"; return false; }
    return load_stacks_into(folder, prefix, suffix, n, w, h, buf.u8(), err);
}

bool Reconstruct::runReconstruction_GE()
{
    GrayCodes grays(scan_w, scan_h, true);
    numOfColBits = grays.getNumOfColBits();
    numberOfImgs = grays.getNumOfImgs();
    if (!ensure_ctx(ctx, lastError)) return false;
    if (!sr) { lastError = "getParameters not called"; return false; }
    const int W = cameraWidth, H = cameraHeight, n = numberOfImgs;
    Pinned imgs, out;
    if (!load_pair(scanFolder, imgPrefix, imgSuffix, n, W, H, imgs, lastError)) return false;
    sr->calParameters();
    slr_calib cal;
    if (!fillCalib(cal) || slr_set_calibration(ctx, &cal) != SLR_OK || !(sr->uploadFromCalibration(ctx) || sr->upload(ctx))) {
        lastError = "calibration incomplete"; warn("Reconstruct", lastError); return false;
    }
    const size_t plane = (size_t)W * H;
    std::vector<const uint8_t *> pl[2];
    for (int c = 0; c < 2; c++) for (int i = 0; i < n; i++) pl[c].push_back(imgs.u8() + ((size_t)c * n + i) * plane);
    if (!out.ensure(plane * 14)) { lastError = "out of page-locked memory"; return false; }
    float *xyz = (float *)out.p;
    uint8_t *has = out.u8() + plane * 12, *col = has + plane;
    if (slr_reconstruct_ge(ctx, pl[0].data(), pl[1].data(), numOfColBits, W, W, H, blackThreshold, whiteThreshold, scan_w, 1,
                           haveColor ? 1 : 0, xyz, has, haveColor ? col : nullptr, SLR_MEM_HOST) != SLR_OK) {
        lastError = slr_last_error(ctx); warn("Reconstruct", lastError); return false;
    }
    delete points3DProjView;
    points3DProjView = new PointCloudImage(scan_w, scan_h, haveColor);
    std::vector<uint8_t> pcol(haveColor ? (size_t)scan_w * scan_h : 0);
    if (slr_pointcloud_from_grid(ctx, xyz, has, haveColor ? col : nullptr, W, H, scan_w, scan_h, points3DProjView->points.data(),
                                 points3DProjView->numOfPointsForPixel.data(), haveColor ? pcol.data() : nullptr,
                                 SLR_MEM_HOST) != SLR_OK) { lastError = slr_last_error(ctx); return false; }
    if (haveColor)
        for (size_t i = 0; i < pcol.size(); i++)
            points3DProjView->color[3 * i] = points3DProjView->color[3 * i + 1] = points3DProjView->color[3 * i + 2] = pcol[i];
    return true;
}

bool Reconstruct::runReconstruction()
{
    GrayCodes grays(scan_w, scan_h, false);
    numOfColBits = grays.getNumOfColBits();
    numOfRowBits = grays.getNumOfRowBits();
    numberOfImgs = grays.getNumOfImgs();
    if (!ensure_ctx(ctx, lastError)) return false;
    const int W = cameraWidth, H = cameraHeight, n = numberOfImgs;
    Pinned imgs;
    if (!load_pair(scanFolder, imgPrefix, imgSuffix, n, W, H, imgs, lastError)) return false;
    slr_calib cal;
    if (!fillCalib(cal) || slr_set_calibration(ctx, &cal) != SLR_OK) { lastError = "calibration incomplete"; return false; }
    const size_t plane = (size_t)W * H;
    std::vector<const uint8_t *> pl[2];
    for (int c = 0; c < 2; c++) for (int i = 0; i < n; i++) pl[c].push_back(imgs.u8() + ((size_t)c * n + i) * plane);
    delete points3DProjView;
    points3DProjView = new PointCloudImage(scan_w, scan_h, haveColor);
    if (slr_reconstruct_gray(ctx, pl[0].data(), pl[1].data(), numOfColBits, numOfRowBits, W, W, H, blackThreshold, whiteThreshold,
                             scan_w, scan_h, points3DProjView->points.data(), points3DProjView->numOfPointsForPixel.data(),
                             SLR_MEM_HOST) != SLR_OK) { lastError = slr_last_error(ctx); warn("Reconstruct", lastError); return false; }
    return true;
}

// ---- MFReconstruct -------------------------------------------------------------------------------------------------------------
constexpr int kInSlots = 4;
struct MFReconstruct::SeriesBuffers { Pinned in[kInSlots], cloud[2]; };

MFReconstruct::MFReconstruct() { cameras = new VirtualCamera[2]; points3DProjView = nullptr; }
MFReconstruct::~MFReconstruct()
{
    delete series;
    delete points3DProjView;
    delete sr;
    delete[] cameras;
    if (ctx) slr_destroy(ctx);
    if (ctx2) slr_destroy(ctx2);
}

void MFReconstruct::getParameters(int scansn, int scanw, int scanh, int camw, int camh, int blackt, int whitet,
                                  const std::string &savePath)
{
    scanSN = scansn; scan_w = scanw; scan_h = scanh; cameraWidth = camw; cameraHeight = camh;
    blackThreshold = blackt; whiteThreshold = whitet; savePath_ = savePath;
    delete sr;
    sr = new stereoRect(savePath, camw, camh);
    sr->getParameters();
    scanFolder[0] = savePath + "/scan/left/";  calibFolder[0] = savePath + "/calib/left/";
    scanFolder[1] = savePath + "/scan/right/"; calibFolder[1] = savePath + "/calib/right/";
    setScan(scansn);
    camerasLoaded = loadCameras();
    if (!camerasLoaded) warn("Get Param", "Load Calibration files failed.");
}

void MFReconstruct::setScan(int sn)
{
    scanSN = sn;
    imgPrefix[0] = scan_prefix(sn, 'L');
    imgPrefix[1] = scan_prefix(sn, 'R');
}

bool MFReconstruct::loadCameras()
{
    for (int i = 0; i < 2; i++) {
        if (!load_camera_files(cameras[i], calibFolder[i], savePath_ + "/calib/", false)) return false;
        cameras[i].height = cameraHeight; cameras[i].width = cameraWidth;
    }
    return true;
}

// calibration + rectification maps of this project on one context
bool MFReconstruct::configure(slr_ctx *c, int sn)
{
    slr_calib cal;
    memset(&cal, 0, sizeof cal);
    cameras[0].fill(cal.cam[0]);
    cameras[1].fill(cal.cam[1]);
    if (sr->Q.empty()) { lastError = "stereo calibration files missing"; warn("Reconstruct", lastError); return false; }
    memcpy(cal.Q, sr->Q.v.data(), sizeof(double) * 16);
    if (!load_transfer(savePath_, sn, cal) || slr_set_calibration(c, &cal) != SLR_OK) {
        lastError = "calibration incomplete"; warn("Reconstruct", lastError); return false;
    }
    return true;
}

bool MFReconstruct::runReconstruction()
{
    std::vector<int> one(1, scanSN);
    PointCloudImage *result = nullptr;
    const bool ok = runReconstructionSeries(one, [&](int, PointCloudImage *pc) { result = pc; return true; });
    if (!ok) { delete result; return false; }
    delete points3DProjView;
    points3DProjView = result;
    return result != nullptr;
}

// A series of scans of one project (same cameras, same calibration): scan_sns[i] -> sink(sn, cloud); the sink owns the cloud.
// Two contexts alternate on the GPU side, up to three scans are being inflated ahead of them; see the file header.  false (and
// lastError) at the first scan that fails; clouds already handed to the sink stay there.
bool MFReconstruct::runReconstructionSeries(const std::vector<int> &scan_sns, const std::function<bool(int, PointCloudImage *)> &sink)
{
    if (!camerasLoaded || !sr) { lastError = "calibration not loaded"; return false; }
    if (scan_sns.empty()) return true;
    if (!ensure_ctx(ctx, lastError, device)) return false;
    const bool two = scan_sns.size() > 1;
    if (two && !ensure_ctx(ctx2, lastError, device)) return false;
    slr_ctx *cx[2] = {ctx, two ? ctx2 : ctx};
    sr->calParameters();
    if (sr->Q.empty()) { lastError = "stereo calibration files missing"; warn("Reconstruct", lastError); return false; }
    for (int s = 0; s < (two ? 2 : 1); s++) {
        if (!(sr->uploadFromCalibration(cx[s]) || sr->upload(cx[s]))) { lastError = "calibration incomplete"; warn("Reconstruct", lastError); return false; }
        slr_set_option(cx[s], SLR_OPT_ASYNC_HOST, two ? 1 : 0);
    }
    const int W = cameraWidth, H = cameraHeight, n = numberOfImgs;
    const size_t plane = (size_t)W * H, cells = (size_t)scan_w * scan_h;
    // Input slots: page-locked buffers the pool inflates scans into AHEAD of the GPU.  One slot is being consumed (scan i - 1, on
    // the GPU) while the others are decoded or wait decoded; two scans ahead keep every worker busy across scan boundaries, a
    // third pays only where one scan's 2 n files cannot occupy the CPUs there are.
    const unsigned cpus = std::max(1u, loader_threads(1 << 20, two) / std::max(1u, loaderShare));
    const int ahead = !two ? 1 : (cpus > (unsigned)(2 * 2 * n) ? 3 : 2);
    const int ns = two ? std::min(kInSlots, ahead + 1) : 1;
    if (!series) series = new SeriesBuffers();
    std::vector<ScanLoad> ins((size_t)ns);
    for (int k = 0; k < ns; k++) ins[(size_t)k].buf = &series->in[k];
    FilePool pool(std::min(cpus, (unsigned)(2 * n * ahead))); // (destroyed before `ins`, which its workers write to)
    auto start_load = [&](size_t i) {
        ScanLoad &in = ins[i % (size_t)ns];
        in.folder[0] = scanFolder[0]; in.folder[1] = scanFolder[1];
        in.prefix[0] = scan_prefix(scan_sns[i], 'L'); in.prefix[1] = scan_prefix(scan_sns[i], 'R');
        in.suffix = imgSuffix; in.n = n; in.w = W; in.h = H; in.sn = scan_sns[i];
        pool.submit(&in);
    };
    struct Slot { Pinned *cloud = nullptr; int sn = -1; bool busy = false; } slot[2];
    slot[0].cloud = &series->cloud[0]; slot[1].cloud = &series->cloud[1];
    auto finish = [&](int s) -> bool {                       // wait for slot s and hand its cloud over
        if (!slot[s].busy) return true;
        slot[s].busy = false;
        if (slr_synchronize(cx[s]) != SLR_OK) { lastError = slr_last_error(cx[s]); warn("Reconstruct", lastError); return false; }
        PointCloudImage *pc = new PointCloudImage(scan_w, scan_h, false);
        memcpy(pc->points.data(), slot[s].cloud->p, cells * 12);
        memcpy(pc->numOfPointsForPixel.data(), slot[s].cloud->u8() + cells * 12, cells);
        return sink(slot[s].sn, pc);
    };
    bool ok = true;
    size_t next = 0;                                         // first scan whose decode has not been started
    const bool trace = getenv("SLR_SERIES_TRACE") != nullptr;   // per-scan wall times of the stages below, on stderr
    auto now = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    for (size_t i = 0; i < scan_sns.size() && ok; i++) {
        const int s = (int)(i & 1) * (two ? 1 : 0);
        const double t0 = now();
        ok = finish(s);                                      // scan i - 2 used this context: its input slot is free again
        if (!ok) break;
        const double t1 = now();
        // input slots in use right now: scan i - 1 (on the GPU) and the scans i .. next - 1 already decoding
        while (next < scan_sns.size() && next < i + (size_t)ns - (i > 0 ? 1 : 0) && (ns > 1 || next == i)) start_load(next++);
        Slot &sl = slot[s];
        ScanLoad &in = ins[i % (size_t)ns];
        in.wait();                                           // (decoded while the GPU worked on the scans before)
        const double t2 = now();
        if (!in.ok(lastError)) { ok = false; break; }
        setScan(scan_sns[i]);
        if (!sl.cloud->ensure(cells * 13)) { lastError = "out of page-locked memory"; ok = false; break; }
        if (!configure(cx[s], scan_sns[i])) { ok = false; break; }
        const uint8_t *pl[2][SLR_MF_PLANES];
        for (int c = 0; c < 2; c++) for (int k = 0; k < SLR_MF_PLANES; k++) pl[c][k] = in.buf->u8() + ((size_t)c * n + k) * plane;
        if (slr_reconstruct_mf_cloud(cx[s], pl[0], pl[1], W, W, H, blackThreshold, 1, scan_w, scan_h, (float *)sl.cloud->p,
                                     sl.cloud->u8() + cells * 12, SLR_MEM_HOST) != SLR_OK) {
            lastError = slr_last_error(cx[s]); warn("Reconstruct", lastError); ok = false; break;
        }
        sl.sn = scan_sns[i]; sl.busy = true;
        if (trace)
            fprintf(stderr, "[series] scan %d: previous cloud %.1f ms, wait for the decoder %.1f ms (its files took %.1f ms on %u threads), "
                    "submit %.1f ms\n", scan_sns[i], t1 - t0, t2 - t1, in.t_end - in.t_start, std::min(cpus, (unsigned)(2 * n * ahead)), now() - t2);
    }
    // drain in scan order
    const int last = (int)((scan_sns.size() - 1) & 1) * (two ? 1 : 0);
    if (two) ok = finish(1 - last) && ok; else (void)0;
    ok = finish(last) && ok;
    for (int s = 0; s < 2; s++) if (slot[s].busy) { (void)slr_synchronize(cx[s]); slot[s].busy = false; }
    for (int s = 0; s < (two ? 2 : 1); s++) slr_set_option(cx[s], SLR_OPT_ASYNC_HOST, 0);
    return ok;
}

}  // namespace duke
