// reconstruct.cpp -- Reconstruct / MFReconstruct of the host mirror (Duke/reconstruct.cpp:99-307, Duke/mfreconstruct.cpp:67-187:
// the public call sequence of the reference; every per-pixel loop is ONE call into libslr_hip.so) and the scan-directory reader
// in front of them (reconstruct.cpp:158-164, mfreconstruct.cpp:119-125).
//
// Loader.  PNG inflate runs at ~200 MB/s per core: ~60 ms per 4096x3000 plane, 1.7 s for the 28 images of one
// multi-frequency scan when done one file after the other as the reference does -- three orders of magnitude more than the GPU
// path.  Here the files of a scan are decoded by a pool of host threads (one file per task) STRAIGHT INTO page-locked memory
// (slr_host_alloc), plane after plane in the layout the C ABI stages from, and a series of scans is pipelined: two contexts with
// SLR_OPT_ASYNC_HOST alternate on the GPU side (upload / reconstruct / download of scan i while scan i + 1 is enqueued), and
// background threads inflate up to three scans AHEAD into four page-locked input slots (round 3; round 2 decoded one scan ahead:
// one plane's inflate, ~60 ms on one core, was the time per scan).  The steady state costs max(decode / 3, upload + kernels +
// download) per scan.
#include "duke.hpp"

#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <sstream>
#include <thread>

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

unsigned loader_threads(int n)
{
    unsigned nt = std::thread::hardware_concurrency();
    if (const char *e = getenv("SLR_LOADER_THREADS")) nt = (unsigned)atoi(e);
    if (nt < 1) nt = 1;
    if (nt > (unsigned)n) nt = (unsigned)(n > 0 ? n : 1);
    return nt;
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
                std::string e1, e2;
                uint8_t *out = dst + (size_t)t * w * h;
                if (imread_gray_into(p.str() + suffix, w, h, out, e1)) continue;
                if (suffix != ".pgm" && imread_gray_into(p.str() + ".pgm", w, h, out, e2)) continue;
                errs[(size_t)t] = e1.compare(0, 11, "cannot open") == 0 ? "Scan Images not found! (" + p.str() + suffix + ")" : e1;
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

bool ensure_ctx(slr_ctx *&ctx, std::string &err)
{
    if (ctx) return true;
    const int st = slr_create(0, &ctx);
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
MFReconstruct::MFReconstruct() { cameras = new VirtualCamera[2]; points3DProjView = nullptr; }
MFReconstruct::~MFReconstruct()
{
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
    if (!ensure_ctx(ctx, lastError)) return false;
    const bool two = scan_sns.size() > 1;
    if (two && !ensure_ctx(ctx2, lastError)) return false;
    slr_ctx *cx[2] = {ctx, two ? ctx2 : ctx};
    sr->calParameters();
    if (sr->Q.empty()) { lastError = "stereo calibration files missing"; warn("Reconstruct", lastError); return false; }
    for (int s = 0; s < (two ? 2 : 1); s++) {
        if (!(sr->uploadFromCalibration(cx[s]) || sr->upload(cx[s]))) { lastError = "calibration incomplete"; warn("Reconstruct", lastError); return false; }
        slr_set_option(cx[s], SLR_OPT_ASYNC_HOST, two ? 1 : 0);
    }
    const int W = cameraWidth, H = cameraHeight, n = numberOfImgs;
    const size_t plane = (size_t)W * H, cells = (size_t)scan_w * scan_h;
    // Input slots: page-locked buffers that background threads inflate scans into AHEAD of the GPU.  Two scans can be on the GPU
    // (the two contexts), so with kInSlots buffers kInSlots - 1 scans are being decoded or wait decoded while one is consumed:
    // the 28 files of ONE scan keep 28 cores busy for the ~60 ms one 12 MB plane takes to inflate, so a single scan ahead bounds
    // the series at that time per scan (round 2: 100 ms); several scans ahead divide it.
    constexpr int kInSlots = 4;
    const int ns = two ? kInSlots : 1;
    struct InSlot { Pinned buf; std::thread th; bool ok = false; std::string err; };
    std::vector<InSlot> ins((size_t)ns);
    struct Joiner {                                           // no path may leave a loader thread behind
        std::vector<InSlot> &v;
        ~Joiner() { for (auto &in : v) if (in.th.joinable()) in.th.join(); }
    } joiner{ins};
    auto start_load = [&](size_t i) {
        InSlot &in = ins[i % (size_t)ns];
        if (in.th.joinable()) in.th.join();
        in.ok = false; in.err.clear();
        const std::string pl = scan_prefix(scan_sns[i], 'L'), pr = scan_prefix(scan_sns[i], 'R');
        in.th = std::thread([this, &in, pl, pr, n, W, H]() {
            try {
                const std::string prefix[2] = {pl, pr};
                in.ok = load_pair(scanFolder, prefix, imgSuffix, n, W, H, in.buf, in.err);
            } catch (const std::exception &ex) { in.ok = false; in.err = std::string("loader: ") + ex.what(); }
            catch (...) { in.ok = false; in.err = "loader: unknown exception"; }
        });
    };
    struct Slot { Pinned cloud; int sn = -1; bool busy = false; } slot[2];
    auto finish = [&](int s) -> bool {                       // wait for slot s and hand its cloud over
        if (!slot[s].busy) return true;
        slot[s].busy = false;
        if (slr_synchronize(cx[s]) != SLR_OK) { lastError = slr_last_error(cx[s]); warn("Reconstruct", lastError); return false; }
        PointCloudImage *pc = new PointCloudImage(scan_w, scan_h, false);
        memcpy(pc->points.data(), slot[s].cloud.p, cells * 12);
        memcpy(pc->numOfPointsForPixel.data(), slot[s].cloud.u8() + cells * 12, cells);
        return sink(slot[s].sn, pc);
    };
    bool ok = true;
    size_t next = 0;                                         // first scan whose decode has not been started
    for (size_t i = 0; i < scan_sns.size() && ok; i++) {
        const int s = (int)(i & 1) * (two ? 1 : 0);
        ok = finish(s);                                      // scan i - 2 used this context: its input slot is free again
        if (!ok) break;
        // input slots in use right now: scan i - 1 (on the GPU) and the scans i .. next - 1 already decoding
        while (next < scan_sns.size() && next < i + (size_t)ns - (i > 0 ? 1 : 0) && (ns > 1 || next == i)) start_load(next++);
        Slot &sl = slot[s];
        InSlot &in = ins[i % (size_t)ns];
        if (in.th.joinable()) in.th.join();                  // (decoded while the GPU worked on the scans before)
        if (!in.ok) { lastError = in.err; ok = false; break; }
        setScan(scan_sns[i]);
        if (!sl.cloud.ensure(cells * 13)) { lastError = "out of page-locked memory"; ok = false; break; }
        if (!configure(cx[s], scan_sns[i])) { ok = false; break; }
        const uint8_t *pl[2][SLR_MF_PLANES];
        for (int c = 0; c < 2; c++) for (int k = 0; k < SLR_MF_PLANES; k++) pl[c][k] = in.buf.u8() + ((size_t)c * n + k) * plane;
        if (slr_reconstruct_mf_cloud(cx[s], pl[0], pl[1], W, W, H, blackThreshold, 1, scan_w, scan_h, (float *)sl.cloud.p,
                                     sl.cloud.u8() + cells * 12, SLR_MEM_HOST) != SLR_OK) {
            lastError = slr_last_error(cx[s]); warn("Reconstruct", lastError); ok = false; break;
        }
        sl.sn = scan_sns[i]; sl.busy = true;
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
