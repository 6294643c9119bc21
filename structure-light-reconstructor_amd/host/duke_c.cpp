// duke_c.cpp -- extern "C" hooks over the C++ host mirror (libslr_host.so) so tests can drive it through ctypes, and
// the body of slr_cli.  duke_run_project mirrors MainWindow::startreconstruct (Duke/mainwindow.cpp:562-652).
#include <string.h>

#include <atomic>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "duke.hpp"

using namespace duke;

extern "C" {

static void set_err(char *err, int err_len, const std::string &msg)
{
    if (err && err_len > 0) { strncpy(err, msg.c_str(), (size_t)err_len - 1); err[err_len - 1] = 0; }
}

// every entry point below is a C boundary: no exception may cross it
#define DUKE_GUARD_BEGIN try {
#define DUKE_GUARD_END(fail_value, err, err_len)                                                      \
    } catch (const std::exception &ex) { set_err(err, err_len, std::string("exception: ") + ex.what()); return fail_value; } \
      catch (...) { set_err(err, err_len, "unknown exception"); return fail_value; }

// mode: 0 = GRAY_ONLY, 1 = GRAY_EPI, 2 = MULTIFREQ_EPI (mainwindow.h:95 codePatternUsed)
// pc_sum [scan_h][scan_w][3], pc_count [scan_h][scan_w] receive the PointCloudImage (may be NULL); out_ply may be NULL/"".
// returns 1 on success, 0 on failure (err receives the message)
int duke_run_project(const char *project, int mode, int sn, int scan_w, int scan_h, int cam_w, int cam_h, int black_thr,
                     int white_thr, int have_color, const char *suffix, const char *out_ply, float *pc_sum,
                     uint8_t *pc_count, uint8_t *pc_color, char *err, int err_len)
{
    DUKE_GUARD_BEGIN
    if (!project || cam_w <= 0 || cam_h <= 0 || scan_w <= 0 || scan_h <= 0 || (long long)cam_w * cam_h > (1ll << 28)) {
        set_err(err, err_len, "bad argument");
        return 0;
    }
    std::string msg;
    PointCloudImage *cloud = nullptr;
    std::unique_ptr<Reconstruct> reconstructor;
    std::unique_ptr<MFReconstruct> mfr;
    bool ok = false;
    if (mode == 0 || mode == 1) {
        reconstructor.reset(new Reconstruct(mode == 1));
        reconstructor->scanSN = sn;
        if (suffix && *suffix) reconstructor->imgSuffix = suffix;
        reconstructor->getParameters(scan_w, scan_h, cam_w, cam_h, false, have_color != 0, project);
        reconstructor->setCalibPath(std::string(project) + "/calib/left/", 0);
        reconstructor->setCalibPath(std::string(project) + "/calib/right/", 1);
        if (reconstructor->loadCameras()) {
            reconstructor->setBlackThreshold(black_thr);
            reconstructor->setWhiteThreshold(white_thr);
            reconstructor->disableRaySampling();
            ok = mode == 0 ? reconstructor->runReconstruction() : reconstructor->runReconstruction_GE();
        } else reconstructor->lastError = "Load Calibration files failed.";
        msg = reconstructor->lastError;
        cloud = reconstructor->points3DProjView;
    } else if (mode == 2) {
        mfr.reset(new MFReconstruct());
        if (suffix && *suffix) mfr->imgSuffix = suffix;
        mfr->getParameters(sn, scan_w, scan_h, cam_w, cam_h, black_thr, white_thr, project);
        ok = mfr->camerasLoaded && mfr->runReconstruction();
        msg = mfr->lastError.empty() && !mfr->camerasLoaded ? "Load Calibration files failed." : mfr->lastError;
        cloud = mfr->points3DProjView;
    } else msg = "mode must be 0 (GRAY_ONLY), 1 (GRAY_EPI) or 2 (MULTIFREQ_EPI)";
    if (ok && cloud) {
        if (pc_sum) memcpy(pc_sum, cloud->points.data(), cloud->points.size() * sizeof(float));
        if (pc_count) memcpy(pc_count, cloud->numOfPointsForPixel.data(), cloud->numOfPointsForPixel.size());
        if (pc_color && !cloud->color.empty()) memcpy(pc_color, cloud->color.data(), cloud->color.size());
        if (out_ply && *out_ply) {
            MeshCreator mc(cloud, reconstructor ? reconstructor->context() : mfr->context());
            if (!mc.exportPlyMesh(out_ply)) { ok = false; msg = "cannot write the mesh"; }
        }
    }
    set_err(err, err_len, msg);
    return ok ? 1 : 0;
    DUKE_GUARD_END(0, err, err_len)
}

// MULTIFREQ_EPI over scans sn_first .. sn_first + n_scans - 1 of one project, pipelined (MFReconstruct::runReconstructionSeries).
// pc_sum [n_scans][scan_h][scan_w][3] / pc_count [n_scans][scan_h][scan_w] (may be NULL); ply_prefix (may be NULL/""):
// <prefix><sn>.ply per scan.  Returns the number of scans completed.
int duke_run_series(const char *project, int sn_first, int n_scans, int scan_w, int scan_h, int cam_w, int cam_h, int black_thr,
                    int white_thr, const char *suffix, const char *ply_prefix, float *pc_sum, uint8_t *pc_count, char *err,
                    int err_len)
{
    DUKE_GUARD_BEGIN
    if (!project || n_scans < 0 || cam_w <= 0 || cam_h <= 0 || scan_w <= 0 || scan_h <= 0 || (long long)cam_w * cam_h > (1ll << 28)) {
        set_err(err, err_len, "bad argument");
        return 0;
    }
    MFReconstruct mfr;
    if (suffix && *suffix) mfr.imgSuffix = suffix;
    mfr.getParameters(sn_first, scan_w, scan_h, cam_w, cam_h, black_thr, white_thr, project);
    if (!mfr.camerasLoaded) { set_err(err, err_len, "Load Calibration files failed."); return 0; }
    std::vector<int> sns;
    for (int i = 0; i < n_scans; i++) sns.push_back(sn_first + i);
    const size_t cells = (size_t)scan_w * scan_h;
    int done = 0;
    const bool ok = mfr.runReconstructionSeries(sns, [&](int sn, PointCloudImage *pc) {
        std::unique_ptr<PointCloudImage> own(pc);
        const size_t i = (size_t)(sn - sn_first);
        if (pc_sum) memcpy(pc_sum + i * cells * 3, pc->points.data(), cells * 12);
        if (pc_count) memcpy(pc_count + i * cells, pc->numOfPointsForPixel.data(), cells);
        if (ply_prefix && *ply_prefix) {
            // (not mfr.context(): the series' contexts run with SLR_OPT_ASYNC_HOST and have the next scan in flight; the export
            //  takes its own context on this thread's current device, which is the one the series runs on)
            MeshCreator mc(pc);
            if (!mc.exportPlyMesh(std::string(ply_prefix) + std::to_string(sn) + ".ply")) return false;
        }
        done++;
        return true;
    });
    set_err(err, err_len, ok ? "" : mfr.lastError);
    return done;
    DUKE_GUARD_END(0, err, err_len)
}

// The same series over SEVERAL GPUs of the node from one process (SURVEY 8e + 8f-1: frames are independent, so a scan directory
// shards by scan): scan sn_first + i runs on devices[i % n_devices] (device ordinals; an ordinal may repeat -- two pipelines on one
// GPU).  One MFReconstruct per entry of `devices`, each on its own host thread with its own two contexts, page-locked slots and
// 1 / n_devices of the decoder threads: every GPU has its own PCIe link, and the host-buffer path is PCIe-bound,
// so n GPUs take n scans in the time one takes one -- as long as the CPUs inflate that fast.  Results land in the scan's slot of
// pc_sum / pc_count and in <ply_prefix><sn>.ply exactly as duke_run_series writes them.  Returns the number of scans completed
// (all devices together); on a failure `err` names the first one and the other pipelines finish the scans they already started.
int duke_run_series_multi(const char *project, int sn_first, int n_scans, int scan_w, int scan_h, int cam_w, int cam_h, int black_thr,
                          int white_thr, const char *suffix, const char *ply_prefix, const int *devices, int n_devices, float *pc_sum,
                          uint8_t *pc_count, char *err, int err_len)
{
    DUKE_GUARD_BEGIN
    if (!project || n_scans < 0 || cam_w <= 0 || cam_h <= 0 || scan_w <= 0 || scan_h <= 0 || (long long)cam_w * cam_h > (1ll << 28) ||
        !devices || n_devices < 1 || n_devices > 64) {
        set_err(err, err_len, "bad argument");
        return 0;
    }
    for (int k = 0; k < n_devices; k++) if (devices[k] < 0) { set_err(err, err_len, "bad device ordinal"); return 0; }
    const size_t cells = (size_t)scan_w * scan_h;
    std::atomic<int> done(0);
    std::mutex em;
    std::string first_err;
    auto pipeline = [&](int k) {
        std::string my_err;
        try {
            MFReconstruct mfr;
            mfr.device = devices[k];
            mfr.loaderShare = (unsigned)n_devices;
            if (suffix && *suffix) mfr.imgSuffix = suffix;
            mfr.getParameters(sn_first, scan_w, scan_h, cam_w, cam_h, black_thr, white_thr, project);
            std::vector<int> sns;
            for (int i = k; i < n_scans; i += n_devices) sns.push_back(sn_first + i);
            if (!mfr.camerasLoaded) my_err = "Load Calibration files failed.";
            else if (!sns.empty()) {
                slr_ctx *numbering = nullptr;                // the mesh export's vertex numbering runs on this pipeline's device too
                if (ply_prefix && *ply_prefix && slr_create(devices[k], &numbering) != SLR_OK) numbering = nullptr;
                const bool ok = mfr.runReconstructionSeries(sns, [&](int sn, PointCloudImage *pc) {
                    std::unique_ptr<PointCloudImage> own(pc);
                    const size_t i = (size_t)(sn - sn_first);
                    if (pc_sum) memcpy(pc_sum + i * cells * 3, pc->points.data(), cells * 12);
                    if (pc_count) memcpy(pc_count + i * cells, pc->numOfPointsForPixel.data(), cells);
                    if (ply_prefix && *ply_prefix) {
                        if (!numbering) return false;
                        MeshCreator mc(pc, numbering);
                        if (!mc.exportPlyMesh(std::string(ply_prefix) + std::to_string(sn) + ".ply")) return false;
                    }
                    done++;
                    return true;
                });
                if (numbering) slr_destroy(numbering);
                if (!ok) my_err = mfr.lastError.empty() ? "series stopped" : mfr.lastError;
            }
        } catch (const std::exception &e) { my_err = e.what(); } catch (...) { my_err = "unknown exception"; }
        if (!my_err.empty()) { std::lock_guard<std::mutex> g(em); if (first_err.empty()) first_err = my_err; }
    };
    std::vector<std::thread> th;
    for (int k = 1; k < n_devices; k++) th.emplace_back(pipeline, k);
    pipeline(0);
    for (auto &t : th) t.join();
    set_err(err, err_len, first_err);
    return done.load();
    DUKE_GUARD_END(0, err, err_len)
}

// PLY / OBJ export of a cloud handed over as arrays (tests: the file-level parity of MeshCreator)
int duke_export_mesh(const char *path, int obj, int w, int h, const float *pc_sum, const uint8_t *pc_count)
{
    DUKE_GUARD_BEGIN
    if (!path || !pc_sum || !pc_count || w <= 0 || h <= 0) return 0;
    PointCloudImage pc(w, h, false);
    memcpy(pc.points.data(), pc_sum, (size_t)w * h * 12);
    memcpy(pc.numOfPointsForPixel.data(), pc_count, (size_t)w * h);
    MeshCreator mc(&pc);
    return (obj ? mc.exportObjMesh(path) : mc.exportPlyMesh(path)) ? 1 : 0;
    DUKE_GUARD_END(0, nullptr, 0)
}

// ---- host-only pieces (no GPU needed) ---------------------------------------------------------------------------
int duke_gray_num_imgs(int scan_w, int scan_h, int use_epi) { return GrayCodes(scan_w, scan_h, use_epi != 0).getNumOfImgs(); }

int duke_gen_graycodes(int scan_w, int scan_h, int use_epi, uint8_t *out)
{
    GrayCodes g(scan_w, scan_h, use_epi != 0);
    g.generateGrays();
    const size_t plane = (size_t)scan_w * scan_h;
    for (int i = 0; i < g.getNumOfImgs(); i++) memcpy(out + plane * i, g.grayCodes[i].d.data(), plane);
    return g.getNumOfImgs();
}

void duke_gen_multifreq(int proj_w, int proj_h, uint8_t *out)
{
    MultiFrequency m(proj_w, proj_h);
    m.generateMutiFreq();
    const size_t plane = (size_t)proj_w * proj_h;
    for (int i = 0; i < 14; i++) memcpy(out + plane * i, m.MultiFreqImages[i].d.data(), plane);
}

int duke_gray_to_dec(const uint8_t *bits, int n)
{
    std::vector<bool> v(bits, bits + n);
    return GrayCodes::grayToDec(v);
}

void duke_init_undistort_rectify_map(const double *M, const double *D, const double *R, const double *P, int W, int H,
                                     int16_t *map_xy, uint16_t *map_frac)
{
    Matd m, d, r, p;
    m.rows = m.cols = 3; m.v.assign(M, M + 9);
    d.rows = 5; d.cols = 1; d.v.assign(D, D + 5);
    r.rows = r.cols = 3; r.v.assign(R, R + 9);
    p.rows = 3; p.cols = 4; p.v.assign(P, P + 12);
    std::vector<int16_t> xy;
    std::vector<uint16_t> fr;
    initUndistortRectifyMap(m, d, r, p, W, H, xy, fr);
    memcpy(map_xy, xy.data(), xy.size() * sizeof(int16_t));
    memcpy(map_frac, fr.data(), fr.size() * sizeof(uint16_t));
}

// stereoRect::getParameters + calParameters on a project directory; outputs R1,R2 (9), P1,P2 (12), Q (16) and the maps
int duke_stereo_rect(const char *project, int W, int H, double *R1, double *R2, double *P1, double *P2, double *Q,
                     int16_t *map11, uint16_t *map12, int16_t *map21, uint16_t *map22)
{
    DUKE_GUARD_BEGIN
    if (!project || W <= 0 || H <= 0 || (long long)W * H > (1ll << 28)) return 0;
    stereoRect sr(project, W, H);
    sr.getParameters();
    sr.calParameters();
    if (sr.Q.empty()) return 0;
    memcpy(R1, sr.R1.v.data(), 72); memcpy(R2, sr.R2.v.data(), 72);
    memcpy(P1, sr.P1.v.data(), 96); memcpy(P2, sr.P2.v.data(), 96);
    memcpy(Q, sr.Q.v.data(), 128);
    if (map11) memcpy(map11, sr.map11.data(), sr.map11.size() * 2);
    if (map12) memcpy(map12, sr.map12.data(), sr.map12.size() * 2);
    if (map21) memcpy(map21, sr.map21.data(), sr.map21.size() * 2);
    if (map22) memcpy(map22, sr.map22.data(), sr.map22.size() * 2);
    return 1;
    DUKE_GUARD_END(0, nullptr, 0)
}

// cv::stereoRectify(M1, D1, M2, D2, size, R, T, flags 0, alpha -1) from arrays (row-major f64: M 3x3, D 5, R 3x3, T 3):
// R1, R2 (9), P1, P2 (12), Q (16) -- what stereoRect::calParameters computes before it builds the maps (bench.py builds the maps
// of verged rigs on the device with slr_init_rectify_maps from these)
int duke_stereo_rectify(const double *M1, const double *D1, const double *M2, const double *D2, const double *R, const double *T,
                        int W, int H, double *R1, double *R2, double *P1, double *P2, double *Q)
{
    DUKE_GUARD_BEGIN
    if (!M1 || !D1 || !M2 || !D2 || !R || !T || !R1 || !R2 || !P1 || !P2 || !Q || W <= 0 || H <= 0) return 0;
    stereoRect sr("", W, H);
    auto set = [](Matd &m, int rows, int cols, const double *src) { m.rows = rows; m.cols = cols; m.v.assign(src, src + (size_t)rows * cols); };
    set(sr.M1, 3, 3, M1); set(sr.D1, 5, 1, D1); set(sr.M2, 3, 3, M2); set(sr.D2, 5, 1, D2); set(sr.R, 3, 3, R); set(sr.T, 3, 1, T);
    sr.calRectification();
    if (sr.Q.empty()) return 0;
    memcpy(R1, sr.R1.v.data(), 72); memcpy(R2, sr.R2.v.data(), 72);
    memcpy(P1, sr.P1.v.data(), 96); memcpy(P2, sr.P2.v.data(), 96);
    memcpy(Q, sr.Q.v.data(), 128);
    return 1;
    DUKE_GUARD_END(0, nullptr, 0)
}

// png: 0 = PGM, 1 = PNG, 2 = Adam7-interlaced PNG (test input)
int duke_imwrite(const char *path, const uint8_t *data, int w, int h, int png)
{
    DUKE_GUARD_BEGIN
    if (!path || !data || w <= 0 || h <= 0) return 0;
    Image8 img;
    img.w = w; img.h = h; img.d.assign(data, data + (size_t)w * h);
    return (png ? imwrite_png(path, img, png == 2) : imwrite_pgm(path, img)) ? 1 : 0;
    DUKE_GUARD_END(0, nullptr, 0)
}

int duke_imread(const char *path, uint8_t *data, int cap, int *w, int *h)
{
    DUKE_GUARD_BEGIN
    if (!path || !w || !h) return 0;
    Image8 img = imread_gray(path);
    if (img.empty()) return 0;
    *w = img.w; *h = img.h;
    if (!data || (size_t)cap < img.d.size()) return -1;
    memcpy(data, img.d.data(), img.d.size());
    return 1;
    DUKE_GUARD_END(0, nullptr, 0)
}

int duke_export_mat(const char *path, const double *m, int rows, int cols) { return exportMat(path, m, rows, cols) ? 1 : 0; }

// VirtualCamera::loadMatrix (through float) -- returns 1 / -1 like the reference
int duke_load_matrix(const char *path, int rows, int cols, float *out)
{
    VirtualCamera vc;
    Matf m;
    const int r = vc.loadMatrix(m, rows, cols, path);
    if (r > 0) memcpy(out, m.v.data(), m.v.size() * sizeof(float));
    return r;
}

// PointCloudImage semantics (addPoint sequence) for tests: pts [n][5] = (i_w, j_h, x, y, z)
void duke_pointcloud_accumulate(int w, int h, const float *pts, int n, float *sum, uint8_t *count, float *mean)
{
    PointCloudImage pc(w, h, false);
    for (int k = 0; k < n; k++) {
        Point3f p; p.x = pts[5 * k + 2]; p.y = pts[5 * k + 3]; p.z = pts[5 * k + 4];
        pc.addPoint((int)pts[5 * k], (int)pts[5 * k + 1], p);
    }
    memcpy(sum, pc.points.data(), pc.points.size() * sizeof(float));
    memcpy(count, pc.numOfPointsForPixel.data(), pc.numOfPointsForPixel.size());
    for (int j = 0; j < h; j++)
        for (int i = 0; i < w; i++) {
            Point3f p;
            const bool ok = pc.getPoint(i, j, p);
            float *m = mean + ((size_t)j * w + i) * 3;
            m[0] = ok ? p.x : 0; m[1] = ok ? p.y : 0; m[2] = ok ? p.z : 0;
        }
}

}  // extern "C"
