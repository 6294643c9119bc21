// duke_c.cpp -- extern "C" hooks over the C++ host mirror (libslr_host.so) so tests can drive it through ctypes, and
// the body of slr_cli.  duke_run_project mirrors MainWindow::startreconstruct (Duke/mainwindow.cpp:562-652).
#include <string.h>

#include <string>

#include "duke.hpp"

using namespace duke;

extern "C" {

// mode: 0 = GRAY_ONLY, 1 = GRAY_EPI, 2 = MULTIFREQ_EPI (mainwindow.h:95 codePatternUsed)
// pc_sum [scan_h][scan_w][3], pc_count [scan_h][scan_w] receive the PointCloudImage (may be NULL); out_ply may be NULL/"".
// returns 1 on success, 0 on failure (err receives the message)
int duke_run_project(const char *project, int mode, int sn, int scan_w, int scan_h, int cam_w, int cam_h, int black_thr,
                     int white_thr, int have_color, const char *suffix, const char *out_ply, float *pc_sum,
                     uint8_t *pc_count, uint8_t *pc_color, char *err, int err_len)
{
    std::string msg;
    PointCloudImage *cloud = nullptr;
    Reconstruct *reconstructor = nullptr;
    MFReconstruct *mfr = nullptr;
    bool ok = false;
    if (mode == 0 || mode == 1) {
        reconstructor = new Reconstruct(mode == 1);
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
    } else {
        mfr = new MFReconstruct();
        if (suffix && *suffix) mfr->imgSuffix = suffix;
        mfr->getParameters(sn, scan_w, scan_h, cam_w, cam_h, black_thr, white_thr, project);
        ok = mfr->camerasLoaded && mfr->runReconstruction();
        msg = mfr->lastError.empty() && !mfr->camerasLoaded ? "Load Calibration files failed." : mfr->lastError;
        cloud = mfr->points3DProjView;
    }
    if (ok && cloud) {
        if (pc_sum) memcpy(pc_sum, cloud->points.data(), cloud->points.size() * sizeof(float));
        if (pc_count) memcpy(pc_count, cloud->numOfPointsForPixel.data(), cloud->numOfPointsForPixel.size());
        if (pc_color && !cloud->color.empty()) memcpy(pc_color, cloud->color.data(), cloud->color.size());
        if (out_ply && *out_ply) { MeshCreator mc(cloud); mc.exportPlyMesh(out_ply); }
    }
    if (err && err_len > 0) { strncpy(err, msg.c_str(), (size_t)err_len - 1); err[err_len - 1] = 0; }
    delete reconstructor;
    delete mfr;
    return ok ? 1 : 0;
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
}

int duke_imwrite(const char *path, const uint8_t *data, int w, int h, int png)
{
    Image8 img;
    img.w = w; img.h = h; img.d.assign(data, data + (size_t)w * h);
    return (png ? imwrite_png(path, img) : imwrite_pgm(path, img)) ? 1 : 0;
}

int duke_imread(const char *path, uint8_t *data, int cap, int *w, int *h)
{
    Image8 img = imread_gray(path);
    if (img.empty()) return 0;
    *w = img.w; *h = img.h;
    if ((size_t)cap < img.d.size()) return -1;
    memcpy(data, img.d.data(), img.d.size());
    return 1;
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
