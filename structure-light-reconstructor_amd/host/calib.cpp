// calib.cpp -- calibration text I/O and stereo rectification of the host mirror.
//   VirtualCamera            Duke/virtualcamera.cpp:25-88     one camera's text files -> slr_camera
//   stereoRect               Duke/stereorect.cpp:3-62         six stereo text files -> cv::stereoRectify + 2 x cv::initUndistortRectifyMap
// The reference's matrices are whitespace-separated text parsed through `float`.  All files of a camera / of the
// stereo pair are described by one table each (name, shape, destination) and read by one routine.
// cv::stereoRectify / cv::initUndistortRectifyMap are restated from OpenCV 2.4's published algorithm (calib3d
// cvStereoRectify, imgproc undistort): PARITY UNPINNED, there is no OpenCV in this image to compare with; an independent
// fp64 NumPy transcription (tests/np_model.py) checks this one to 1e-12.
#include "duke.hpp"

#include <float.h>
#include <math.h>
#include <string.h>

#include <fstream>
#include <iostream>

namespace duke {

void warn(const std::string &title, const std::string &msg)      // QMessageBox::warning stand-in
{
    std::cerr << "[" << title << "] " << msg << std::endl;
}

// rows x cols numbers, each parsed as a float; missing numbers read as 0 like an exhausted istream does.  false: no such file.
bool read_matrix_text(const std::string &file, int rows, int cols, std::vector<float> &out)
{
    std::ifstream in(file.c_str());
    if (!in) return false;
    out.assign((size_t)rows * cols, 0.0f);
    for (float &x : out) {
        float t = 0;
        if (in >> t) x = t; else break;
    }
    return true;
}

// ---- VirtualCamera -------------------------------------------------------------------------------------------------------
VirtualCamera::VirtualCamera() { fc[0] = fc[1] = cc[0] = cc[1] = 0; }

int VirtualCamera::loadMatrix(Matf &matrix, int rows, int cols, const std::string &file)
{
    std::vector<float> v;
    if (!read_matrix_text(file, rows, cols, v)) return -1;
    matrix.rows = rows; matrix.cols = cols; matrix.v.swap(v);
    return 1;
}
void VirtualCamera::loadDistortion(const std::string &path) { loadMatrix(distortion, 5, 1, path); }
bool VirtualCamera::loadCameraMatrix(const std::string &path)
{
    Matf K;
    if (loadMatrix(K, 3, 3, path) < 0) { warn("Matrix not found", "File: '" + path + "' need to be added."); return false; }
    fc[0] = K.at(0, 0); fc[1] = K.at(1, 1); cc[0] = K.at(0, 2); cc[1] = K.at(1, 2);
    return true;
}
void VirtualCamera::loadRotationMatrix(const std::string &path) { loadMatrix(rotationMatrix, 3, 3, path); }
void VirtualCamera::loadTranslationVector(const std::string &path) { loadMatrix(translationVector, 3, 1, path); }
void VirtualCamera::loadFundamentalMatrix(const std::string &path) { loadMatrix(fundamentalMatrix, 3, 3, path); }
void VirtualCamera::loadHomoMatrix(const std::string &path, int i) { loadMatrix(i == 1 ? homoMat1 : homoMat2, 3, 3, path); }

void VirtualCamera::fill(slr_camera &o) const
{
    memset(&o, 0, sizeof o);
    o.fc[0] = fc[0]; o.fc[1] = fc[1]; o.cc[0] = cc[0]; o.cc[1] = cc[1];
    for (int i = 0; i < 5 && i < (int)distortion.v.size(); i++) o.k[i] = distortion.v[i];
    for (int i = 0; i < 9 && i < (int)rotationMatrix.v.size(); i++) o.R[i] = rotationMatrix.v[i];
    for (int i = 0; i < 3 && i < (int)translationVector.v.size(); i++) o.t[i] = translationVector.v[i];
    if (rotationMatrix.v.empty()) { o.R[0] = o.R[4] = o.R[8] = 1.0f; }
}

// the files of one camera below <calib folder>/ plus the pair's shared ones below <project>/calib/ (reconstruct.cpp:99-148,
// mfreconstruct.cpp:67-108).  Only the camera matrix is mandatory, as in the reference.
bool load_camera_files(VirtualCamera &cam, const std::string &camFolder, const std::string &projectCalib, bool withHomographies)
{
    if (!cam.loadCameraMatrix(camFolder + "cam_matrix.txt")) return false;
    cam.loadDistortion(camFolder + "cam_distortion.txt");
    cam.loadRotationMatrix(camFolder + "cam_rotation_matrix.txt");
    cam.loadTranslationVector(camFolder + "cam_trans_vectror.txt");           // (sic: the file name the calibration step writes)
    cam.loadFundamentalMatrix(projectCalib + "fundamental_stereo.txt");
    if (withHomographies) { cam.loadHomoMatrix(projectCalib + "H1_mat.txt", 1); cam.loadHomoMatrix(projectCalib + "H2_mat.txt", 2); }
    return true;
}

// ---- stereoRect ------------------------------------------------------------------------------------------------------------
stereoRect::stereoRect(const std::string &projectPath, int width, int height) : w(width), h(height), ppath(projectPath) {}

void stereoRect::loadMatrix(Matd &matrix, int rows, int cols, const std::string &file)
{
    std::vector<float> v;
    if (!read_matrix_text(file, rows, cols, v)) return;
    matrix.rows = rows; matrix.cols = cols;
    matrix.v.assign(v.begin(), v.end());                     // float, then widened (stereorect.cpp:57-59)
}

void stereoRect::getParameters()
{
    struct Item { Matd stereoRect::*dst; int rows, cols; const char *file; };
    static const Item items[] = {
        {&stereoRect::M1, 3, 3, "/calib/left/cam_stereo.txt"},  {&stereoRect::D1, 5, 1, "/calib/left/distortion_stereo.txt"},
        {&stereoRect::M2, 3, 3, "/calib/right/cam_stereo.txt"}, {&stereoRect::D2, 5, 1, "/calib/right/distortion_stereo.txt"},
        {&stereoRect::R, 3, 3, "/calib/R_stereo.txt"},          {&stereoRect::T, 3, 1, "/calib/T_stereo.txt"},
    };
    for (const Item &it : items) loadMatrix(this->*(it.dst), it.rows, it.cols, ppath + it.file);
}

static void rodrigues_vec2mat(const double r[3], double Rm[9])
{
    double theta = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    if (theta < DBL_EPSILON) { for (int i = 0; i < 9; i++) Rm[i] = (i % 4 == 0) ? 1 : 0; return; }
    double c = cos(theta), s = sin(theta), c1 = 1. - c, it = 1. / theta;
    double x = r[0] * it, y = r[1] * it, z = r[2] * it;
    const double rrt[9] = {x * x, x * y, x * z, x * y, y * y, y * z, x * z, y * z, z * z};
    const double rx[9] = {0, -z, y, z, 0, -x, -y, x, 0};
    for (int k = 0; k < 9; k++) Rm[k] = c * ((k % 4 == 0) ? 1 : 0) + c1 * rrt[k] + s * rx[k];
}

// Nearest rotation to a 3x3 matrix: U * V^T of its singular value decomposition, which is what cvRodrigues2 (OpenCV 2.4.9
// calib3d) does to its input before it reads the angle off it -- a rotation parsed from a text file with six significant
// digits is only orthonormal to ~1e-6.  One-sided Jacobi iteration (Hestenes): plane rotations applied from the right make
// the columns of A * V mutually orthogonal; their norms are the singular values and A * V / sigma = U, so U * V^T = sum over k
// of (normalised column k of A V) (column k of V)^T.  A rotation has three singular values near 1: no rank handling needed.
static void nearest_rotation(const double A[9], double Rn[9])
{
    double B[9], V[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    memcpy(B, A, sizeof B);
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = 0;
        for (int p = 0; p < 2; p++)
            for (int q = p + 1; q < 3; q++) {
                double app = 0, aqq = 0, apq = 0;
                for (int i = 0; i < 3; i++) { app += B[i * 3 + p] * B[i * 3 + p]; aqq += B[i * 3 + q] * B[i * 3 + q]; apq += B[i * 3 + p] * B[i * 3 + q]; }
                off = fmax(off, fabs(apq) / sqrt(app * aqq + DBL_MIN));
                if (fabs(apq) <= 1e-17 * sqrt(app * aqq)) continue;
                const double zeta = (aqq - app) / (2 * apq);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1 + zeta * zeta));
                const double cs = 1 / sqrt(1 + t * t), sn = cs * t;
                for (int i = 0; i < 3; i++) {
                    const double bp = B[i * 3 + p], bq = B[i * 3 + q];
                    B[i * 3 + p] = cs * bp - sn * bq; B[i * 3 + q] = sn * bp + cs * bq;
                    const double vp = V[i * 3 + p], vq = V[i * 3 + q];
                    V[i * 3 + p] = cs * vp - sn * vq; V[i * 3 + q] = sn * vp + cs * vq;
                }
            }
        if (off < 1e-16) break;
    }
    for (int i = 0; i < 9; i++) Rn[i] = 0;
    for (int k = 0; k < 3; k++) {
        const double nk = sqrt(B[k] * B[k] + B[3 + k] * B[3 + k] + B[6 + k] * B[6 + k]);
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) Rn[i * 3 + j] += (B[i * 3 + k] / nk) * V[j * 3 + k];
    }
}

static void rodrigues_mat2vec(const double Rin[9], double r[3])
{
    double Rm[9];
    nearest_rotation(Rin, Rm);
    double rx = Rm[7] - Rm[5], ry = Rm[2] - Rm[6], rz = Rm[3] - Rm[1];
    double s = sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
    double c = (Rm[0] + Rm[4] + Rm[8] - 1) * 0.5;
    c = c > 1. ? 1. : c < -1. ? -1. : c;
    double theta = acos(c);
    if (s < 1e-5) {
        if (c > 0) { r[0] = r[1] = r[2] = 0; return; }
        double t;
        t = (Rm[0] + 1) * 0.5; r[0] = sqrt(t > 0 ? t : 0);
        t = (Rm[4] + 1) * 0.5; r[1] = sqrt(t > 0 ? t : 0) * (Rm[1] < 0 ? -1. : 1.);
        t = (Rm[8] + 1) * 0.5; r[2] = sqrt(t > 0 ? t : 0) * (Rm[2] < 0 ? -1. : 1.);
        if (fabs(r[0]) < fabs(r[1]) && fabs(r[0]) < fabs(r[2]) && (Rm[5] > 0) != (r[1] * r[2] > 0)) r[2] = -r[2];
        double n = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
        theta /= n;
        r[0] *= theta; r[1] *= theta; r[2] *= theta;
        return;
    }
    double vth = 1 / (2 * s) * theta;
    r[0] = rx * vth; r[1] = ry * vth; r[2] = rz * vth;
}

static void mat3mul(const double A[9], const double B[9], double C[9], bool bT)
{
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += A[i * 3 + k] * (bT ? B[j * 3 + k] : B[k * 3 + j]);
            C[i * 3 + j] = s;
        }
}

// cvUndistortPoints for one point, identity R and P, results stored as float like the CV_32FC2 matrix
static void undistort_corner(double u, double v, const Matd &A, const Matd &D, float &ox, float &oy)
{
    double fx = A.at(0, 0), fy = A.at(1, 1), cx = A.at(0, 2), cy = A.at(1, 2);
    double k[5] = {D.v[0], D.v[1], D.v[2], D.v[3], D.v[4]};
    double x = (u - cx) / fx, y = (v - cy) / fy, x0 = x, y0 = y;
    for (int j = 0; j < 5; j++) {
        double r2 = x * x + y * y;
        double icdist = 1. / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
        double dx = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x);
        double dy = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y;
        x = (x0 - dx) * icdist;
        y = (y0 - dy) * icdist;
    }
    ox = (float)x; oy = (float)y;
}

void stereoRect::calParameters()
{
    calRectification();
    if (Q.empty()) return;
    initUndistortRectifyMap(M1, D1, R1, P1, w, h, map11, map12);
    initUndistortRectifyMap(M2, D2, R2, P2, w, h, map21, map22);
}

// cv::stereoRectify(flags 0, alpha -1) alone: R1, R2, P1, P2, Q from M1, D1, M2, D2, R, T (the maps can then be built on the
// device: uploadFromCalibration)
void stereoRect::calRectification()
{
    if (M1.empty() || M2.empty() || D1.empty() || D2.empty() || R.empty() || T.empty()) return;
    const int nx = w, ny = h;
    double om[3], r_r[9], t[3], uu[3] = {0, 0, 0}, ww[3], wR[9], Ri[9];
    rodrigues_mat2vec(R.v.data(), om);
    for (int i = 0; i < 3; i++) om[i] *= -0.5;               // average rotation
    rodrigues_vec2mat(om, r_r);
    for (int i = 0; i < 3; i++) t[i] = r_r[i * 3] * T.v[0] + r_r[i * 3 + 1] * T.v[1] + r_r[i * 3 + 2] * T.v[2];
    const int idx = fabs(t[0]) > fabs(t[1]) ? 0 : 1;
    const double c = t[idx], nt = sqrt(t[0] * t[0] + t[1] * t[1] + t[2] * t[2]);
    uu[idx] = c > 0 ? 1 : -1;
    ww[0] = t[1] * uu[2] - t[2] * uu[1]; ww[1] = t[2] * uu[0] - t[0] * uu[2]; ww[2] = t[0] * uu[1] - t[1] * uu[0];
    const double nw = sqrt(ww[0] * ww[0] + ww[1] * ww[1] + ww[2] * ww[2]);
    if (nw > 0.0) { const double sc = acos(fabs(c) / nt) / nw; ww[0] *= sc; ww[1] *= sc; ww[2] *= sc; }
    rodrigues_vec2mat(ww, wR);
    R1.rows = R1.cols = R2.rows = R2.cols = 3; R1.v.resize(9); R2.v.resize(9);
    mat3mul(wR, r_r, Ri, true);  memcpy(R1.v.data(), Ri, sizeof Ri);       // R1 = wR * r_r^T
    mat3mul(wR, r_r, Ri, false); memcpy(R2.v.data(), Ri, sizeof Ri);       // R2 = wR * r_r
    for (int i = 0; i < 3; i++) t[i] = Ri[i * 3] * T.v[0] + Ri[i * 3 + 1] * T.v[1] + Ri[i * 3 + 2] * T.v[2];

    double fc_new = DBL_MAX, ccx[2] = {0, 0}, ccy[2] = {0, 0};
    for (int k = 0; k < 2; k++) {
        const Matd &A = k == 0 ? M1 : M2;
        const double dk1 = (k == 0 ? D1 : D2).v[0];
        double fc = A.at(idx ^ 1, idx ^ 1);
        if (dk1 < 0) fc *= 1 + dk1 * ((double)nx * nx + (double)ny * ny) / (4 * fc * fc);
        fc_new = fc < fc_new ? fc : fc_new;
    }
    for (int k = 0; k < 2; k++) {
        const Matd &A = k == 0 ? M1 : M2, &Dk = k == 0 ? D1 : D2, &Rk = k == 0 ? R1 : R2;
        double ax = 0, ay = 0;
        for (int i = 0; i < 4; i++) {
            const int j = (i < 2) ? 0 : 1;
            float px, py;
            undistort_corner((float)((i % 2) * (nx - 1)), (float)(j * (ny - 1)), A, Dk, px, py);
            const double X = Rk.v[0] * px + Rk.v[1] * py + Rk.v[2], Y = Rk.v[3] * px + Rk.v[4] * py + Rk.v[5];
            const double Z = Rk.v[6] * px + Rk.v[7] * py + Rk.v[8];
            ax += (double)(float)(fc_new * X / Z);            // cvProjectPoints2 into a CV_32FC2 matrix, cc = 0
            ay += (double)(float)(fc_new * Y / Z);
        }
        ccx[k] = (nx - 1) / 2 - ax / 4;                      // integer division of (nx-1)/2 as in the source
        ccy[k] = (ny - 1) / 2 - ay / 4;
    }
    if (idx == 0) ccy[0] = ccy[1] = (ccy[0] + ccy[1]) * 0.5;  // flags = 0: horizontal stereo keeps separate cx
    else ccx[0] = ccx[1] = (ccx[0] + ccx[1]) * 0.5;
    P1.rows = P2.rows = 3; P1.cols = P2.cols = 4; P1.v.assign(12, 0.0); P2.v.assign(12, 0.0);
    P1.at(0, 0) = P1.at(1, 1) = fc_new; P1.at(0, 2) = ccx[0]; P1.at(1, 2) = ccy[0]; P1.at(2, 2) = 1;
    P2 = P1; P2.at(0, 2) = ccx[1]; P2.at(1, 2) = ccy[1]; P2.at(idx, 3) = t[idx] * fc_new;
    // alpha = -1 -> no zoom (s = 1), newImageSize = imageSize
    Q.rows = Q.cols = 4; Q.v.assign(16, 0.0);
    Q.at(0, 0) = 1; Q.at(0, 3) = -ccx[0]; Q.at(1, 1) = 1; Q.at(1, 3) = -ccy[0]; Q.at(2, 3) = fc_new;
    Q.at(3, 2) = -1. / t[idx];
    Q.at(3, 3) = (idx == 0 ? ccx[0] - ccx[1] : ccy[0] - ccy[1]) / t[idx];
}

void initUndistortRectifyMap(const Matd &M, const Matd &D, const Matd &R, const Matd &P, int W, int H,
                             std::vector<int16_t> &map_xy, std::vector<uint16_t> &map_frac)
{
    double A[9], ir[9];
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += P.at(r, k) * R.at(k, c);
            A[r * 3 + c] = s;
        }
    const double det = A[0] * (A[4] * A[8] - A[5] * A[7]) - A[1] * (A[3] * A[8] - A[5] * A[6]) + A[2] * (A[3] * A[7] - A[4] * A[6]);
    const double d = 1. / det;
    ir[0] = (A[4] * A[8] - A[5] * A[7]) * d; ir[1] = (A[2] * A[7] - A[1] * A[8]) * d; ir[2] = (A[1] * A[5] - A[2] * A[4]) * d;
    ir[3] = (A[5] * A[6] - A[3] * A[8]) * d; ir[4] = (A[0] * A[8] - A[2] * A[6]) * d; ir[5] = (A[2] * A[3] - A[0] * A[5]) * d;
    ir[6] = (A[3] * A[7] - A[4] * A[6]) * d; ir[7] = (A[1] * A[6] - A[0] * A[7]) * d; ir[8] = (A[0] * A[4] - A[1] * A[3]) * d;
    const double u0 = M.at(0, 2), v0 = M.at(1, 2), fx = M.at(0, 0), fy = M.at(1, 1);
    const double k1 = D.v[0], k2 = D.v[1], p1 = D.v[2], p2 = D.v[3], k3 = D.v.size() > 4 ? D.v[4] : 0;
    map_xy.resize((size_t)W * H * 2);
    map_frac.resize((size_t)W * H);
    for (int i = 0; i < H; i++) {
        double _x = i * ir[1] + ir[2], _y = i * ir[4] + ir[5], _w = i * ir[7] + ir[8];
        for (int j = 0; j < W; j++, _x += ir[0], _y += ir[3], _w += ir[6]) {
            const double iw = 1. / _w, x = _x * iw, y = _y * iw;
            const double x2 = x * x, y2 = y * y, r2 = x2 + y2, _2xy = 2 * x * y;
            const double kr = 1 + ((k3 * r2 + k2) * r2 + k1) * r2;
            const double u = fx * (x * kr + p1 * _2xy + p2 * (r2 + 2 * x2)) + u0;
            const double v = fy * (y * kr + p1 * (r2 + 2 * y2) + p2 * _2xy) + v0;
            const long iu = lrint(u * 32), iv = lrint(v * 32);           // cvRound: half to even
            const size_t m = (size_t)i * W + j;
            map_xy[2 * m] = (int16_t)(iu >> 5);
            map_xy[2 * m + 1] = (int16_t)(iv >> 5);
            map_frac[m] = (uint16_t)((iv & 31) * 32 + (iu & 31));
        }
    }
}

bool stereoRect::upload(slr_ctx *ctx)
{
    if (map11.empty() || map21.empty()) return false;
    return slr_set_rectify_maps(ctx, 0, map11.data(), map12.data(), w, h, SLR_MEM_HOST) == SLR_OK &&
           slr_set_rectify_maps(ctx, 1, map21.data(), map22.data(), w, h, SLR_MEM_HOST) == SLR_OK;
}

// the same maps generated on the device (slr_init_rectify_maps): no 2 x 74 MB upload; bit-identical to upload()
bool stereoRect::uploadFromCalibration(slr_ctx *ctx)
{
    if (M1.empty() || M2.empty() || R1.empty() || R2.empty() || P1.empty() || P2.empty()) return false;
    double d1[5] = {0, 0, 0, 0, 0}, d2[5] = {0, 0, 0, 0, 0};
    for (size_t i = 0; i < 5 && i < D1.v.size(); i++) d1[i] = D1.v[i];
    for (size_t i = 0; i < 5 && i < D2.v.size(); i++) d2[i] = D2.v[i];
    return slr_init_rectify_maps(ctx, 0, M1.v.data(), d1, R1.v.data(), P1.v.data(), w, h) == SLR_OK &&
           slr_init_rectify_maps(ctx, 1, M2.v.data(), d2, R2.v.data(), P2.v.data(), w, h) == SLR_OK;
}

bool stereoRect::doStereoRectify(slr_ctx *ctx, Image8 &img, bool isleft)
{
    if (img.empty() || img.w != w || img.h != h) return false;
    Image8 out;
    out.w = w; out.h = h; out.d.resize(img.d.size());
    if (slr_remap_u8(ctx, isleft ? 0 : 1, img.d.data(), w, out.d.data(), w, w, h, SLR_MEM_HOST) != SLR_OK) return false;
    img = out;
    return true;
}


}  // namespace duke
