// duke.cpp -- see duke.hpp.  Host-side C++ (g++), links libslr_hip.so (C ABI) and zlib.  Nothing in this file
// touches a pixel of the hot path: decode / rectify / match / triangulate are single calls into the HIP library.
#include "duke.hpp"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <zlib.h>

#include <fstream>
#include <iostream>
#include <sstream>
#include <thread>
#include <atomic>

namespace duke {

static void warn(const std::string &title, const std::string &msg)      // QMessageBox::warning stand-in
{
    std::cerr << "[" << title << "] " << msg << std::endl;
}

// ------------------------------------------------------------------------------------------------------------
// image I/O (cv::imread(path, 0) / cv::imwrite)
// ------------------------------------------------------------------------------------------------------------
static bool read_file(const std::string &path, std::vector<uint8_t> &out)
{
    std::ifstream f(path.c_str(), std::ios::binary);
    if (!f) return false;
    f.seekg(0, std::ios::end);
    std::streamoff n = f.tellg();
    f.seekg(0);
    out.resize((size_t)n);
    if (n) f.read((char *)out.data(), n);
    return (bool)f;
}

static uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | (p[1] << 16) | (p[2] << 8) | p[3]; }

static Image8 decode_pgm(const std::vector<uint8_t> &buf)
{
    Image8 img;
    size_t pos = 2;
    int vals[3], got = 0;
    while (got < 3 && pos < buf.size()) {
        while (pos < buf.size() && (buf[pos] == ' ' || buf[pos] == '\n' || buf[pos] == '\r' || buf[pos] == '\t')) pos++;
        if (pos < buf.size() && buf[pos] == '#') { while (pos < buf.size() && buf[pos] != '\n') pos++; continue; }
        int v = 0, any = 0;
        while (pos < buf.size() && buf[pos] >= '0' && buf[pos] <= '9') { v = v * 10 + (buf[pos] - '0'); pos++; any = 1; }
        if (!any) return img;
        vals[got++] = v;
    }
    pos++;                                               // single whitespace after maxval
    if (got < 3 || vals[2] > 255 || pos + (size_t)vals[0] * vals[1] > buf.size()) return img;
    img.w = vals[0]; img.h = vals[1];
    img.d.assign(buf.begin() + pos, buf.begin() + pos + (size_t)img.w * img.h);
    return img;
}

static int paeth(int a, int b, int c)
{
    int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

static Image8 decode_png(const std::vector<uint8_t> &buf)
{
    Image8 img;
    if (buf.size() < 33) return img;
    size_t pos = 8;
    int w = 0, h = 0, depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat;
    while (pos + 12 <= buf.size()) {
        uint32_t len = be32(&buf[pos]);
        const uint8_t *type = &buf[pos + 4];
        if (pos + 12 + len > buf.size()) return img;
        if (!memcmp(type, "IHDR", 4)) {
            w = (int)be32(&buf[pos + 8]); h = (int)be32(&buf[pos + 12]);
            depth = buf[pos + 16]; ctype = buf[pos + 17]; interlace = buf[pos + 20];
        } else if (!memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), buf.begin() + pos + 8, buf.begin() + pos + 8 + len);
        } else if (!memcmp(type, "IEND", 4)) break;
        pos += 12 + len;
    }
    int ch = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (!w || !h || !ch || interlace || (depth != 8 && depth != 16)) return img;
    const int bpp = ch * depth / 8;
    const size_t stride = (size_t)w * bpp;
    std::vector<uint8_t> raw((stride + 1) * h);
    uLongf rawlen = (uLongf)raw.size();
    if (uncompress(raw.data(), &rawlen, idat.data(), (uLong)idat.size()) != Z_OK || rawlen != raw.size()) return img;
    std::vector<uint8_t> cur(stride), prev(stride, 0);
    img.w = w; img.h = h; img.d.resize((size_t)w * h);
    for (int y = 0; y < h; y++) {
        const uint8_t *src = &raw[(stride + 1) * y];
        const int ft = src[0];
        for (size_t i = 0; i < stride; i++) {
            int a = i >= (size_t)bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= (size_t)bpp ? prev[i - bpp] : 0, x = src[1 + i];
            switch (ft) { case 1: x += a; break; case 2: x += b; break; case 3: x += (a + b) >> 1; break; case 4: x += paeth(a, b, c); break; default: break; }
            cur[i] = (uint8_t)x;
        }
        const int bs = depth / 8;                            // 16-bit: keep the high byte
        for (int x = 0; x < w; x++) {
            const uint8_t *p = &cur[(size_t)x * bpp];
            int g;
            if (ch <= 2) g = p[0];
            else g = (p[0] * 4899 + p[bs] * 9617 + p[2 * bs] * 1868 + 8192) >> 14;     // OpenCV RGB2GRAY fixed point
            img.d[(size_t)y * w + x] = (uint8_t)g;
        }
        prev.swap(cur);
    }
    return img;
}

Image8 imread_gray(const std::string &path)
{
    std::vector<uint8_t> buf;
    if (!read_file(path, buf) || buf.size() < 4) return Image8();
    if (buf[0] == 'P' && buf[1] == '5') return decode_pgm(buf);
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (buf.size() >= 8 && !memcmp(buf.data(), sig, 8)) return decode_png(buf);
    return Image8();
}

bool imwrite_pgm(const std::string &path, const Image8 &img)
{
    std::ofstream f(path.c_str(), std::ios::binary);
    if (!f) return false;
    f << "P5\n" << img.w << " " << img.h << "\n255\n";
    f.write((const char *)img.d.data(), (std::streamsize)img.d.size());
    return (bool)f;
}

static void png_chunk(std::ofstream &f, const char *type, const std::vector<uint8_t> &data)
{
    uint8_t len[4] = {(uint8_t)(data.size() >> 24), (uint8_t)(data.size() >> 16), (uint8_t)(data.size() >> 8), (uint8_t)data.size()};
    f.write((const char *)len, 4);
    f.write(type, 4);
    if (!data.empty()) f.write((const char *)data.data(), (std::streamsize)data.size());
    uLong crc = crc32(0L, (const Bytef *)type, 4);
    if (!data.empty()) crc = crc32(crc, data.data(), (uInt)data.size());
    uint8_t c[4] = {(uint8_t)(crc >> 24), (uint8_t)(crc >> 16), (uint8_t)(crc >> 8), (uint8_t)crc};
    f.write((const char *)c, 4);
}

bool imwrite_png(const std::string &path, const Image8 &img)
{
    std::ofstream f(path.c_str(), std::ios::binary);
    if (!f) return false;
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    f.write((const char *)sig, 8);
    std::vector<uint8_t> ihdr(13, 0);
    for (int i = 0; i < 4; i++) { ihdr[i] = (uint8_t)(img.w >> (24 - 8 * i)); ihdr[4 + i] = (uint8_t)(img.h >> (24 - 8 * i)); }
    ihdr[8] = 8;                                             // depth 8, colour type 0 (grey)
    png_chunk(f, "IHDR", ihdr);
    std::vector<uint8_t> raw((size_t)(img.w + 1) * img.h);
    for (int y = 0; y < img.h; y++) {
        raw[(size_t)(img.w + 1) * y] = 0;
        memcpy(&raw[(size_t)(img.w + 1) * y + 1], &img.d[(size_t)img.w * y], img.w);
    }
    uLongf clen = compressBound((uLong)raw.size());
    std::vector<uint8_t> comp(clen);
    if (compress2(comp.data(), &clen, raw.data(), (uLong)raw.size(), 6) != Z_OK) return false;
    comp.resize(clen);
    png_chunk(f, "IDAT", comp);
    png_chunk(f, "IEND", std::vector<uint8_t>());
    return (bool)f;
}

bool exportMat(const std::string &path, const double *m, int rows, int cols)
{
    std::ofstream out(path.c_str());
    if (!out) return false;
    for (int r = 0; r < rows; r++) {
        for (int c = 0; c < cols; c++) out << m[r * cols + c] << "\t";
        out << "\n";
    }
    return true;
}

// ------------------------------------------------------------------------------------------------------------
// VirtualCamera (virtualcamera.cpp:25-88)
// ------------------------------------------------------------------------------------------------------------
VirtualCamera::VirtualCamera() { fc[0] = fc[1] = cc[0] = cc[1] = 0; }

int VirtualCamera::loadMatrix(Matf &matrix, int rows, int cols, const std::string &file)
{
    std::ifstream in1(file.c_str());
    if (!in1) return -1;
    matrix.rows = rows; matrix.cols = cols;
    matrix.v.assign((size_t)rows * cols, 0.0f);
    for (int i = 0; i < rows; i++)
        for (int j = 0; j < cols; j++) {
            float val = 0;
            in1 >> val;                                      // parsed through float (Q14)
            matrix.at(i, j) = val;
        }
    return 1;
}
void VirtualCamera::loadDistortion(const std::string &path) { loadMatrix(distortion, 5, 1, path); }
bool VirtualCamera::loadCameraMatrix(const std::string &path)
{
    std::ifstream probe(path.c_str());
    if (!probe) { warn("Matrix not found", "File: '" + path + "' need to be added."); return false; }
    Matf cam;
    loadMatrix(cam, 3, 3, path);
    cc[0] = cam.at(0, 2); cc[1] = cam.at(1, 2);
    fc[0] = cam.at(0, 0); fc[1] = cam.at(1, 1);
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

// ------------------------------------------------------------------------------------------------------------
// stereoRect (stereorect.cpp) -- cv::stereoRectify / cv::initUndistortRectifyMap restated from OpenCV 2.4's published
// algorithm (calib3d/calibration.cpp cvStereoRectify, imgproc/undistort.cpp).  PARITY UNPINNED: no OpenCV here to
// compare with; differences vs 2.4.9: Rodrigues(matrix) skips the SVD re-orthonormalisation of R.
// ------------------------------------------------------------------------------------------------------------
stereoRect::stereoRect(const std::string &projectPath, int width, int height) : w(width), h(height), ppath(projectPath) {}

void stereoRect::loadMatrix(Matd &matrix, int rows, int cols, const std::string &file)
{
    std::ifstream in1(file.c_str());
    if (!in1) return;
    matrix.rows = rows; matrix.cols = cols;
    matrix.v.assign((size_t)rows * cols, 0.0);
    for (int i = 0; i < rows; i++)
        for (int j = 0; j < cols; j++) {
            float val = 0;
            in1 >> val;                                      // float, then stored as f64 (stereorect.cpp:57-59)
            matrix.at(i, j) = val;
        }
}

void stereoRect::getParameters()
{
    loadMatrix(M1, 3, 3, ppath + "/calib/left/cam_stereo.txt");
    loadMatrix(D1, 5, 1, ppath + "/calib/left/distortion_stereo.txt");
    loadMatrix(M2, 3, 3, ppath + "/calib/right/cam_stereo.txt");
    loadMatrix(D2, 5, 1, ppath + "/calib/right/distortion_stereo.txt");
    loadMatrix(R, 3, 3, ppath + "/calib/R_stereo.txt");
    loadMatrix(T, 3, 1, ppath + "/calib/T_stereo.txt");
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

static void rodrigues_mat2vec(const double Rm[9], double r[3])
{
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
    initUndistortRectifyMap(M1, D1, R1, P1, w, h, map11, map12);
    initUndistortRectifyMap(M2, D2, R2, P2, w, h, map21, map22);
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

// ------------------------------------------------------------------------------------------------------------
// PointCloudImage (pointcloudimage.cpp:3-123)
// ------------------------------------------------------------------------------------------------------------
PointCloudImage::PointCloudImage(int imageW, int imageH, bool colorFlag) : w(imageW), h(imageH)
{
    points.assign((size_t)w * h * 3, 0.0f);
    if (colorFlag) color.assign((size_t)w * h * 3, 0);
    numOfPointsForPixel.assign((size_t)w * h, 0);
}
bool PointCloudImage::setPoint(int i_w, int j_h, Point3f p)
{
    if (i_w >= w || j_h >= h) return false;
    float *d = &points[((size_t)j_h * w + i_w) * 3];
    d[0] = p.x; d[1] = p.y; d[2] = p.z;
    numOfPointsForPixel[(size_t)j_h * w + i_w] = 1;
    return true;
}
bool PointCloudImage::setPoint(int i_w, int j_h, Point3f p, const int cg[3])
{
    if (i_w >= w || j_h >= h) return false;
    setPoint(i_w, j_h, p);
    if (!color.empty())
        for (int k = 0; k < 3; k++) color[((size_t)j_h * w + i_w) * 3 + k] = (uint8_t)(cg[k] < 0 ? 0 : cg[k] > 255 ? 255 : cg[k]);
    return true;
}
bool PointCloudImage::getPoint(int i_w, int j_h, Point3f &out) const
{
    if (i_w >= w || j_h >= h) return false;
    const uint8_t num = numOfPointsForPixel[(size_t)j_h * w + i_w];
    if (num == 0) return false;
    const float *s = &points[((size_t)j_h * w + i_w) * 3];
    const double dn = (double)(float)num;                    // Vec3d / float (pointcloudimage.cpp:62)
    out.x = (float)((double)s[0] / dn); out.y = (float)((double)s[1] / dn); out.z = (float)((double)s[2] / dn);
    return true;
}
bool PointCloudImage::getPoint(int i_w, int j_h, Point3f &out, int colorOut[3]) const
{
    if (!getPoint(i_w, j_h, out)) return false;
    const uint8_t num = numOfPointsForPixel[(size_t)j_h * w + i_w];
    if (!color.empty())
        for (int k = 0; k < 3; k++) colorOut[k] = (int)lrint((double)color[((size_t)j_h * w + i_w) * 3 + k] / (double)(float)num);
    else { colorOut[0] = colorOut[1] = colorOut[2] = 100; }  // pointcloudimage.cpp:49 "(100,100,100)" comma expression -> x=100
    return true;
}
bool PointCloudImage::addPoint(int i_w, int j_h, Point3f p)
{
    if (i_w >= w || j_h >= h) return false;
    const size_t o = (size_t)j_h * w + i_w;
    const uint8_t num = numOfPointsForPixel[o];
    if (num == 0) return setPoint(i_w, j_h, p);
    points[o * 3] = p.x + points[o * 3]; points[o * 3 + 1] = p.y + points[o * 3 + 1]; points[o * 3 + 2] = p.z + points[o * 3 + 2];
    numOfPointsForPixel[o] = (uint8_t)(num + 1);
    return true;
}
bool PointCloudImage::addPoint(int i_w, int j_h, Point3f p, const int cg[3])
{
    if (i_w >= w || j_h >= h) return false;
    const size_t o = (size_t)j_h * w + i_w;
    if (numOfPointsForPixel[o] == 0) return setPoint(i_w, j_h, p, cg);
    addPoint(i_w, j_h, p);
    if (color.empty()) return false;
    for (int k = 0; k < 3; k++) { int v = cg[k] + color[o * 3 + k]; color[o * 3 + k] = (uint8_t)(v > 255 ? 255 : v); }
    return true;
}
void PointCloudImage::exportXYZ(const char *path, bool exportOffPixels, bool colorFlag) const
{
    std::ofstream out(path);
    for (int i = 0; i < w; i++)
        for (int j = 0; j < h; j++) {
            const uint8_t num = numOfPointsForPixel[(size_t)j * w + i];
            if (!exportOffPixels && num == 0) continue;
            Point3f p; int c[3] = {0, 0, 0};
            getPoint(i, j, p, c);
            if (exportOffPixels && num == 0) { p = Point3f(); c[0] = c[1] = c[2] = 0; }
            out << p.x << " " << p.y << " " << p.z;
            if (colorFlag && !color.empty()) out << " " << c[2] << " " << c[1] << " " << c[0] << "\n";
            else out << "\n";
        }
}

// ------------------------------------------------------------------------------------------------------------
// encoders (graycodes.cpp:22-128, multifrequency.cpp:14-33)
// ------------------------------------------------------------------------------------------------------------
GrayCodes::GrayCodes(int scanW, int scanH, bool useepi) : useEpi(useepi), height(scanH), width(scanW) { calNumOfImgs(); }
void GrayCodes::calNumOfImgs()
{
    numOfColImgs = (int)ceil(log((double)width) / log(2.0));
    numOfRowImgs = (int)ceil(log((double)height) / log(2.0));
    numOfImgs = useEpi ? 2 + 2 * numOfColImgs : 2 + 2 * numOfColImgs + 2 * numOfRowImgs;
}
void GrayCodes::generateGrays()
{
    grayCodes.assign(numOfImgs, Image8());
    for (auto &g : grayCodes) { g.w = width; g.h = height; g.d.assign((size_t)width * height, 0); }
    std::fill(grayCodes[0].d.begin(), grayCodes[0].d.end(), 255);
    for (int j = 0; j < width; j++) {
        int num = j, prevRem = j % 2;
        for (int k = 0; k < numOfColImgs; k++) {
            num /= 2;
            const int rem = num % 2;
            const uint8_t a = (rem != prevRem) ? 255 : 0, b = a ? 0 : 255;
            Image8 &pa = grayCodes[2 * numOfColImgs - 2 * k], &pb = grayCodes[2 * numOfColImgs - 2 * k + 1];
            for (int i = 0; i < height; i++) { pa.d[(size_t)i * width + j] = a; pb.d[(size_t)i * width + j] = b; }
            prevRem = rem;
        }
    }
    if (!useEpi)
        for (int i = 0; i < height; i++) {
            int num = i, prevRem = i % 2;
            for (int k = 0; k < numOfRowImgs; k++) {
                num /= 2;
                const int rem = num % 2;
                const uint8_t a = (rem != prevRem) ? 255 : 0, b = a ? 0 : 255;
                Image8 &pa = grayCodes[2 * numOfRowImgs - 2 * k + 2 * numOfColImgs];
                Image8 &pb = grayCodes[2 * numOfRowImgs - 2 * k + 2 * numOfColImgs + 1];
                for (int j = 0; j < width; j++) { pa.d[(size_t)i * width + j] = a; pb.d[(size_t)i * width + j] = b; }
                prevRem = rem;
            }
        }
}
int GrayCodes::grayToDec(const std::vector<bool> &gray)
{
    int dec = 0;
    bool tmp = gray[0];
    const int n = (int)gray.size();
    if (tmp) dec += 1 << (n - 1);
    for (int i = 1; i < n; i++) {
        tmp = (tmp != gray[i]);
        if (tmp) dec += 1 << (n - i - 1);
    }
    return dec;
}

MultiFrequency::MultiFrequency(int projwidth, int projheight) : projW(projwidth), projH(projheight) {}
void MultiFrequency::generateMutiFreq()
{
    static const int frequency[3] = {70, 64, 59};
    const double PI = 3.1416;                               // multifrequency.h:5
    for (int i = 0; i < 14; i++) { MultiFreqImages[i].w = projW; MultiFreqImages[i].h = projH; MultiFreqImages[i].d.assign((size_t)projW * projH, i == 0 ? 255 : 0); }
    for (int f = 0; f < 3; f++)
        for (int phi = 0; phi < 4; phi++) {
            Image8 &t = MultiFreqImages[4 * f + phi + 2];
            for (int x = 0; x < projW; x++) {
                const float v = 135 + 79 * cosf((float)(PI * 2 * (double)x * (double)frequency[f] / (double)projW + PI * (double)phi / 2));
                const uint8_t g = (uint8_t)v;
                for (int y = 0; y < projH; y++) t.d[(size_t)y * projW + x] = g;
            }
        }
}

// ------------------------------------------------------------------------------------------------------------
// Reconstruct / MFReconstruct: the reference's call sequence; the per-pixel work is one C-ABI call
// ------------------------------------------------------------------------------------------------------------
static bool load_transfer(const std::string &savePath, int scanSN, slr_calib &cal)
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

// scan-directory reader (reconstruct.cpp:158-164, mfreconstruct.cpp:119-125): <folder><prefix><i><suffix>, 8-bit grey.
// PNG inflate runs at ~200 MB/s per core, i.e. ~60 ms per 4096x3000 plane and 1.7 s for a 28-image stereo frame if
// done one file after the other as the reference does -- three orders of magnitude more than the GPU path.  The files of
// a stack are therefore decoded by a pool of host threads (one file per task, SLR_LOADER_THREADS or the core count).
static bool load_stack(const std::string &folder, const std::string &prefix, const std::string &suffix, int n, int w, int h,
                       std::vector<Image8> &imgs, std::string &err)
{
    imgs.clear();
    imgs.resize(n > 0 ? n : 0);
    std::vector<std::string> errs(n > 0 ? n : 0);
    std::atomic<int> next(0);
    auto work = [&]() {
        for (int i = next.fetch_add(1); i < n; i = next.fetch_add(1)) {
            std::ostringstream p;
            p << folder << prefix << i;
            Image8 img = imread_gray(p.str() + suffix);
            if (img.empty() && suffix != ".pgm") img = imread_gray(p.str() + ".pgm");
            if (img.empty()) { errs[i] = "Scan Images not found! (" + p.str() + suffix + ")"; continue; }
            if (img.w != w || img.h != h) { errs[i] = "image size differs from the configured camera size"; continue; }
            imgs[i] = std::move(img);
        }
    };
    unsigned nt = std::thread::hardware_concurrency();
    if (const char *e = getenv("SLR_LOADER_THREADS")) nt = (unsigned)atoi(e);
    if (nt < 1) nt = 1;
    if (nt > (unsigned)n) nt = (unsigned)(n > 0 ? n : 1);
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < nt; t++) pool.emplace_back(work);
    work();
    for (auto &t : pool) t.join();
    for (int i = 0; i < n; i++)                              // first failing file in index order, like the sequential loop
        if (!errs[i].empty()) { err = errs[i]; warn("Load Images", err); imgs.clear(); return false; }
    return true;
}

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
    if (EPI) { delete sr; sr = new stereoRect(savePath_, camw, camh); sr->getParameters(); }
    std::ostringstream sn;
    sn << scanSN;
    scanFolder[0] = savePath + "/scan/left/";  imgPrefix[0] = sn.str() + "/L";
    scanFolder[1] = savePath + "/scan/right/"; imgPrefix[1] = sn.str() + "/R";
}

bool Reconstruct::loadCameras()
{
    bool loaded = false;
    for (int i = 0; i < 2; i++) {
        loaded = cameras[i].loadCameraMatrix(calibFolder[i] + "cam_matrix.txt");
        if (!loaded) break;
        cameras[i].loadDistortion(calibFolder[i] + "cam_distortion.txt");
        cameras[i].loadRotationMatrix(calibFolder[i] + "cam_rotation_matrix.txt");
        cameras[i].loadTranslationVector(calibFolder[i] + "cam_trans_vectror.txt");
        cameras[i].loadFundamentalMatrix(savePath_ + "/calib/fundamental_stereo.txt");
        cameras[i].loadHomoMatrix(savePath_ + "/calib/H1_mat.txt", 1);
        cameras[i].loadHomoMatrix(savePath_ + "/calib/H2_mat.txt", 2);
        cameras[i].height = 0; cameras[i].width = 0;
    }
    return loaded;
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

bool Reconstruct::loadCamImgs(int cam, std::vector<Image8> &imgs)
{
    return load_stack(scanFolder[cam], imgPrefix[cam], imgSuffix, numberOfImgs, cameraWidth, cameraHeight, imgs, lastError);
}

static bool ensure_ctx(slr_ctx *&ctx, std::string &err)
{
    if (ctx) return true;
    const int st = slr_create(0, &ctx);
    if (st != SLR_OK) { err = std::string("no GPU context: ") + slr_status_string(st); warn("Reconstruct", err); return false; }
    return true;
}

bool Reconstruct::runReconstruction_GE()
{
    GrayCodes grays(scan_w, scan_h, true);
    numOfColBits = grays.getNumOfColBits();
    numberOfImgs = grays.getNumOfImgs();
    if (!ensure_ctx(ctx, lastError)) return false;
    std::vector<Image8> imgs[2];
    for (int i = 0; i < 2; i++)
        if (!loadCamImgs(i, imgs[i])) return false;
    if (!sr) { lastError = "getParameters not called"; return false; }
    sr->calParameters();
    slr_calib cal;
    if (!fillCalib(cal) || slr_set_calibration(ctx, &cal) != SLR_OK || !(sr->uploadFromCalibration(ctx) || sr->upload(ctx))) {
        lastError = "calibration incomplete"; warn("Reconstruct", lastError); return false;
    }
    const int W = cameraWidth, H = cameraHeight;
    std::vector<const uint8_t *> pl[2];
    for (int c = 0; c < 2; c++) for (auto &im : imgs[c]) pl[c].push_back(im.d.data());
    std::vector<float> xyz((size_t)W * H * 3);
    std::vector<uint8_t> has((size_t)W * H), col(haveColor ? (size_t)W * H : 0);
    if (slr_reconstruct_ge(ctx, pl[0].data(), pl[1].data(), numOfColBits, W, W, H, blackThreshold, whiteThreshold, scan_w, 1,
                           haveColor ? 1 : 0, xyz.data(), has.data(), haveColor ? col.data() : nullptr, SLR_MEM_HOST) != SLR_OK) {
        lastError = slr_last_error(ctx); warn("Reconstruct", lastError); return false;
    }
    delete points3DProjView;
    points3DProjView = new PointCloudImage(scan_w, scan_h, haveColor);
    std::vector<uint8_t> pcol(haveColor ? (size_t)scan_w * scan_h : 0);
    if (slr_pointcloud_from_grid(ctx, xyz.data(), has.data(), haveColor ? col.data() : nullptr, W, H, scan_w, scan_h,
                                 points3DProjView->points.data(), points3DProjView->numOfPointsForPixel.data(),
                                 haveColor ? pcol.data() : nullptr, SLR_MEM_HOST) != SLR_OK) { lastError = slr_last_error(ctx); return false; }
    if (haveColor)
        for (size_t i = 0; i < pcol.size(); i++) points3DProjView->color[3 * i] = points3DProjView->color[3 * i + 1] = points3DProjView->color[3 * i + 2] = pcol[i];
    return true;
}

bool Reconstruct::runReconstruction()
{
    GrayCodes grays(scan_w, scan_h, false);
    numOfColBits = grays.getNumOfColBits();
    numOfRowBits = grays.getNumOfRowBits();
    numberOfImgs = grays.getNumOfImgs();
    if (!ensure_ctx(ctx, lastError)) return false;
    std::vector<Image8> imgs[2];
    for (int i = 0; i < 2; i++)
        if (!loadCamImgs(i, imgs[i])) return false;
    slr_calib cal;
    if (!fillCalib(cal) || slr_set_calibration(ctx, &cal) != SLR_OK) { lastError = "calibration incomplete"; return false; }
    const int W = cameraWidth, H = cameraHeight;
    std::vector<const uint8_t *> pl[2];
    for (int c = 0; c < 2; c++) for (auto &im : imgs[c]) pl[c].push_back(im.d.data());
    delete points3DProjView;
    points3DProjView = new PointCloudImage(scan_w, scan_h, haveColor);
    if (slr_reconstruct_gray(ctx, pl[0].data(), pl[1].data(), numOfColBits, numOfRowBits, W, W, H, blackThreshold, whiteThreshold,
                             scan_w, scan_h, points3DProjView->points.data(), points3DProjView->numOfPointsForPixel.data(),
                             SLR_MEM_HOST) != SLR_OK) { lastError = slr_last_error(ctx); warn("Reconstruct", lastError); return false; }
    return true;
}

MFReconstruct::MFReconstruct() { cameras = new VirtualCamera[2]; points3DProjView = nullptr; }
MFReconstruct::~MFReconstruct()
{
    delete points3DProjView;
    delete sr;
    delete[] cameras;
    if (ctx) slr_destroy(ctx);
}

void MFReconstruct::getParameters(int scansn, int scanw, int scanh, int camw, int camh, int blackt, int whitet,
                                  const std::string &savePath)
{
    scanSN = scansn; scan_w = scanw; scan_h = scanh; cameraWidth = camw; cameraHeight = camh;
    blackThreshold = blackt; whiteThreshold = whitet; savePath_ = savePath;
    delete sr;
    sr = new stereoRect(savePath, camw, camh);
    sr->getParameters();
    std::ostringstream sn;
    sn << scanSN;
    scanFolder[0] = savePath + "/scan/left/";  imgPrefix[0] = sn.str() + "/L"; calibFolder[0] = savePath + "/calib/left/";
    scanFolder[1] = savePath + "/scan/right/"; imgPrefix[1] = sn.str() + "/R"; calibFolder[1] = savePath + "/calib/right/";
    camerasLoaded = loadCameras();
    if (!camerasLoaded) warn("Get Param", "Load Calibration files failed.");
}

bool MFReconstruct::loadCameras()
{
    bool loaded = false;
    for (int i = 0; i < 2; i++) {
        loaded = cameras[i].loadCameraMatrix(calibFolder[i] + "cam_matrix.txt");
        if (!loaded) break;
        cameras[i].loadDistortion(calibFolder[i] + "cam_distortion.txt");
        cameras[i].loadRotationMatrix(calibFolder[i] + "cam_rotation_matrix.txt");
        cameras[i].loadTranslationVector(calibFolder[i] + "cam_trans_vectror.txt");
        cameras[i].loadFundamentalMatrix(savePath_ + "/calib/fundamental_stereo.txt");
        cameras[i].height = cameraHeight; cameras[i].width = cameraWidth;
    }
    return loaded;
}

bool MFReconstruct::loadCamImgs(int cam, std::vector<Image8> &imgs)
{
    return load_stack(scanFolder[cam], imgPrefix[cam], imgSuffix, numberOfImgs, cameraWidth, cameraHeight, imgs, lastError);
}

bool MFReconstruct::runReconstruction()
{
    if (!camerasLoaded || !sr) { lastError = "calibration not loaded"; return false; }
    if (!ensure_ctx(ctx, lastError)) return false;
    std::vector<Image8> imgs[2];
    for (int i = 0; i < 2; i++)
        if (!loadCamImgs(i, imgs[i])) return false;
    sr->calParameters();
    slr_calib cal;
    memset(&cal, 0, sizeof cal);
    cameras[0].fill(cal.cam[0]);
    cameras[1].fill(cal.cam[1]);
    if (sr->Q.empty()) { lastError = "stereo calibration files missing"; warn("Reconstruct", lastError); return false; }
    memcpy(cal.Q, sr->Q.v.data(), sizeof(double) * 16);
    if (!load_transfer(savePath_, scanSN, cal) || slr_set_calibration(ctx, &cal) != SLR_OK || !(sr->uploadFromCalibration(ctx) || sr->upload(ctx))) {
        lastError = "calibration incomplete"; warn("Reconstruct", lastError); return false;
    }
    const int W = cameraWidth, H = cameraHeight;
    const uint8_t *pl[2][SLR_MF_PLANES];
    for (int c = 0; c < 2; c++) for (int i = 0; i < SLR_MF_PLANES; i++) pl[c][i] = imgs[c][i].d.data();
    std::vector<float> xyz((size_t)W * H * 3);
    std::vector<uint8_t> has((size_t)W * H);
    if (slr_reconstruct_mf(ctx, pl[0], pl[1], W, W, H, blackThreshold, 1, xyz.data(), has.data(), SLR_MEM_HOST) != SLR_OK) {
        lastError = slr_last_error(ctx); warn("Reconstruct", lastError); return false;
    }
    delete points3DProjView;
    points3DProjView = new PointCloudImage(scan_w, scan_h, false);
    if (slr_pointcloud_from_grid(ctx, xyz.data(), has.data(), nullptr, W, H, scan_w, scan_h, points3DProjView->points.data(),
                                 points3DProjView->numOfPointsForPixel.data(), nullptr, SLR_MEM_HOST) != SLR_OK) {
        lastError = slr_last_error(ctx); return false;
    }
    return true;
}

// ------------------------------------------------------------------------------------------------------------
// MeshCreator (meshcreator.cpp:16-172): vertex index 0 doubles as "no vertex", as in the reference
// ------------------------------------------------------------------------------------------------------------
MeshCreator::MeshCreator(PointCloudImage *in) : cloud(in), w(in->getWidth()), h(in->getHeight()) { pixelNum.assign((size_t)w * h, 0); }

void MeshCreator::exportPlyMesh(const std::string &path)
{
    std::ofstream out1(path.c_str());
    Point3f point;
    int color[3];
    int vertexCount = 0;
    for (int i = 0; i < w; i++)
        for (int j = 0; j < h; j++) {
            if (cloud->getPoint(i, j, point)) { pixelNum[access(i, j)] = vertexCount; vertexCount++; }
            else pixelNum[access(i, j)] = 0;
        }
    auto faces = [&](bool emit) {
        int n = 0;
        for (int i = 0; i < w; i++)
            for (int j = 0; j < h; j++) {
                int v1 = pixelNum[access(i, j)], v2 = (i < w - 1) ? pixelNum[access(i + 1, j)] : 0;
                int v3 = (j < h - 1) ? pixelNum[access(i, j + 1)] : 0;
                if (v1 != 0 && v2 != 0 && v3 != 0) { n++; if (emit) out1 << "3 " << v1 << " " << v2 << " " << v3 << "\n"; }
                v3 = (j > 0 && i < w - 1) ? pixelNum[access(i + 1, j - 1)] : 0;
                if (v1 != 0 && v2 != 0 && v3 != 0) { n++; if (emit) out1 << "3 " << v1 << " " << v3 << " " << v2 << "\n"; }
            }
        return n;
    };
    const int facesCount = faces(false);
    out1 << "ply\nformat ascii 1.0\nelement vertex " << vertexCount << "\nproperty float x\nproperty float y\nproperty float z\n"
         << "property uchar red\nproperty uchar green\nproperty uchar blue\nelement face " << facesCount
         << "\nproperty list uchar int vertex_indices\nend_header\n";
    for (int i = 0; i < w; i++)
        for (int j = 0; j < h; j++)
            if (cloud->getPoint(i, j, point, color))
                out1 << point.x << " " << point.y << " " << point.z << " " << color[2] << " " << color[1] << " " << color[0] << "\n";
    faces(true);
}

void MeshCreator::exportObjMesh(const std::string &path)
{
    std::ofstream out1(path.c_str());
    Point3f point;
    int count = 1;                                           // OBJ indices start at 1 (meshcreator.cpp:21-33)
    for (int i = 0; i < w; i++)
        for (int j = 0; j < h; j++) {
            if (cloud->getPoint(i, j, point)) { pixelNum[access(i, j)] = count++; out1 << "v " << point.x << " " << point.y << " " << point.z << "\n"; }
            else pixelNum[access(i, j)] = 0;
        }
    for (int i = 0; i < w; i++)
        for (int j = 0; j < h; j++) {
            int v1 = pixelNum[access(i, j)], v2 = (i < w - 1) ? pixelNum[access(i + 1, j)] : 0;
            int v3 = (j < h - 1) ? pixelNum[access(i, j + 1)] : 0;
            if (v1 != 0 && v2 != 0 && v3 != 0) out1 << "f " << v1 << "/" << v1 << " " << v2 << "/" << v2 << " " << v3 << "/" << v3 << "\n";
            v3 = (j > 0 && i < w - 1) ? pixelNum[access(i + 1, j - 1)] : 0;
            if (v1 != 0 && v2 != 0 && v3 != 0) out1 << "f " << v1 << "/" << v1 << " " << v3 << "/" << v3 << " " << v2 << "/" << v2 << "\n";
        }
}

}  // namespace duke
