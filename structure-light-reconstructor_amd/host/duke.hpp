// duke.hpp -- dependency-free C++ host layer that mirrors the reference application's classes for the hot path
// (same class and method names, argument meaning and error behaviour), on top of the C ABI of libslr_hip.so.
// No Qt, no OpenCV: QString -> std::string, cv::Mat -> flat std::vector, QMessageBox -> message on stderr + false.
//
// Mirrors (all under /root/reference/Duke/):
//   VirtualCamera      virtualcamera.h:12-41,  virtualcamera.cpp:25-88
//   stereoRect         stereorect.h:13-29,     stereorect.cpp:3-62
//   PointCloudImage    pointcloudimage.h:8-35, pointcloudimage.cpp:3-164
//   GrayCodes          graycodes.h:14-45,      graycodes.cpp:3-138
//   MultiFrequency     multifrequency.h:11-24, multifrequency.cpp:5-39
//   Reconstruct        reconstruct.h:14-101,   reconstruct.cpp (public surface)
//   MFReconstruct      mfreconstruct.h:12-66,  mfreconstruct.cpp (public surface)
//   MeshCreator        meshcreator.h,          meshcreator.cpp:16-172
// The dense per-pixel work is NOT here: it is one slr_reconstruct_* call into the HIP library.
#pragma once

#include <stdint.h>

#include <functional>
#include <string>
#include <vector>

#include "slr.h"

namespace duke {

// ---- small containers -------------------------------------------------------------------------------------
struct Image8 {                       // cv::Mat CV_8U, row-major, pitch == w
    int w = 0, h = 0;
    std::vector<uint8_t> d;
    bool empty() const { return d.empty(); }
};
struct Matd {                         // cv::Mat CV_64F
    int rows = 0, cols = 0;
    std::vector<double> v;
    double &at(int r, int c) { return v[(size_t)r * cols + c]; }
    double at(int r, int c) const { return v[(size_t)r * cols + c]; }
    bool empty() const { return v.empty(); }
};
struct Matf {                         // cv::Mat CV_32F
    int rows = 0, cols = 0;
    std::vector<float> v;
    float &at(int r, int c) { return v[(size_t)r * cols + c]; }
    float at(int r, int c) const { return v[(size_t)r * cols + c]; }
    bool empty() const { return v.empty(); }
};
struct Point3f { float x = 0, y = 0, z = 0; };

// cv::imread(path, 0): 8-bit greyscale.  PGM (P5) and PNG (grey / grey+alpha / RGB / RGBA at 8 or 16 bits, non-interlaced or
// Adam7; chunk CRCs and every size are verified -- image_io.cpp).  Colour is converted like OpenCV:
// (R*4899 + G*9617 + B*1868 + 8192) >> 14.  Never throws; an unreadable file is an empty image.
Image8 imread_gray(const std::string &path);
// the same, decoded straight into a caller buffer of w * h bytes (e.g. page-locked memory); the file must have that size
bool imread_gray_into(const std::string &path, int w, int h, uint8_t *dst, std::string &err);
bool imwrite_pgm(const std::string &path, const Image8 &img);
bool imwrite_png(const std::string &path, const Image8 &img, bool adam7 = false);
// Utilities::exportMat (utilities.cpp:364-378): default ostream precision (6 significant digits, Q14), tab separated
bool exportMat(const std::string &path, const double *m, int rows, int cols);

// ---- VirtualCamera ------------------------------------------------------------------------------------------
class VirtualCamera {
public:
    VirtualCamera();
    void loadDistortion(const std::string &path);
    bool loadCameraMatrix(const std::string &path);          // false + message when the file is missing
    void loadRotationMatrix(const std::string &path);
    void loadTranslationVector(const std::string &path);
    void loadFundamentalMatrix(const std::string &path);
    void loadHomoMatrix(const std::string &path, int i);
    int loadMatrix(Matf &matrix, int rows, int cols, const std::string &file);   // parses through `float` (Q14)

    Matf distortion, rotationMatrix, translationVector, fundamentalMatrix, homoMat1, homoMat2;
    Point3f position;
    float fc[2], cc[2];
    int width = 0, height = 0;
    void fill(slr_camera &out) const;                        // -> C ABI record
};

// ---- stereoRect ---------------------------------------------------------------------------------------------
class stereoRect {
public:
    stereoRect(const std::string &projectPath, int width, int height);
    void getParameters();                                    // 6 text files, parsed via float into f64 (stereorect.cpp:46-62)
    void calParameters();                                    // cv::stereoRectify(flags 0, alpha -1) + 2x initUndistortRectifyMap, restated
    void calRectification();                                 // the stereoRectify half alone (R1, R2, P1, P2, Q)
    bool doStereoRectify(slr_ctx *ctx, Image8 &img, bool isleft);   // cv::remap on the GPU (slr_remap_u8)
    bool upload(slr_ctx *ctx);                               // slr_set_rectify_maps for both cameras
    bool uploadFromCalibration(slr_ctx *ctx);                // slr_init_rectify_maps: the maps are built on the device
    Matd R1, P1, R2, P2, Q;
    Matd M1, D1, M2, D2, R, T;
    std::vector<int16_t> map11, map21;                       // CV_16SC2
    std::vector<uint16_t> map12, map22;                      // CV_16UC1
    int w, h;
private:
    std::string ppath;
    void loadMatrix(Matd &matrix, int rows, int cols, const std::string &file);
};
// cv::initUndistortRectifyMap(M, D, R, P, size, CV_16SC2) restated (SURVEY 8c-3 i)
void initUndistortRectifyMap(const Matd &M, const Matd &D, const Matd &R, const Matd &P, int W, int H,
                             std::vector<int16_t> &map_xy, std::vector<uint16_t> &map_frac);

// ---- PointCloudImage ----------------------------------------------------------------------------------------
class PointCloudImage {
public:
    PointCloudImage(int imageW, int imageH, bool color);
    bool setPoint(int i_w, int j_h, Point3f point);
    bool setPoint(int i_w, int j_h, Point3f point, const int colorgray[3]);
    bool getPoint(int i_w, int j_h, Point3f &pointOut) const;
    bool getPoint(int i_w, int j_h, Point3f &pointOut, int colorOut[3]) const;
    bool addPoint(int i_w, int j_h, Point3f point);
    bool addPoint(int i_w, int j_h, Point3f point, const int colorgray[3]);
    void exportXYZ(const char *path, bool exportOffPixels = true, bool colorFlag = true) const;
    int getWidth() const { return w; }
    int getHeight() const { return h; }
    // storage is public so the C ABI can fill it directly (slr_pointcloud_from_grid / slr_ray_triangulate output)
    std::vector<float> points;                // [h][w][3] sums
    std::vector<uint8_t> numOfPointsForPixel; // [h][w]
    std::vector<uint8_t> color;               // [h][w][3] or empty
private:
    int w, h;
};

// ---- pattern encoders ---------------------------------------------------------------------------------------
class GrayCodes {
public:
    GrayCodes(int scanW, int scanH, bool useepi);
    int getNumOfImgs() const { return numOfImgs; }
    int getNumOfRowBits() const { return numOfRowImgs; }
    int getNumOfColBits() const { return numOfColImgs; }
    void generateGrays();
    static int grayToDec(const std::vector<bool> &gray);
    std::vector<Image8> grayCodes;
    bool useEpi;
private:
    void calNumOfImgs();
    int numOfImgs, numOfRowImgs, numOfColImgs, height, width;
};

class MultiFrequency {
public:
    MultiFrequency(int projwidth = 1280, int projheight = 1024);
    void generateMutiFreq();
    int getNumOfImgs() const { return 14; }
    Image8 MultiFreqImages[14];
private:
    int projW, projH;
};

// ---- the two reconstruction classes ---------------------------------------------------------------------------
class Reconstruct {
public:
    explicit Reconstruct(bool useEpi);
    ~Reconstruct();
    bool loadCameras();
    bool runReconstruction();                 // GRAY_ONLY
    bool runReconstruction_GE();              // GRAY_EPI
    VirtualCamera *cameras;
    std::string *calibFolder;
    PointCloudImage *points3DProjView;
    void setBlackThreshold(int val) { blackThreshold = val; }
    void setWhiteThreshold(int val) { whiteThreshold = val; }
    void setCalibPath(const std::string &path1st, int cam_no);
    void enableRaySampling() { raySampling_ = true; }    // set but never read in the reference either
    void disableRaySampling() { raySampling_ = false; }
    void getParameters(int scanw, int scanh, int camw, int camh, bool autocontrast, bool havecolor,
                       const std::string &savePath);
    std::string savePath_;
    int scanSN = 0;
    std::string imgSuffix = ".png";
    std::string lastError;
    slr_ctx *context() const { return ctx; }             // (for MeshCreator: export on the device the scan ran on)
private:
    bool EPI;
    stereoRect *sr = nullptr;
    slr_ctx *ctx = nullptr;
    bool fillCalib(slr_calib &cal);
    std::string scanFolder[2], imgPrefix[2];
    int numberOfImgs = 0, numOfColBits = 0, numOfRowBits = 0;
    int blackThreshold = 40, whiteThreshold = 0;
    bool pathSet = false, autoContrast_ = false, raySampling_ = false, haveColor = false;
    int cameraWidth = 0, cameraHeight = 0, scan_w = 0, scan_h = 0;
};

class MFReconstruct {
public:
    MFReconstruct();
    ~MFReconstruct();
    void getParameters(int scansn, int scanw, int scanh, int camw, int camh, int blackt, int whitet,
                       const std::string &savePath);
    bool runReconstruction();
    // a series of scans of this project, pipelined over two contexts (PNG decode of scan i+1 into page-locked memory while the
    // GPU works on scan i): sink(sn, cloud) receives every PointCloudImage in order and owns it; false + lastError on failure
    bool runReconstructionSeries(const std::vector<int> &scan_sns, const std::function<bool(int, PointCloudImage *)> &sink);
    PointCloudImage *points3DProjView;
    std::string imgSuffix = ".png";
    std::string lastError;
    bool camerasLoaded = false;
    // which GPU this reconstructor's contexts live on (set before the first run; default 0), and the share of the process's CPU
    // budget its loader pool may take (1 / loaderShare: n reconstructors of a multi-GPU series divide the decoder threads)
    int device = 0;
    unsigned loaderShare = 1;
    slr_ctx *context() const { return ctx; }             // (for MeshCreator: export on the device the scan ran on)
private:
    bool loadCameras();
    void setScan(int sn);
    bool configure(slr_ctx *c, int sn);
    slr_ctx *ctx2 = nullptr;
    struct SeriesBuffers;                                    // page-locked input / output slots of runReconstructionSeries, kept
    SeriesBuffers *series = nullptr;                         // across calls (pinning 4 x 344 MB costs more than decoding a scan)
    int scanSN = 0, numberOfImgs = 14, blackThreshold = 40, whiteThreshold = 0;
    int cameraWidth = 0, cameraHeight = 0, scan_w = 0, scan_h = 0;
    std::string savePath_, calibFolder[2], scanFolder[2], imgPrefix[2];
    VirtualCamera *cameras;
    stereoRect *sr = nullptr;
    slr_ctx *ctx = nullptr;
};

// ---- MeshCreator (meshcreator.cpp:16-172; vertex numbering by slr_prefix_index on the GPU) ----------------------------
class MeshCreator {
public:
    // numbering_ctx: the context the vertex numbering runs on -- normally the reconstructor's (Reconstruct::context()), so the export
    // uses the device the scan was reconstructed on.  nullptr: a context is created for the export on the calling thread's
    // current HIP device.
    explicit MeshCreator(PointCloudImage *in, slr_ctx *numbering_ctx = nullptr);
    bool exportObjMesh(const std::string &path);     // false: no GPU context for the vertex numbering, or the file cannot be written
    bool exportPlyMesh(const std::string &path);
private:
    PointCloudImage *cloud;
    slr_ctx *ctx;
    int w, h;
};

}  // namespace duke
