/*
 * slr_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE ONLY).
 *
 * A plain-C restatement of the per-pixel hot path of DrawZeroPoint/Structure-Light-Reconstructor
 * ("Duke"), used only as the CHECKER by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg.  Nothing under structure-light-reconstructor_amd/ (the product) may include, link, import or
 * execute this code.
 *
 * PARITY STATUS: **parity unpinned** at the OpenCV-2.4.9 boundary.  The reference has no tests, golden
 * vectors or fixtures and cannot be built here (Qt5 + OpenCV 2.4.9 + Win32/MSVC),
 * so no oracle/_ref exists.  The oracle is pinned only by known-answer anchors derived from the reference
 * source text itself (KA1..KA8, tests/test_oracle_known_answers.py).  The arithmetic that lives in
 * un-vendored OpenCV 2.4.9 (cv::remap fixed-point bilinear, cv::Mat GEMM accumulation order,
 * cv::initUndistortRectifyMap) is restated from its published algorithm.
 * Also unpinned: the reference was built with MSVC2010/x87; this restatement fixes strict IEEE
 * f32/f64 evaluation with the C++ overload types the source text selects (atan(float)->float etc.).
 *
 * All citations are file:line under /root/reference/Duke/.
 */
#ifndef SLR_ORACLE_H
#define SLR_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* mfreconstruct.cpp:5  "float PI = 3.1416;"  (Q3) */
#define SLRO_PI_F 3.1416f
/* mfreconstruct.cpp:22 numberOfImgs = 14 */
#define SLRO_MF_PLANES 14

/* virtualcamera.h:27-37 -- the fields the hot path reads.  k[4] (k3) is carried but ignored by
 * undistortPoints (utilities.cpp:66). */
typedef struct slro_camera {
    float fc[2];
    float cc[2];
    float k[5];
    float R[9];   /* row-major 3x3 rotationMatrix */
    float t[3];   /* translationVector */
} slro_camera;

/* ---- encoders (a17) ---------------------------------------------------------------------------- */
/* graycodes.cpp:22-30  ceil(log(n)/log(2)) in double */
int  slro_gray_num_bits(int n);
/* multifrequency.cpp:14-33 : 14 planes, each projH x projW, contiguous [14][projH][projW] */
void slro_gen_multifreq(int projW, int projH, uint8_t *planes);
/* graycodes.cpp:55-114 : 2+2*ncol(+2*nrow) planes [n][scanH][scanW]; returns number of planes */
int  slro_gen_graycodes(int scanW, int scanH, int use_epi, uint8_t *planes);
/* graycodes.cpp:116-128 : bits[0] is the MSB */
int  slro_gray_to_dec(const uint8_t *bits, int nbits);

/* ---- a13 : cv::remap(CV_16SC2, CV_16UC1, INTER_LINEAR, BORDER_CONSTANT 0)  stereorect.cpp:26-34 -- */
void slro_remap_u8(const uint8_t *src, int src_pitch, int W, int H,
                   const int16_t *map_xy, const uint16_t *map_frac,
                   uint8_t *dst, int dst_pitch);

/* ---- a14 (next-row f4, restated OpenCV 2.4 initUndistortRectifyMap, CV_16SC2 output) ---------- */
void slro_init_undistort_rectify_map(const double M[9], const double D[5], const double R[9],
                                     const double P[12], int W, int H,
                                     int16_t *map_xy, uint16_t *map_frac);

/* ---- a1 + a2 + a3 : MFReconstruct::computeShadows / getPhase / decodePatterns
 *      mfreconstruct.cpp:190-269.  planes[14]: 0 white, 1 black, 4c+2+s fringe.
 *      phase[H*W] f32 (0.0f where mask==0), valid[H*W] u8 (mask && all three P defined; Q5 rule). */
void slro_mf_decode(const uint8_t *const planes[SLRO_MF_PLANES], int pitch, int W, int H,
                    int black_thr, float *phase, uint8_t *valid);
/* single-pixel pieces, exposed for the known-answer tests (KA2, KA3) */
int   slro_wrapped_phase(int G1, int G2, int G3, int G4, float *P);   /* returns 0 if undefined (Q5) */
float slro_heterodyne(const double P[3]);

/* ---- a5/a6/a7 : Gray decode.  reconstruct.cpp:56-97,210-227,325-407.
 *      planes: 0 white,1 black, 2c+2 / 2c+3 col bit c (MSB first) pattern/inverse,
 *      row bits at 2c+2+2*ncol.  n_row_bits==0 -> GRAY_EPI (code_y may be NULL).
 *      code = -1 where invalid. */
void slro_gray_decode(const uint8_t *const *planes, int n_col_bits, int n_row_bits,
                      int pitch, int W, int H, int black_thr, int white_thr,
                      int scan_w, int scan_h,
                      int32_t *code_x, int32_t *code_y, uint8_t *valid);

/* ---- a11 : Utilities::undistortPoints utilities.cpp:58-94 ---------------------------------------- */
void slro_undistort_point(float px, float py, const slro_camera *cam, float *ox, float *oy);

/* the two matrix products of a matched pixel, exposed for the literal-cost model (slr_literal.cpp):
 * Q (4x4 f64) * p (4x1 f64), then x/w, y/w, z/w narrowed  mfreconstruct.cpp:299-311 / reconstruct.cpp:570-582;
 * matCoordTrans (3x4 f32) * [X;1]: f64 accumulate, narrowed once  mfreconstruct.cpp:316-322 */
void slro_reproject(const double Q[16], const double p[4], float out[3]);
void slro_apply_T(const float *T, const float in[3], float out[3]);

/* ---- a4 : MFReconstruct::triangulation mfreconstruct.cpp:272-334 (natural [H][W] output, Q11 lives
 *      in slro_pointcloud_from_grid).  T: 3x4 row-major f32 or NULL (scanSN==0).
 *      xyz [H][W][3] (0 where no point), has [H][W], match_k [H][W] (-1 where none; may be NULL). */
void slro_mf_triangulate(const float *phaseL, const uint8_t *validL,
                         const float *phaseR, const uint8_t *validR, int W, int H,
                         const slro_camera *camL, const slro_camera *camR,
                         const double Q[16], const float *T,
                         float *xyz, uint8_t *has, int32_t *match_k);
/* bounded-sample variant for the cpu_baseline leg: rows [row0,row1) only */
void slro_mf_triangulate_rows(const float *phaseL, const uint8_t *validL,
                              const float *phaseR, const uint8_t *validR, int W, int H,
                              int row0, int row1,
                              const slro_camera *camL, const slro_camera *camR,
                              const double Q[16], const float *T,
                              float *xyz, uint8_t *has, int32_t *match_k);

/* ---- a8 : Reconstruct::triangulation_ge reconstruct.cpp:555-611.  whiteL/whiteR rectified white
 *      planes or NULL (haveColor false); color [H][W] u8 grey or NULL. */
void slro_ge_triangulate(const int32_t *codeL, const uint8_t *validL,
                         const int32_t *codeR, const uint8_t *validR, int W, int H,
                         const double Q[16], const float *T,
                         const uint8_t *whiteL, const uint8_t *whiteR,
                         float *xyz, uint8_t *has, uint8_t *color, int32_t *match_k);

/* ---- a7 scatter + a9/a10/a12 : GRAY_ONLY buckets and ray-ray midpoint triangulation
 *      reconstruct.cpp:56-74,310-322,417-481; utilities.cpp:19-28,47-56,399-425.
 *      Buckets as CSR over keys x*scan_h+y (Q9): offsets[scan_w*scan_h+1], items = packed
 *      (col | row<<16) in the reference's column-major traversal order. */
void slro_gray_bucket(const int32_t *code_x, const int32_t *code_y, const uint8_t *valid,
                      int W, int H, int scan_w, int scan_h,
                      int32_t *offsets, uint32_t *items);
void slro_ray_triangulate(const int32_t *offL, const uint32_t *itemsL,
                          const int32_t *offR, const uint32_t *itemsR,
                          const slro_camera *camL, const slro_camera *camR, const float *T,
                          int scan_w, int scan_h,
                          float *xyz_sum /*[scan_h][scan_w][3]*/, uint8_t *count /*[scan_h][scan_w]*/);
/* utilities.cpp:399-425, returns 0 when rejected */
int  slro_line_line_intersection(const float p1[3], const float v1[3],
                                 const float p2[3], const float v2[3], float out[3]);
/* reconstruct.cpp:310-322 */
void slro_cam2world(const slro_camera *cam, float p[3]);

/* ---- a15 : PointCloudImage (pointcloudimage.cpp:3-97) + Q11 host adaptor ------------------------
 *      Fill a PointCloudImage(w=scan_w,h=scan_h) from a natural [H][W] grid exactly as the
 *      reference's addPoint(i=row, j=col, p) would: points[j][i] iff i<scan_w && j<scan_h.
 *      pc_sum [scan_h][scan_w][3], pc_count [scan_h][scan_w], pc_color [scan_h][scan_w] or NULL. */
void slro_pointcloud_from_grid(const float *xyz, const uint8_t *has, const uint8_t *color,
                               int W, int H, int scan_w, int scan_h,
                               float *pc_sum, uint8_t *pc_count, uint8_t *pc_color);
/* getPoint: sum / (float)count, pointcloudimage.cpp:56-67 ; out [n][3] */
void slro_pointcloud_get(const float *pc_sum, const uint8_t *pc_count, int n, float *out);

/* ---- BUILD EXTENSION, NO REFERENCE COUNTERPART (BASELINE config 5; parity unpinned by construction) --------
 *      fp64 model of the generalised n_freq x n_step decode that slr_mfn_decode implements: planes are IEEE
 *      binary16 bit patterns, planes[0] white, planes[1] black, planes[2 + f*n_step + k]; phase in f64
 *      (0 where shadow-masked), valid = mask && every frequency has modulation.  Not a restatement of anything in
 *      /root/reference: the reference is hard-wired to 3 x 4 steps of u8 (mfreconstruct.cpp:21-22,237-242). */
void slro_mfn_decode_f64(const uint16_t *const *planes, int n_freq, int n_step, int pitch, int W, int H,
                         double black_thr, double *phase, uint8_t *valid);

/* ... through the rectification (slr_mfn_rectify_decode): cv::remap's geometry, the bilinear sample exact in f64; rows [row0, row1)
 * of full [H][W] outputs */
void slro_mfn_rect_decode_f64(const uint16_t *const *planes, int n_freq, int n_step, int pitch, int W, int H,
                              double black_thr, const int16_t *map_xy, const uint16_t *map_frac, int row0, int row1,
                              double *phase, uint8_t *valid);

/* ---- second evaluation model: the reference's MSVC2010 x87 build (slr_oracle_x87.c; sensitivity analysis only) --------------
 *      x87 = 0: strict IEEE, what every function above computes; x87 = 1: 53-bit x87 stack under /fp:precise (see that file).
 *      atab: 511 floats from slro_atan_table (mode 0 glibc atanf, 1 MSVC x86's (float)atan((double)q), 2 / 3 +-1 ulp, >= 4 random). */
void  slro_atan_table(int mode, float tab[511]);
int   slro_wrapped_phase_ev(int G1, int G2, int G3, int G4, const float *atab, int x87, double *P);
float slro_heterodyne_ev(const double P[3], int x87);
void  slro_mf_decode_ev(const uint8_t *const planes[SLRO_MF_PLANES], int pitch, int W, int H, int black_thr,
                        const float *atab, int x87, float *phase, uint8_t *valid);
void  slro_mf_triangulate_rows_ev(const float *phaseL, const uint8_t *validL, const float *phaseR, const uint8_t *validR,
                                  int W, int H, int row0, int row1, const slro_camera *camL, const slro_camera *camR,
                                  const double Q[16], const float *T, int x87, float *xyz, uint8_t *has, int32_t *match_k);
int   slro_line_line_intersection_x87(const float p1[3], const float v1[3], const float p2[3], const float v2[3], float out[3]);
void  slro_ray_triangulate_x87(const int32_t *offL, const uint32_t *itemsL, const int32_t *offR, const uint32_t *itemsR,
                               const slro_camera *camL, const slro_camera *camR, const float *T, int scan_w, int scan_h,
                               float *xyz_sum, uint8_t *count);
/* the device's constant-divisor form of :268's division under the x87 model, counted against the division (slr_oracle_x87.c) */
long  slro_x87_quotient_mismatches(long lo, long hi);

#ifdef __cplusplus
}
#endif
#endif
