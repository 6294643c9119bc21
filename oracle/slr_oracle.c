/*
 * slr_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE ONLY; see slr_oracle.h header for the rules and
 * for the "parity unpinned" statement).  Build: make -C oracle   (gcc -O2 -ffp-contract=off).
 *
 * Every function cites the reference lines it restates (paths under /root/reference/Duke/).
 * Arithmetic types follow the reference's source text operation by operation; nothing is
 * "improved".  No code is copied: containers (cv::Mat, cv::vector per pixel) are replaced by flat
 * arrays and every loop is re-expressed.
 */
#include "slr_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------------ */
/* encoders (a17)                                                                                    */
/* ------------------------------------------------------------------------------------------------ */

/* graycodes.cpp:24-25 : (int)ceil(log(double(width))/log(2.0)) */
int slro_gray_num_bits(int n)
{
    return (int)ceil(log((double)n) / log(2.0));
}

/* multifrequency.cpp:3 frequency[] ; multifrequency.h:5 "#define PI 3.1416" (a double here, unlike the
 * float global of mfreconstruct.cpp) ; multifrequency.cpp:27
 *   temp(h,w) = 135+79*cos(float(PI*2*w*frequency[f]/projW+PI*phi/2))   -> cos(float) is cosf,
 *   79*float -> float, 135+float -> float, stored into uchar (truncation). */
void slro_gen_multifreq(int projW, int projH, uint8_t *planes)
{
    static const int freq[3] = {70, 64, 59};
    const double PI_D = 3.1416;
    const size_t plane = (size_t)projW * (size_t)projH;
    memset(planes, 255, plane);          /* multifrequency.cpp:16 white */
    memset(planes + plane, 0, plane);    /* multifrequency.cpp:17 black */
    for (int f = 0; f < 3; f++) {
        for (int phi = 0; phi < 4; phi++) {
            uint8_t *dst = planes + (size_t)(4 * f + phi + 2) * plane;
            for (int w = 0; w < projW; w++) {
                double arg = PI_D * 2 * (double)w * (double)freq[f] / (double)projW
                             + PI_D * (double)phi / 2;
                float v = 135 + 79 * cosf((float)arg);
                uint8_t g = (uint8_t)v;
                for (int h = 0; h < projH; h++) dst[(size_t)h * projW + w] = g;
            }
        }
    }
}

/* graycodes.cpp:55-114.  Column bit k (k=0 is the LSB of the Gray word) goes to plane 2n-2k, its
 * inverse to 2n-2k+1; row bits likewise offset by 2n.  flag = (rem != prevRem). */
int slro_gen_graycodes(int scanW, int scanH, int use_epi, uint8_t *planes)
{
    const int ncol = slro_gray_num_bits(scanW);
    const int nrow = slro_gray_num_bits(scanH);
    const int nimg = use_epi ? 2 + 2 * ncol : 2 + 2 * ncol + 2 * nrow;
    const size_t plane = (size_t)scanW * (size_t)scanH;
    memset(planes, 255, plane);
    memset(planes + plane, 0, plane);
    for (int j = 0; j < scanW; j++) {
        int num = j, prevRem = j % 2;
        for (int k = 0; k < ncol; k++) {
            num = num / 2;
            int rem = num % 2;
            int flag = (rem != prevRem);
            uint8_t a = (uint8_t)(flag * 255), b = (uint8_t)(a > 0 ? 0 : 255);
            uint8_t *pa = planes + (size_t)(2 * ncol - 2 * k) * plane;
            uint8_t *pb = planes + (size_t)(2 * ncol - 2 * k + 1) * plane;
            for (int i = 0; i < scanH; i++) {
                pa[(size_t)i * scanW + j] = a;
                pb[(size_t)i * scanW + j] = b;
            }
            prevRem = rem;
        }
    }
    if (!use_epi) {
        for (int i = 0; i < scanH; i++) {
            int num = i, prevRem = i % 2;
            for (int k = 0; k < nrow; k++) {
                num = num / 2;
                int rem = num % 2;
                int flag = (rem != prevRem);
                uint8_t a = (uint8_t)(flag * 255), b = (uint8_t)(a > 0 ? 0 : 255);
                uint8_t *pa = planes + (size_t)(2 * nrow - 2 * k + 2 * ncol) * plane;
                uint8_t *pb = planes + (size_t)(2 * nrow - 2 * k + 2 * ncol + 1) * plane;
                for (int j = 0; j < scanW; j++) {
                    pa[(size_t)i * scanW + j] = a;
                    pb[(size_t)i * scanW + j] = b;
                }
                prevRem = rem;
            }
        }
    }
    return nimg;
}

/* graycodes.cpp:116-128 : running XOR from the MSB; dec += 2^(n-i-1) when the running bit is set */
int slro_gray_to_dec(const uint8_t *bits, int nbits)
{
    int dec = 0;
    int tmp = bits[0] ? 1 : 0;
    if (tmp) dec += (int)powf(2.0f, (float)(nbits - 1));
    for (int i = 1; i < nbits; i++) {
        tmp = (tmp == (bits[i] ? 1 : 0)) ? 0 : 1;      /* utilities.cpp:11-17 XOR */
        if (tmp) dec += (int)powf(2.0f, (float)(nbits - i - 1));
    }
    return dec;
}

/* ------------------------------------------------------------------------------------------------ */
/* a13 : cv::remap fixed-point bilinear (OpenCV 2.4.9 imgproc, restated; SURVEY 8c-3 ii)             */
/* ------------------------------------------------------------------------------------------------ */
void slro_remap_u8(const uint8_t *src, int src_pitch, int W, int H,
                   const int16_t *map_xy, const uint16_t *map_frac,
                   uint8_t *dst, int dst_pitch)
{
    for (int y = 0; y < H; y++) {
        for (int x = 0; x < W; x++) {
            const size_t m = (size_t)y * W + x;
            const int sx = map_xy[2 * m], sy = map_xy[2 * m + 1];
            const int f = map_frac[m] & 1023;             /* INTER_TAB_SIZE2-1 */
            const int fx = f & 31, fy = f >> 5;
            /* weights: float tab (1-fx/32)(1-fy/32).. * 32768 rounded to i16; exact integers. The one
             * saturating entry (fx=fy=0 -> {32767,0,0,1}) gives the same u8 result. */
            const int w00 = (32 - fx) * (32 - fy) * 32, w01 = fx * (32 - fy) * 32;
            const int w10 = (32 - fx) * fy * 32,        w11 = fx * fy * 32;
            int v;
            if (sx >= W || sx + 1 < 0 || sy >= H || sy + 1 < 0) {
                v = 0;                                       /* BORDER_CONSTANT, borderValue 0 */
            } else {
                const int x0ok = (sx >= 0 && sx < W), x1ok = (sx + 1 >= 0 && sx + 1 < W);
                const int y0ok = (sy >= 0 && sy < H), y1ok = (sy + 1 >= 0 && sy + 1 < H);
                const int s00 = (x0ok && y0ok) ? src[(size_t)sy * src_pitch + sx] : 0;
                const int s01 = (x1ok && y0ok) ? src[(size_t)sy * src_pitch + sx + 1] : 0;
                const int s10 = (x0ok && y1ok) ? src[(size_t)(sy + 1) * src_pitch + sx] : 0;
                const int s11 = (x1ok && y1ok) ? src[(size_t)(sy + 1) * src_pitch + sx + 1] : 0;
                v = (s00 * w00 + s01 * w01 + s10 * w10 + s11 * w11 + 16384) >> 15;
                if (v > 255) v = 255;
            }
            dst[(size_t)y * dst_pitch + x] = (uint8_t)v;
        }
    }
}

/* a14 / f4 : cv::initUndistortRectifyMap(M, D, R, P, size, CV_16SC2) restated from OpenCV 2.4
 * imgproc/undistort.cpp (SURVEY 8c-3 i).  iR = (P[:,0:3]*R)^-1 by adjugate/det (3x3 closed form). */
static long slro_cvround(double v) { return lrint(v); }   /* round-half-even under default rounding */

void slro_init_undistort_rectify_map(const double M[9], const double D[5], const double R[9],
                                     const double P[12], int W, int H,
                                     int16_t *map_xy, uint16_t *map_frac)
{
    double A[9], ir[9];
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += P[r * 4 + k] * R[k * 3 + c];
            A[r * 3 + c] = s;
        }
    double det = A[0] * (A[4] * A[8] - A[5] * A[7]) - A[1] * (A[3] * A[8] - A[5] * A[6])
               + A[2] * (A[3] * A[7] - A[4] * A[6]);
    double d = 1. / det;
    ir[0] = (A[4] * A[8] - A[5] * A[7]) * d; ir[1] = (A[2] * A[7] - A[1] * A[8]) * d;
    ir[2] = (A[1] * A[5] - A[2] * A[4]) * d; ir[3] = (A[5] * A[6] - A[3] * A[8]) * d;
    ir[4] = (A[0] * A[8] - A[2] * A[6]) * d; ir[5] = (A[2] * A[3] - A[0] * A[5]) * d;
    ir[6] = (A[3] * A[7] - A[4] * A[6]) * d; ir[7] = (A[1] * A[6] - A[0] * A[7]) * d;
    ir[8] = (A[0] * A[4] - A[1] * A[3]) * d;

    const double u0 = M[2], v0 = M[5], fx = M[0], fy = M[4];
    const double k1 = D[0], k2 = D[1], p1 = D[2], p2 = D[3], k3 = D[4];
    for (int i = 0; i < H; i++) {
        double _x = i * ir[1] + ir[2], _y = i * ir[4] + ir[5], _w = i * ir[7] + ir[8];
        for (int j = 0; j < W; j++, _x += ir[0], _y += ir[3], _w += ir[6]) {
            double w = 1. / _w, x = _x * w, y = _y * w;
            double x2 = x * x, y2 = y * y;
            double r2 = x2 + y2, _2xy = 2 * x * y;
            double kr = 1 + ((k3 * r2 + k2) * r2 + k1) * r2;
            double u = fx * (x * kr + p1 * _2xy + p2 * (r2 + 2 * x2)) + u0;
            double v = fy * (y * kr + p1 * (r2 + 2 * y2) + p2 * _2xy) + v0;
            long iu = slro_cvround(u * 32), iv = slro_cvround(v * 32);
            size_t m = (size_t)i * W + j;
            map_xy[2 * m]     = (int16_t)(iu >> 5);
            map_xy[2 * m + 1] = (int16_t)(iv >> 5);
            map_frac[m] = (uint16_t)((iv & 31) * 32 + (iu & 31));
        }
    }
}

/* ------------------------------------------------------------------------------------------------ */
/* a1..a3 : multi-frequency decode                                                                   */
/* ------------------------------------------------------------------------------------------------ */

/* mfreconstruct.cpp:246-261 (Q1,Q2,Q3,Q5).  G are ints; (G4-G2)/(G1-G3) is C integer division;
 * atan(float(..)) resolves to the float overload; PI, 2*PI, 3*PI/2, PI/2 are f32 expressions. */
int slro_wrapped_phase(int G1, int G2, int G3, int G4, float *P)
{
    const float PI = SLRO_PI_F;
    if (G4 == G2 && G1 > G3)      { *P = 0.0f;        return 1; }
    else if (G4 == G2 && G1 < G3) { *P = PI;          return 1; }
    else if (G1 == G3 && G4 > G2) { *P = 3 * PI / 2;  return 1; }
    else if (G1 == G3 && G4 < G2) { *P = PI / 2;      return 1; }
    else if (G1 == G3 && G4 == G2) { *P = 0.0f;       return 0; }  /* Q5: UB in the reference; rule */
    else if (G1 < G3)             { *P = atanf((float)((G4 - G2) / (G1 - G3))) + PI;     return 1; }
    else if (G1 > G3 && G4 > G2)  { *P = atanf((float)((G4 - G2) / (G1 - G3))) + 2 * PI; return 1; }
    else                          { *P = atanf((float)((G4 - G2) / (G1 - G3)));          return 1; }
}

/* mfreconstruct.cpp:265-268 (Q4): P[] double; P12,P23 computed in f64 and narrowed once; P123 and
 * phase are pure f32. */
float slro_heterodyne(const double P[3])
{
    const float PI = SLRO_PI_F;
    float P12  = (float)((P[0] > P[1]) ? (P[0] - P[1]) : (P[0] - P[1] + (double)(2 * PI)));
    float P23  = (float)((P[1] > P[2]) ? (P[1] - P[2]) : (P[1] - P[2] + (double)(2 * PI)));
    float P123 = (P12 > P23) ? (P12 - P23) : (P12 - P23 + 2 * PI);
    return P123 / (2 * PI) * 255;
}

void slro_mf_decode(const uint8_t *const planes[SLRO_MF_PLANES], int pitch, int W, int H,
                    int black_thr, float *phase, uint8_t *valid)
{
    for (int row = 0; row < H; row++) {
        for (int col = 0; col < W; col++) {
            const size_t s = (size_t)row * pitch + col, o = (size_t)row * W + col;
            /* mfreconstruct.cpp:198-204 computeShadows */
            float whiteVal = (float)planes[0][s], blackVal = (float)planes[1][s];
            int mask = (whiteVal - blackVal > (float)black_thr) ? 1 : 0;
            float ph = 0.0f;
            int ok = mask;
            if (mask) {                                   /* mfreconstruct.cpp:219-223 */
                double P[3];
                for (int c = 0; c < 3; c++) {             /* mfreconstruct.cpp:237-262 */
                    float Pf;
                    int G1 = planes[4 * c + 2][s], G2 = planes[4 * c + 3][s];
                    int G3 = planes[4 * c + 4][s], G4 = planes[4 * c + 5][s];
                    if (!slro_wrapped_phase(G1, G2, G3, G4, &Pf)) ok = 0;
                    P[c] = Pf;
                }
                ph = slro_heterodyne(P);
            }
            phase[o] = ph;
            valid[o] = (uint8_t)ok;
        }
    }
}

/* ------------------------------------------------------------------------------------------------ */
/* a5..a7 : Gray decode                                                                              */
/* ------------------------------------------------------------------------------------------------ */
void slro_gray_decode(const uint8_t *const *planes, int n_col_bits, int n_row_bits,
                      int pitch, int W, int H, int black_thr, int white_thr,
                      int scan_w, int scan_h,
                      int32_t *code_x, int32_t *code_y, uint8_t *valid)
{
    uint8_t bits[32];
    for (int row = 0; row < H; row++) {
        for (int col = 0; col < W; col++) {
            const size_t s = (size_t)row * pitch + col, o = (size_t)row * W + col;
            float whiteVal = (float)planes[0][s], blackVal = (float)planes[1][s];   /* reconstruct.cpp:218-224 */
            int mask = (whiteVal - blackVal > (float)black_thr) ? 1 : 0;
            int xDec = -1, yDec = -1;
            if (mask) {
                int error = 0;
                for (int c = 0; c < n_col_bits; c++) {    /* reconstruct.cpp:333-346 / 387-400 */
                    double v1 = planes[c * 2 + 2][s], v2 = planes[c * 2 + 3][s];
                    if (fabs(v1 - v2) < (double)white_thr) error = 1;
                    bits[c] = (v1 > v2) ? 1 : 0;
                }
                xDec = slro_gray_to_dec(bits, n_col_bits);
                if (n_row_bits > 0) {                     /* reconstruct.cpp:349-366 */
                    for (int c = 0; c < n_row_bits; c++) {
                        double v1 = planes[c * 2 + 2 + n_col_bits * 2][s];
                        double v2 = planes[c * 2 + 2 + n_col_bits * 2 + 1][s];
                        if (fabs(v1 - v2) < (double)white_thr) error = 1;
                        bits[c] = (v1 > v2) ? 1 : 0;
                    }
                    yDec = slro_gray_to_dec(bits, n_row_bits);
                    if (yDec > scan_h || xDec > scan_w) error = 1;       /* Q9: '>' not '>=' */
                } else {
                    if (xDec > scan_w) error = 1;         /* reconstruct.cpp:403 */
                }
                if (error) { mask = 0; xDec = -1; yDec = -1; }
            }
            code_x[o] = xDec;
            if (code_y) code_y[o] = (n_row_bits > 0) ? yDec : -1;
            valid[o] = (uint8_t)mask;
        }
    }
}

/* ------------------------------------------------------------------------------------------------ */
/* a11 : undistortPoints  utilities.cpp:58-94   (all f64 except the two marked narrowings)           */
/* ------------------------------------------------------------------------------------------------ */
void slro_undistort_point(float px, float py, const slro_camera *cam, float *ox, float *oy)
{
    double k[5] = {cam->k[0], cam->k[1], cam->k[2], cam->k[3], 0};   /* :62-66, k[4]=0 */
    double fx = cam->fc[0], fy = cam->fc[1];
    double ifx = 1. / fx, ify = 1. / fy;
    double cx = cam->cc[0], cy = cam->cc[1];
    double x = px, y = py, x0, y0;
    x0 = x = (x - cx) * ifx;
    y0 = y = (y - cy) * ify;
    for (int jj = 0; jj < 5; jj++) {
        double r2 = x * x + y * y;
        double icdist = 1. / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
        double deltaX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x);
        double deltaY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y;
        x = (x0 - deltaX) * icdist;
        y = (y0 - deltaY) * icdist;
    }
    *ox = (float)((double)(float)(x * fx) + cx);     /* :93  (float)(x*fx)+cx -> Point2f */
    *oy = (float)((double)(float)(y * fy) + cy);
}

/* cv::Mat 3x4 f32 * 4x1 f32 : OpenCV f32 GEMM accumulates in f64 and narrows (SURVEY 8c-3 iii) */
void slro_apply_T(const float *T, const float in[3], float out[3])
{
    const float p[4] = {in[0], in[1], in[2], 1.0f};
    for (int r = 0; r < 3; r++) {
        double s = 0;
        for (int c = 0; c < 4; c++) s += (double)T[r * 4 + c] * (double)p[c];
        out[r] = (float)s;
    }
}

/* Q (4x4 f64) * p (4x1 f64), then x/w..  mfreconstruct.cpp:299-311 / reconstruct.cpp:570-582 */
void slro_reproject(const double Q[16], const double p[4], float out[3])
{
    double r[4];
    for (int i = 0; i < 4; i++) {
        double s = 0;
        for (int c = 0; c < 4; c++) s += Q[i * 4 + c] * p[c];
        r[i] = s;
    }
    out[0] = (float)(r[0] / r[3]);
    out[1] = (float)(r[1] / r[3]);
    out[2] = (float)(r[2] / r[3]);
}

/* ------------------------------------------------------------------------------------------------ */
/* a4 : MF match + triangulate  mfreconstruct.cpp:284-333                                            */
/* ------------------------------------------------------------------------------------------------ */
void slro_mf_triangulate_rows(const float *phaseL, const uint8_t *validL,
                              const float *phaseR, const uint8_t *validR, int W, int H,
                              int row0, int row1,
                              const slro_camera *camL, const slro_camera *camR,
                              const double Q[16], const float *T,
                              float *xyz, uint8_t *has, int32_t *match_k)
{
    (void)H;
    for (int i = row0; i < row1; i++) {
        for (int j = 0; j < W; j++) {
            const size_t o = (size_t)i * W + j;
            xyz[3 * o] = xyz[3 * o + 1] = xyz[3 * o + 2] = 0.0f;
            has[o] = 0;
            if (match_k) match_k[o] = -1;
            if (!validL[o]) continue;                                  /* :287 */
            const float pl = phaseL[o];
            for (int k = 0; k < W; k++) {                              /* :289 */
                const size_t r = (size_t)i * W + k;
                if (!validR[r]) continue;                              /* :292 */
                if (fabs(pl - phaseR[r]) < 0.1) {                      /* :295 (f32 diff vs 0.1 double) */
                    float ulx, uly, urx, ury, X[3];
                    slro_undistort_point((float)j, (float)i, camL, &ulx, &uly);   /* :297 */
                    slro_undistort_point((float)k, (float)i, camR, &urx, &ury);   /* :298 */
                    double p[4] = {ulx, uly, (double)(float)(ulx - urx), 1};     /* :299 */
                    slro_reproject(Q, p, X);
                    if (T) { float Y[3]; slro_apply_T(T, X, Y); X[0] = Y[0]; X[1] = Y[1]; X[2] = Y[2]; }
                    xyz[3 * o] = X[0]; xyz[3 * o + 1] = X[1]; xyz[3 * o + 2] = X[2];
                    has[o] = 1;
                    if (match_k) match_k[o] = k;
                    break;                                             /* :327 first match wins (Q7) */
                }
            }
        }
    }
}

void slro_mf_triangulate(const float *phaseL, const uint8_t *validL,
                         const float *phaseR, const uint8_t *validR, int W, int H,
                         const slro_camera *camL, const slro_camera *camR,
                         const double Q[16], const float *T,
                         float *xyz, uint8_t *has, int32_t *match_k)
{
    slro_mf_triangulate_rows(phaseL, validL, phaseR, validR, W, H, 0, H, camL, camR, Q, T,
                             xyz, has, match_k);
}

/* ------------------------------------------------------------------------------------------------ */
/* a8 : GE match + triangulate  reconstruct.cpp:555-611                                              */
/* ------------------------------------------------------------------------------------------------ */
void slro_ge_triangulate(const int32_t *codeL, const uint8_t *validL,
                         const int32_t *codeR, const uint8_t *validR, int W, int H,
                         const double Q[16], const float *T,
                         const uint8_t *whiteL, const uint8_t *whiteR,
                         float *xyz, uint8_t *has, uint8_t *color, int32_t *match_k)
{
    for (int i = 0; i < H; i++) {
        int kstart = 0;                                                /* :556 */
        for (int j = 0; j < W; j++) {
            const size_t o = (size_t)i * W + j;
            xyz[3 * o] = xyz[3 * o + 1] = xyz[3 * o + 2] = 0.0f;
            has[o] = 0;
            if (color) color[o] = 0;
            if (match_k) match_k[o] = -1;
            if (!validL[o]) continue;                                  /* :559 */
            for (int k = kstart; k < W; k++) {                         /* :561 */
                const size_t r = (size_t)i * W + k;
                if (!validR[r]) continue;
                if (codeL[o] == codeR[r]) {                            /* :565 */
                    double p[4] = {(double)j, (double)i, (double)(j - k), 1};   /* :570 */
                    float X[3];
                    slro_reproject(Q, p, X);
                    if (T) { float Y[3]; slro_apply_T(T, X, Y); X[0] = Y[0]; X[1] = Y[1]; X[2] = Y[2]; }
                    xyz[3 * o] = X[0]; xyz[3 * o + 1] = X[1]; xyz[3 * o + 2] = X[2];
                    has[o] = 1;
                    if (color && whiteL && whiteR)                     /* :598 (Q12) */
                        color[o] = (uint8_t)(((int)whiteL[o] + (int)whiteR[r]) / 2);
                    if (match_k) match_k[o] = k;
                    kstart = k;                                        /* :604 */
                    break;
                }
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------------ */
/* a7 scatter : decodePaterns reconstruct.cpp:61-72 (column-major traversal), key x*scan_h+y (Q9)    */
/* ------------------------------------------------------------------------------------------------ */
void slro_gray_bucket(const int32_t *code_x, const int32_t *code_y, const uint8_t *valid,
                      int W, int H, int scan_w, int scan_h,
                      int32_t *offsets, uint32_t *items)
{
    const long nb = (long)scan_w * scan_h;
    memset(offsets, 0, sizeof(int32_t) * (size_t)(nb + 1));
    for (int col = 0; col < W; col++)
        for (int row = 0; row < H; row++) {
            const size_t o = (size_t)row * W + col;
            if (!valid[o]) continue;
            long key = (long)code_x[o] * scan_h + code_y[o];
            if (key >= nb) continue;           /* Q9: OOB write in the reference; dropped by rule */
            offsets[key + 1]++;
        }
    for (long b = 0; b < nb; b++) offsets[b + 1] += offsets[b];
    int32_t *cur = (int32_t *)malloc(sizeof(int32_t) * (size_t)nb);
    memcpy(cur, offsets, sizeof(int32_t) * (size_t)nb);
    for (int col = 0; col < W; col++)
        for (int row = 0; row < H; row++) {
            const size_t o = (size_t)row * W + col;
            if (!valid[o]) continue;
            long key = (long)code_x[o] * scan_h + code_y[o];
            if (key >= nb) continue;
            items[cur[key]++] = (uint32_t)col | ((uint32_t)row << 16);
        }
    free(cur);
}

/* reconstruct.cpp:310-322 : tmp = -R^T t ; tmpPoint = R^T p ; p = tmp + tmpPoint.
 * OpenCV f32 GEMM: f64 accumulation, d = (float)(s*alpha). */
void slro_cam2world(const slro_camera *cam, float p[3])
{
    float tmp[3], tp[3];
    for (int r = 0; r < 3; r++) {
        double s = 0, s2 = 0;
        for (int k = 0; k < 3; k++) {
            s  += (double)cam->R[k * 3 + r] * (double)cam->t[k];
            s2 += (double)cam->R[k * 3 + r] * (double)p[k];
        }
        tmp[r] = (float)(s * -1.0);
        tp[r]  = (float)s2;
    }
    p[0] = tmp[0] + tp[0];
    p[1] = tmp[1] + tp[1];
    p[2] = tmp[2] + tp[2];
}

/* utilities.cpp:399-425 ; Vec3f::dot accumulates in f32 left to right from 0 */
static float slro_dot3(const float a[3], const float b[3])
{
    float s = 0;
    s += a[0] * b[0];
    s += a[1] * b[1];
    s += a[2] * b[2];
    return s;
}

int slro_line_line_intersection(const float p1[3], const float v1[3],
                                const float p2[3], const float v2[3], float out[3])
{
    float v12[3] = {p1[0] - p2[0], p1[1] - p2[1], p1[2] - p2[2]};
    float v1_dot_v1 = slro_dot3(v1, v1);
    float v2_dot_v2 = slro_dot3(v2, v2);
    float v1_dot_v2 = slro_dot3(v1, v2);
    float v12_dot_v1 = slro_dot3(v12, v1);
    float v12_dot_v2 = slro_dot3(v12, v2);
    float denom = v1_dot_v1 * v2_dot_v2 - v1_dot_v2 * v1_dot_v2;
    if (fabsf(denom) < 0.1) return 0;                                 /* :414 */
    float s =  (v1_dot_v2 / denom) * v12_dot_v2 - (v2_dot_v2 / denom) * v12_dot_v1;
    float t = -(v1_dot_v2 / denom) * v12_dot_v1 + (v1_dot_v1 / denom) * v12_dot_v2;
    for (int c = 0; c < 3; c++) {
        float a = p1[c] + s * v1[c];
        float b = p2[c] + t * v2[c];
        out[c] = (float)(0.5 * (double)(a + b));                      /* :420-422 */
    }
    return 1;
}

/* per camera pixel: reconstruct.cpp:440-445 -> unit ray from camera centre */
static void slro_pixel_ray(uint32_t item, const slro_camera *cam, const float pos[3], float ray[3])
{
    float ux, uy, pt[3];
    slro_undistort_point((float)(item & 0xFFFFu), (float)(item >> 16), cam, &ux, &uy);
    pt[0] = (ux - cam->cc[0]) / cam->fc[0];                           /* utilities.cpp:51-53 */
    pt[1] = (uy - cam->cc[1]) / cam->fc[1];
    pt[2] = 1;
    slro_cam2world(cam, pt);
    ray[0] = pos[0] - pt[0]; ray[1] = pos[1] - pt[1]; ray[2] = pos[2] - pt[2];
    /* utilities.cpp:19-25 : sqrt(float) overload -> f32 ; max(0.000001, mag) in f64 ; /= (float) */
    double mag = sqrtf(ray[0] * ray[0] + ray[1] * ray[1] + ray[2] * ray[2]);
    float dv = (float)(0.000001 > mag ? 0.000001 : mag);
    ray[0] /= dv; ray[1] /= dv; ray[2] /= dv;
}

/* a9 : Reconstruct::triangulation reconstruct.cpp:428-480 with PointCloudImage::addPoint
 * (pointcloudimage.cpp:86-97) folded in: first hit sets sum,count=1 ; later hits add, count=(u8)(c+1)
 * (a wrap to 0 makes the next hit a "set" again, as in the reference). */
void slro_ray_triangulate(const int32_t *offL, const uint32_t *itemsL,
                          const int32_t *offR, const uint32_t *itemsR,
                          const slro_camera *camL, const slro_camera *camR, const float *T,
                          int scan_w, int scan_h, float *xyz_sum, uint8_t *count)
{
    float posL[3] = {0, 0, 0}, posR[3] = {0, 0, 0};
    slro_cam2world(camL, posL);                                       /* reconstruct.cpp:239-240 */
    slro_cam2world(camR, posR);
    memset(xyz_sum, 0, sizeof(float) * 3 * (size_t)scan_w * scan_h);
    memset(count, 0, (size_t)scan_w * scan_h);
    for (int i = 0; i < scan_w; i++)
        for (int j = 0; j < scan_h; j++) {
            const long b = (long)i * scan_h + j;
            const int n1 = offL[b + 1] - offL[b], n2 = offR[b + 1] - offR[b];
            if (n1 == 0 || n2 == 0) continue;
            const size_t o = (size_t)j * scan_w + i;                  /* addPoint(i_w=i, j_h=j) */
            for (int c1 = 0; c1 < n1; c1++) {
                float r1[3];
                slro_pixel_ray(itemsL[offL[b] + c1], camL, posL, r1);
                for (int c2 = 0; c2 < n2; c2++) {
                    float r2[3], X[3];
                    slro_pixel_ray(itemsR[offR[b] + c2], camR, posR, r2);
                    if (!slro_line_line_intersection(posL, r1, posR, r2, X)) continue;
                    if (T) { float Y[3]; slro_apply_T(T, X, Y); X[0] = Y[0]; X[1] = Y[1]; X[2] = Y[2]; }
                    uint8_t num = count[o];
                    if (num == 0) {
                        xyz_sum[3 * o] = X[0]; xyz_sum[3 * o + 1] = X[1]; xyz_sum[3 * o + 2] = X[2];
                        count[o] = 1;
                    } else {
                        xyz_sum[3 * o]     = X[0] + xyz_sum[3 * o];
                        xyz_sum[3 * o + 1] = X[1] + xyz_sum[3 * o + 1];
                        xyz_sum[3 * o + 2] = X[2] + xyz_sum[3 * o + 2];
                        count[o] = (uint8_t)(num + 1);
                    }
                }
            }
        }
}

/* ------------------------------------------------------------------------------------------------ */
/* a15 + Q11 : PointCloudImage adaptor                                                               */
/* ------------------------------------------------------------------------------------------------ */
void slro_pointcloud_from_grid(const float *xyz, const uint8_t *has, const uint8_t *color,
                               int W, int H, int scan_w, int scan_h,
                               float *pc_sum, uint8_t *pc_count, uint8_t *pc_color)
{
    memset(pc_sum, 0, sizeof(float) * 3 * (size_t)scan_w * scan_h);
    memset(pc_count, 0, (size_t)scan_w * scan_h);
    if (pc_color) memset(pc_color, 0, (size_t)scan_w * scan_h);
    for (int i = 0; i < H; i++)
        for (int j = 0; j < W; j++) {
            const size_t o = (size_t)i * W + j;
            if (!has[o]) continue;
            if (i >= scan_w || j >= scan_h) continue;                 /* pointcloudimage.cpp:88 */
            const size_t d = (size_t)j * scan_w + i;                  /* row=j_h, col=i_w */
            pc_sum[3 * d] = xyz[3 * o]; pc_sum[3 * d + 1] = xyz[3 * o + 1]; pc_sum[3 * d + 2] = xyz[3 * o + 2];
            pc_count[d] = 1;
            if (pc_color && color) pc_color[d] = color[o];
        }
}

void slro_pointcloud_get(const float *pc_sum, const uint8_t *pc_count, int n, float *out)
{
    for (int i = 0; i < n; i++) {
        if (pc_count[i] == 0) { out[3 * i] = out[3 * i + 1] = out[3 * i + 2] = 0.0f; continue; }
        double dnum = (double)(float)pc_count[i];                     /* Vec3d / float */
        out[3 * i]     = (float)((double)pc_sum[3 * i] / dnum);
        out[3 * i + 1] = (float)((double)pc_sum[3 * i + 1] / dnum);
        out[3 * i + 2] = (float)((double)pc_sum[3 * i + 2] / dnum);
    }
}

/* ------------------------------------------------------------------------------------------------ */
/* build extension (no reference counterpart): fp64 model of the generalised n_freq x n_step decode   */
/* ------------------------------------------------------------------------------------------------ */
static double slro_half_to_double(uint16_t h)
{
    const int sign = h >> 15, e = (h >> 10) & 31, m = h & 1023;
    double v;
    if (e == 0) v = ldexp((double)m, -24);
    else if (e == 31) v = m ? NAN : INFINITY;
    else v = ldexp((double)(m + 1024), e - 25);
    return sign ? -v : v;
}

void slro_mfn_decode_f64(const uint16_t *const *planes, int n_freq, int n_step, int pitch, int W, int H,
                         double black_thr, double *phase, uint8_t *valid)
{
    const double PI = 3.14159265358979323846, TWO_PI = 2 * PI;
    double D[16];
    for (int row = 0; row < H; row++)
        for (int col = 0; col < W; col++) {
            const size_t s = (size_t)row * pitch + col, o = (size_t)row * W + col;
            const int mask = slro_half_to_double(planes[0][s]) - slro_half_to_double(planes[1][s]) > black_thr;
            int ok = mask;
            for (int f = 0; f < n_freq; f++) {
                double S = 0, C = 0;
                for (int k = 0; k < n_step; k++) {
                    const double I = slro_half_to_double(planes[2 + f * n_step + k][s]);
                    S += I * sin(TWO_PI * k / n_step);
                    C += I * cos(TWO_PI * k / n_step);
                }
                double p = atan2(-S, C);
                if (p < 0) p += TWO_PI;
                if (!(S * S + C * C > (0.25 * n_step) * (0.25 * n_step))) ok = 0;   /* modulation < 0.5 grey levels */
                D[f] = p;
            }
            for (int lvl = 1; lvl < n_freq; lvl++)
                for (int i = 0; i + lvl < n_freq; i++)
                    D[i] = (D[i] > D[i + 1]) ? (D[i] - D[i + 1]) : (D[i] - D[i + 1] + TWO_PI);
            phase[o] = mask ? D[0] / TWO_PI * 255 : 0.0;
            valid[o] = (uint8_t)ok;
        }
}

/* the same model THROUGH the rectification (slr_mfn_rectify_decode; build extension, no reference counterpart): cv::remap's
 * geometry as stereorect.cpp:26-34 uses it (CV_16SC2 + CV_16UC1 maps, 5-bit fractions, taps (sx, sy) .. (sx+1, sy+1),
 * BORDER_CONSTANT 0) with the bilinear sample evaluated in f64 -- exact: taps are binary16, weights integers below 2^11 --
 * and the decode of slro_mfn_decode_f64 on those samples.  rows [row0, row1) only (phase / valid are full [H][W] arrays). */
void slro_mfn_rect_decode_f64(const uint16_t *const *planes, int n_freq, int n_step, int pitch, int W, int H,
                              double black_thr, const int16_t *map_xy, const uint16_t *map_frac, int row0, int row1,
                              double *phase, uint8_t *valid)
{
    const double PI = 3.14159265358979323846, TWO_PI = 2 * PI;
    double D[16];
    for (int row = row0; row < row1; row++)
        for (int col = 0; col < W; col++) {
            const size_t o = (size_t)row * W + col;
            const int sx = map_xy[2 * o], sy = map_xy[2 * o + 1];
            const int f = map_frac[o] & 1023, fx = f & 31, fy = f >> 5;
            const double w[4] = {(double)((32 - fx) * (32 - fy)), (double)(fx * (32 - fy)), (double)((32 - fx) * fy), (double)(fx * fy)};
            size_t t[4];
            int in[4];
            for (int k = 0; k < 4; k++) {
                const int x = sx + (k & 1), y = sy + (k >> 1);
                in[k] = x >= 0 && x < W && y >= 0 && y < H;
                t[k] = in[k] ? (size_t)y * pitch + x : 0;
            }
#define SLRO_SAMPLE(P) ((((in[0] ? slro_half_to_double((P)[t[0]]) : 0.0) * w[0] + (in[1] ? slro_half_to_double((P)[t[1]]) : 0.0) * w[1]) \
                       + ((in[2] ? slro_half_to_double((P)[t[2]]) : 0.0) * w[2] + (in[3] ? slro_half_to_double((P)[t[3]]) : 0.0) * w[3])) / 1024.0)
            const int mask = SLRO_SAMPLE(planes[0]) - SLRO_SAMPLE(planes[1]) > black_thr;
            int ok = mask;
            for (int fq = 0; fq < n_freq; fq++) {
                double S = 0, C = 0;
                for (int k = 0; k < n_step; k++) {
                    const double I = SLRO_SAMPLE(planes[2 + fq * n_step + k]);
                    S += I * sin(TWO_PI * k / n_step);
                    C += I * cos(TWO_PI * k / n_step);
                }
                double p = atan2(-S, C);
                if (p < 0) p += TWO_PI;
                if (!(S * S + C * C > (0.25 * n_step) * (0.25 * n_step))) ok = 0;
                D[fq] = p;
            }
#undef SLRO_SAMPLE
            for (int lvl = 1; lvl < n_freq; lvl++)
                for (int i = 0; i + lvl < n_freq; i++)
                    D[i] = (D[i] > D[i + 1]) ? (D[i] - D[i + 1]) : (D[i] - D[i + 1] + TWO_PI);
            phase[o] = mask ? D[0] / TWO_PI * 255 : 0.0;
            valid[o] = (uint8_t)ok;
        }
}
