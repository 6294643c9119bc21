/*
 * slr_oracle_x87.c -- CPU ORACLE, second evaluation model (TEST INFRASTRUCTURE ONLY; same rules as slr_oracle.h).
 *
 * The reference binary is an MSVC2010 32-bit Debug build (Duke/Duke.pro:35-70, the committed build directory
 * build-Duke-Desktop_Qt_5_3_MSVC2010_OpenGL_32bit-Debug): x87 code under /fp:precise with the CRT's default 53-bit
 * precision control.  There an expression's operations run on the x87 stack with a 53-bit significand and a value is
 * rounded to its declared type only where it is ASSIGNED, CAST or PASSED to a function -- a float expression that feeds a
 * double (or a compound float expression) keeps its extra bits.  slr_oracle.c is strict IEEE (every f32 operation rounds to
 * f32).  This file restates the same reference lines under the x87 model, so that the difference can be COUNTED
 * (oracle/x87_sensitivity.py); it cannot be pinned either -- no MSVC2010 here -- but it bounds what
 * "parity unpinned" can cost.
 *
 * Model ("x87-53"): C doubles ARE the x87 stack at PC=53 for the value ranges of this path (no denormals, no overflow);
 * a single f32 operation whose result is stored at once is the same in both models (double rounding 53 -> 24 bits is
 * innocuous for + - * / sqrt since 53 >= 2*24 + 2), so only COMPOUND expressions differ.  Per reference line:
 *   mfreconstruct.cpp:257-261  P[count] = atan(float(q)) + PI        float + float evaluated at 53 bits, stored to a DOUBLE:
 *                                                                     the sum is exact, no f32 rounding (strict: rounded to f32)
 *   mfreconstruct.cpp:251      P[count] = 3*PI/2                      exact in 53 bits (strict: 3*PI rounds to f32 first)
 *   mfreconstruct.cpp:265-266  P12, P23                               double arithmetic, narrowed once: as the strict oracle,
 *                                                                     but on the P above
 *   mfreconstruct.cpp:267      P123 = P12 - P23 (+ 2*PI)              one rounding to f32 at the store (strict: two in the + branch)
 *   mfreconstruct.cpp:268      phase = P123/(2*PI)*255                division and product at 53 bits, one rounding (strict: two)
 *   mfreconstruct.cpp:295      fabs(cam1Pix[0] - cam2Pix[0]) < 0.1    the difference of two floats at 53 bits is (nearly always)
 *                                                                     exact and goes to fabs(double) unrounded
 *   mfreconstruct.cpp:299      camPixelUDL.x - camPixelUDR.x          stored to a double: unrounded
 *   utilities.cpp:21           sqrt(v0*v0 + v1*v1 + v2*v2)            sum of squares at 53 bits, rounded once when passed to
 *                                                                     sqrt(float)
 *   utilities.cpp:51-52        (p.x - cc.x) / fc.x                    one rounding at the store
 *   utilities.cpp:404-408      Vec3f::dot: s += a[i]*b[i]             the product is exact in 53 bits; one rounding per step
 *   utilities.cpp:412          denom = a*b - c*c                      one rounding
 *   utilities.cpp:417-418      s, t                                   quotients and products at 53 bits, one rounding
 * atan: MSVC2010's x86 <math.h> defines atanf(x) as ((float)atan((double)x)) and <cmath>'s atan(float) calls it -- i.e. the
 * correctly rounded float of a double-precision atan (up to a ~2^-29 chance of a double-rounding miss per value); glibc's
 * atanf is a separate implementation.  slro_atan_table builds either, and +-1-ulp perturbations of it.
 */
#include "slr_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* tab[q + 255] = atan of the integer quotient q in [-255, 255] (Q1: the quotient is a C integer division).
 * mode 0: glibc atanf (what slr_oracle.c calls); 1: (float)atan((double)q) -- MSVC2010 x86's atanf;
 * 2 / 3: mode 1 with every |entry| moved one ulp away from / towards zero; >= 4: mode 1 with a seeded random -1 / 0 / +1 ulp
 * per |q| (seed = mode).  Odd symmetry is kept (both libms are odd-symmetric). */
void slro_atan_table(int mode, float tab[511])
{
    unsigned long long st = 0x9E3779B97F4A7C15ull * (unsigned long long)(mode + 1);
    tab[255] = 0.0f;
    for (int q = 1; q <= 255; q++) {
        float a = mode == 0 ? atanf((float)q) : (float)atan((double)q);
        if (mode == 2) a = nextafterf(a, 4.0f);
        else if (mode == 3) a = nextafterf(a, 0.0f);
        else if (mode >= 4) {
            st ^= st << 13; st ^= st >> 7; st ^= st << 17;
            const int r = (int)((st >> 33) % 3ull);
            if (r == 1) a = nextafterf(a, 4.0f); else if (r == 2) a = nextafterf(a, 0.0f);
        }
        tab[255 + q] = a;
        tab[255 - q] = -a;
    }
}

/* mfreconstruct.cpp:246-261 under either model.  Returns 0 for the undefined case (Q5 rule as in slr_oracle.c). */
int slro_wrapped_phase_ev(int G1, int G2, int G3, int G4, const float *atab, int x87, double *P)
{
    const float PI = SLRO_PI_F;
    if (G4 == G2 && G1 > G3)       { *P = 0.0; return 1; }
    else if (G4 == G2 && G1 < G3)  { *P = (double)PI; return 1; }
    else if (G1 == G3 && G4 > G2)  { *P = x87 ? 3.0 * (double)PI / 2.0 : (double)(3 * PI / 2); return 1; }
    else if (G1 == G3 && G4 < G2)  { *P = (double)(PI / 2); return 1; }
    else if (G1 == G3 && G4 == G2) { *P = 0.0; return 0; }
    const float a = atab[(G4 - G2) / (G1 - G3) + 255];
    if (G1 < G3)                   *P = x87 ? (double)a + (double)PI : (double)(a + PI);
    else if (G1 > G3 && G4 > G2)   *P = x87 ? (double)a + 2.0 * (double)PI : (double)(a + 2 * PI);
    else                           *P = (double)a;
    return 1;
}

/* mfreconstruct.cpp:265-268 under either model */
float slro_heterodyne_ev(const double P[3], int x87)
{
    const float PI = SLRO_PI_F;
    const float P12 = (float)((P[0] > P[1]) ? (P[0] - P[1]) : (P[0] - P[1] + (double)(2 * PI)));
    const float P23 = (float)((P[1] > P[2]) ? (P[1] - P[2]) : (P[1] - P[2] + (double)(2 * PI)));
    if (!x87) {
        const float P123 = (P12 > P23) ? (P12 - P23) : (P12 - P23 + 2 * PI);
        return P123 / (2 * PI) * 255;
    }
    const double d = (double)P12 - (double)P23;
    const float P123 = (float)((P12 > P23) ? d : d + 2.0 * (double)PI);
    return (float)((double)P123 / (2.0 * (double)PI) * 255.0);
}

void slro_mf_decode_ev(const uint8_t *const planes[SLRO_MF_PLANES], int pitch, int W, int H, int black_thr,
                       const float *atab, int x87, float *phase, uint8_t *valid)
{
    for (int row = 0; row < H; row++)
        for (int col = 0; col < W; col++) {
            const size_t s = (size_t)row * pitch + col, o = (size_t)row * W + col;
            const float whiteVal = (float)planes[0][s], blackVal = (float)planes[1][s];
            const int mask = (whiteVal - blackVal > (float)black_thr) ? 1 : 0;
            float ph = 0.0f;
            int ok = mask;
            if (mask) {
                double P[3];
                for (int c = 0; c < 3; c++) {
                    const int G1 = planes[4 * c + 2][s], G2 = planes[4 * c + 3][s];
                    const int G3 = planes[4 * c + 4][s], G4 = planes[4 * c + 5][s];
                    if (!slro_wrapped_phase_ev(G1, G2, G3, G4, atab, x87, &P[c])) ok = 0;
                }
                ph = slro_heterodyne_ev(P, x87);
            }
            phase[o] = ph;
            valid[o] = (uint8_t)ok;
        }
}

/* The device code (decode_common.hpp, het_finish_x87) replaces the f64 division of mfreconstruct.cpp:268 under this model by a
 * multiply and two fused multiply-adds with constants (Markstein's quotient by a constant).  This restates THOSE operations --
 * test infrastructure for the identity, not a second decode -- and counts the integers d in [lo, hi), d = P123 * 2^24 before
 * the f32 rounding of its store, for which they do not reproduce (float)((double)P123 / (2.0 * PI) * 255.0) bit for bit. */
long slro_x87_quotient_mismatches(long lo, long hi)
{
    const float PI = SLRO_PI_F;
    const double c = (double)(2 * PI), cq = c * 16777216.0, rq = (1.0 / c) * (1.0 / 16777216.0);
    long bad = 0;
    for (long d = lo; d < hi; d++) {
        const float F123 = (float)d;                               /* the one rounding of the x87 store, at the 2^24 scale */
        const float P123 = F123 * (1.0f / 16777216.0f);            /* exact */
        const float want = (float)((double)P123 / (2.0 * (double)PI) * 255.0);
        const double x = (double)F123;
        const double q0 = x * rq;
        const double r = fma(-q0, cq, x);
        const double q = fma(r, rq, q0);
        const float got = (float)(q * 255.0);
        if (memcmp(&got, &want, sizeof got) != 0) bad++;
    }
    return bad;
}

/* mfreconstruct.cpp:284-333 under either model: the match predicate (:295) and the disparity (:299) */
void slro_mf_triangulate_rows_ev(const float *phaseL, const uint8_t *validL, const float *phaseR, const uint8_t *validR,
                                 int W, int H, int row0, int row1, const slro_camera *camL, const slro_camera *camR,
                                 const double Q[16], const float *T, int x87, float *xyz, uint8_t *has, int32_t *match_k)
{
    (void)H;
    for (int i = row0; i < row1; i++)
        for (int j = 0; j < W; j++) {
            const size_t o = (size_t)i * W + j;
            xyz[3 * o] = xyz[3 * o + 1] = xyz[3 * o + 2] = 0.0f;
            has[o] = 0;
            if (match_k) match_k[o] = -1;
            if (!validL[o]) continue;
            const float pl = phaseL[o];
            for (int k = 0; k < W; k++) {
                const size_t r = (size_t)i * W + k;
                if (!validR[r]) continue;
                const double diff = x87 ? (double)pl - (double)phaseR[r] : (double)(float)(pl - phaseR[r]);
                if (fabs(diff) < 0.1) {
                    float ulx, uly, urx, ury, X[3];
                    slro_undistort_point((float)j, (float)i, camL, &ulx, &uly);
                    slro_undistort_point((float)k, (float)i, camR, &urx, &ury);
                    double p[4] = {ulx, uly, x87 ? (double)ulx - (double)urx : (double)(float)(ulx - urx), 1};
                    slro_reproject(Q, p, X);
                    if (T) { float Y[3]; slro_apply_T(T, X, Y); X[0] = Y[0]; X[1] = Y[1]; X[2] = Y[2]; }
                    xyz[3 * o] = X[0]; xyz[3 * o + 1] = X[1]; xyz[3 * o + 2] = X[2];
                    has[o] = 1;
                    if (match_k) match_k[o] = k;
                    break;
                }
            }
        }
}

/* ---- GRAY_ONLY: utilities.cpp:19-25, 51-53, 399-425 under the x87 model ------------------------------------------------- */
static float dot3_x87(const float a[3], const float b[3])
{
    float s = 0;                                   /* Matx::dot: s += a[i]*b[i], the product exact at 53 bits */
    for (int i = 0; i < 3; i++) s = (float)((double)s + (double)a[i] * (double)b[i]);
    return s;
}

int slro_line_line_intersection_x87(const float p1[3], const float v1[3], const float p2[3], const float v2[3], float out[3])
{
    const float v12[3] = {p1[0] - p2[0], p1[1] - p2[1], p1[2] - p2[2]};
    const float a = dot3_x87(v1, v1), c = dot3_x87(v2, v2), b = dot3_x87(v1, v2);
    const float d1 = dot3_x87(v12, v1), d2 = dot3_x87(v12, v2);
    const float denom = (float)((double)a * (double)c - (double)b * (double)b);
    if (fabsf(denom) < 0.1) return 0;
    const float s = (float)(((double)b / (double)denom) * (double)d2 - ((double)c / (double)denom) * (double)d1);
    const float t = (float)(-((double)b / (double)denom) * (double)d1 + ((double)a / (double)denom) * (double)d2);
    for (int k = 0; k < 3; k++) {                  /* Point3f operators: one stored f32 operation each, as in the strict model */
        const float x = p1[k] + s * v1[k];
        const float y = p2[k] + t * v2[k];
        out[k] = (float)(0.5 * (double)(x + y));
    }
    return 1;
}

static void pixel_ray_x87(uint32_t item, const slro_camera *cam, const float pos[3], float ray[3])
{
    float ux, uy, pt[3];
    slro_undistort_point((float)(item & 0xFFFFu), (float)(item >> 16), cam, &ux, &uy);
    pt[0] = (float)(((double)ux - (double)cam->cc[0]) / (double)cam->fc[0]);
    pt[1] = (float)(((double)uy - (double)cam->cc[1]) / (double)cam->fc[1]);
    pt[2] = 1;
    slro_cam2world(cam, pt);                       /* OpenCV's GEMM: library code, not the application's x87 code */
    ray[0] = pos[0] - pt[0]; ray[1] = pos[1] - pt[1]; ray[2] = pos[2] - pt[2];
    const float ss = (float)((double)ray[0] * (double)ray[0] + (double)ray[1] * (double)ray[1] + (double)ray[2] * (double)ray[2]);
    const double mag = (double)(float)sqrt((double)ss);          /* sqrt(float) -> sqrtf = (float)sqrt((double)x) */
    const float dv = (float)(0.000001 > mag ? 0.000001 : mag);
    ray[0] /= dv; ray[1] /= dv; ray[2] /= dv;
}

/* reconstruct.cpp:428-480 with the x87 forms above (structure as slro_ray_triangulate) */
void slro_ray_triangulate_x87(const int32_t *offL, const uint32_t *itemsL, const int32_t *offR, const uint32_t *itemsR,
                              const slro_camera *camL, const slro_camera *camR, const float *T, int scan_w, int scan_h,
                              float *xyz_sum, uint8_t *count)
{
    float posL[3] = {0, 0, 0}, posR[3] = {0, 0, 0};
    slro_cam2world(camL, posL);
    slro_cam2world(camR, posR);
    memset(xyz_sum, 0, sizeof(float) * 3 * (size_t)scan_w * scan_h);
    memset(count, 0, (size_t)scan_w * scan_h);
    for (int i = 0; i < scan_w; i++)
        for (int j = 0; j < scan_h; j++) {
            const long b = (long)i * scan_h + j;
            const int n1 = offL[b + 1] - offL[b], n2 = offR[b + 1] - offR[b];
            if (n1 == 0 || n2 == 0) continue;
            const size_t o = (size_t)j * scan_w + i;
            for (int c1 = 0; c1 < n1; c1++) {
                float r1[3];
                pixel_ray_x87(itemsL[offL[b] + c1], camL, posL, r1);
                for (int c2 = 0; c2 < n2; c2++) {
                    float r2[3], X[3];
                    pixel_ray_x87(itemsR[offR[b] + c2], camR, posR, r2);
                    if (!slro_line_line_intersection_x87(posL, r1, posR, r2, X)) continue;
                    if (T) { float Y[3]; slro_apply_T(T, X, Y); X[0] = Y[0]; X[1] = Y[1]; X[2] = Y[2]; }
                    const uint8_t num = count[o];
                    if (num == 0) {
                        xyz_sum[3 * o] = X[0]; xyz_sum[3 * o + 1] = X[1]; xyz_sum[3 * o + 2] = X[2];
                        count[o] = 1;
                    } else {
                        xyz_sum[3 * o] = X[0] + xyz_sum[3 * o];
                        xyz_sum[3 * o + 1] = X[1] + xyz_sum[3 * o + 1];
                        xyz_sum[3 * o + 2] = X[2] + xyz_sum[3 * o + 2];
                        count[o] = (uint8_t)(num + 1);
                    }
                }
            }
        }
}
