// kernels_mfn.hip -- BUILD EXTENSION, no reference counterpart (BASELINE config 5, SURVEY 8d: "parity unpinned"):
// generalised multi-frequency phase-shift decode for n_freq frequencies x n_step equally spaced shifts with fp16
// image planes and f32 accumulation.  gfx950 (MI355X) only.
//
// The reference is hard-wired to 3 frequencies x 4 steps of u8 (mfreconstruct.cpp:21-22,237-242); this kernel keeps
// its STRUCTURE -- shadow mask from white/black (:190-207), one wrapped phase per frequency, a heterodyne cascade of
// neighbouring differences with the reference's "a > b ? a-b : a-b+2pi" rule (:265-267), output scaled to 0..255
// (:268) so that K4's 0.1 threshold keeps its meaning -- and replaces what cannot generalise:
//   wrapped phase   I_k = A + B cos(phi + 2 pi k / N)  =>  phi = atan2(-sum I_k sin(2 pi k/N), sum I_k cos(2 pi k/N)),
//                   brought to [0, 2 pi)                   (the reference's 4-step integer-quotient atan is Q1/Q2)
//   PI              the f32 nearest to pi                   (the reference's 3.1416f is Q3)
//   cascade         level 0: P_0..P_{F-1};  level l+1: D_i = wrap(D_i - D_{i+1});  the single value of level F-1
// Pure streaming: (2 + F*N) fp16 reads + 4 + 1 bytes written per pixel (73 B/px at 4 x 8) -> HBM-bound.
#include "slr_device.hpp"

#include <hip/hip_fp16.h>
#include <math.h>

#pragma clang diagnostic ignored "-Wpass-failed"   // the run-time (FS = NS = 0) variant cannot fully unroll its plane loops

namespace slr {

constexpr float kTruePI = 3.14159265358979323846f;
constexpr float kTrue2PI = 2 * kTruePI;

struct MfnPlanes { const uint16_t *p[SLR_MFN_MAX_PLANES]; };
struct MfnTrig { float cs[SLR_MFN_MAX_STEPS], sn[SLR_MFN_MAX_STEPS]; };

__device__ __forceinline__ float h2f(unsigned short b) { return __half2float(__ushort_as_half(b)); }

// V pixels per thread: 4 (8-byte loads; W % 4 == 0, aligned planes) or 1.  FS x NS != 0: compile-time frequency and
// step counts -- the plane loop unrolls completely and all 2 + FS*NS loads of a thread are in flight at once (the
// run-time loop keeps one load per iteration in flight, which costs ~25 % of the HBM rate at 4 x 8).
template <int V, int FS, int NS>
__global__ __launch_bounds__(256) void mfn_decode_kernel(MfnPlanes pl, MfnTrig tr, int n_freq_rt, int n_step_rt, int pitch, int W,
                                                         int H, float black_thr, float *__restrict__ phase,
                                                         uint8_t *__restrict__ valid)
{
    const int n_freq = FS ? FS : n_freq_rt, n_step = NS ? NS : n_step_rt;
    const unsigned gpr = (unsigned)(W / V);
    const unsigned total = gpr * (unsigned)H;
    for (unsigned g = blockIdx.x * 256u + threadIdx.x; g < total; g += gridDim.x * 256u) {
        const unsigned row = g / gpr, col0 = (g - row * gpr) * V;
        const size_t so = (size_t)row * pitch + col0, oo = (size_t)row * W + col0;
        float wh[V], bk[V];
        auto load = [&](int p, float out[V]) {
            if constexpr (V == 4) {
                const uint2 w = *reinterpret_cast<const uint2 *>(pl.p[p] + so);
                out[0] = h2f((unsigned short)(w.x & 0xFFFFu)); out[1] = h2f((unsigned short)(w.x >> 16));
                out[2] = h2f((unsigned short)(w.y & 0xFFFFu)); out[3] = h2f((unsigned short)(w.y >> 16));
            } else {
                out[0] = h2f(pl.p[p][so]);
            }
        };
        load(0, wh);
        load(1, bk);
        float D[V][SLR_MFN_MAX_FREQ];
        bool ok[V];
        const float mod2 = (0.25f * n_step) * (0.25f * n_step);        // (B_min * N / 2)^2 with B_min = 0.5
#pragma unroll
        for (int v = 0; v < V; v++) ok[v] = wh[v] - bk[v] > black_thr;
#pragma unroll
        for (int f = 0; f < n_freq; f++) {
            float S[V], C[V];
#pragma unroll
            for (int v = 0; v < V; v++) S[v] = C[v] = 0.0f;
#pragma unroll
            for (int k = 0; k < n_step; k++) {
                float I[V];
                load(2 + f * n_step + k, I);
#pragma unroll
                for (int v = 0; v < V; v++) { S[v] += I[v] * tr.sn[k]; C[v] += I[v] * tr.cs[k]; }
            }
#pragma unroll
            for (int v = 0; v < V; v++) {
                float p = atan2f(-S[v], C[v]);
                if (p < 0.0f) p += kTrue2PI;
                // modulation B = 2/N * |DFT bin| below half a grey level: the phase is noise (cf. Q5) -> invalid
                ok[v] = ok[v] && (S[v] * S[v] + C[v] * C[v] > mod2);
#pragma unroll
                for (int q = 0; q < SLR_MFN_MAX_FREQ; q++) if (q == f) D[v][q] = p;   // static indexing keeps D in registers
            }
        }
#pragma unroll
        for (int lvl = 1; lvl < SLR_MFN_MAX_FREQ; lvl++) {
            if (lvl < n_freq) {
#pragma unroll
                for (int i = 0; i + 1 < SLR_MFN_MAX_FREQ; i++) {
                    if (i + lvl < n_freq) {
#pragma unroll
                        for (int v = 0; v < V; v++) {
                            const float a = D[v][i], b = D[v][i + 1];
                            D[v][i] = (a > b) ? (a - b) : (a - b + kTrue2PI);
                        }
                    }
                }
            }
        }
        float out[V];
        unsigned vw = 0;
#pragma unroll
        for (int v = 0; v < V; v++) {
            out[v] = (wh[v] - bk[v] > black_thr) ? D[v][0] / kTrue2PI * 255 : 0.0f;
            vw |= (ok[v] ? 1u : 0u) << (8 * v);
        }
        if constexpr (V == 4) {
            *reinterpret_cast<float4 *>(phase + oo) = make_float4(out[0], out[1], out[2], out[3]);
            *reinterpret_cast<unsigned *>(valid + oo) = vw;
        } else {
            phase[oo] = out[0];
            valid[oo] = (uint8_t)vw;
        }
    }
}

hipError_t launch_mfn_decode(const uint16_t *const *planes, int n_freq, int n_step, int pitch, int W, int H, float black_thr,
                             float *phase, uint8_t *valid, hipStream_t s)
{
    MfnPlanes pl;
    const int np = 2 + n_freq * n_step;
    bool a4 = W % 4 == 0 && pitch % 4 == 0 && (uintptr_t)phase % 16 == 0 && (uintptr_t)valid % 4 == 0;
    for (int i = 0; i < SLR_MFN_MAX_PLANES; i++) {
        pl.p[i] = i < np ? planes[i] : nullptr;
        if (i < np) a4 = a4 && (uintptr_t)planes[i] % 8 == 0;
    }
    MfnTrig tr;
    for (int k = 0; k < SLR_MFN_MAX_STEPS; k++) {
        const double a = 2.0 * 3.14159265358979323846 * k / (double)n_step;
        tr.cs[k] = k < n_step ? (float)cos(a) : 0.0f;
        tr.sn[k] = k < n_step ? (float)sin(a) : 0.0f;
    }
    const size_t groups = a4 ? (size_t)(W / 4) * H : (size_t)W * H;
    const unsigned blocks = (unsigned)((groups + 255) / 256 < 65536 ? (groups + 255) / 256 : 65536);
#define SLR_MFN(V, FS, NS)                                                                                       \
    SLR_LAUNCH((mfn_decode_kernel<V, FS, NS>), dim3(blocks ? blocks : 1), dim3(256), 0, s, pl, tr, n_freq, n_step, \
                       pitch, W, H, black_thr, phase, valid)
    if (a4 && n_freq == 4 && n_step == 8) SLR_MFN(4, 4, 8);          // BASELINE config 5
    else if (a4 && n_freq == 3 && n_step == 4) SLR_MFN(4, 3, 4);     // the reference's own pattern count
    else if (a4) SLR_MFN(4, 0, 0);
    else SLR_MFN(1, 0, 0);
#undef SLR_MFN
    return hipGetLastError();
}

}  // namespace slr
