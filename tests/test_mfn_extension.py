"""BUILD EXTENSION (BASELINE config 5): generalised n_freq x n_step fp16 decode, slr_mfn_decode.  The reference has no
counterpart (3 x 4 steps of u8 only, mfreconstruct.cpp:21-22,237-242), so parity is UNPINNED by construction: the
GPU kernel (f32) is checked against the fp64 model in oracle/ and the model against the analytic phase of the
synthetic scene."""
import numpy as np
import pytest
import torch


def _circ(a, b, period=255.0):
    d = np.abs(a - b) % period
    return np.minimum(d, period - d)


@pytest.mark.parametrize("F,N", [(4, 8), (3, 4), (2, 3), (6, 16)])
def test_f64_model_recovers_the_analytic_phase(oracle, synth, F, N):
    W, H = 256, 40
    st = synth.render_mfn_stack(W, H, F, N, noise=0.0)
    uL, uR, _ = synth.projector_columns(W, H, W)
    for cam, u in enumerate((uL, uR)):
        ph, v = oracle.mfn_decode_f64(st[cam].numpy(), F, N, 40.0)
        fr = synth.MFN_FREQ
        c = [1.0]                                         # cascade coefficients: alternating binomials
        for _ in range(F - 1):
            c = [a - b for a, b in zip(c + [0.0], [0.0] + c)]
        beat = sum(ci * fi for ci, fi in zip(c, fr[:F]))
        exp = (u.numpy() * beat / W % 1.0) * 255.0
        ok = v.astype(bool)
        assert ok.sum() > 0.5 * ok.size
        # fp16 pixel quantisation (0.125 grey levels above 128) limits the accuracy; binomial error growth with F
        assert np.percentile(_circ(ph[ok], exp[ok]), 99) < 0.15 * 2 ** F


def test_model_plane_order_and_mask(oracle):
    """hand-made 1-pixel stacks: shift k of frequency f lives at plane 2 + f*N + k; mask is white - black > thr"""
    F, N = 2, 4
    def stack(ph0, ph1, white=200.0, black=50.0):
        p = np.zeros((2 + F * N, 1, 1), np.float16)
        p[0], p[1] = white, black
        for k in range(N):
            p[2 + k] = 100 + 50 * np.cos(ph0 + 2 * np.pi * k / N)
            p[2 + N + k] = 100 + 50 * np.cos(ph1 + 2 * np.pi * k / N)
        return p
    ph, v = oracle.mfn_decode_f64(stack(2.0, 0.5), F, N, 40.0)
    assert v[0, 0] == 1 and abs(ph[0, 0] - 1.5 / (2 * np.pi) * 255) < 0.05
    ph, v = oracle.mfn_decode_f64(stack(0.5, 2.0), F, N, 40.0)              # a < b: + 2 pi
    assert abs(ph[0, 0] - (2 * np.pi - 1.5) / (2 * np.pi) * 255) < 0.05
    ph, v = oracle.mfn_decode_f64(stack(2.0, 0.5, white=80.0), F, N, 40.0)  # shadow
    assert v[0, 0] == 0 and ph[0, 0] == 0.0
    flat = stack(2.0, 0.5); flat[2:2 + N] = 100.0                            # frequency 0 has no modulation
    ph, v = oracle.mfn_decode_f64(flat, F, N, 40.0)
    assert v[0, 0] == 0


@pytest.mark.gpu
@pytest.mark.parametrize("W,H,F,N", [(256, 64, 4, 8), (130, 33, 3, 5), (64, 16, 6, 16), (512, 48, 2, 3), (128, 16, 3, 4)])
def test_gpu_mfn_decode_vs_f64_model(ctx, oracle, synth, W, H, F, N):
    st = synth.render_mfn_stack(W, H, F, N, noise=0.5, seed=W)
    for cam in range(2):
        exp, ev = oracle.mfn_decode_f64(st[cam].numpy(), F, N, 40.0)
        for planes in (st[cam].numpy(), st[cam].cuda()):                 # host and device entry
            ph, v = ctx.mfn_decode(planes, F, N, 40.0)
            ctx.synchronize()
            ph = ph.cpu().numpy() if hasattr(ph, "cpu") else ph
            v = v.cpu().numpy() if hasattr(v, "cpu") else v
            assert np.array_equal(v, ev)
            d = np.abs(ph.astype(np.float64) - exp)
            tol = 2e-3 + 1e-4 * np.abs(exp)                              # north_star: 1e-4 relative (+ f32 floor)
            bad = d > tol
            # the only admissible large differences are wrap flips at an a > b comparison decided differently in
            # f32 and f64 (|diff| = a multiple of 255): vanishing fraction
            flips = bad & (_circ(ph, exp) <= tol)
            assert (bad & ~flips).sum() == 0, float(d[bad & ~flips].max())
            assert flips.sum() <= 1e-3 * ph.size


@pytest.mark.gpu
def test_gpu_mfn_then_match_triangulate(ctx, oracle, synth):
    """config-5 path: generalised decode of both cameras, then the unchanged K4 (match + Q triangulation)"""
    from util import calib_parts, bits_equal
    W, H, F, N = 512, 24, 4, 8
    calib, _ = synth.make_calibration(W, H)
    ctx.set_calibration(calib)
    st = synth.render_mfn_stack(W, H, F, N, noise=0.25, seed=3).cuda()
    dec = [ctx.mfn_decode(st[cam], F, N, 40.0) for cam in range(2)]
    xyz, has, mk = ctx.mf_triangulate(dec[0][0], dec[0][1], dec[1][0], dec[1][1])
    ctx.synchronize()
    camL, camR, Q, T = calib_parts(oracle, calib)
    hp = [(d[0].cpu().numpy(), d[1].cpu().numpy()) for d in dec]
    exyz, ehas, emk = oracle.mf_triangulate(hp[0][0], hp[0][1], hp[1][0], hp[1][1], camL, camR, Q, T)
    assert bits_equal(mk.cpu().numpy(), emk) and bits_equal(has.cpu().numpy(), ehas) and bits_equal(xyz.cpu().numpy(), exyz)
    assert ehas.sum() > 0.3 * ehas.size


@pytest.mark.gpu
@pytest.mark.parametrize("algo", [0, 1, 3])
def test_gpu_row_bands_equal_the_whole_frame(ctx, oracle, synth, slr, algo):
    """config-5 sharding: decode + match of row bands (slr_mf_triangulate_rows: absolute rows for the reprojection and the
    undistortion tables) concatenated == the whole frame in one call == the oracle, for every K4 form"""
    from util import calib_parts, bits_equal
    W, H, F, N = 384, 37, 4, 8
    calib, _ = synth.make_calibration(W, H, with_T=True)
    ctx.set_calibration(calib)
    ctx.set_option(slr.capi.OPT_MF_MATCH_ALGO, algo)
    st = synth.render_mfn_stack(W, H, F, N, noise=0.25, seed=9).cuda()
    full = [ctx.mfn_decode(st[cam], F, N, 40.0) for cam in range(2)]
    fx, fh, fk = ctx.mf_triangulate(full[0][0], full[0][1], full[1][0], full[1][1])
    parts = []
    for r0, r1 in ((0, 10), (10, 11), (11, 30), (30, 37)):
        dec = [ctx.mfn_decode([st[cam, p, r0:r1] for p in range(2 + F * N)], F, N, 40.0) for cam in range(2)]
        parts.append(ctx.mf_triangulate(dec[0][0], dec[0][1], dec[1][0], dec[1][1], row0=r0, image_h=H))
    ctx.synchronize()
    ctx.set_option(slr.capi.OPT_MF_MATCH_ALGO, 0)
    bx = torch.cat([p[0] for p in parts]); bh = torch.cat([p[1] for p in parts]); bk = torch.cat([p[2] for p in parts])
    assert torch.equal(bx, fx) and torch.equal(bh, fh) and torch.equal(bk, fk)
    camL, camR, Q, T = calib_parts(oracle, calib)
    exyz, ehas, emk = oracle.mf_triangulate(full[0][0].cpu().numpy(), full[0][1].cpu().numpy(), full[1][0].cpu().numpy(),
                                            full[1][1].cpu().numpy(), camL, camR, Q, T)
    assert bits_equal(fk.cpu().numpy(), emk) and bits_equal(fx.cpu().numpy(), exyz)
