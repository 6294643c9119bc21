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


# ---- through the rectification (slr_mfn_rectify_decode): BASELINE config 5 on the north_star path ------------------------------
def _close_to_model(ph, v, exp, ev):
    """the comparison of test_gpu_mfn_decode_vs_f64_model, with the valid flags allowed to differ where the f32 and f64 modulation /
    shadow tests sit on their thresholds (counted)"""
    d = np.abs(ph.astype(np.float64) - exp)
    tol = 2e-3 + 1e-4 * np.abs(exp)
    bad = d > tol
    flips = bad & (_circ(ph, exp) <= tol)
    vdiff = v != ev
    assert (bad & ~flips & ~vdiff).sum() == 0, float(d[bad & ~flips & ~vdiff].max())
    assert flips.sum() <= 1e-3 * ph.size and vdiff.sum() <= 1e-4 * ph.size + 1, (int(flips.sum()), int(vdiff.sum()))


def test_f64_rect_model_is_remap_then_decode(oracle, synth):
    """the rectified model == the unrectified model on planes remapped in f64 (identity maps: the input itself; half-pixel maps:
    the mean of two neighbours; out-of-image taps: 0)"""
    W, H, F, N = 64, 20, 3, 4
    st = synth.render_mfn_stack(W, H, F, N, noise=0.0)[0].numpy()
    mx, mf = synth.identity_maps(W, H)
    p0, v0 = oracle.mfn_decode_f64(st, F, N, 40.0)
    p1, v1 = oracle.mfn_rect_decode_f64(st, F, N, 40.0, mx.numpy(), mf.numpy())
    assert np.array_equal(p0, p1) and np.array_equal(v0, v1)
    mx, mf = synth.identity_maps(W, H, fx=16)                          # half a pixel to the right
    p2, v2 = oracle.mfn_rect_decode_f64(st, F, N, 40.0, mx.numpy(), mf.numpy())
    sh = st.astype(np.float64)
    half = np.zeros_like(sh)
    half[:, :, :-1] = 0.5 * (sh[:, :, :-1] + sh[:, :, 1:])
    half[:, :, -1] = 0.5 * sh[:, :, -1]                                # the right neighbour is outside: BORDER_CONSTANT 0
    # (not representable as float16 planes, so decode by hand with the same formulas on one row)
    row = 7
    for col in (0, 5, W - 1):
        I = half[:, row, col]
        mask = I[0] - I[1] > 40.0
        assert bool(v2[row, col]) <= bool(mask)
        if mask:
            D = []
            for f in range(F):
                k = np.arange(N)
                S, Cc = (I[2 + f * N:2 + (f + 1) * N] * np.sin(2 * np.pi * k / N)).sum(), (I[2 + f * N:2 + (f + 1) * N] * np.cos(2 * np.pi * k / N)).sum()
                p = np.arctan2(-S, Cc)
                D.append(p + 2 * np.pi if p < 0 else p)
            for lvl in range(1, F):
                for i in range(F - lvl):
                    D[i] = D[i] - D[i + 1] if D[i] > D[i + 1] else D[i] - D[i + 1] + 2 * np.pi
            assert abs(p2[row, col] - D[0] / (2 * np.pi) * 255) < 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize("W,H,F,N", [(256, 64, 4, 8), (132, 33, 3, 5), (64, 16, 6, 16), (131, 20, 2, 3)])
def test_gpu_mfn_rectify_decode_vs_f64_model(ctx, oracle, synth, W, H, F, N):
    st = synth.render_mfn_stack(W, H, F, N, noise=0.5, seed=W + 1)
    for cam in range(2):
        mx, mf = synth.make_rectify_maps(W, H, cam, strength=3.0)      # strong enough that taps leave the image at the borders
        ctx.set_rectify_maps(cam, mx.numpy(), mf.numpy())
        exp, ev = oracle.mfn_rect_decode_f64(st[cam].numpy(), F, N, 40.0, mx.numpy(), mf.numpy())
        for planes in (st[cam].numpy(), st[cam].cuda()):                # host and device entry
            ph, v = ctx.mfn_rectify_decode(cam, planes, F, N, 40.0)
            ctx.synchronize()
            ph = ph.cpu().numpy() if hasattr(ph, "cpu") else ph
            v = v.cpu().numpy() if hasattr(v, "cpu") else v
            _close_to_model(ph, v, exp, ev)
            assert (v != 0).sum() > 0.3 * v.size
    # identity maps: the rectifying decode IS the plain decode (w00 = 1024: the sample is the tap, exactly)
    mx, mf = synth.identity_maps(W, H)
    ctx.set_rectify_maps(0, mx.numpy(), mf.numpy())
    a = ctx.mfn_rectify_decode(0, st[0].cuda(), F, N, 40.0)
    b = ctx.mfn_decode(st[0].cuda(), F, N, 40.0)
    ctx.synchronize()
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


@pytest.mark.gpu
def test_gpu_mfn_rectified_row_bands_with_source_windows(ctx, oracle, synth):
    """config 5's 8-GPU split: every band decodes its destination rows from ONLY the source rows slr_rectify_source_rows names
    (a copy of that window, nothing else of the frame on the 'device'), then matches them with absolute rows; the bands
    concatenated == the whole frame, bit for bit -- decode and XYZ"""
    W, H, F, N = 384, 61, 4, 8
    np_ = 2 + F * N
    calib, _ = synth.make_calibration(W, H, with_T=True)
    ctx.set_calibration(calib)
    maps = [synth.make_rectify_maps(W, H, cam, strength=4.0) for cam in range(2)]
    for cam in range(2):
        ctx.set_rectify_maps(cam, maps[cam][0].numpy(), maps[cam][1].numpy())
    st = synth.render_mfn_stack(W, H, F, N, noise=0.25, seed=11).cuda()
    full = [ctx.mfn_rectify_decode(cam, st[cam], F, N, 40.0) for cam in range(2)]
    fx, fh, fk = ctx.mf_triangulate(full[0][0], full[0][1], full[1][0], full[1][1])
    ctx.synchronize()
    world = 8
    band = (H + world - 1) // world
    parts, windows = [], []
    for r in range(world):
        r0, r1 = min(H, r * band), min(H, (r + 1) * band)
        if r1 <= r0:
            continue
        dec = []
        for cam in range(2):
            s0, sn = ctx.rectify_source_rows(cam, r0, r1 - r0)
            windows.append((r0, r1, s0, sn))
            assert 0 <= s0 and s0 + sn <= H and sn >= 1
            window = [st[cam, p, s0:s0 + sn].clone() for p in range(np_)]     # what this GPU would hold of the frame
            dec.append(ctx.mfn_rectify_decode(cam, window, F, N, 40.0, W=W, H=H, row0=r0, rows=r1 - r0, src_row0=s0))
        parts.append((r0, r1, dec, ctx.mf_triangulate(dec[0][0], dec[0][1], dec[1][0], dec[1][1], row0=r0, image_h=H)))
    ctx.synchronize()
    for cam in range(2):
        assert torch.equal(torch.cat([p[2][cam][0] for p in parts]), full[cam][0])
        assert torch.equal(torch.cat([p[2][cam][1] for p in parts]), full[cam][1])
    assert torch.equal(torch.cat([p[3][0] for p in parts]), fx) and torch.equal(torch.cat([p[3][1] for p in parts]), fh)
    assert torch.equal(torch.cat([p[3][2] for p in parts]), fk)
    assert any(sn < H for (_, _, _, sn) in windows) and fh.float().mean().item() > 0.2
    # the windows are the oracle-side min / max of the maps over the band
    for (r0, r1, s0, sn), cam in zip(windows[:2], (0, 1)):
        sy = maps[cam][0].numpy()[r0:r1, :, 1].astype(np.int64)
        sx = maps[cam][0].numpy()[r0:r1, :, 0].astype(np.int64)
        touch = ~((sx >= W) | (sx + 1 < 0) | (sy >= H) | (sy + 1 < 0))
        assert s0 == max(0, sy[touch].min()) and s0 + sn - 1 == min(H - 1, sy[touch].max() + 1)
    # a window one row short of what the band needs must change the result (the proof that the window is really used)
    r0, r1, s0, sn = windows[2]
    cam = 0
    short = [st[cam, p, s0 + 1:s0 + sn].clone() for p in range(np_)]
    d2 = ctx.mfn_rectify_decode(cam, short, F, N, 40.0, W=W, H=H, row0=r0, rows=r1 - r0, src_row0=s0 + 1)
    ctx.synchronize()
    assert not torch.equal(d2[0], full[cam][0][r0:r1])


@pytest.mark.gpu
@pytest.mark.parametrize("W,H,strength", [(1024, 200, 1.0), (512, 77, 6.0), (2048, 64, 40.0), (64, 16, 1.0), (8, 3, 1.0), (4096, 1030, 3.0)])
def test_gpu_mfn_rectify_tiled_form_equals_the_gather_form(ctx, slr, synth, W, H, strength):
    """the LDS-tiled form of the 4 x 8 rectifying decode (contiguous stack, W % 8 == 0: what config 5 runs) against the per-pixel
    gather form (SLR_OPT_DEBUG_FLAGS bit 0), bit for bit: ragged last tiles, borders, maps whose tile boxes do not fit the LDS image
    (strength 40: those tiles run the gather code -- inside the register-staged kernel, in the LDS-DMA form's fix-up pass), row bands
    with source windows.  Flags 0 / 256: the LDS-DMA form (ring of 3 / 4 plane groups; + 128: 512 threads x 2 pixels), 2: the register-staged tile form."""
    F, N = 4, 8
    st = synth.render_mfn_stack(W, H, F, N, noise=0.5, seed=W + H).cuda()
    for cam in range(2):
        mx, mf = synth.make_rectify_maps(W, H, cam, strength=strength)
        ctx.set_rectify_maps(cam, mx.numpy(), mf.numpy())
    res = {}
    for flags in (1, 0, 2, 256, 128, 384):                  # gather; LDS-DMA 256 x 4, ring of 3 (shipped); register-staged tiles; ring of 4; the 512-thread LDS-DMA forms
        ctx.set_option(slr.capi.OPT_DEBUG_FLAGS, flags)
        out = []
        for cam in range(2):
            out.append(ctx.mfn_rectify_decode(cam, st[cam], F, N, 40.0))
            r0, r1 = H // 3, max(H // 3 + 1, 2 * H // 3)
            s0, sn = ctx.rectify_source_rows(cam, r0, r1 - r0)
            if sn > 0:
                win = st[cam][:, s0:s0 + sn].contiguous()
                out.append(ctx.mfn_rectify_decode(cam, win, F, N, 40.0, W=W, H=H, row0=r0, rows=r1 - r0, src_row0=s0))
        ctx.synchronize()
        res[flags] = out
    ctx.set_option(slr.capi.OPT_DEBUG_FLAGS, 0)
    for flags in (0, 2, 256, 128, 384):
        assert len(res[flags]) == len(res[1])
        for a, b in zip(res[flags], res[1]):
            assert torch.equal(a[1], b[1]) and torch.equal(a[0].view(torch.int32), b[0].view(torch.int32)), flags
    assert res[0][0][1].float().mean().item() > (0.2 if strength < 10 else 0.0)
