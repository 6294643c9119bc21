"""SLR_OPT_EVAL_MODEL = 1: the multi-frequency path as the reference's own MSVC2010 x87 / fp:precise binary evaluates its source text
(include/slr.h) against the oracle's x87 model (oracle/slr_oracle_x87.c), bit for bit; and back to the default."""
import numpy as np
import pytest
import torch

from util import bits_equal, calib_parts, forms

pytestmark = pytest.mark.gpu

BLACK = 40


@pytest.fixture()
def xctx(slr):
    c = slr.Context(0)
    c.set_option(slr.capi.OPT_EVAL_MODEL, 1)
    yield c
    c.close()


def _all_quotient_planes(rng):
    n = np.arange(-255, 256)[:, None] * np.ones((1, 511), int)
    d = np.ones((511, 1), int) * np.arange(-255, 256)[None, :]
    pl = rng.integers(0, 256, size=(14, 511, 511), dtype=np.uint8)
    pl[0], pl[1] = 230, 20
    pl[2], pl[4] = np.maximum(d, 0), np.maximum(-d, 0)
    pl[5], pl[3] = np.maximum(n, 0), np.maximum(-n, 0)
    return pl


def test_x87_decode_every_quotient_and_branch(xctx, ctx, oracle, slr):
    rng = np.random.default_rng(17)
    tab = oracle.atan_table(1)
    pl = _all_quotient_planes(rng)
    for planes in (pl, np.ascontiguousarray(np.pad(pl, ((0, 0), (0, 0), (0, 1)))), np.ascontiguousarray(pl[:, :, :508])):
        e_ph, e_v = oracle.mf_decode_ev(planes, BLACK, tab, 1)
        ph, v = xctx.mf_decode(planes, BLACK)
        assert bits_equal(v, e_v) and bits_equal(ph, e_ph)
        # the two models really differ on this image, and the default context still computes the strict one
        s_ph, s_v = oracle.mf_decode(planes, BLACK)
        assert not bits_equal(e_ph, s_ph)
        ph0, v0 = ctx.mf_decode(planes, BLACK)
        assert bits_equal(ph0, s_ph) and bits_equal(v0, s_v)
    # few grey levels: the equality branches and the undefined case (Q5)
    few = (rng.integers(0, 4, size=(14, 64, 128)) * 60).astype(np.uint8)
    few[0], few[1] = 250, rng.integers(0, 255, size=(64, 128))
    e_ph, e_v = oracle.mf_decode_ev(few, BLACK, tab, 1)
    ph, v = xctx.mf_decode(few, BLACK)
    assert bits_equal(v, e_v) and bits_equal(ph, e_ph) and (e_v == 0).any()


def test_x87_option_switches_back(slr, oracle):
    rng = np.random.default_rng(18)
    pl = _all_quotient_planes(rng)
    c = slr.Context(0)
    try:
        strict = oracle.mf_decode(pl, BLACK)
        x87 = oracle.mf_decode_ev(pl, BLACK, oracle.atan_table(1), 1)
        for mode in (0, 1, 0, 1, 1, 0):
            c.set_option(slr.capi.OPT_EVAL_MODEL, mode)
            ph, v = c.mf_decode(pl, BLACK)
            want = x87 if mode else strict
            assert bits_equal(ph, want[0]) and bits_equal(v, want[1]), mode
        with pytest.raises(slr.capi.SlrError):
            c.set_option(slr.capi.OPT_EVAL_MODEL, 2)
    finally:
        c.close()


@pytest.mark.parametrize("W,H,rect", [(320, 240, True), (320, 240, False), (1024, 96, True), (100, 37, False), (4100, 8, True)])
def test_x87_whole_path_against_the_x87_oracle(xctx, oracle, synth, W, H, rect):
    """fused rectify + decode, match (predicate on the exact difference) and triangulation (unrounded disparity) == the oracle's
    x87 chain: valid masks, match columns and XYZ bit for bit -- and not the strict chain's"""
    tab = oracle.atan_table(1)
    calib, _ = synth.make_calibration(W, H, with_T=(W == 320))
    xctx.set_calibration(calib)
    camL, camR, Q, T = calib_parts(oracle, calib)
    maps = [synth.make_rectify_maps(W, H, cam) for cam in range(2)]
    if rect:
        for cam in range(2):
            xctx.set_rectify_maps(cam, maps[cam][0].numpy(), maps[cam][1].numpy())
    st = synth.render_mf_stack(W, H, seed=99 + W)
    dec, dec_s = [], []
    for cam in range(2):
        pl = st[cam].numpy()
        if rect:
            pl = np.stack([oracle.remap_u8(pl[p], maps[cam][0].numpy(), maps[cam][1].numpy()) for p in range(14)])
        dec.append(oracle.mf_decode_ev(pl, BLACK, tab, 1))
        dec_s.append(oracle.mf_decode(pl, BLACK))
    exyz, ehas, emk = oracle.mf_triangulate_ev(dec[0][0], dec[0][1], dec[1][0], dec[1][1], camL, camR, Q, 1, T=T)
    xyz, has = xctx.reconstruct_mf(st[0].cuda(), st[1].cuda(), BLACK, rect)
    xctx.synchronize()
    assert bits_equal(has.cpu().numpy(), ehas)
    assert bits_equal(xyz.cpu().numpy(), exyz)
    # the separate entries too (host buffers): decode with valid bytes, match with the columns
    for cam in range(2):
        ph, v = xctx.mf_decode(st[cam].numpy(), BLACK, rectify_cam=cam if rect else None)
        assert bits_equal(ph, dec[cam][0]) and bits_equal(v, dec[cam][1])
    x2, h2, mk = xctx.mf_triangulate(dec[0][0], dec[0][1], dec[1][0], dec[1][1])
    assert bits_equal(mk, emk) and bits_equal(h2, ehas) and bits_equal(x2, exyz)
    if W >= 320 and H >= 96:
        sxyz, shas, smk = oracle.mf_triangulate(dec_s[0][0], dec_s[0][1], dec_s[1][0], dec_s[1][1], camL, camR, Q, T)
        assert not bits_equal(sxyz, exyz)               # (the models differ: this test would notice a silent strict path)


def test_x87_fullsize_decode_and_sampled_match(xctx, oracle, synth):
    """BASELINE's size: the rectifying decode of one camera and the match on 64 rows, x87 model, against the oracle"""
    W, H = 4096, 3000
    tab = oracle.atan_table(1)
    dev = torch.device("cuda", 0)
    rig = synth.make_verged_rig(W, H, 0.2, -0.15)
    xctx.set_calibration(rig["calib"])
    synth.install_verged_maps(xctx, rig, W, H)
    camL, camR, Q, T = calib_parts(oracle, rig["calib"])
    st = synth.render_mf_stack(W, H, seed=1234, noise=2, device=dev)
    ph, vd = [], []
    for cam in range(2):
        p, v = xctx.mf_decode(st[cam], BLACK, rectify_cam=cam)
        xctx.synchronize()                               # (the kernels run on the ctx stream: torch must not read before they are done)
        ph.append(p.cpu().numpy()); vd.append(v.cpu().numpy())
    mx, mf = xctx.get_rectify_maps(0, W, H)
    pl = np.stack([oracle.remap_u8(st[0, p].cpu().numpy(), mx, mf) for p in range(14)])
    e_ph, e_v = oracle.mf_decode_ev(pl, BLACK, tab, 1)
    assert bits_equal(vd[0], e_v) and bits_equal(ph[0], e_ph)
    r0, r1 = 1468, 1532
    x, h, mk = xctx.mf_triangulate(ph[0][r0:r1], vd[0][r0:r1], ph[1][r0:r1], vd[1][r0:r1], row0=r0, image_h=H)
    ex, eh, emk = oracle.mf_triangulate_ev(ph[0], vd[0], ph[1], vd[1], camL, camR, Q, 1, T=T, rows=(r0, r1))
    assert bits_equal(np.asarray(mk), emk[r0:r1]) and bits_equal(np.asarray(h), eh[r0:r1]) and bits_equal(np.asarray(x), ex[r0:r1])


# ---- GRAY_ONLY under the x87 model: normalize / pixelToImageSpace / line_lineIntersection (utilities.cpp:19-28, 47-56, 399-425) ----
@pytest.mark.parametrize("W,H,scan_w,scan_h,with_T", [(160, 120, 64, 48, False), (96, 64, 40, 33, True), (1024, 768, 300, 200, False)])
def test_x87_ray_triangulate_against_the_x87_oracle(xctx, ctx, oracle, synth, W, H, scan_w, scan_h, with_T):
    """K6 with SLR_OPT_EVAL_MODEL = 1 (unit-ray tables and the ray-ray midpoints as the x87 binary rounds them) == the oracle's x87
    restatement, bit for bit; the default context still computes the strict model, and the two differ on these scenes"""
    calib, _ = synth.make_calibration(W, H, with_T=with_T, baseline=400.0, theta=0.6)
    xctx.set_calibration(calib)
    ctx.set_calibration(calib)
    camL, camR, _, T = calib_parts(oracle, calib)
    st = synth.render_gray_stack(W, H, scan_w, scan_h, seed=41, noise=2, rows=True)
    ncol, nrow = synth.gray_num_bits(scan_w), synth.gray_num_bits(scan_h)
    dec = [oracle.gray_decode(st[c].numpy(), ncol, nrow, BLACK, 0, scan_w, scan_h) for c in range(2)]
    offL, itL = oracle.gray_bucket(dec[0][0], dec[0][1], dec[0][2], scan_w, scan_h)
    offR, itR = oracle.gray_bucket(dec[1][0], dec[1][1], dec[1][2], scan_w, scan_h)
    exyz, ecnt = oracle.ray_triangulate_x87(offL, itL, offR, itR, camL, camR, scan_w, scan_h, T)
    sxyz, scnt = oracle.ray_triangulate(offL, itL, offR, itR, camL, camR, scan_w, scan_h, T)
    xyz, cnt = xctx.ray_triangulate(dec[0][0], dec[0][1], dec[0][2], dec[1][0], dec[1][1], dec[1][2], scan_w, scan_h)
    assert bits_equal(cnt, ecnt) and bits_equal(xyz, exyz)
    assert (ecnt > 1).any()
    x0, c0 = ctx.ray_triangulate(dec[0][0], dec[0][1], dec[0][2], dec[1][0], dec[1][1], dec[1][2], scan_w, scan_h)
    assert bits_equal(c0, scnt) and bits_equal(x0, sxyz)
    assert not bits_equal(sxyz, exyz)                      # (the models differ here: a silent strict path would be noticed)
    # the whole-path entry (decode + buckets + K6 from the planes) under the x87 model
    x2, c2 = xctx.reconstruct_gray(st[0].cuda(), st[1].cuda(), ncol, nrow, BLACK, 0, scan_w, scan_h)
    xctx.synchronize()
    assert bits_equal(c2.cpu().numpy(), ecnt) and bits_equal(x2.cpu().numpy(), exyz)


def test_x87_line_line_intersections(xctx, oracle):
    """Utilities::line_lineIntersection under the x87 model on rays of many scales, around the |denom| < 0.1 rejection and nearly
    perpendicular ones, against the oracle's restatement (slro_line_line_intersection_x87) on a sample, and against an independent
    NumPy float64 transcription of the same rules on all of them"""
    rng = np.random.default_rng(78)
    n = 200000
    f = np.float32

    def unit(m):
        v = rng.standard_normal((m, 3)); return v / np.linalg.norm(v, axis=1, keepdims=True)
    v1 = unit(n); v2 = unit(n)
    q = n // 4
    v1[q:2 * q] *= np.exp2(rng.integers(-20, 12, (q, 1))); v2[q:2 * q] *= np.exp2(rng.integers(-20, 12, (q, 1)))
    ang = np.arcsin(np.sqrt(np.linspace(0.0995, 0.1005, q)))                                             # sin^2 around 0.1
    perp = np.cross(v1[2 * q:3 * q], unit(q)); perp /= np.linalg.norm(perp, axis=1, keepdims=True)
    v2[2 * q:3 * q] = v1[2 * q:3 * q] * np.cos(ang)[:, None] + perp * np.sin(ang)[:, None]
    v1 = v1.astype(f); v2 = v2.astype(f)
    p1 = np.array([3.5, -20.25, 1000.0], f); p2 = np.array([-410.0, 7.0, 955.5], f)
    D = np.float64
    with np.errstate(all="ignore"):
        def d3(a, b):                                       # Vec3f::dot: product exact at 53 bits, one rounding to f32 per step
            s = np.zeros(a.shape[:-1] if a.ndim > 1 else b.shape[:-1], f)
            for i in range(3):
                s = (s.astype(D) + a[..., i].astype(D) * b[..., i].astype(D)).astype(f)
            return s
        v12 = (p1 - p2)[None]
        a = d3(v1, v1); c = d3(v2, v2); b = d3(v1, v2); d = d3(v12, v1); e = d3(v12, v2)
        den = (a.astype(D) * c.astype(D) - b.astype(D) * b.astype(D)).astype(f)
        hit = ~(np.abs(den) < f(0.1))
        bq, cq, aq = b.astype(D) / den.astype(D), c.astype(D) / den.astype(D), a.astype(D) / den.astype(D)
        s_ = (bq * e.astype(D) - cq * d.astype(D)).astype(f)
        t_ = (-bq * d.astype(D) + aq * e.astype(D)).astype(f)
        exp = f(0.5) * ((p1[None] + s_[:, None] * v1) + (p2[None] + t_[:, None] * v2))
    exp[~hit] = 0
    for i in rng.integers(0, n, 3000):                       # the NumPy transcription is the oracle's
        ok_i, o_i = oracle.line_line_intersection_x87(p1, v1[i], p2, v2[i])
        assert ok_i == bool(hit[i])
        if ok_i and np.isfinite(o_i).all():
            assert bits_equal(o_i, exp[i]), i
    out, ok = xctx.line_line_intersections(p1, p2, v1, v2)
    assert np.array_equal(ok != 0, hit)
    fin = np.isfinite(exp).all(axis=1)
    assert fin.sum() > 0.9 * n and 0.3 * q < hit[2 * q:3 * q].sum() < 0.7 * q
    assert bits_equal(out[fin], exp[fin])


def _flip_pairs(rng, n):
    """n (left phase t, right phase b) pairs of floats, |t| < 0.25, whose EXACT difference lies between the midpoint of the two
    floats around 0.1 and the double 0.1: the x87 predicate accepts them, the f32 difference rounds up to 0.1f and the strict one
    does not.  (No such pair exists with |t| >= 0.25: Sterbenz.)"""
    f = np.float32
    lo = (np.float64(np.nextafter(f(0.1), f(0))) + np.float64(f(0.1))) / 2
    out = []
    while len(out) < n:
        t = f(rng.random() * 0.5 - 0.25)
        side = 1.0 if rng.random() < 0.5 else -1.0
        b0 = np.array(f(np.float64(t) + side * 0.1))
        for u in rng.permutation(7) - 3:
            b = (b0.view(np.int32) + np.int32(u)).view(f)[()]
            if lo < abs(np.float64(t) - np.float64(b)) < 0.1:
                out.append((t, b))
                break
    return out


@pytest.mark.parametrize("W", [700, 4096, 3000 // 4 * 4, 8192])
def test_x87_match_predicate_on_adversarial_rows(xctx, ctx, oracle, synth, slr, W):
    """mfreconstruct.cpp:295 under the x87 model -- fabs of the EXACT difference < 0.1 -- where it differs from the strict
    predicate: a left phase inside (-0.25, 0.25) and a right phase whose exact difference to it lies just below 0.1 while the f32
    difference rounds up to 0.1f.  Rows 0-11: one such pair per row (the edge candidate in an early column, an exact match in a
    later one: the x87 search stops at the first, the strict one at the second), the left phase varying along the row in three
    well separated levels; rows 12-17: left phases of 0.25 and more with candidates an ulp or two around the edge (the lean
    kernel's f32 fast path: both models agree there); the rest random.  The lean kernel, the general forms and the sweep against the
    oracle's x87 search; the strict context against the strict oracle on the same rows."""
    rng = np.random.default_rng(W + 5)
    H = 24
    f = np.float32
    calib, _ = synth.make_calibration(W, H, with_T=(W == 700))
    camL, camR, Q, T = calib_parts(oracle, calib)
    for c in (xctx, ctx):
        c.set_calibration(calib)
    phL = (rng.random((H, W)) * 600 + 50).astype(f)
    phR = (rng.random((H, W)) * 600 + 1000).astype(f)                               # far from every left phase
    for r in range(12):
        pairs = _flip_pairs(rng, 3)
        lv = np.array([q[0] for q in pairs], f)
        # three levels at least 0.25 apart would not fit (-0.25, 0.25): keep the levels in separate column ranges of the left row and
        # give each its own edge candidate; a candidate may also match another level's pixels -- the oracle decides, both models see it
        seg = W // 3
        for m_, (t, b) in enumerate(pairs):
            phL[r, m_ * seg:(m_ + 1) * seg if m_ < 2 else W] = t
            k1 = int(rng.integers(0, W // 2))
            phR[r, k1] = b
            phR[r, W // 2 + int(rng.integers(0, W // 2))] = t                      # an exact match further right
    for r in range(12, 18):
        big = (rng.random(W) * 400 + 0.25).astype(f)
        if r >= 16:
            big[:] = f(0.25) if r == 16 else np.nextafter(f(0.25), f(0))           # the boundary of the fast path itself
        side = np.where(rng.random(W) < 0.5, 1.0, -1.0)
        edge = ((big.astype(np.float64) + side * 0.1).astype(f).view(np.int32) + rng.integers(-2, 3, W).astype(np.int32)).view(f)
        phL[r] = big
        phR[r] = edge[rng.permutation(W)]
    vL = (rng.random((H, W)) < 0.97).astype(np.uint8)
    vR = np.ones((H, W), np.uint8)
    exyz, ehas, emk = oracle.mf_triangulate_ev(phL, vL, phR, vR, camL, camR, Q, 1, T=T)
    sxyz, shas, smk = oracle.mf_triangulate(phL, vL, phR, vR, camL, camR, Q, T)
    assert (emk[:12] != smk[:12]).sum() > W                  # the predicates differ on the flip rows ...
    assert bits_equal(emk[12:18], smk[12:18]) and ehas[12:16].sum() > 0                # ... and not where |phase| >= 0.25
    for algo in forms(xctx, slr, slr.capi.OPT_MF_MATCH_ALGO, (0, 4, 3, 1), required=(0, 4, 3, 1)):
        xctx.set_option(slr.capi.OPT_MF_MATCH_ALGO, algo)
        xyz, has, mk = xctx.mf_triangulate(phL, vL, phR, vR)
        assert bits_equal(mk, emk), algo
        assert bits_equal(has, ehas) and bits_equal(xyz, exyz), algo
    xctx.set_option(slr.capi.OPT_MF_MATCH_ALGO, 0)
    xyz, has, mk = ctx.mf_triangulate(phL, vL, phR, vR)
    assert bits_equal(mk, smk) and bits_equal(has, shas) and bits_equal(xyz, sxyz)
    # the whole-path form of the call (no valid bytes, NaN-folded, device arrays, no match columns): what the batch entry launches
    nan = f(np.nan)
    fl = np.where(vL != 0, phL, nan).astype(f)
    ones = np.ones((H, W), np.uint8)
    x2, h2, _ = xctx.mf_triangulate(torch.from_numpy(fl).cuda(), torch.from_numpy(ones).cuda(), torch.from_numpy(phR).cuda(),
                                    torch.from_numpy(ones).cuda(), want_match=False)
    xctx.synchronize()
    assert bits_equal(h2.cpu().numpy(), ehas) and bits_equal(x2.cpu().numpy(), exyz)
