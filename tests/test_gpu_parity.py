"""GPU parity tests proper: the HIP path (through the C ABI, include/slr.h) against the CPU oracle on the same
seeded inputs.  Bar (BASELINE.json north_star): Gray-code indices / masks / match columns bit-exact; phase and
XYZ within 1e-4 relative -- the tolerance is written in util.assert_float_parity.  In practice phase is asserted
BIT-exact (atanf comes from a host-filled table, all other ops are IEEE f32/f64 without contraction).
"""
import numpy as np
import pytest
import torch

from util import assert_float_parity, bits_equal, calib_parts, forms, np_of

pytestmark = pytest.mark.gpu

BLACK = 40


# ---------------------------------------------------------------------------------------------------------
# K2
# ---------------------------------------------------------------------------------------------------------
def _all_quotient_planes(rng):
    """511x511 image whose pixel (r,c) has n = r-255, d = c-255 at frequency 0: every (n,d) pair, i.e. every
    branch of mfreconstruct.cpp:246-261 and every integer quotient.  Frequencies 1,2 are random."""
    n = np.arange(-255, 256)[:, None] * np.ones((1, 511), int)
    d = np.ones((511, 1), int) * np.arange(-255, 256)[None, :]
    pl = rng.integers(0, 256, size=(14, 511, 511), dtype=np.uint8)
    pl[0] = 230
    pl[1] = 20
    pl[2] = np.maximum(d, 0)      # G1
    pl[4] = np.maximum(-d, 0)     # G3  -> G1-G3 = d
    pl[5] = np.maximum(n, 0)      # G4
    pl[3] = np.maximum(-n, 0)     # G2  -> G4-G2 = n
    return pl


def test_mf_decode_every_quotient_and_branch(ctx, oracle):
    rng = np.random.default_rng(7)
    pl = _all_quotient_planes(rng)
    exp_ph, exp_v = oracle.mf_decode(pl, BLACK)
    ph, v = ctx.mf_decode(pl, BLACK)                 # W=511: generic 1-px kernel, host staging
    assert bits_equal(v, exp_v)
    assert bits_equal(ph, exp_ph)
    # same content padded to W=512 -> 16-px vector kernel; and W=508 crop -> 4-px kernel
    pad = np.zeros((14, 511, 512), np.uint8)
    pad[:, :, :511] = pl
    e2, ev2 = oracle.mf_decode(pad, BLACK)
    g2, gv2 = ctx.mf_decode(pad, BLACK)
    assert bits_equal(gv2, ev2) and bits_equal(g2, e2)
    crop = np.ascontiguousarray(pl[:, :, :508])
    e3, ev3 = oracle.mf_decode(crop, BLACK)
    g3, gv3 = ctx.mf_decode(crop, BLACK)
    assert bits_equal(gv3, ev3) and bits_equal(g3, e3)
    # the 5 special branches really occur and the Q5 pixel is invalid
    assert exp_v[255, 255] == 0 and exp_v[255, 300] == 1


@pytest.mark.parametrize("W,H", [(640, 480), (64, 48), (100, 37), (1, 1), (16, 1), (3, 5)])
def test_mf_decode_synth_host_and_device(ctx, oracle, synth, W, H):
    st = synth.render_mf_stack(W, H, seed=1234 + W)
    for cam in range(2):
        pl = st[cam].numpy()
        exp_ph, exp_v = oracle.mf_decode(pl, BLACK)
        ph, v = ctx.mf_decode(pl, BLACK)
        assert bits_equal(v, exp_v) and bits_equal(ph, exp_ph)
        dph, dv = ctx.mf_decode(st[cam].cuda(), BLACK)
        ctx.synchronize()
        assert bits_equal(np_of(dv), exp_v) and bits_equal(np_of(dph), exp_ph)


def test_mf_decode_pitch_and_random(ctx, oracle):
    rng = np.random.default_rng(11)
    H, W, pitch = 33, 96, 128
    pl = rng.integers(0, 256, size=(14, H, pitch), dtype=np.uint8)
    exp_ph, exp_v = oracle.mf_decode(pl, BLACK, W=W)
    ph, v = ctx.mf_decode(pl, BLACK, W=W)
    assert bits_equal(v, exp_v) and bits_equal(ph, exp_ph)
    # thresholds at the edge: white-black == thr is NOT lit ('>' in mfreconstruct.cpp:201)
    pl[0, :, :] = 100
    pl[1, :, :] = 60
    _, v2 = ctx.mf_decode(pl, 40, W=W)
    assert v2.sum() == 0
    _, v3 = ctx.mf_decode(pl, 39, W=W)
    e3 = oracle.mf_decode(pl, 39, W=W)[1]
    assert bits_equal(v3, e3) and v3.sum() > 0


# ---------------------------------------------------------------------------------------------------------
# K1 and the fused rectify + decode
# ---------------------------------------------------------------------------------------------------------
def test_remap_identity_and_half_pixel(ctx, oracle, synth):
    rng = np.random.default_rng(3)
    H, W = 60, 84
    img = rng.integers(0, 256, size=(H, W), dtype=np.uint8)
    for cam, (dx, dy, fx, fy) in enumerate([(0, 0, 0, 0), (0, 0, 16, 0)]):
        mx, mf = synth.identity_maps(W, H, dx, dy, fx, fy)
        ctx.set_rectify_maps(cam, mx.numpy(), mf.numpy())
    out0 = ctx.remap_u8(0, img)
    assert np.array_equal(out0, img)                               # KA4
    out1 = ctx.remap_u8(1, img)
    a = img.astype(np.int64)
    b = np.concatenate([a[:, 1:], np.zeros((H, 1), np.int64)], axis=1)
    assert np.array_equal(out1, ((a * 16384 + b * 16384 + 16384) >> 15).astype(np.uint8))   # KA5
    assert np.array_equal(out1, oracle.remap_u8(img, mx.numpy(), mf.numpy()))


@pytest.mark.parametrize("W,H", [(640, 480), (130, 67), (4096, 200)])
def test_init_rectify_maps_on_device_matches_oracle(ctx, oracle, W, H):
    """cv::initUndistortRectifyMap restated on the device (running sums along a row, f64, cvRound half-to-even) ==
    the oracle's host restatement, bit for bit, incl. strong distortion and a rotated R"""
    f = 0.9 * W
    M = np.array([[f, 0, W / 2 - 3.3], [0, f * 1.01, H / 2 + 1.7], [0, 0, 1]])
    D = np.array([-0.28, 0.11, 0.0013, -0.0009, 0.02])
    a, b = 0.02, -0.013
    R = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]]) @ \
        np.array([[1, 0, 0], [0, np.cos(b), -np.sin(b)], [0, np.sin(b), np.cos(b)]])
    P = np.array([[f * 0.97, 0, W / 2 + 5.0, -40.0 * f], [0, f * 0.97, H / 2, 0], [0, 0, 1, 0]])
    exy, efr = oracle.init_undistort_rectify_map(M, D, R, P, W, H)
    for cam in range(2):
        ctx.init_rectify_maps(cam, M, D, R, P, W, H)
        xy, fr = ctx.get_rectify_maps(cam, W, H)
        assert np.array_equal(xy, exy) and np.array_equal(fr, efr)
    # and the installed maps drive the remap
    src = np.random.default_rng(1).integers(0, 256, (H, W)).astype(np.uint8)
    assert np.array_equal(ctx.remap_u8(0, src), oracle.remap_u8(src, exy, efr))


@pytest.mark.parametrize("dx,dy,fx,fy", [(-3, -2, 7, 19), (5, 4, 31, 31), (-1, 0, 0, 13), (0, -1, 30, 0), (700, 0, 3, 3)])
def test_remap_border_constant(ctx, oracle, synth, dx, dy, fx, fy):
    """taps that leave the image read 0 (BORDER_CONSTANT): partially-outside 2x2 footprints on every side and a
    fully-outside map; standalone remap and the fused rectify+decode must agree with the oracle"""
    W, H = 128, 40
    st = synth.render_mf_stack(W, H, seed=3)
    mx, mf = synth.identity_maps(W, H, dx, dy, fx, fy)
    mx, mf = mx.numpy(), mf.numpy()
    for cam in range(2):
        ctx.set_rectify_maps(cam, mx, mf)
    raw = st[0].numpy()
    rect = np.stack([oracle.remap_u8(raw[p], mx, mf) for p in range(14)])
    assert (rect[0] == 0).any()
    for p in (0, 7):
        assert np.array_equal(ctx.remap_u8(1, raw[p]), rect[p])
    exp_ph, exp_v = oracle.mf_decode(rect, BLACK)
    ph, v = ctx.mf_decode(raw, BLACK, rectify_cam=0)
    assert bits_equal(v, exp_v) and bits_equal(ph, exp_ph)


@pytest.mark.parametrize("W,H", [(640, 480), (101, 67), (64, 48), (200, 77)])
def test_remap_and_fused_rectify_decode(ctx, oracle, synth, slr, W, H):
    st = synth.render_mf_stack(W, H, seed=99)
    maps = [synth.make_rectify_maps(W, H, cam, strength=3.0) for cam in range(2)]
    for cam in range(2):
        ctx.set_rectify_maps(cam, maps[cam][0].numpy(), maps[cam][1].numpy())
    for cam in range(2):
        mx, mf = maps[cam][0].numpy(), maps[cam][1].numpy()
        raw = st[cam].numpy()
        rect = np.stack([oracle.remap_u8(raw[p], mx, mf) for p in range(14)])
        got = ctx.remap_u8(cam, raw[5])
        assert np.array_equal(got, rect[5])
        gdev = ctx.remap_u8(cam, st[cam, 5].cuda())
        ctx.synchronize()
        assert np.array_equal(np_of(gdev), rect[5])
        exp_ph, exp_v = oracle.mf_decode(rect, BLACK)
        for algo in forms(ctx, slr, slr.capi.OPT_RECT_DECODE_ALGO, (0, 1, 2, 3, 4, 5, 6), required=(0, 1, 5, 6)):   # auto, gather, [64x16, ring, 128x8 two rounds: FORMS=all], 128x8, 64x8
            ctx.set_option(slr.capi.OPT_RECT_DECODE_ALGO, algo)
            ph, v = ctx.mf_decode(raw, BLACK, rectify_cam=cam)
            assert bits_equal(v, exp_v) and bits_equal(ph, exp_ph), algo
        ctx.set_option(slr.capi.OPT_RECT_DECODE_ALGO, 0)


def test_fused_rectify_decode_wild_maps(ctx, oracle, synth, slr):
    """maps whose tile bounding boxes do not fit the LDS budget (random scatter) or leave the image entirely must
    take the per-workgroup fallback and still match the oracle"""
    rng = np.random.default_rng(9)
    W, H = 192, 80
    st = synth.render_mf_stack(W, H, seed=4)
    raw = st[0].numpy()
    mxs = []
    mx = np.stack([rng.integers(-20, W + 20, size=(H, W)), rng.integers(-20, H + 20, size=(H, W))], -1).astype(np.int16)
    mxs.append(mx)                                                   # random scatter: huge boxes -> fallback
    ys, xs = np.mgrid[0:H, 0:W]
    mxs.append(np.stack([xs + 5000, ys], -1).astype(np.int16))       # everything outside
    mxs.append(np.stack([(xs * 3) % W, (ys * 2) % H], -1).astype(np.int16))   # 3x/2x stretch: wide boxes
    mxs.append(np.stack([W - 1 - xs, H - 1 - ys], -1).astype(np.int16))       # mirrored (sx decreasing)
    for mx in mxs:
        mf = rng.integers(0, 1024, size=(H, W)).astype(np.uint16)
        ctx.set_rectify_maps(0, np.ascontiguousarray(mx), mf)
        rect = np.stack([oracle.remap_u8(raw[p], mx, mf) for p in range(14)])
        exp_ph, exp_v = oracle.mf_decode(rect, BLACK)
        for algo in forms(ctx, slr, slr.capi.OPT_RECT_DECODE_ALGO, (0, 1, 2, 3, 4, 5, 6), required=(0, 1, 5, 6)):
            ctx.set_option(slr.capi.OPT_RECT_DECODE_ALGO, algo)
            ph, v = ctx.mf_decode(raw, BLACK, rectify_cam=0)
            assert bits_equal(v, exp_v) and bits_equal(ph, exp_ph), algo
        ctx.set_option(slr.capi.OPT_RECT_DECODE_ALGO, 0)
        assert np.array_equal(ctx.remap_u8(0, raw[3]), rect[3])


@pytest.mark.parametrize("W,H", [(200, 77), (448, 203)])
def test_fused_rectify_decode_pipeline_many_tiles_per_workgroup(ctx, oracle, synth, slr, W, H, monkeypatch):
    """the persistent, register-prefetching form with only 8 / 16 workgroups: every workgroup walks many tiles, among
    them tiles whose box does not fit (random-scatter patch -> gather fallback), tiles completely outside the source,
    ragged right/bottom tiles (W % 64 != 0, H % 8 != 0) -- the prefetch of tile t+1 must never leak into tile t"""
    rng = np.random.default_rng(W)
    st = synth.render_mf_stack(W, H, seed=11)
    raw = st[0].numpy()
    mxt, mft = synth.make_rectify_maps(W, H, 0, strength=3.0)
    mx, mf = mxt.numpy().copy(), mft.numpy().copy()
    mx[20:45, 70:150] = np.stack([rng.integers(-9, W + 9, size=(25, 80)), rng.integers(-9, H + 9, size=(25, 80))], -1)
    mx[50:60, 0:64, 0] += 6000                                       # one tile row segment entirely outside
    ctx.set_rectify_maps(0, np.ascontiguousarray(mx), np.ascontiguousarray(mf))
    rect = np.stack([oracle.remap_u8(raw[p], mx, mf) for p in range(14)])
    exp_ph, exp_v = oracle.mf_decode(rect, BLACK)
    for res in ("8", "16", "40"):
        ctx.set_option(slr.capi.OPT_DEBUG_RECT_RESIDENT, int(res))
        for algo in forms(ctx, slr, slr.capi.OPT_RECT_DECODE_ALGO, (0, 2, 3, 4, 5, 6), required=(0, 5, 6)):
            ctx.set_option(slr.capi.OPT_RECT_DECODE_ALGO, algo)
            ph, v = ctx.mf_decode(raw, BLACK, rectify_cam=0)
            assert bits_equal(v, exp_v) and bits_equal(ph, exp_ph), (res, algo)
    ctx.set_option(slr.capi.OPT_DEBUG_RECT_RESIDENT, 0)
    ctx.set_option(slr.capi.OPT_RECT_DECODE_ALGO, 0)


# ---------------------------------------------------------------------------------------------------------
# K3 / K3'
# ---------------------------------------------------------------------------------------------------------
def test_gray_roundtrip_every_column(ctx, oracle):
    """KA1: generateGrays -> decode returns the column, for 1280 (11 bits) and 4096 (12 bits)"""
    for w in (1280, 4096, 640, 100):
        g = oracle.gen_graycodes(w, 4, True)
        n = oracle.gray_num_bits(w)
        cx, _, v = ctx.gray_decode(g, n, 0, BLACK, 0, w, 4)
        assert v.all() and np.array_equal(cx, np.broadcast_to(np.arange(w, dtype=np.int32), (4, w)))


@pytest.mark.parametrize("W,H,rows", [(640, 480, False), (640, 480, True), (97, 31, True), (64, 48, False)])
def test_gray_decode_parity(ctx, oracle, synth, W, H, rows):
    scan_w, scan_h = 600, 450
    st = synth.render_gray_stack(W, H, scan_w, scan_h, seed=5, noise=6, rows=rows)
    ncol = synth.gray_num_bits(scan_w)
    nrow = synth.gray_num_bits(scan_h) if rows else 0
    for cam in range(2):
        for white_thr in (0, 9):
            pl = st[cam].numpy()
            ex, ey, ev = oracle.gray_decode(pl, ncol, nrow, BLACK, white_thr, scan_w, scan_h)
            cx, cy, v = ctx.gray_decode(pl, ncol, nrow, BLACK, white_thr, scan_w, scan_h)
            assert bits_equal(v, ev) and bits_equal(cx, ex)
            if rows:
                assert bits_equal(cy, ey)
            dcx, dcy, dv = ctx.gray_decode(st[cam].cuda(), ncol, nrow, BLACK, white_thr, scan_w, scan_h)
            ctx.synchronize()
            assert bits_equal(np_of(dv), ev) and bits_equal(np_of(dcx), ex)
    assert 0 < ev.mean() < 1


def test_config3_gray_plus_phase_share_white_black(ctx, oracle, synth):
    """BASELINE config 3: one 38-plane stack per camera = white, black, 12 Gray bit pairs, 12 fringe planes; the Gray and
    the multi-frequency decode both read the SAME white/black planes (plane pointers, no copies).  The reference has
    no Gray+phase fusion (three exclusive modes), so each decode is checked against the oracle on its own."""
    W, H, scan_w = 512, 96, 4096
    ncol = synth.gray_num_bits(scan_w)
    assert ncol == 12
    g = synth.render_gray_stack(W, H, scan_w, seed=5)                 # [2][26][H][W]
    m = synth.render_mf_stack(W, H, seed=5)                           # [2][14][H][W]
    for cam in range(2):
        stack = torch.cat([g[cam], m[cam, 2:]], 0).contiguous()       # 26 + 12 = 38 planes
        assert stack.shape[0] == 38
        gray_planes = [stack[i] for i in range(26)]
        mf_planes = [stack[0], stack[1]] + [stack[26 + i] for i in range(12)]
        ex, _, ev = oracle.gray_decode(np.stack([p.numpy() for p in gray_planes]), ncol, 0, BLACK, 0, scan_w, 0)
        eph, epv = oracle.mf_decode(np.stack([p.numpy() for p in mf_planes]), BLACK)
        dev = stack.cuda()
        cx, _, v = ctx.gray_decode([dev[i] for i in range(26)], ncol, 0, BLACK, 0, scan_w, 0)
        ph, pv = ctx.mf_decode([dev[0], dev[1]] + [dev[26 + i] for i in range(12)], BLACK)
        ctx.synchronize()
        assert bits_equal(np_of(cx), ex) and bits_equal(np_of(v), ev)
        assert bits_equal(np_of(pv), epv) and bits_equal(np_of(ph), eph)
        assert ev.sum() > 0.5 * ev.size and epv.sum() > 0.3 * epv.size


def test_gray_range_check_uses_greater_than(ctx, oracle):
    """Q9: xDec == scan_w passes (reconstruct.cpp:403 uses '>')"""
    w = 64
    g = oracle.gen_graycodes(w, 2, True)
    n = oracle.gray_num_bits(w)
    for scan_w in (62, 40, 10):
        ex, _, ev = oracle.gray_decode(g, n, 0, BLACK, 0, scan_w, 2)
        cx, _, v = ctx.gray_decode(g, n, 0, BLACK, 0, scan_w, 2)
        assert bits_equal(cx, ex) and bits_equal(v, ev)
        assert v[0, scan_w] == 1 and v[0, scan_w + 1] == 0


def test_gray_rectify_decode(ctx, oracle, synth, slr):
    W, H, scan_w = 320, 200, 300
    st = synth.render_gray_stack(W, H, scan_w, seed=8, noise=3)
    ncol = synth.gray_num_bits(scan_w)
    for cam in range(2):
        mx, mf = synth.make_rectify_maps(W, H, cam, strength=2.0)
        ctx.set_rectify_maps(cam, mx.numpy(), mf.numpy())
    for cam in range(2):
        mx, mf = synth.make_rectify_maps(W, H, cam, strength=2.0)
        raw = st[cam].numpy()
        rect = np.stack([oracle.remap_u8(raw[p], mx.numpy(), mf.numpy()) for p in range(raw.shape[0])])
        ex, _, ev = oracle.gray_decode(rect, ncol, 0, BLACK, 4, scan_w, 0)
        for algo, flags in ((0, 0), (1, 0), (6, 16), (5, 0), (6, 0)):   # auto, direct gather, LDS tiles 64x4 (debug flag), 128x8, 64x8
            ctx.set_option(slr.capi.OPT_RECT_DECODE_ALGO, algo)
            ctx.set_option(slr.capi.OPT_DEBUG_FLAGS, flags)
            cx, _, v = ctx.gray_decode(raw, ncol, 0, BLACK, 4, scan_w, 0, rectify_cam=cam)
            ctx.set_option(slr.capi.OPT_DEBUG_FLAGS, 0)
            assert bits_equal(cx, ex) and bits_equal(v, ev), (algo, flags)
        ctx.set_option(slr.capi.OPT_RECT_DECODE_ALGO, 0)
    # rows too (GRAY_ONLY never rectifies in the reference, but the entry point allows it), odd size, wild map
    W2, H2, sw, sh = 132, 37, 100, 90
    st2 = synth.render_gray_stack(W2, H2, sw, sh, seed=9, noise=3, rows=True)
    nc, nr = synth.gray_num_bits(sw), synth.gray_num_bits(sh)
    rng = np.random.default_rng(3)
    maps = [synth.make_rectify_maps(W2, H2, 0, strength=4.0),
            (torch.from_numpy(np.stack([rng.integers(-9, W2 + 9, (H2, W2)), rng.integers(-9, H2 + 9, (H2, W2))], -1).astype(np.int16)),
             torch.from_numpy(rng.integers(0, 1024, (H2, W2)).astype(np.uint16).view(np.int16)).view(torch.uint16))]
    for mx, mf in maps:
        mxn, mfn = mx.numpy(), mf.numpy()
        ctx.set_rectify_maps(0, mxn, mfn)
        raw = st2[0].numpy()
        rect = np.stack([oracle.remap_u8(raw[p], mxn, mfn) for p in range(raw.shape[0])])
        ex, ey, ev = oracle.gray_decode(rect, nc, nr, BLACK, 3, sw, sh)
        cx, cy, v = ctx.gray_decode(raw, nc, nr, BLACK, 3, sw, sh, rectify_cam=0)
        assert bits_equal(cx, ex) and bits_equal(cy, ey) and bits_equal(v, ev)


# ---------------------------------------------------------------------------------------------------------
# K4
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("W,H,with_T", [(640, 480, False), (320, 100, True), (67, 9, True)])
def test_mf_triangulate_parity(ctx, oracle, synth, W, H, with_T):
    calib, _ = synth.make_calibration(W, H, with_T=with_T)
    ctx.set_calibration(calib)
    camL, camR, Q, T = calib_parts(oracle, calib)
    st = synth.render_mf_stack(W, H, seed=21, noise=0 if W == 640 else 2)
    phL, vL = oracle.mf_decode(st[0].numpy(), BLACK)
    phR, vR = oracle.mf_decode(st[1].numpy(), BLACK)
    exyz, ehas, emk = oracle.mf_triangulate(phL, vL, phR, vR, camL, camR, Q, T)
    xyz, has, mk = ctx.mf_triangulate(phL, vL, phR, vR)
    assert bits_equal(has, ehas) and bits_equal(mk, emk)            # first-match semantics, exact
    assert ehas.sum() > 0
    nb = assert_float_parity(xyz, exyz, 1e-4, "mf xyz")
    assert nb == 0, "XYZ expected bit-exact, %d elements differ" % nb
    dx, dh, dk = ctx.mf_triangulate(*[torch.from_numpy(a).cuda() for a in (phL, vL, phR, vR)])
    ctx.synchronize()
    assert bits_equal(np_of(dk), emk) and bits_equal(np_of(dx), exyz)


def test_mf_triangulate_threshold_edges(ctx, oracle, synth):
    """|dphi| < 0.1 strict, first k wins, invalid right pixels are skipped (mfreconstruct.cpp:292-295)"""
    W, H = 16, 2
    calib, _ = synth.make_calibration(W, H)
    ctx.set_calibration(calib)
    camL, camR, Q, T = calib_parts(oracle, calib)
    phL = np.full((H, W), 10.0, np.float32)
    phR = np.full((H, W), 50.0, np.float32)
    vL = np.ones((H, W), np.uint8)
    vR = np.ones((H, W), np.uint8)
    phR[0, 3] = 10.0 + np.float32(0.1)                # not < 0.1 after f32 subtraction? decided by the oracle
    phR[0, 5] = np.nextafter(np.float32(10.1), np.float32(0))
    phR[0, 7] = 10.0
    phR[1, 2] = 10.05
    vR[1, 2] = 0                                      # skipped
    phR[1, 9] = 9.95
    vL[1, 4] = 0
    exyz, ehas, emk = oracle.mf_triangulate(phL, vL, phR, vR, camL, camR, Q, T)
    xyz, has, mk = ctx.mf_triangulate(phL, vL, phR, vR)
    assert bits_equal(mk, emk) and bits_equal(has, ehas) and bits_equal(xyz, exyz)
    assert emk[1, 0] == 9 and emk[1, 4] == -1


@pytest.mark.parametrize("W", [16, 255, 256, 300, 700, 1500, 2100, 4096, 5000, 8192, 9001, 16390])
def test_mf_match_sweep_and_indexed_forms_agree(ctx, oracle, synth, slr, W):
    """the O(log W) indexed K4 must return exactly what the literal linear sweep returns (and the oracle):
    adversarial rows -- few distinct values, dense clusters inside one 0.2 window, +-0, NaN, huge magnitudes"""
    rng = np.random.default_rng(W)
    H = 12
    calib, _ = synth.make_calibration(W, H, with_T=True)
    ctx.set_calibration(calib)
    camL, camR, Q, T = calib_parts(oracle, calib)
    phL = (rng.random((H, W)) * 600 - 150).astype(np.float32)
    phR = (rng.random((H, W)) * 600 - 150).astype(np.float32)
    vals = (rng.random(7) * 500).astype(np.float32)
    phL[1] = vals[rng.integers(0, 7, W)]; phR[1] = vals[rng.integers(0, 7, W)]            # few distinct values
    phL[2] = 100 + rng.random(W) * 0.3; phR[2] = 100 + rng.random(W) * 0.3                 # one dense cluster
    phL[3] = 42.0; phR[3] = 42.0 + np.float32(0.1)                                         # threshold edge
    phR[4] = np.where(rng.random(W) < 0.5, 0.0, -0.0); phL[4] = np.where(rng.random(W) < 0.5, 0.05, -0.05)
    phR[5, ::3] = np.nan; phL[5, ::5] = np.nan                                             # NaN never matches
    phL[6] = 1e7 + rng.integers(0, 4, W); phR[6] = 1e7 + rng.integers(0, 4, W)             # ulp(1e7) = 1
    phL[7] = np.round(phL[7]); phR[7] = np.round(phR[7]) + np.float32(0.0999)
    phL[8] = np.sort(phL[8]); phR[8] = np.sort(phR[8])
    phL[9] = phR[9][::-1]
    vL = (rng.random((H, W)) < 0.9).astype(np.uint8)
    vR = (rng.random((H, W)) < 0.9).astype(np.uint8)
    vR[10] = 0
    vL[11] = 0
    exyz, ehas, emk = oracle.mf_triangulate(phL, vL, phR, vR, camL, camR, Q, T)
    out = {}
    algos = forms(ctx, slr, slr.capi.OPT_MF_MATCH_ALGO, (1, 2, 3, 0), required=(0, 1, 3))
    for algo in algos:
        ctx.set_option(slr.capi.OPT_MF_MATCH_ALGO, algo)
        out[algo] = ctx.mf_triangulate(phL, vL, phR, vR)
    ctx.set_option(slr.capi.OPT_MF_MATCH_ALGO, 0)
    for algo in algos:
        xyz, has, mk = out[algo]
        assert bits_equal(mk, emk), "algo %d" % algo
        assert bits_equal(has, ehas) and bits_equal(xyz, exyz), "algo %d" % algo
    assert ehas[1].sum() > 0 and ehas[6].sum() > 0 and ehas[10].sum() == 0 and ehas[11].sum() == 0


@pytest.mark.parametrize("W", [1024, 4096])
def test_mf_match_fine_bin_edges(ctx, oracle, synth, slr, W):
    """round 5 (written for a K4 with 1/16-wide bins over [-32, 288), measured slower and not kept -- the rows stay): phases on a
    1/16 grid and one float beside it, pairs exactly 0.1 / 0.125 apart across bin borders, values piled up at -32, 288 and far
    beyond, clusters spanning many bins, 4096 copies of one value, +-inf / NaN, values around zero; against the oracle and the
    literal sweep"""
    rng = np.random.default_rng(7 * W)
    H = 12
    calib, _ = synth.make_calibration(W, H, with_T=True)
    ctx.set_calibration(calib)
    camL, camR, Q, T = calib_parts(oracle, calib)
    f32 = np.float32
    grid = (rng.integers(-40 * 16, 300 * 16, (H, W)) / 16.0).astype(f32)
    phL = grid.copy(); phR = (rng.integers(-40 * 16, 300 * 16, (H, W)) / 16.0).astype(f32)
    phR[0] = np.nextafter(phL[0][::-1].copy(), f32(1e9)); phL[0] = np.nextafter(phL[0], f32(-1e9))       # one float beside the grid
    phR[1] = phL[1] + f32(0.1); phR[1, ::2] = phL[1, ::2] - f32(0.1)                                   # the threshold, across two bins
    phR[2] = phL[2] + f32(0.125); phR[2, ::3] = np.nextafter(phL[2, ::3] + f32(0.1), f32(-1e9))
    phL[3] = (-32 + rng.integers(-3, 4, W) / 16.0 + rng.random(W) * 0.01).astype(f32); phR[3] = (-32 + rng.integers(-3, 4, W) / 16.0).astype(f32)
    phL[4] = (288 + rng.integers(-3, 4, W) / 16.0 - rng.random(W) * 0.01).astype(f32); phR[4] = (288 + rng.integers(-3, 4, W) / 16.0).astype(f32)
    phL[5] = (rng.choice([-1e6, -500.0, -33.0, 300.0, 511.9, 4e5], W) + rng.random(W) * 0.2).astype(f32)
    phR[5] = (rng.choice([-1e6, -500.0, -33.0, 300.0, 511.9, 4e5], W) + rng.random(W) * 0.2).astype(f32)
    phL[6] = (100 + rng.random(W) * 0.75).astype(f32); phR[6] = (100 + rng.random(W) * 0.75).astype(f32)  # a cluster over 12 fine bins
    phL[7] = f32(17.03125); phR[7] = f32(17.03125); phR[7, : W // 2] = f32(17.2)                            # one value, thousands of copies
    phL[8] = np.sort((rng.random(W) * 255).astype(f32)); phR[8] = np.sort((rng.random(W) * 255).astype(f32))   # the reference's range, monotone
    phR[9] = (rng.random(W) * 255).astype(f32); phL[9] = phR[9][rng.permutation(W)] + f32(0.0999)
    phL[10] = np.where(rng.random(W) < 0.3, np.nan, phL[10]).astype(f32); phR[10] = np.where(rng.random(W) < 0.3, np.inf, phR[10]).astype(f32)
    phL[11] = (rng.random(W) * 0.5 - 0.25).astype(f32); phR[11] = (rng.random(W) * 0.5 - 0.25).astype(f32)     # around zero (the x87 predicate's f64 path)
    vL = (rng.random((H, W)) < 0.95).astype(np.uint8)
    vR = (rng.random((H, W)) < 0.95).astype(np.uint8)
    exyz, ehas, emk = oracle.mf_triangulate(phL, vL, phR, vR, camL, camR, Q, T)
    for algo in (0, 1):
        ctx.set_option(slr.capi.OPT_MF_MATCH_ALGO, algo)
        xyz, has, mk = ctx.mf_triangulate(phL, vL, phR, vR)
        ctx.set_option(slr.capi.OPT_MF_MATCH_ALGO, 0)
        assert bits_equal(mk, emk), "algo %d" % algo
        assert bits_equal(has, ehas) and bits_equal(xyz, exyz), "algo %d" % algo
    for r in (0, 1, 2, 3, 4, 6, 7, 8, 9, 11):
        assert ehas[r].sum() > 0, r


# ---------------------------------------------------------------------------------------------------------
# K5
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("W,H,with_T,color", [(640, 480, False, True), (320, 64, True, False), (70, 5, False, True)])
def test_ge_triangulate_parity(ctx, oracle, synth, W, H, with_T, color):
    scan_w = W
    calib, _ = synth.make_calibration(W, H, with_T=with_T)
    ctx.set_calibration(calib)
    _, _, Q, T = calib_parts(oracle, calib)
    st = synth.render_gray_stack(W, H, scan_w, seed=31, noise=3)
    ncol = synth.gray_num_bits(scan_w)
    cL, _, vL = oracle.gray_decode(st[0].numpy(), ncol, 0, BLACK, 0, scan_w, 0)
    cR, _, vR = oracle.gray_decode(st[1].numpy(), ncol, 0, BLACK, 0, scan_w, 0)
    wl = st[0, 0].numpy() if color else None
    wr = st[1, 0].numpy() if color else None
    exyz, ehas, ecol, emk = oracle.ge_triangulate(cL, vL, cR, vR, Q, T, wl, wr)
    xyz, has, col, mk = ctx.ge_triangulate(cL, vL, cR, vR, wl, wr)
    assert bits_equal(mk, emk) and bits_equal(has, ehas)
    assert ehas.sum() > 0.3 * ehas.size
    nb = assert_float_parity(xyz, exyz, 1e-4, "ge xyz")
    assert nb == 0
    if color:
        assert bits_equal(col, ecol)


def test_ge_kstart_semantics_adversarial(ctx, oracle, synth):
    """random, non-monotone codes: the sequential kstart carry (reconstruct.cpp:556,561,604) must be reproduced
    exactly by the wave-parallel speculative walk (many fix-up rounds)"""
    rng = np.random.default_rng(17)
    W, H = 300, 40
    calib, _ = synth.make_calibration(W, H)
    ctx.set_calibration(calib)
    _, _, Q, T = calib_parts(oracle, calib)
    for ncodes in (3, 17, 300):
        cL = rng.integers(0, ncodes, size=(H, W), dtype=np.int32)
        cR = rng.integers(0, ncodes, size=(H, W), dtype=np.int32)
        vL = (rng.random((H, W)) < 0.8).astype(np.uint8)
        vR = (rng.random((H, W)) < 0.8).astype(np.uint8)
        cR[3] = np.sort(cR[3])[::-1]
        cL[4] = np.sort(cL[4])
        cR[4] = np.sort(cR[4])
        exyz, ehas, _, emk = oracle.ge_triangulate(cL, vL, cR, vR, Q, T)
        xyz, has, _, mk = ctx.ge_triangulate(cL, vL, cR, vR)
        assert bits_equal(mk, emk), ncodes
        assert bits_equal(has, ehas) and bits_equal(xyz, exyz)


def test_ge_reprojection_known_answer(ctx, synth):
    """KA7: Z = f*Tx / ((cx1-cx2) - d), X = (j-cx1)*Z/f, Y = (i-cy)*Z/f for a constant-disparity scene"""
    W, H, d = 128, 8, 11
    calib, info = synth.make_calibration(W, H)
    ctx.set_calibration(calib)
    code = np.broadcast_to(np.arange(W, dtype=np.int32), (H, W)).copy()
    cL = code + 1000
    cR = code + 1000 + d          # right pixel k shows what left pixel k+d shows -> j-k = d
    v = np.ones((H, W), np.uint8)
    xyz, has, _, mk = ctx.ge_triangulate(cL, v, cR, v)
    j = np.arange(d, W)
    assert np.array_equal(mk[2, d:], j - d) and not has[:, :d].any()
    f, cx1, cx2, cy, Tx = info["f"], info["cx1"], info["cx2"], info["cy"], info["Tx"]
    Wq = (-1.0 / Tx) * d + (cx1 - cx2) / Tx
    assert np.allclose(xyz[2, d:, 2], f / Wq, rtol=1e-6)
    assert np.allclose(xyz[2, d:, 0], (j - cx1) / Wq, rtol=1e-5, atol=1e-3)
    assert np.allclose(xyz[2, d:, 1], (2 - cy) / Wq, rtol=1e-5)


# ---------------------------------------------------------------------------------------------------------
# K3' scatter + K6
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("W,H,scan_w,scan_h,with_T", [(160, 120, 64, 48, False), (96, 64, 40, 33, True), (1024, 768, 300, 200, False), (1000, 300, 1000, 300, True)])
def test_ray_triangulate_parity(ctx, oracle, synth, W, H, scan_w, scan_h, with_T):
    calib, _ = synth.make_calibration(W, H, with_T=with_T, baseline=400.0, theta=0.6)
    ctx.set_calibration(calib)
    camL, camR, _, T = calib_parts(oracle, calib)
    st = synth.render_gray_stack(W, H, scan_w, scan_h, seed=41, noise=2, rows=True)
    ncol, nrow = synth.gray_num_bits(scan_w), synth.gray_num_bits(scan_h)
    dec = [oracle.gray_decode(st[c].numpy(), ncol, nrow, BLACK, 0, scan_w, scan_h) for c in range(2)]
    offL, itL = oracle.gray_bucket(dec[0][0], dec[0][1], dec[0][2], scan_w, scan_h)
    offR, itR = oracle.gray_bucket(dec[1][0], dec[1][1], dec[1][2], scan_w, scan_h)
    exyz, ecnt = oracle.ray_triangulate(offL, itL, offR, itR, camL, camR, scan_w, scan_h, T)
    xyz, cnt = ctx.ray_triangulate(dec[0][0], dec[0][1], dec[0][2], dec[1][0], dec[1][1], dec[1][2], scan_w, scan_h)
    assert bits_equal(cnt, ecnt)
    assert (ecnt > 1).any(), "want buckets with several pairs so the accumulation order matters"
    nb = assert_float_parity(xyz, exyz, 1e-4, "ray xyz")
    assert nb == 0
    # getPoint: sum / count
    got = ctx.pointcloud_get(xyz, cnt)
    assert bits_equal(got, oracle.pointcloud_get(exyz, ecnt))


def test_line_line_intersections_on_rays_of_every_scale(ctx, oracle):
    """Utilities::line_lineIntersection as K6 evaluates it (three quotients over one refined reciprocal inside a guarded exponent
    range, the plain division outside): unit rays, rays scaled by 2^-40 .. 2^20, nearly and exactly perpendicular ones (b -> 0),
    nearly parallel ones around the 0.1 threshold -- bit for bit against the f32 expression (NumPy, one rounding per operation,
    itself checked against the C oracle on a sample)"""
    rng = np.random.default_rng(77)
    n = 400000
    f = np.float32
    def unit(m):
        v = rng.standard_normal((m, 3)); return v / np.linalg.norm(v, axis=1, keepdims=True)
    v1 = unit(n); v2 = unit(n)
    q = n // 8
    v1[q:2 * q] *= np.exp2(rng.integers(-40, 21, (q, 1))); v2[q:2 * q] *= np.exp2(rng.integers(-40, 21, (q, 1)))
    w = np.cross(v1[2 * q:3 * q], unit(q)); w /= np.linalg.norm(w, axis=1, keepdims=True)
    v2[2 * q:3 * q] = w + v1[2 * q:3 * q] * np.exp2(rng.integers(-140, -10, (q, 1)).astype(np.float64))   # b from ~2^-140 up
    v2[3 * q:3 * q + q // 2] = np.cross(v1[3 * q:3 * q + q // 2], np.array([0.0, 0.0, 1.0]))              # axis-aligned cases: b == 0 exactly
    v1[3 * q:3 * q + q // 2] = np.round(v1[3 * q:3 * q + q // 2] * 4) / 4
    v2[3 * q:3 * q + q // 2] = np.cross(v1[3 * q:3 * q + q // 2], np.array([0.0, 0.0, 1.0]))
    ang = np.arcsin(np.sqrt(np.linspace(0.08, 0.12, q)))                                                 # sin^2 around 0.1
    v2[4 * q:5 * q] = v1[4 * q:5 * q] * np.cos(ang)[:, None] + np.cross(v1[4 * q:5 * q], unit(q)) * np.sin(ang)[:, None]
    v1[5 * q:6 * q] *= np.exp2(rng.integers(5, 40, (q, 1))); v2[5 * q:6 * q] *= np.exp2(rng.integers(-30, 0, (q, 1)))
    v1 = v1.astype(f); v2 = v2.astype(f)
    p1 = np.array([3.5, -20.25, 1000.0], f); p2 = np.array([-410.0, 7.0, 955.5], f)
    with np.errstate(all="ignore"):
        d3 = lambda a, b: (f(0) + a[..., 0] * b[..., 0] + a[..., 1] * b[..., 1]) + a[..., 2] * b[..., 2]   # utilities.cpp:399-425, f32 per operation
        v12 = p1 - p2
        a = d3(v1, v1); c = d3(v2, v2); b = d3(v1, v2); d = d3(v12[None], v1); e = d3(v12[None], v2)
        den = a * c - b * b
        hit = ~(np.abs(den) < f(0.1))
        s_ = (b / den) * e - (c / den) * d
        t_ = -(b / den) * d + (a / den) * e
        exp = f(0.5) * ((p1[None] + s_[:, None] * v1) + (p2[None] + t_[:, None] * v2))
    exp[~hit] = 0
    for i in rng.integers(0, n, 3000):                       # the NumPy expression is the oracle's
        ok_i, o_i = oracle.line_line_intersection(p1, v1[i], p2, v2[i])
        assert ok_i == bool(hit[i])
        if ok_i and np.isfinite(o_i).all(): assert bits_equal(o_i, exp[i]), i
    out, ok = ctx.line_line_intersections(p1, p2, v1, v2)
    assert np.array_equal(ok != 0, hit)
    fin = np.isfinite(exp).all(axis=1)
    assert fin.sum() > 0.9 * n and (hit & fin).sum() > 0.5 * n
    assert bits_equal(out[fin], exp[fin])
    assert np.array_equal(np.isnan(out[~fin]), np.isnan(exp[~fin]))
    assert (np.abs(b[hit]) < 2.0 ** -60).sum() > 1000 and (b[hit] == 0).sum() > 100, "want the plain-division side of the guard too"


def test_ray_count_wraps_like_uchar(ctx, oracle, synth):
    """> 255 pairs in one bucket: the u8 counter wraps and the next hit restarts the sum (pointcloudimage.cpp:90-95)"""
    W, H, scan_w, scan_h = 40, 20, 4, 4
    calib, _ = synth.make_calibration(W, H, baseline=400.0, theta=0.6)
    ctx.set_calibration(calib)
    camL, camR, _, T = calib_parts(oracle, calib)
    cx = np.zeros((H, W), np.int32)
    cy = np.zeros((H, W), np.int32)
    cx[:, W // 2:] = 2
    cy[:, W // 2:] = 4                 # y == scan_h aliases bucket (3,0)  (Q9)
    v = np.ones((H, W), np.uint8)
    vR = np.zeros((H, W), np.uint8)
    vR[:, :2] = 1                      # 40 right pixels x 400 left pixels in bucket (0,0) = 16000 pairs
    vR[:2, W // 2:W // 2 + 3] = 1
    offL, itL = oracle.gray_bucket(cx, cy, v, scan_w, scan_h)
    offR, itR = oracle.gray_bucket(cx, cy, vR, scan_w, scan_h)
    exyz, ecnt = oracle.ray_triangulate(offL, itL, offR, itR, camL, camR, scan_w, scan_h, T)
    xyz, cnt = ctx.ray_triangulate(cx, cy, v, cx, cy, vR, scan_w, scan_h)
    assert bits_equal(cnt, ecnt) and bits_equal(xyz, exyz)
    assert ecnt[0, 3] > 0               # the aliased bucket (x=3, y=0)


def test_ray_buckets_of_every_form(ctx, oracle, synth):
    """K6 has three forms: buckets of up to 16 items (sorting network, one wave per 64 cells), longer ones staged through LDS
    256 cells at a time (up to 3072 items per camera and run), and one cell beyond that on a single lane.  One frame with
    all three -- a left bucket of ~3800 pixels, buckets of 600 and of 17..40, and ordinary ones of 1..16 -- against the oracle."""
    W, H, scan_w, scan_h = 96, 48, 40, 16
    calib, _ = synth.make_calibration(W, H, baseline=400.0, theta=0.6)
    ctx.set_calibration(calib)
    camL, camR, _, T = calib_parts(oracle, calib)
    rng = np.random.default_rng(12)
    cx = np.zeros((H, W), np.int32)
    cy = np.zeros((H, W), np.int32)
    cx[:, 13:] = 7; cy[:, 13:] = 2                           # cell (2,7): 83 x 46 pixels of the left camera
    cxL, cyL = cx.copy(), cy.copy()
    cxL[:2, :] = rng.integers(0, scan_w, size=(2, W)); cyL[:2, :] = scan_h - 1      # ordinary buckets in the last cell row
    v = np.ones((H, W), np.uint8)
    cxR, cyR = cx.copy(), cy.copy()
    far = rng.random((H, W)) < 0.5
    cxR[far] = rng.integers(0, scan_w, size=int(far.sum())); cyR[far] = scan_h - 1   # ~60 pixels per cell there: staged
    mid = (~far) & (rng.random((H, W)) < 0.3)
    cxR[mid] = rng.integers(0, 12, size=int(mid.sum())); cyR[mid] = 5                # and a row of cells with a few each
    cxL[2:4, :] = rng.integers(0, 12, size=(2, W)); cyL[2:4, :] = 5
    vR = (rng.random((H, W)) < 0.35).astype(np.uint8)
    offL, itL = oracle.gray_bucket(cxL, cyL, v, scan_w, scan_h)
    offR, itR = oracle.gray_bucket(cxR, cyR, vR, scan_w, scan_h)
    lenL, lenR = np.diff(offL), np.diff(offR)
    both = (lenL > 0) & (lenR > 0)
    assert (lenL[both] > 3072).any() and ((lenR[both] > 16) & (lenR[both] < 3072)).any() and ((lenL[both] <= 16) & (lenR[both] <= 16)).any()
    assert ((lenR[both] > 12) & (lenR[both] <= 16) & (lenL[both] <= 16)).any(), "want a right bucket of 13..16 (the register rows)"
    exyz, ecnt = oracle.ray_triangulate(offL, itL, offR, itR, camL, camR, scan_w, scan_h, T)
    xyz, cnt = ctx.ray_triangulate(cxL, cyL, v, cxR, cyR, vR, scan_w, scan_h)
    assert bits_equal(cnt, ecnt) and bits_equal(xyz, exyz)


# ---------------------------------------------------------------------------------------------------------
# PointCloudImage adaptor (Q11) and whole-path drop-ins
# ---------------------------------------------------------------------------------------------------------
def test_pointcloud_from_grid_transpose_and_crop(ctx, oracle):
    rng = np.random.default_rng(2)
    W, H = 50, 30
    xyz = rng.standard_normal((H, W, 3)).astype(np.float32)
    has = (rng.random((H, W)) < 0.6).astype(np.uint8)
    color = rng.integers(0, 256, size=(H, W), dtype=np.uint8)
    for scan_w, scan_h in [(64, 64), (20, 40), (30, 50), (7, 3)]:
        es, ec, ek = oracle.pointcloud_from_grid(xyz, has, scan_w, scan_h, color)
        s, c, k = ctx.pointcloud_from_grid(xyz, has, scan_w, scan_h, color)
        assert bits_equal(c, ec) and bits_equal(s, es) and bits_equal(k, ek)


def test_reconstruct_mf_whole_path(ctx, oracle, synth):
    W, H = 320, 240
    calib, _ = synth.make_calibration(W, H, with_T=True)
    ctx.set_calibration(calib)
    camL, camR, Q, T = calib_parts(oracle, calib)
    st = synth.render_mf_stack(W, H, seed=77)
    maps = [synth.make_rectify_maps(W, H, cam) for cam in range(2)]
    for cam in range(2):
        ctx.set_rectify_maps(cam, maps[cam][0].numpy(), maps[cam][1].numpy())
    for rectify in (False, True):
        dec = []
        for cam in range(2):
            pl = st[cam].numpy()
            if rectify:
                pl = np.stack([oracle.remap_u8(pl[p], maps[cam][0].numpy(), maps[cam][1].numpy()) for p in range(14)])
            dec.append(oracle.mf_decode(pl, BLACK))
        exyz, ehas, _ = oracle.mf_triangulate(dec[0][0], dec[0][1], dec[1][0], dec[1][1], camL, camR, Q, T)
        xyz, has = ctx.reconstruct_mf(st[0].numpy(), st[1].numpy(), BLACK, rectify)
        assert bits_equal(has, ehas) and bits_equal(xyz, exyz)
        dxyz, dhas = ctx.reconstruct_mf(st[0].cuda(), st[1].cuda(), BLACK, rectify)
        ctx.synchronize()
        assert bits_equal(np_of(dhas), ehas) and bits_equal(np_of(dxyz), exyz)
    # batch fast path: two frames
    stack = torch.stack([st, synth.render_mf_stack(W, H, seed=78)]).cuda()
    bx, bh = ctx.reconstruct_mf_batch(stack, BLACK, True)
    ctx.synchronize()
    assert bits_equal(np_of(bh[0]), ehas) and bits_equal(np_of(bx[0]), exyz)
    assert not bits_equal(np_of(bx[1]), exyz)


def test_reconstruct_mf_under_every_decode_and_match_form(ctx, oracle, synth, slr):
    """inside slr_reconstruct_mf* the valid flag travels in the phase (NaN for invalid pixels, no separate valid bytes):
    every fused-decode form and every K4 form must give the oracle's cloud, also on pixels whose phase is undefined (Q5:
    flat fringes), shadowed (mask) or outside the rectified image"""
    W, H = 384, 96
    calib, _ = synth.make_calibration(W, H, with_T=True)
    ctx.set_calibration(calib)
    camL, camR, Q, T = calib_parts(oracle, calib)
    st = synth.render_mf_stack(W, H, seed=91, noise=2).numpy().copy()
    st[0, 2:, 10:30, 40:90] = 77                      # flat fringes under full light: n = d = 0 for all frequencies (Q5)
    st[1, 2:, 50:60, 200:260] = 120
    st[0, 0, 60:80, 300:340] = st[0, 1, 60:80, 300:340]   # white == black: shadow mask
    maps = [synth.make_rectify_maps(W, H, cam, strength=3.0) for cam in range(2)]
    for cam in range(2):
        ctx.set_rectify_maps(cam, maps[cam][0].numpy(), maps[cam][1].numpy())
    for rectify in (True, False):
        dec = []
        for cam in range(2):
            pl = st[cam]
            if rectify:
                pl = np.stack([oracle.remap_u8(pl[p], maps[cam][0].numpy(), maps[cam][1].numpy()) for p in range(14)])
            dec.append(oracle.mf_decode(pl, BLACK))
        assert (dec[0][1] == 0).sum() > 500 and (dec[0][1] == 1).sum() > 500
        exyz, ehas, _ = oracle.mf_triangulate(dec[0][0], dec[0][1], dec[1][0], dec[1][1], camL, camR, Q, T)
        for ralgo in (forms(ctx, slr, slr.capi.OPT_RECT_DECODE_ALGO, (0, 1, 2, 3, 4, 5, 6), required=(0, 1, 5, 6)) if rectify else (0,)):
            for malgo in forms(ctx, slr, slr.capi.OPT_MF_MATCH_ALGO, (0, 1, 2, 3), required=(0, 1, 3)):
                ctx.set_option(slr.capi.OPT_RECT_DECODE_ALGO, ralgo)
                ctx.set_option(slr.capi.OPT_MF_MATCH_ALGO, malgo)
                xyz, has = ctx.reconstruct_mf(st[0], st[1], BLACK, rectify)
                assert bits_equal(has, ehas) and bits_equal(xyz, exyz), (rectify, ralgo, malgo)
    ctx.set_option(slr.capi.OPT_RECT_DECODE_ALGO, 0)
    ctx.set_option(slr.capi.OPT_MF_MATCH_ALGO, 0)


def test_reconstruct_ge_and_gray_whole_path(ctx, oracle, synth):
    W, H, scan_w, scan_h = 256, 160, 256, 160
    calib, _ = synth.make_calibration(W, H, baseline=400.0, theta=0.6)
    ctx.set_calibration(calib)
    camL, camR, Q, T = calib_parts(oracle, calib)
    maps = [synth.make_rectify_maps(W, H, cam) for cam in range(2)]
    for cam in range(2):
        ctx.set_rectify_maps(cam, maps[cam][0].numpy(), maps[cam][1].numpy())
    ncol, nrow = synth.gray_num_bits(scan_w), synth.gray_num_bits(scan_h)
    st = synth.render_gray_stack(W, H, scan_w, seed=51, noise=2)
    for rectify in (False, True):
        dec, white = [], []
        for cam in range(2):
            pl = st[cam].numpy()
            if rectify:
                pl = np.stack([oracle.remap_u8(pl[p], maps[cam][0].numpy(), maps[cam][1].numpy())
                               for p in range(pl.shape[0])])
            white.append(pl[0])
            dec.append(oracle.gray_decode(pl, ncol, 0, BLACK, 3, scan_w, 0))
        exyz, ehas, ecol, _ = oracle.ge_triangulate(dec[0][0], dec[0][2], dec[1][0], dec[1][2], Q, T, white[0], white[1])
        xyz, has, col = ctx.reconstruct_ge(st[0].numpy(), st[1].numpy(), ncol, BLACK, 3, scan_w, rectify, True)
        assert bits_equal(has, ehas) and bits_equal(xyz, exyz) and bits_equal(col, ecol)
    # GRAY_ONLY
    st2 = synth.render_gray_stack(W, H, scan_w, scan_h, seed=52, noise=2, rows=True)
    dec = [oracle.gray_decode(st2[c].numpy(), ncol, nrow, BLACK, 0, scan_w, scan_h) for c in range(2)]
    offL, itL = oracle.gray_bucket(dec[0][0], dec[0][1], dec[0][2], scan_w, scan_h)
    offR, itR = oracle.gray_bucket(dec[1][0], dec[1][1], dec[1][2], scan_w, scan_h)
    exyz, ecnt = oracle.ray_triangulate(offL, itL, offR, itR, camL, camR, scan_w, scan_h, T)
    xyz, cnt = ctx.reconstruct_gray(st2[0].numpy(), st2[1].numpy(), ncol, nrow, BLACK, 0, scan_w, scan_h)
    assert bits_equal(cnt, ecnt) and bits_equal(xyz, exyz)


@pytest.mark.parametrize("W,H,scan_w,scan_h,white", [(256, 160, 64, 40, 0), (1000, 37, 300, 33, 3), (260, 50, 70, 20, 0), (2048, 300, 640, 100, 2)])
def test_gray_only_fused_decode_and_count(ctx, oracle, synth, slr, W, H, scan_w, scan_h, white):
    """slr_reconstruct_gray decodes inside the bucket histogram (one kernel per camera, runs of equal cells over a thread's 4
    pixels and over lanes): against the oracle chain and against the two-kernel form (SLR_OPT_DEBUG_FLAGS bit 3); widths that
    are no multiple of 256 pixels (waves straddling rows), shadows, a white threshold, codes beyond the projector (Q9)"""
    calib, _ = synth.make_calibration(W, H, baseline=400.0, theta=0.6)
    ctx.set_calibration(calib)
    camL, camR, _, T = calib_parts(oracle, calib)
    ncol, nrow = synth.gray_num_bits(scan_w), synth.gray_num_bits(scan_h)
    st = synth.render_gray_stack(W, H, scan_w, scan_h, seed=58, noise=3, rows=True).numpy().copy()
    st[0, 2:, H // 3:H // 2, W // 4:W // 2] = 255                   # codes past scan_w / scan_h: dropped or aliased like ac() does
    st[1, 3::2, : H // 5, : W // 3] = 0
    dec = [oracle.gray_decode(st[c], ncol, nrow, BLACK, white, scan_w, scan_h) for c in range(2)]
    offL, itL = oracle.gray_bucket(dec[0][0], dec[0][1], dec[0][2], scan_w, scan_h)
    offR, itR = oracle.gray_bucket(dec[1][0], dec[1][1], dec[1][2], scan_w, scan_h)
    exyz, ecnt = oracle.ray_triangulate(offL, itL, offR, itR, camL, camR, scan_w, scan_h, T)
    assert (ecnt > 0).sum() > 50
    got = {}
    for flags in (0, 8):
        ctx.set_option(slr.capi.OPT_DEBUG_FLAGS, flags)
        try:
            got[flags] = ctx.reconstruct_gray(st[0], st[1], ncol, nrow, BLACK, white, scan_w, scan_h)
        finally:
            ctx.set_option(slr.capi.OPT_DEBUG_FLAGS, 0)
        assert bits_equal(got[flags][1], ecnt) and bits_equal(got[flags][0], exyz), flags


# ---------------------------------------------------------------------------------------------------------
# error behaviour (reference: bool + QMessageBox; here: status codes, never an abort)
# ---------------------------------------------------------------------------------------------------------
def test_error_paths(slr, synth):
    c = slr.Context(0)
    pl = np.zeros((14, 8, 16), np.uint8)
    with pytest.raises(slr.SlrError) as e:
        c.mf_decode(pl, 40, rectify_cam=0)
    assert e.value.status == slr.capi.ERR_NOT_CONFIGURED
    with pytest.raises(slr.SlrError) as e:
        c.mf_triangulate(np.zeros((8, 16), np.float32), np.zeros((8, 16), np.uint8),
                         np.zeros((8, 16), np.float32), np.zeros((8, 16), np.uint8))
    assert e.value.status == slr.capi.ERR_NOT_CONFIGURED
    mx, mf = synth.identity_maps(32, 8)
    c.set_rectify_maps(0, mx.numpy(), mf.numpy())
    with pytest.raises(slr.SlrError) as e:
        c.mf_decode(pl, 40, rectify_cam=0)             # map size != image size
    assert e.value.status == slr.capi.ERR_INVALID_ARG
    with pytest.raises(slr.SlrError) as e:
        c.mf_decode(pl, 40, W=32)                      # pitch < W
    assert e.value.status == slr.capi.ERR_INVALID_ARG
    c.close()


def test_async_host_option_double_buffering(slr, synth, oracle):
    """SLR_OPT_ASYNC_HOST: host-buffer calls only enqueue; two contexts fed alternately (the double-buffered loader of
    SURVEY 8f-1) must give the results of the synchronous call once synchronised"""
    W, H = 256, 64
    calib, _ = synth.make_calibration(W, H)
    frames = [synth.render_mf_stack(W, H, seed=900 + f) for f in range(4)]
    pinned = [[f[c].contiguous().pin_memory().numpy() for c in range(2)] for f in frames]
    ref = slr.Context(0)
    ref.set_calibration(calib)
    exp = [ref.reconstruct_mf(p[0], p[1], BLACK, False) for p in pinned]
    ref.close()
    ctxs = [slr.Context(0) for _ in range(2)]
    outs = [(torch.empty((H, W, 3), dtype=torch.float32).pin_memory().numpy(), torch.empty((H, W), dtype=torch.uint8).pin_memory().numpy())
            for _ in range(4)]
    for c in ctxs:
        c.set_calibration(calib)
        c.set_option(slr.capi.OPT_ASYNC_HOST, 1)
    for f in range(4):
        ctxs[f % 2].reconstruct_mf(pinned[f][0], pinned[f][1], BLACK, False, xyz=outs[f][0], has=outs[f][1])
    for c in ctxs:
        c.synchronize()
    for f in range(4):
        assert bits_equal(outs[f][1], exp[f][1]) and bits_equal(outs[f][0], exp[f][0]), f
    for c in ctxs:
        c.close()
