"""Pins the CPU oracle to the known-answer anchors that can be derived from the reference SOURCE TEXT alone
(SURVEY.md 8c-4, KA1..KA8).  The reference ships no tests or golden vectors, so this is the strongest pin that
exists: parity stays "unpinned" at the OpenCV 2.4.9 boundary (see oracle/slr_oracle.h)."""
import numpy as np

PI = np.float32(3.1416)            # Duke/mfreconstruct.cpp:5


def test_ka1_gray_roundtrip(oracle):
    for w, nbits in ((1280, 11), (4096, 12), (1024, 10), (640, 10)):
        assert oracle.gray_num_bits(w) == nbits                    # graycodes.cpp:24
        g = oracle.gen_graycodes(w, 3, True)
        assert g.shape[0] == 2 + 2 * nbits                         # graycodes.cpp:27
        cx, _, v = oracle.gray_decode(g, nbits, 0, 40, 0, w, 3)
        assert v.all() and np.array_equal(cx[1], np.arange(w))
    g = oracle.gen_graycodes(64, 48, False)                        # rows too: graycodes.cpp:87-111
    cx, cy, v = oracle.gray_decode(g, 6, 6, 40, 0, 64, 48)
    assert v.all() and np.array_equal(cy[:, 5], np.arange(48)) and np.array_equal(cx[7], np.arange(64))
    assert oracle.gray_to_dec([1, 0, 0]) == 7 and oracle.gray_to_dec([1, 1, 0]) == 4   # graycodes.cpp:116-128


def test_ka2_getphase_branch_table(oracle):
    """mfreconstruct.cpp:246-261 incl. the integer division (Q1) and the swapped quadrant offsets (Q2)"""
    table = [
        ((200, 135, 70, 135), True, np.float32(0)),
        ((70, 135, 200, 135), True, PI),
        ((135, 70, 135, 200), True, np.float32(3) * PI / np.float32(2)),
        ((135, 200, 135, 70), True, PI / np.float32(2)),
        ((100, 120, 180, 160), True, PI),                          # q = 40/-80 = 0
        ((180, 100, 100, 190), True, np.float32(7.068598)),        # q = 90/80 = 1, +2PI
        ((180, 190, 100, 100), True, np.float32(-0.7853982)),      # q = -90/80 = -1, no offset
    ]
    for g, ok, exp in table:
        got_ok, got = oracle.wrapped_phase(*g)
        assert got_ok == ok and got == exp, (g, got, exp)
    assert oracle.wrapped_phase(135, 135, 135, 135)[0] is False    # Q5: mask cleared, P undefined


def test_ka3_heterodyne(oracle):
    f = np.float32
    assert oracle.heterodyne([0, float(PI), float(f(1.5708))]) == f(63.75)
    assert oracle.heterodyne([float(f(7.068598)), 0, float(PI)]) == f(159.37492)
    # the ill-conditioned cliff: strict '>' and the per-step f32 narrowing
    assert oracle.heterodyne([float(f(7.068598)), float(PI), float(f(-0.7853982))]) == f(254.99998)
    tiny = oracle.heterodyne([float(f(3) * PI / f(2)), float(PI), float(PI / f(2))])
    assert 0 < tiny < 1e-5


def test_ka4_ka5_remap(oracle):
    rng = np.random.default_rng(0)
    H, W = 17, 23
    img = rng.integers(0, 256, size=(H, W), dtype=np.uint8)
    ys, xs = np.mgrid[0:H, 0:W]
    mxy = np.stack([xs, ys], -1).astype(np.int16)
    assert np.array_equal(oracle.remap_u8(img, mxy, np.zeros((H, W), np.uint16)), img)          # KA4 identity
    half = oracle.remap_u8(img, mxy, np.full((H, W), 16, np.uint16))                             # fx=16, fy=0
    a = img.astype(np.int64)
    b = np.concatenate([a[:, 1:], np.zeros((H, 1), np.int64)], 1)
    assert np.array_equal(half, ((a * 16384 + b * 16384 + 16384) >> 15).astype(np.uint8))        # KA5
    # fully outside -> 0 ; one row above the image -> only the lower taps contribute
    off = mxy.copy()
    off[..., 1] -= 1
    up = oracle.remap_u8(img, off, np.full((H, W), 16 << 5, np.uint16))                          # fy=16
    assert np.array_equal(up[0], ((a[0] * 16384 + 16384) >> 15).astype(np.uint8))
    off[..., 0] += 1000
    assert not oracle.remap_u8(img, off, np.zeros((H, W), np.uint16)).any()


def test_ka6_line_line_intersection(oracle):
    s = np.float32(1 / np.sqrt(2))
    ok, p = oracle.line_line_intersection([0, 0, 0], [0, 0, 1], [1, 0, 0], [-s, 0, s])
    assert ok and np.allclose(p, [0, 0, 1], atol=1e-6)
    ok, _ = oracle.line_line_intersection([0, 0, 0], [0, 0, 1], [1, 0, 0], [0, 0, 1])            # parallel
    assert not ok
    th = np.deg2rad(18.0)                                                                        # < 18.4 deg: dropped
    ok, _ = oracle.line_line_intersection([0, 0, 0], [0, 0, 1], [1, 0, 0], [np.sin(th), 0, np.cos(th)])
    assert not ok
    th = np.deg2rad(19.0)
    ok, _ = oracle.line_line_intersection([0, 0, 0], [0, 0, 1], [1, 0, 0], [-np.sin(th), 0, np.cos(th)])
    assert ok


def test_ka7_q_reprojection(oracle):
    f, cx1, cx2, cy, Tx = 800.0, 321.5, 318.0, 240.25, -95.0
    Q = np.array([[1, 0, 0, -cx1], [0, 1, 0, -cy], [0, 0, 0, f], [0, 0, -1 / Tx, (cx1 - cx2) / Tx]])
    W, H, d = 64, 4, 9
    code = np.broadcast_to(np.arange(W, dtype=np.int32), (H, W)).copy()
    v = np.ones((H, W), np.uint8)
    xyz, has, _, mk = oracle.ge_triangulate(code, v, code + d, v, Q)
    j = np.arange(d, W)
    assert np.array_equal(mk[1, d:], j - d)
    Z = f * Tx / ((cx1 - cx2) - d)
    assert np.allclose(xyz[1, d:, 2], Z, rtol=1e-6)
    assert np.allclose(xyz[1, d:, 0], (j - cx1) * Z / f, rtol=1e-5, atol=1e-4)
    assert np.allclose(xyz[1, d:, 1], (1 - cy) * Z / f, rtol=1e-5)


def test_ka8_encode_decode_regression(oracle):
    """the reference's own ideal patterns (W=1280) through getPhase: 562 distinct values spanning
    -178.19 .. +474.71, first samples 255,255,255,255,254.29,1.14,13.77,18.82,223.13,255"""
    mf = oracle.gen_multifreq(1280, 2)
    assert mf[0].min() == 255 and mf[1].max() == 0
    ph, v = oracle.mf_decode(mf, 40)
    assert v.all()
    assert len(np.unique(ph)) == 562
    assert abs(ph.min() + 178.19151) < 1e-3 and abs(ph.max() - 474.71353) < 1e-3
    assert np.allclose(ph[0, :10], [255, 255, 255, 255, 254.28806, 1.1415693, 13.769984, 18.816868, 223.12508, 255],
                       rtol=0, atol=1e-4)


def test_undistort_fixed_point_and_k3_ignored(oracle):
    """utilities.cpp:58-94: 5 iterations, k[4] forced to 0"""
    cam_a = oracle.Camera.make((800, 810), (320, 240), (-0.1, 0.02, 1e-3, -5e-4, 0.0))
    cam_b = oracle.Camera.make((800, 810), (320, 240), (-0.1, 0.02, 1e-3, -5e-4, 9.0))
    assert oracle.undistort_point(100, 50, cam_a) == oracle.undistort_point(100, 50, cam_b)
    cam0 = oracle.Camera.make((800, 810), (320, 240), (0, 0, 0, 0, 0))
    x, y = oracle.undistort_point(100, 50, cam0)
    assert abs(x - 100) < 1e-4 and abs(y - 50) < 1e-4
    x, y = oracle.undistort_point(600, 400, cam_a)                 # barrel distortion: undistorted point moves outwards
    assert x > 600 and y > 400


def test_pointcloud_adaptor_q11(oracle):
    """addPoint(i=row, j=col) on PointCloudImage(scan_w, scan_h): transposed, cropped to i<scan_w, j<scan_h"""
    H, W = 6, 9
    xyz = np.arange(H * W * 3, dtype=np.float32).reshape(H, W, 3)
    has = np.ones((H, W), np.uint8)
    has[2, 3] = 0
    s, c, _ = oracle.pointcloud_from_grid(xyz, has, 4, 5)
    assert s.shape == (5, 4, 3)
    assert c[3, 2] == 0 and c.sum() == 4 * 5 - 1
    assert np.array_equal(s[4, 1], xyz[1, 4])                      # points[j][i] = p(i, j)
    got = oracle.pointcloud_get(s * 3, (c * 3).astype(np.uint8))
    assert np.allclose(got[4, 1], xyz[1, 4])
