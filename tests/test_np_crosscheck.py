"""the C oracle cross-checked by an independent NumPy transcription of the compat-mode
pseudo-spec (tests/np_model.py).  Two separately written restatements of the reference must agree BIT FOR BIT on
random inputs that reach every branch (equal samples, zero denominators, the heterodyne cliff, borders)."""
import numpy as np
import pytest

import np_model as M
from util import bits_equal


def _cam(O, fc, cc, k):
    return O.Camera.make(list(fc), list(cc), list(k) + [0.0] * (5 - len(k))), dict(fc=fc, cc=cc, k=k)


def test_wrapped_phase_all_quotients(oracle):
    """all 511x511 (n, d) pairs through both models"""
    n = np.arange(-255, 256)
    nn, dd = np.meshgrid(n, n, indexing="ij")
    # realise (n, d) as sample quadruples: G4-G2 = n, G1-G3 = d
    G2 = np.where(nn < 0, -nn, 0); G4 = G2 + nn
    G3 = np.where(dd < 0, -dd, 0); G1 = G3 + dd
    got, ok = M.wrapped_phase(G1, G2, G3, G4)
    for a in range(0, 511, 7):
        for b in range(0, 511, 5):
            o_ok, o_p = oracle.wrapped_phase(int(G1[a, b]), int(G2[a, b]), int(G3[a, b]), int(G4[a, b]))
            assert o_ok == bool(ok[a, b])
            if o_ok:
                assert np.float32(o_p).view(np.uint32) == got[a, b].view(np.uint32), (nn[a, b], dd[a, b])


@pytest.mark.parametrize("seed,levels", [(1, 256), (2, 6), (3, 2)])
def test_mf_decode_random_planes(oracle, seed, levels):
    """few grey levels -> many exact-equality branches and Q5 pixels; 256 levels -> generic"""
    rng = np.random.default_rng(seed)
    H, W = 48, 64
    planes = (rng.integers(0, levels, (14, H, W)) * (255 // (levels - 1))).astype(np.uint8)
    planes[0] = rng.integers(100, 256, (H, W)); planes[1] = rng.integers(0, 120, (H, W))
    ph_o, v_o = oracle.mf_decode(planes, 40)
    ph_n, v_n = M.mf_decode(planes, 40)
    assert np.array_equal(v_o, v_n)
    assert bits_equal(ph_o, ph_n)
    assert 0 < v_o.sum() < v_o.size


def test_heterodyne_cliff(oracle):
    P = np.array([[7.068598, 3.1416, -0.7853982], [3 * 3.1416 / 2, 3.1416, 3.1416 / 2]], np.float32)
    P[1] = [M.PI_3_2, M.PI, M.PI_1_2]
    for row in P:
        n = M.heterodyne(*(np.array([v], np.float32) for v in row))[0]
        o = oracle.heterodyne([float(v) for v in row])
        assert np.float32(o).view(np.uint32) == n.view(np.uint32)


@pytest.mark.parametrize("nrow", [0, 6])
def test_gray_decode_random(oracle, nrow):
    rng = np.random.default_rng(5)
    H, W, ncol = 40, 96, 7
    n = 2 + 2 * ncol + 2 * nrow
    planes = rng.integers(0, 256, (n, H, W)).astype(np.uint8)
    planes[0] = rng.integers(100, 256, (H, W)); planes[1] = rng.integers(0, 120, (H, W))
    for wt in (0, 9):
        o = oracle.gray_decode(planes, ncol, nrow, 40, wt, 100, 50)
        m = M.gray_decode(planes, ncol, nrow, 40, wt, 100, 50)
        assert np.array_equal(o[2], m[2])
        ok = o[2].astype(bool)
        assert np.array_equal(o[0][ok], m[0][ok])
        if nrow:
            assert np.array_equal(o[1][ok], m[1][ok])
        assert 0 < ok.sum() < ok.size


def test_remap_random_maps_with_borders(oracle):
    rng = np.random.default_rng(9)
    H, W = 37, 53
    src = rng.integers(0, 256, (H, W)).astype(np.uint8)
    xy = np.stack([rng.integers(-3, W + 3, (H, W)), rng.integers(-3, H + 3, (H, W))], -1).astype(np.int16)
    fr = rng.integers(0, 1024, (H, W)).astype(np.uint16)
    fr[:4] = 0                                             # the saturated {32768,0,0,0} entry
    assert np.array_equal(oracle.remap_u8(src, xy, fr), M.remap_u8(src, xy, fr))


def test_undistort_and_mf_triangulate(oracle):
    rng = np.random.default_rng(11)
    H, W = 12, 160
    camL, nL = _cam(oracle, (900.0, 905.0), (80.5, 6.25), (-0.21, 0.09, 0.0013, -0.0007))
    camR, nR = _cam(oracle, (910.0, 902.0), (79.0, 5.75), (-0.18, 0.05, -0.0009, 0.0011))
    px = rng.uniform(0, W, 50).astype(np.float32); py = rng.uniform(0, H, 50).astype(np.float32)
    ox, oy = M.undistort(px, py, nL["fc"], nL["cc"], nL["k"])
    for a in range(50):
        o = oracle.undistort_point(float(px[a]), float(py[a]), camL)
        assert o[0].view(np.uint32) == ox[a].view(np.uint32) and o[1].view(np.uint32) == oy[a].view(np.uint32)
    # phases on a coarse lattice so that several right pixels satisfy |d| < 0.1 (first-match matters)
    phL = (rng.integers(0, 400, (H, W)) * 0.07).astype(np.float32)
    phR = (rng.integers(0, 400, (H, W)) * 0.07).astype(np.float32)
    vL = (rng.random((H, W)) < 0.8).astype(np.uint8); vR = (rng.random((H, W)) < 0.8).astype(np.uint8)
    Q = np.array([[1, 0, 0, -80.5], [0, 1, 0, -6.0], [0, 0, 0, 900.0], [0, 0, 1 / 120.0, 0.0125]], np.float64)
    T = rng.normal(size=(3, 4)).astype(np.float32)
    for tt in (None, T):
        xo, ho, ko = oracle.mf_triangulate(phL, vL, phR, vR, camL, camR, Q, tt)
        xn, hn, kn = M.mf_triangulate(phL, vL, phR, vR, nL, nR, Q, tt)
        assert np.array_equal(ko, kn) and np.array_equal(ho, hn)
        assert bits_equal(xo, xn)
        assert ho.sum() > 100


def test_ge_triangulate_and_pointcloud(oracle):
    rng = np.random.default_rng(13)
    H, W = 10, 120
    cL = np.sort(rng.integers(0, 60, (H, W)), axis=1).astype(np.int32)
    cR = np.sort(rng.integers(0, 60, (H, W)), axis=1).astype(np.int32)
    cR[:, 50:60] = rng.integers(0, 60, (H, 10))            # non-monotone stretch: exercises the kstart carry
    vL = (rng.random((H, W)) < 0.85).astype(np.uint8); vR = (rng.random((H, W)) < 0.85).astype(np.uint8)
    wL = rng.integers(0, 256, (H, W)).astype(np.uint8); wR = rng.integers(0, 256, (H, W)).astype(np.uint8)
    Q = np.array([[1, 0, 0, -60.5], [0, 1, 0, -5.0], [0, 0, 0, 700.0], [0, 0, 1 / 90.0, 0.02]], np.float64)
    xo, ho, co, ko = oracle.ge_triangulate(cL, vL, cR, vR, Q, None, wL, wR)
    xn, hn, cn, kn = M.ge_triangulate(cL, vL, cR, vR, Q, None, wL, wR)
    assert np.array_equal(ko, kn) and np.array_equal(ho, hn) and np.array_equal(co, cn)
    assert bits_equal(xo, xn)
    for sw, sh in ((8, 100), (16, 200)):                   # Q11: transposed, cropped at (scan_w rows, scan_h cols)
        so, cnt_o, _ = oracle.pointcloud_from_grid(xo, ho, sw, sh)
        sn, cnt_n = M.pointcloud_from_grid(xn, hn, sw, sh)
        assert np.array_equal(cnt_o, cnt_n) and bits_equal(so, sn)
