"""Pins that do NOT come from this repository's own restatements: pieces of the oracle and of the host mirror against
INDEPENDENT third-party implementations that happen to be installed in the image (SciPy, Pillow: see also
test_host_mirror.py::test_png_and_pgm_codec_against_pillow).  None of them is OpenCV 2.4.9 -- the reference's own dependency
cannot be had here -- so each check states what it can and cannot show.

  * cv::Rodrigues (inside cv::stereoRectify, stereorect.cpp:41) as restated in tests/np_model.py (which pins host/calib.cpp)
    vs scipy.spatial.transform.Rotation: a quaternion-based implementation -- agreement to 1e-12 incl. angles near 0 and pi.
  * cv::remap's fixed-point bilinear interpolation (stereorect.cpp:26-34; oracle/slr_oracle.c) vs
    scipy.ndimage.map_coordinates(order=1, mode="grid-constant") in float64 at the coordinates the CV_16SC2 + CV_16UC1 maps encode:
    the same geometry (x / y and fx / fy not swapped, the right tap order, BORDER_CONSTANT 0) -- within one grey level, which is
    the most a float interpolation can say about a 5-bit-fraction fixed-point one.
  * the host mirror's stereoRectify output: R1, R2 are rotations (SciPy accepts them, det 1) and they rectify: the rotated baseline
    is parallel to the x axis, both new cameras share their rows."""
import importlib
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import np_model as M  # noqa: E402

Rotation = pytest.importorskip("scipy.spatial.transform").Rotation
ndimage = pytest.importorskip("scipy.ndimage")


def test_rodrigues_against_scipy_rotation():
    rng = np.random.default_rng(3)
    vecs = [rng.standard_normal(3) * s for s in (1e-9, 1e-4, 0.1, 1.0, 2.5) for _ in range(20)]
    vecs += [np.array([np.pi - 1e-7, 0, 0]), np.array([0, 3.0, 0.5]), np.zeros(3), np.array([0.2, -0.3, 0.1])]
    for r in vecs:
        theta = np.linalg.norm(r)
        if theta > np.pi:                                    # the canonical rotation vector has |r| <= pi
            r = r * ((theta - 2 * np.pi) / theta)
        Rm = M.rodrigues_to_matrix(r)
        assert np.allclose(Rm, Rotation.from_rotvec(r).as_matrix(), atol=1e-12, rtol=0), r
        back = M.rodrigues_to_vector(Rm)
        ref = Rotation.from_matrix(Rm).as_rotvec()
        # (at theta = pi the sign of the axis is a convention: compare the rotations; below sin(theta) = 1e-5 cvRodrigues2 returns the
        #  zero vector -- its documented precision floor, restated in np_model -- so tiny angles agree to 2e-5 only)
        tol = 2.1e-5 if abs(np.sin(np.linalg.norm(r))) < 1e-4 else 1e-9          # (near 0 AND near pi)
        assert np.allclose(M.rodrigues_to_matrix(back), Rotation.from_rotvec(ref).as_matrix(), atol=tol, rtol=0), r
        if np.linalg.norm(r) < np.pi - 1e-3:
            assert np.allclose(back, ref, atol=tol, rtol=0), r


def test_remap_geometry_against_scipy_map_coordinates():
    import oracle as O
    rng = np.random.default_rng(4)
    H, W = 61, 83
    # a smooth image (interpolation differences stay small and a swapped axis or fraction shows as tens of grey levels) + noise
    yy, xx = np.mgrid[0:H, 0:W]
    src = (127 + 90 * np.sin(xx / 5.0) * np.cos(yy / 7.0) + rng.integers(-6, 7, (H, W))).clip(0, 255).astype(np.uint8)
    # maps of a small rotation + shift, with integer parts that leave the image on every side (BORDER_CONSTANT)
    th = 0.07
    sx = (xx - W / 2) * np.cos(th) - (yy - H / 2) * np.sin(th) + W / 2 + 2.3
    sy = (xx - W / 2) * np.sin(th) + (yy - H / 2) * np.cos(th) + H / 2 - 1.7
    ix, iy = np.floor(sx * 32 + 0.5).astype(np.int64), np.floor(sy * 32 + 0.5).astype(np.int64)   # cv::convertMaps: round to 1/32
    xy = np.stack([ix >> 5, iy >> 5], -1).astype(np.int16)
    fr = (((iy & 31) << 5) | (ix & 31)).astype(np.uint16)
    got = O.remap_u8(src, xy, fr).astype(np.float64)
    # float64 bilinear at the SAME 1/32-quantised coordinates, zeros outside the image
    ref = ndimage.map_coordinates(src.astype(np.float64), [iy / 32.0, ix / 32.0], order=1, mode="grid-constant", cval=0.0)
    assert np.abs(got - ref).max() <= 1.0 + 1e-9, np.abs(got - ref).max()
    assert np.abs(got - ref).mean() < 0.3
    outside = (sx < -1) | (sy < -1) | (sx > W) | (sy > H)
    assert outside.any() and (got[outside] == 0).all()
    # the check can fail: swapped fractions or axes are far outside the tolerance on this image
    swapped = ndimage.map_coordinates(src.astype(np.float64), [ix / 32.0 * H / W, iy / 32.0 * W / H], order=1, mode="grid-constant", cval=0.0)
    assert np.abs(got - swapped).max() > 20
    fr_sw = (((ix & 31) << 5) | (iy & 31)).astype(np.uint16)
    assert np.abs(O.remap_u8(src, xy, fr_sw).astype(np.float64) - ref).max() > 1.0


def test_stereo_rectify_outputs_are_rotations_that_rectify():
    rng = np.random.default_rng(5)
    for _ in range(10):
        f = 2400.0 + rng.uniform(-200, 200)
        M1 = np.array([[f, 0, 2048 + rng.uniform(-30, 30)], [0, f * 1.001, 1500 + rng.uniform(-30, 30)], [0, 0, 1]])
        M2 = np.array([[f * 0.99, 0, 2040.0], [0, f * 0.992, 1510.0], [0, 0, 1]])
        D1 = np.array([-0.1, 0.02, 1e-4, -2e-4, 0.0]); D2 = np.array([-0.12, 0.03, -1e-4, 1e-4, 0.0])
        R = Rotation.from_rotvec(rng.standard_normal(3) * 0.08).as_matrix()
        T = np.array([-250.0 + rng.uniform(-20, 20), rng.uniform(-8, 8), rng.uniform(-8, 8)])
        R1, R2, P1, P2, Q = M.stereo_rectify(M1, D1, M2, D2, R, T, 4096, 3000)
        for Rk in (R1, R2):
            assert abs(np.linalg.det(Rk) - 1) < 1e-12 and np.allclose(Rk @ Rk.T, np.eye(3), atol=1e-12)
            assert np.allclose(Rotation.from_matrix(Rk).as_matrix(), Rk, atol=1e-12)       # SciPy takes it as a rotation unchanged
        assert np.allclose(R2 @ R @ R1.T, np.eye(3), atol=1e-12)            # both cameras end up with the SAME orientation
        t = R2 @ T                                                          # ... and the baseline along x
        assert abs(t[1]) < 1e-9 * abs(t[0]) and abs(t[2]) < 1e-9 * abs(t[0])
        assert P1[1, 1] == P2[1, 1] and P1[1, 2] == P2[1, 2]                # same rows: fy and cy shared
        assert np.isclose(P2[0, 3], P2[0, 0] * t[0]) and np.isclose(Q[3, 2], -1.0 / t[0])


def test_init_undistort_rectify_map_inverts_the_camera_model_by_root_finding():
    """cv::initUndistortRectifyMap (stereorect.cpp:42-43; oracle/slr_oracle.c, bit-identical to the device builder): the map sends a
    rectified pixel (u, v) to the raw pixel that sees the same ray.  Checked from the other side with a general-purpose solver:
    scipy.optimize.root finds the UNDISTORTED normalised point whose Brown-Conrady distortion (the formula of OpenCV's documentation)
    lands on the map's source pixel; rotated by R and projected by P it must give (u, v) again -- to the 1/32-pixel quantisation of
    the CV_16SC2 + CV_16UC1 map pair."""
    import oracle as O
    optimize = pytest.importorskip("scipy.optimize")
    W, H = 160, 120
    Mx = np.array([[210.0, 0, 81.3], [0, 214.0, 58.9], [0, 0, 1]])
    D = np.array([-0.21, 0.07, 1.5e-3, -9e-4, 0.012])
    R = Rotation.from_rotvec([0.03, -0.05, 0.02]).as_matrix()
    P = np.array([[200.0, 0, 80.0, 0], [0, 200.0, 60.0, 0], [0, 0, 1, 0]])
    xy, fr = O.init_undistort_rectify_map(Mx, D, R, P, W, H)
    sx = xy[..., 0] + (fr & 31) / 32.0
    sy = xy[..., 1] + (fr >> 5) / 32.0

    def distort(p):
        x, y = p
        r2 = x * x + y * y
        kr = 1 + D[0] * r2 + D[1] * r2 * r2 + D[4] * r2 * r2 * r2
        return np.array([x * kr + 2 * D[2] * x * y + D[3] * (r2 + 2 * x * x), y * kr + D[2] * (r2 + 2 * y * y) + 2 * D[3] * x * y])

    worst = 0.0
    for v in range(3, H, 13):
        for u in range(2, W, 11):
            target = np.array([(sx[v, u] - Mx[0, 2]) / Mx[0, 0], (sy[v, u] - Mx[1, 2]) / Mx[1, 1]])
            sol = optimize.root(lambda p: distort(p) - target, target, tol=1e-12)
            assert np.abs(distort(sol.x) - target).max() < 1e-10
            ray = R @ np.array([sol.x[0], sol.x[1], 1.0])              # the raw camera's ray in the rectified camera's frame
            uu = P[0, 0] * ray[0] / ray[2] + P[0, 2]
            vv = P[1, 1] * ray[1] / ray[2] + P[1, 2]
            worst = max(worst, abs(uu - u), abs(vv - v))
    # half a step of the 1/32 grid, magnified by the local scale of the map (focal ratio and distortion: < 1.2 here)
    assert worst < 0.6 / 32 * 1.2 + 1e-6, worst
