"""Committed golden fixtures (tests/golden/*.npz, written by tests/golden/make_golden.py): inputs are raw u8
planes + calibration, expected outputs are the CPU oracle's.  CPU half: the oracle still reproduces them.  GPU half:
the HIP path, through the C ABI, reproduces them -- integers bit-exact, floats within 1e-4 relative (asserted
bit-exact in practice).  Nothing here reads /root/reference."""
import importlib
import os

import numpy as np
import pytest

from util import assert_float_parity, bits_equal

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    return dict(np.load(os.path.join(G, name)))


def _oracle_cams(O, g):
    cams = []
    for row in g["cams"]:
        cams.append(O.Camera.make(row[0:2], row[2:4], row[4:9], row[9:18].reshape(3, 3), row[18:21]))
    return cams[0], cams[1], g["Q"], (g["T"] if int(g["has_T"]) else None)


def _slr_calib(slr, g):
    cams = [slr.make_camera(r[0:2], r[2:4], r[4:9], r[9:18].reshape(3, 3), r[18:21]) for r in g["cams"]]
    return slr.make_calib(cams[0], cams[1], g["Q"], g["T"] if int(g["has_T"]) else None)


# ------------------------------------------------------------------------------------------------ CPU half
def test_oracle_reproduces_mf_golden(oracle):
    g = _load("mf_64x48.npz")
    camL, camR, Q, T = _oracle_cams(oracle, g)
    black = int(g["black"])
    for tag in ("raw", "rect"):
        dec = []
        for cam in range(2):
            pl = g["stack"][cam]
            if tag == "rect":
                pl = np.stack([oracle.remap_u8(pl[p], g["map_xy"][cam], g["map_frac"][cam]) for p in range(14)])
                assert np.array_equal(pl, g["rectified_cam%d" % cam])
            ph, v = oracle.mf_decode(pl, black)
            assert bits_equal(ph, g["phase_%s_cam%d" % (tag, cam)]) and bits_equal(v, g["valid_%s_cam%d" % (tag, cam)])
            dec.append((ph, v))
        xyz, has, mk = oracle.mf_triangulate(dec[0][0], dec[0][1], dec[1][0], dec[1][1], camL, camR, Q, T)
        assert bits_equal(has, g["has_" + tag]) and bits_equal(mk, g["match_" + tag]) and bits_equal(xyz, g["xyz_" + tag])
        assert has.sum() > 100
    s, c, _ = oracle.pointcloud_from_grid(g["xyz_rect"], g["has_rect"], 40, 50)
    assert bits_equal(s, g["pc_sum"]) and bits_equal(c, g["pc_count"])


def test_oracle_reproduces_ka8_and_gray_goldens(oracle):
    k = _load("ka8_ideal_1280.npz")
    ph, v = oracle.mf_decode(k["planes"], 40)
    assert bits_equal(ph, k["phase"]) and bits_equal(v, k["valid"]) and len(np.unique(ph)) == 562
    g = _load("gray_96x40.npz")
    camL, camR, Q, T = _oracle_cams(oracle, g)
    ncol, nrow, sw, sh = int(g["ncol"]), int(g["nrow"]), int(g["scan_w"]), int(g["scan_h"])
    dec = [oracle.gray_decode(g["stack"][c], ncol, 0, int(g["black"]), int(g["white_thr"]), sw, 0) for c in range(2)]
    for c in range(2):
        assert bits_equal(dec[c][0], g["ge_code_cam%d" % c]) and bits_equal(dec[c][2], g["ge_valid_cam%d" % c])
    xyz, has, col, mk = oracle.ge_triangulate(dec[0][0], dec[0][2], dec[1][0], dec[1][2], Q, T,
                                              g["stack"][0, 0], g["stack"][1, 0])
    assert bits_equal(xyz, g["ge_xyz"]) and bits_equal(has, g["ge_has"]) and bits_equal(col, g["ge_color"])
    assert bits_equal(mk, g["ge_match"]) and has.sum() > 100
    assert (g["go_count"] > 0).sum() > 50


# ------------------------------------------------------------------------------------------------ GPU half
@pytest.mark.gpu
def test_hip_reproduces_mf_golden(ctx, slr):
    g = _load("mf_64x48.npz")
    ctx.set_calibration(_slr_calib(slr, g))
    for cam in range(2):
        ctx.set_rectify_maps(cam, np.ascontiguousarray(g["map_xy"][cam]), np.ascontiguousarray(g["map_frac"][cam]))
    black = int(g["black"])
    for cam in range(2):
        raw = np.ascontiguousarray(g["stack"][cam])
        assert np.array_equal(ctx.remap_u8(cam, raw[4]), g["rectified_cam%d" % cam][4])
        ph, v = ctx.mf_decode(raw, black)
        assert bits_equal(ph, g["phase_raw_cam%d" % cam]) and bits_equal(v, g["valid_raw_cam%d" % cam])
        ph, v = ctx.mf_decode(raw, black, rectify_cam=cam)
        assert bits_equal(ph, g["phase_rect_cam%d" % cam]) and bits_equal(v, g["valid_rect_cam%d" % cam])
    for tag, rectify in (("raw", False), ("rect", True)):
        xyz, has = ctx.reconstruct_mf(np.ascontiguousarray(g["stack"][0]), np.ascontiguousarray(g["stack"][1]), black, rectify)
        assert bits_equal(has, g["has_" + tag])
        assert assert_float_parity(xyz, g["xyz_" + tag], 1e-4, "golden xyz " + tag) == 0
        _, _, mk = ctx.mf_triangulate(g["phase_%s_cam0" % tag], g["valid_%s_cam0" % tag],
                                      g["phase_%s_cam1" % tag], g["valid_%s_cam1" % tag])
        assert bits_equal(mk, g["match_" + tag])
    s, c, _ = ctx.pointcloud_from_grid(g["xyz_rect"], g["has_rect"], 40, 50)
    assert bits_equal(s, g["pc_sum"]) and bits_equal(c, g["pc_count"])


@pytest.mark.gpu
def test_hip_reproduces_ka8_and_gray_goldens(ctx, slr):
    k = _load("ka8_ideal_1280.npz")
    ph, v = ctx.mf_decode(k["planes"], 40)
    assert bits_equal(ph, k["phase"]) and bits_equal(v, k["valid"])
    g = _load("gray_96x40.npz")
    ctx.set_calibration(_slr_calib(slr, g))
    ncol, nrow, sw, sh = int(g["ncol"]), int(g["nrow"]), int(g["scan_w"]), int(g["scan_h"])
    black, wthr = int(g["black"]), int(g["white_thr"])
    st = [np.ascontiguousarray(g["stack"][c]) for c in range(2)]
    for c in range(2):
        cx, _, vv = ctx.gray_decode(st[c], ncol, 0, black, wthr, sw, 0)
        assert bits_equal(cx, g["ge_code_cam%d" % c]) and bits_equal(vv, g["ge_valid_cam%d" % c])
        cx, cy, vv = ctx.gray_decode(st[c], ncol, nrow, black, wthr, sw, sh)
        assert bits_equal(cx, g["go_codex_cam%d" % c]) and bits_equal(cy, g["go_codey_cam%d" % c])
        assert bits_equal(vv, g["go_valid_cam%d" % c])
    xyz, has, col = ctx.reconstruct_ge(st[0], st[1], ncol, black, wthr, sw, False, True)
    assert bits_equal(has, g["ge_has"]) and bits_equal(col, g["ge_color"])
    assert assert_float_parity(xyz, g["ge_xyz"], 1e-4, "golden ge xyz") == 0
    xs, cnt = ctx.reconstruct_gray(st[0], st[1], ncol, nrow, black, wthr, sw, sh)
    assert bits_equal(cnt, g["go_count"])
    assert assert_float_parity(xs, g["go_xyz_sum"], 1e-4, "golden gray-only xyz") == 0
