"""SLR_OPT_EVAL_MODEL = 1: the multi-frequency path as the reference's own MSVC2010 x87 / fp:precise binary evaluates its source text
(include/slr.h) against the oracle's x87 model (oracle/slr_oracle_x87.c), bit for bit; and back to the default."""
import numpy as np
import pytest
import torch

from util import bits_equal, calib_parts

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
