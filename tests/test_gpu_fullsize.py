"""BASELINE.json's full size (4096x3000) on the GPU: direct comparison with the oracle where the oracle finishes in
seconds (decode, rectify, Gray, GE), sampled rows + size-independent properties where it does not (the O(W^2) MF
match: indexed form == literal sweep on the whole frame; oracle on a handful of rows)."""
import numpy as np
import pytest
import torch

from util import bits_equal, calib_parts, np_of

pytestmark = pytest.mark.gpu
W, H, BLACK = 4096, 3000, 40


@pytest.fixture(scope="module")
def scene(synth):
    dev = torch.device("cuda", 0)
    st = synth.render_mf_stack(W, H, seed=1234, device=dev)
    maps = [synth.make_rectify_maps(W, H, cam, device=dev) for cam in range(2)]
    calib, info = synth.make_calibration(W, H)
    torch.cuda.synchronize()
    return st, maps, calib, info


def test_fullsize_mf_decode_and_fused_rectify(ctx, oracle, scene):
    st, maps, calib, _ = scene
    for cam in range(2):
        ctx.set_rectify_maps(cam, maps[cam][0], maps[cam][1])
    pl = st[0].cpu().numpy()
    eph, ev = oracle.mf_decode(pl, BLACK)
    ph, v = ctx.mf_decode(st[0], BLACK)
    ctx.synchronize()
    assert bits_equal(np_of(v), ev) and bits_equal(np_of(ph), eph)
    mx, mf = maps[1][0].cpu().numpy(), maps[1][1].cpu().numpy()
    raw = st[1].cpu().numpy()
    rect = np.stack([oracle.remap_u8(raw[p], mx, mf) for p in range(14)])
    eph, ev = oracle.mf_decode(rect, BLACK)
    ph, v = ctx.mf_decode(st[1], BLACK, rectify_cam=1)
    r5 = ctx.remap_u8(1, st[1, 5])
    ctx.synchronize()
    assert np.array_equal(np_of(r5), rect[5])
    assert bits_equal(np_of(v), ev) and bits_equal(np_of(ph), eph)
    assert 0.5 < ev.mean() < 1.0


def test_fullsize_mf_match_forms_agree_and_oracle_rows(ctx, oracle, scene, slr):
    st, maps, calib, _ = scene
    ctx.set_calibration(calib)
    for cam in range(2):
        ctx.set_rectify_maps(cam, maps[cam][0], maps[cam][1])
    dec = [ctx.mf_decode(st[cam], BLACK, rectify_cam=cam) for cam in range(2)]
    out = {}
    for algo in (3, 2, 1):
        ctx.set_option(slr.capi.OPT_MF_MATCH_ALGO, algo)
        out[algo] = ctx.mf_triangulate(dec[0][0], dec[0][1], dec[1][0], dec[1][1])
    ctx.set_option(slr.capi.OPT_MF_MATCH_ALGO, 0)
    ctx.synchronize()
    for algo in (2, 3):
        for a, b in zip(out[1], out[algo]):
            assert torch.equal(a, b)                   # whole frame: indexed forms == literal sweep, bit for bit
    xyz, has, mk = [np_of(t) for t in out[2]]
    assert 0.05 < has.mean() < 1.0
    camL, camR, Q, T = calib_parts(oracle, calib)
    phL, vL, phR, vR = [np_of(t) for t in (dec[0][0], dec[0][1], dec[1][0], dec[1][1])]
    for r in (0, 1, 777, 1500, 2999):
        exyz, ehas, emk = oracle.mf_triangulate(phL, vL, phR, vR, camL, camR, Q, T, rows=(r, r + 1))
        assert bits_equal(mk[r], emk[r]) and bits_equal(has[r], ehas[r]) and bits_equal(xyz[r], exyz[r]), r
    # whole-path entry point == the three stages
    x2, h2 = ctx.reconstruct_mf(st[0], st[1], BLACK, True)
    ctx.synchronize()
    assert torch.equal(x2, out[2][0]) and torch.equal(h2, out[2][1])


def test_fullsize_gray_decode_and_ge(ctx, oracle, synth, scene):
    _, _, calib, info = scene
    ctx.set_calibration(calib)
    dev = torch.device("cuda", 0)
    g = synth.render_gray_stack(W, H, W, seed=1234, noise=2, device=dev)
    ncol = synth.gray_num_bits(W)
    assert ncol == 12 and g.shape[1] == 26
    torch.cuda.synchronize()
    dec, edec = [], []
    for cam in range(2):
        cx, _, v = ctx.gray_decode(g[cam], ncol, 0, BLACK, 0, W, 0)
        ex, _, ev = oracle.gray_decode(g[cam].cpu().numpy(), ncol, 0, BLACK, 0, W, 0)
        ctx.synchronize()
        assert bits_equal(np_of(cx), ex) and bits_equal(np_of(v), ev)
        dec.append((cx, v))
        edec.append((ex, ev))
    _, _, Q, T = calib_parts(oracle, calib)
    exyz, ehas, _, emk = oracle.ge_triangulate(edec[0][0], edec[0][1], edec[1][0], edec[1][1], Q, T)
    xyz, has, _, mk = ctx.ge_triangulate(dec[0][0], dec[0][1], dec[1][0], dec[1][1])
    ctx.synchronize()
    assert bits_equal(np_of(mk), emk) and bits_equal(np_of(has), ehas) and bits_equal(np_of(xyz), exyz)
    assert ehas.mean() > 0.5
    # size-independent property (KA7): depth from disparity, Z = f*Tx/((cx1-cx2) - d)
    d = (np.arange(W)[None, :] - emk).astype(np.float64)
    Z = info["f"] * info["Tx"] / ((info["cx1"] - info["cx2"]) - d)
    sel = ehas.astype(bool)
    assert np.allclose(np_of(xyz)[..., 2][sel], Z[sel], rtol=1e-5)


def test_fullsize_whole_path_pair_launch_equals_step_by_step(ctx, scene):
    """slr_reconstruct_mf_batch (both cameras' fused rectify+decode in ONE launch, then K4) == the three separate calls,
    bit for bit, on the bench workload; two frames in the batch so that the frame stride is exercised too"""
    st, maps, calib, _ = scene
    ctx.set_calibration(calib)
    for cam in range(2):
        ctx.set_rectify_maps(cam, maps[cam][0], maps[cam][1])
    st2 = torch.stack([st, torch.flip(st, dims=[0])]).contiguous()       # frame 1 = cameras swapped
    torch.cuda.synchronize()            # the ctx has its own stream: inputs made by torch must be complete first
    xyz, has = ctx.reconstruct_mf_batch(st2, BLACK, True)
    ctx.synchronize()
    for f in range(2):
        dec = [ctx.mf_decode(st2[f, cam], BLACK, rectify_cam=cam) for cam in range(2)]
        ex, eh, _ = ctx.mf_triangulate(dec[0][0], dec[0][1], dec[1][0], dec[1][1])
        ctx.synchronize()
        assert torch.equal(has[f], eh) and torch.equal(xyz[f], ex), f
    assert 0.2 < has[0].float().mean().item() < 1.0
