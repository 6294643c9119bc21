"""Randomised shape / pitch / alignment sweep of the device entry points against the oracle (seeded, so reproducible):
odd widths (no dword path), pitch > W, plane pointers that are not dword aligned, single-row and single-column images,
rectification maps that wander outside the source.  Everything must still be bit-exact."""
import numpy as np
import pytest
import torch

from util import bits_equal, calib_parts, forms, np_of

pytestmark = pytest.mark.gpu
BLACK = 40


def _dev_planes(planes_np, pitch, offset):
    """device copies of [N][H][W] planes with the given row pitch, each plane starting `offset` bytes into its own
    allocation (offset % 4 != 0 -> unaligned plane pointers)"""
    N, H, W = planes_np.shape
    out, keep = [], []
    for p in range(N):
        buf = torch.zeros(offset + H * pitch + 8, dtype=torch.uint8, device="cuda")
        view = buf[offset:offset + H * pitch].view(H, pitch)
        view[:, :W] = torch.from_numpy(planes_np[p]).cuda()
        out.append(view)
        keep.append(buf)
    return out, keep


@pytest.mark.parametrize("seed", range(12))
def test_fuzz_decode_and_fused_rectify(ctx, oracle, synth, slr, seed):
    rng = np.random.default_rng(1000 + seed)
    W = int(rng.choice([1, 2, 3, 5, 31, 64, 100, 129, 200, 257, 380]))
    H = int(rng.choice([1, 2, 7, 8, 9, 33, 70]))
    pitch = W + int(rng.choice([0, 0, 1, 3, 4, 28]))
    offset = int(rng.choice([0, 0, 1, 2, 4, 6]))
    raw = rng.integers(0, 256, size=(14, H, W), dtype=np.uint8)
    raw[0] = rng.integers(90, 256, size=(H, W)); raw[1] = rng.integers(0, 110, size=(H, W))
    planes, keep = _dev_planes(raw, pitch, offset)
    # unfused
    eph, ev = oracle.mf_decode(raw, BLACK)
    ph, v = ctx.mf_decode(planes, BLACK, W=W)
    ctx.synchronize()
    assert bits_equal(np_of(v), ev) and bits_equal(np_of(ph), eph), (W, H, pitch, offset)
    # fused with a smooth map plus a random-jitter band
    mxt, mft = synth.make_rectify_maps(W, H, seed % 2, strength=4.0)
    mx, mf = mxt.numpy().copy(), mft.numpy().copy()
    if H > 4:
        mx[H // 2] = np.stack([rng.integers(-5, W + 5, W), rng.integers(-5, H + 5, W)], -1)
    ctx.set_rectify_maps(0, np.ascontiguousarray(mx), np.ascontiguousarray(mf))
    rect = np.stack([oracle.remap_u8(raw[p], mx, mf) for p in range(14)])
    eph, ev = oracle.mf_decode(rect, BLACK)
    for algo in forms(ctx, slr, slr.capi.OPT_RECT_DECODE_ALGO, (0, 1, 2, 3, 4, 5, 6), required=(0, 1, 5, 6)):
        ctx.set_option(slr.capi.OPT_RECT_DECODE_ALGO, algo)
        ph, v = ctx.mf_decode(planes, BLACK, W=W, rectify_cam=0)
        ctx.synchronize()
        assert bits_equal(np_of(v), ev) and bits_equal(np_of(ph), eph), (W, H, pitch, offset, algo)
    ctx.set_option(slr.capi.OPT_RECT_DECODE_ALGO, 0)
    del keep


@pytest.mark.parametrize("seed", range(8))
def test_fuzz_gray_decode_and_fused_rectify(ctx, oracle, synth, slr, seed):
    rng = np.random.default_rng(2000 + seed)
    W = int(rng.choice([3, 31, 64, 100, 132, 260]))
    H = int(rng.choice([1, 5, 8, 19, 40]))
    ncol = int(rng.integers(1, 13)); nrow = int(rng.choice([0, 0, 3, 9]))
    n = 2 + 2 * ncol + 2 * nrow
    pitch = W + int(rng.choice([0, 4, 5]))
    offset = int(rng.choice([0, 0, 2, 4]))
    raw = rng.integers(0, 256, size=(n, H, W), dtype=np.uint8)
    raw[0] = rng.integers(90, 256, size=(H, W)); raw[1] = rng.integers(0, 110, size=(H, W))
    planes, keep = _dev_planes(raw, pitch, offset)
    scan_w, scan_h, wt = int(rng.integers(1, 1 << ncol) + 1), int(rng.integers(1, 600)), int(rng.choice([0, 0, 7]))
    ex, ey, ev = oracle.gray_decode(raw, ncol, nrow, BLACK, wt, scan_w, scan_h)
    cx, cy, v = ctx.gray_decode(planes, ncol, nrow, BLACK, wt, scan_w, scan_h, W=W)
    ctx.synchronize()
    assert bits_equal(np_of(v), ev) and bits_equal(np_of(cx), ex)
    if nrow:
        assert bits_equal(np_of(cy), ey)
    mxt, mft = synth.make_rectify_maps(W, H, seed % 2, strength=4.0)
    ctx.set_rectify_maps(1, mxt.numpy(), mft.numpy())
    rect = np.stack([oracle.remap_u8(raw[p], mxt.numpy(), mft.numpy()) for p in range(n)])
    ex, ey, ev = oracle.gray_decode(rect, ncol, nrow, BLACK, wt, scan_w, scan_h)
    for algo, flags in ((0, 0), (1, 0), (6, 16), (5, 0), (6, 0)):     # auto, direct gather, LDS tiles 64x4 (debug flag), 128x8, 64x8
        ctx.set_option(slr.capi.OPT_RECT_DECODE_ALGO, algo)
        ctx.set_option(slr.capi.OPT_DEBUG_FLAGS, flags)
        cx, cy, v = ctx.gray_decode(planes, ncol, nrow, BLACK, wt, scan_w, scan_h, W=W, rectify_cam=1)
        ctx.synchronize()
        ctx.set_option(slr.capi.OPT_DEBUG_FLAGS, 0)
        assert bits_equal(np_of(v), ev) and bits_equal(np_of(cx), ex), (W, H, ncol, nrow, algo, flags)
    ctx.set_option(slr.capi.OPT_RECT_DECODE_ALGO, 0)
    del keep


@pytest.mark.parametrize("seed", range(10))
def test_fuzz_match_kernels(ctx, oracle, synth, slr, seed):
    """K4 (all three forms) and K5 on random rows: quantised phases / codes so that ties, duplicates and empty lists occur"""
    rng = np.random.default_rng(3000 + seed)
    W = int(rng.choice([1, 2, 5, 63, 64, 65, 255, 257, 600, 1025, 2049]))
    H = int(rng.choice([1, 3, 6]))
    calib, _ = synth.make_calibration(max(W, 8), max(H, 8), with_T=bool(seed % 2))
    ctx.set_calibration(calib)
    camL, camR, Q, T = calib_parts(oracle, calib)
    q = float(rng.choice([0.03, 0.07, 1.0, 13.0]))
    phL = (rng.integers(-50, 400, (H, W)) * q).astype(np.float32)
    phR = (rng.integers(-50, 400, (H, W)) * q).astype(np.float32)
    vL = (rng.random((H, W)) < 0.85).astype(np.uint8); vR = (rng.random((H, W)) < 0.85).astype(np.uint8)
    exyz, ehas, emk = oracle.mf_triangulate(phL, vL, phR, vR, camL, camR, Q, T)
    for algo in forms(ctx, slr, slr.capi.OPT_MF_MATCH_ALGO, (0, 1, 2, 3), required=(0, 1, 3)):
        ctx.set_option(slr.capi.OPT_MF_MATCH_ALGO, algo)
        xyz, has, mk = ctx.mf_triangulate(phL, vL, phR, vR)
        assert bits_equal(mk, emk) and bits_equal(has, ehas) and bits_equal(xyz, exyz), (W, H, algo)
    ctx.set_option(slr.capi.OPT_MF_MATCH_ALGO, 0)
    ncodes = int(rng.choice([1, 3, 40, 700]))
    cL = rng.integers(0, ncodes, (H, W)).astype(np.int32); cR = rng.integers(0, ncodes, (H, W)).astype(np.int32)
    if seed % 3 == 0:
        cL.sort(axis=1); cR.sort(axis=1)
    exyz, ehas, _, emk = oracle.ge_triangulate(cL, vL, cR, vR, Q, T)
    xyz, has, _, mk = ctx.ge_triangulate(cL, vL, cR, vR)
    assert bits_equal(mk, emk) and bits_equal(has, ehas) and bits_equal(xyz, exyz), (W, H, ncodes)


@pytest.mark.parametrize("seed", range(12))
def test_fuzz_ray_buckets(ctx, oracle, synth, slr, seed):
    """K3' + K6 on random codes: bucket lengths from 0 to hundreds in one frame (the sorting-network form, its register rows 12..15,
    the staged form and their mixtures inside one wave / workgroup), invalid pixels, codes beyond the projector (Q9), with and
    without the transfer matrix"""
    rng = np.random.default_rng(9000 + seed)
    W, H = int(rng.integers(16, 260)), int(rng.integers(8, 120))
    scan_w, scan_h = int(rng.integers(4, 80)), int(rng.integers(4, 60))
    calib, _ = synth.make_calibration(W, H, with_T=bool(seed % 2), baseline=400.0, theta=0.6)
    ctx.set_calibration(calib)
    camL, camR, _, T = calib_parts(oracle, calib)
    def codes():
        kind = rng.integers(0, 3)
        if kind == 0:                                         # smooth (a camera looking at the projector): short buckets
            cx = (np.arange(W)[None, :] * scan_w // W + rng.integers(-1, 2, (H, W))).astype(np.int32)
            cy = (np.arange(H)[:, None] * scan_h // H + rng.integers(-1, 2, (H, W))).astype(np.int32)
        elif kind == 1:                                       # uniform: lengths around W H / cells
            cx = rng.integers(0, scan_w + 2, (H, W)).astype(np.int32); cy = rng.integers(0, scan_h + 2, (H, W)).astype(np.int32)
        else:                                                 # a few cells take most pixels: long buckets beside empty ones
            hot = rng.integers(0, scan_w, 5), rng.integers(0, scan_h, 5)
            pick = rng.integers(0, 5, (H, W))
            cx = np.where(rng.random((H, W)) < 0.7, hot[0][pick], rng.integers(0, scan_w, (H, W))).astype(np.int32)
            cy = np.where(rng.random((H, W)) < 0.7, hot[1][pick], rng.integers(0, scan_h, (H, W))).astype(np.int32)
        v = (rng.random((H, W)) < rng.choice([0.3, 0.8, 1.0])).astype(np.uint8)
        return np.clip(cx, 0, None), np.clip(cy, 0, None), v
    cxL, cyL, vL = codes()
    cxR, cyR, vR = codes()
    offL, itL = oracle.gray_bucket(cxL, cyL, vL, scan_w, scan_h)
    offR, itR = oracle.gray_bucket(cxR, cyR, vR, scan_w, scan_h)
    exyz, ecnt = oracle.ray_triangulate(offL, itL, offR, itR, camL, camR, scan_w, scan_h, T)
    xyz, cnt = ctx.ray_triangulate(cxL, cyL, vL, cxR, cyR, vR, scan_w, scan_h)
    assert bits_equal(cnt, ecnt) and bits_equal(xyz, exyz), (W, H, scan_w, scan_h)


@pytest.mark.parametrize("seed", range(6))
def test_fuzz_mf_batch_frame_groups_on_verged_rigs(ctx, slr, synth, seed):
    """the grouped launches of slr_reconstruct_mf_batch (round 4: the cameras of up to 8 frames per fused-decode launch -- 16 ticket pools
    per XCD --, a group of frames per match launch) on the keystone maps of small verged rigs, where tiles are split into parts, pools
    hold few entries and workgroups find nothing to do: every group size against frame-by-frame launches, bit for bit"""
    rng = np.random.default_rng(4000 + seed)
    W, H = [(1040, 524), (528, 260), (2064, 120), (4096, 64), (3008, 75), (2560, 301)][seed]
    frames = int(rng.integers(2, 10))
    rig = synth.make_verged_rig(W, H, float(rng.uniform(0.1, 0.3)), float(rng.uniform(-0.2, -0.05)))
    ctx.set_calibration(rig["calib"])
    synth.install_verged_maps(ctx, rig, W, H)
    stack = torch.stack([synth.render_mf_stack(W, H, seed=int(rng.integers(1 << 20)), noise=2) for _ in range(frames)]).cuda()
    res = {}
    try:
        for mg, dg in ((1, 1), (8, 8), (int(rng.integers(2, 8)), int(rng.integers(2, 8))), (8, 1)):
            ctx.set_option(slr.capi.OPT_MF_BATCH_GROUP, mg)
            ctx.set_option(slr.capi.OPT_MF_BATCH_DECODE_GROUP, dg)
            x, h = ctx.reconstruct_mf_batch(stack, 40, True)
            ctx.synchronize()
            res[(mg, dg)] = (x.clone(), h.clone())
    finally:
        ctx.set_option(slr.capi.OPT_MF_BATCH_GROUP, 8)
        ctx.set_option(slr.capi.OPT_MF_BATCH_DECODE_GROUP, 8)
    ref = res[(1, 1)]
    for key, (x, h) in res.items():
        assert torch.equal(x.view(torch.int32), ref[0].view(torch.int32)) and torch.equal(h, ref[1]), (W, H, frames, key)
    assert ref[1].float().mean().item() > 0.02
