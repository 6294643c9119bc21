"""Round-2 kernel forms that only engage at sizes the small parity tests do not reach, against the oracle, bit for bit:
  * mf_match_lean_kernel (K4, rows of 513..1024 and 2049..4096 pixels) vs the general binned form and the oracle;
  * ge_match_lean_kernel (K5, rows of 2049..4096 pixels): counting-sort lists, rows deferred to the general kernel (codes >= 8192,
    lists longer than 32 columns), the kstart carry on adversarial rows, colour, T;
  * gray_rect_decode_dma_kernel (fused rectify + Gray decode, LDS-DMA form): odd and even phase counts, column-only and
    column+row stacks, every tile shape it is built for, many tiles per workgroup, borders."""
import numpy as np
import pytest
import torch

from util import bits_equal, calib_parts, forms, np_of

pytestmark = pytest.mark.gpu
BLACK = 40


# ---------------------------------------------------------------------------------------------------------
# K4 lean
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("W,H,with_T,q", [(516, 5, True, 0.07), (1024, 4, False, 1.0), (2052, 3, True, 0.03), (4096, 3, False, 0.07),
                                          (4096, 2, True, 13.0), (3000, 3, True, 0.05)])
def test_k4_lean_vs_general_and_oracle(ctx, slr, oracle, synth, W, H, with_T, q):
    rng = np.random.default_rng(W + H)
    calib, _ = synth.make_calibration(max(W, 8), max(H, 8), with_T=with_T)
    ctx.set_calibration(calib)
    camL, camR, Q, T = calib_parts(oracle, calib)
    phL = (rng.integers(-50, 400, (H, W)) * q).astype(np.float32)
    phR = (rng.integers(-50, 400, (H, W)) * q).astype(np.float32)
    phR[0] = np.sort(phR[0]); phL[0] = np.sort(phL[0])                 # a monotone row (the usual case)
    phR[-1, ::3] = np.float32(7.25)                                     # one value in a third of the row: a long duplicate run
    phL[-1, 5] = np.float32(7.3); phL[-1, 6] = np.float32(7.3500004)    # around the 0.1 threshold
    phL[-1, 7] = np.float32(600.0); phR[-1, 1] = np.float32(-600.0)     # clamped bins
    vL = (rng.random((H, W)) < 0.85).astype(np.uint8); vR = (rng.random((H, W)) < 0.85).astype(np.uint8)
    phL[1 % H, 9] = np.nan; phR[1 % H, 11] = np.nan
    exyz, ehas, emk = oracle.mf_triangulate(phL, vL, phR, vR, camL, camR, Q, T)
    try:
        for algo in forms(ctx, slr, slr.capi.OPT_MF_MATCH_ALGO, (0, 3, 2, 4, 5, 6, 8, 9), required=(0, 3, 4, 8)):   # lean (auto), general binned, sorted (FORMS=all), the lean shapes, 8 = lean with the hash dedup (= auto), 9 = without it (FORMS=all)
            ctx.set_option(slr.capi.OPT_MF_MATCH_ALGO, algo)
            xyz, has, mk = ctx.mf_triangulate(phL, vL, phR, vR)
            assert bits_equal(mk, emk) and bits_equal(has, ehas) and bits_equal(xyz, exyz), (W, algo)
    finally:
        ctx.set_option(slr.capi.OPT_MF_MATCH_ALGO, 0)
    assert ehas.sum() > 0


@pytest.mark.parametrize("W,with_T,x87", [(4096, False, False), (2052, True, False), (4096, True, True)])
def test_k4_lean_index_without_the_hash_light_and_heavy_rows(ctx, slr, oracle, synth, W, with_T, x87):
    """round 6: the lean K4's index with and without the hash dedup (SLR_OPT_MF_MATCH_ALGO 9, FORMS=all builds).  Rows whose fullest
    0.25-wide bin holds exactly 32 / 33 / many more right pixels (the no-hash form's in-kernel switch to the dedup is at > 32),
    duplicates that are not neighbours, duplicates across thread boundaries, an all-equal row, an empty row: every form == the
    general binned kernel == the oracle, bit for bit"""
    rng = np.random.default_rng(W)
    H = 8
    calib, _ = synth.make_calibration(max(W, 8), 8, with_T=with_T)
    ctx.set_calibration(calib)
    camL, camR, Q, T = calib_parts(oracle, calib)
    base = (np.arange(W, dtype=np.float32) * np.float32(0.0625) + np.float32(3.0)).astype(np.float32)   # 4 pixels per bin: a light row
    phR = np.tile(base, (H, 1)); phL = np.tile(base + np.float32(0.03), (H, 1))
    # row 1: one bin with exactly 32 entries of distinct values (light), row 2: 33 (heavy), the rest spread out
    for r, n in ((1, 32), (2, 33)):
        phR[r] = base * np.float32(4.0)                                  # one pixel per bin ...
        cols = rng.choice(W, n, replace=False)
        phR[r, cols] = np.float32(700.0) + np.arange(n, dtype=np.float32) * np.float32(0.005)   # ... and n of them in the clamped top bin
        phL[r] = phR[r][::-1].copy()
    phR[3] = np.float32(12.5); phL[3] = np.float32(12.55)               # all equal: one value, column 0 wins everywhere
    phR[4, :] = rng.choice(np.float32([5.0, 5.05, 5.1, 5.2, 90.0]), W)   # five values, thousands of non-adjacent duplicates
    phL[4, :] = rng.choice(np.float32([5.0, 5.12, 5.3, 89.95, 40.0]), W)
    phR[5, 3:-1:4] = phR[5, 4::4]                                        # equal pairs across the 4-pixel thread boundary
    vL = np.ones((H, W), np.uint8); vR = np.ones((H, W), np.uint8)
    vR[6] = 0                                                             # nothing to match against
    vL[7] = (rng.random(W) < 0.5); vR[7] = (rng.random(W) < 0.5)
    if x87:
        ctx.set_option(slr.capi.OPT_EVAL_MODEL, 1)
    try:
        exyz, ehas, emk = oracle.mf_triangulate_ev(phL, vL, phR, vR, camL, camR, Q, x87, T)
        for algo in forms(ctx, slr, slr.capi.OPT_MF_MATCH_ALGO, (0, 8, 3, 9), required=(0, 8, 3)):
            ctx.set_option(slr.capi.OPT_MF_MATCH_ALGO, algo)
            xyz, has, mk = ctx.mf_triangulate(phL, vL, phR, vR)
            assert bits_equal(mk, emk) and bits_equal(has, ehas) and bits_equal(xyz, exyz), (W, algo)
        assert ehas[3].all() and (emk[3] == 0).all() and ehas[6].sum() == 0 and ehas[1].sum() > 0 and ehas[2].sum() > 0
    finally:
        ctx.set_option(slr.capi.OPT_MF_MATCH_ALGO, 0)
        ctx.set_option(slr.capi.OPT_EVAL_MODEL, 0)


def test_k4_wide_rows_flat_phases_are_not_quadratic(ctx, slr, oracle, synth):
    """rows of 4097..8192 pixels take mf_match_wide_kernel, whose index has no dedup of equal phases -- a flat or saturated
    row would make every query scan the whole row.  Rows with an overfull bin are now listed by that kernel and matched by the chunked
    kernel (hash dedup) behind it: a frame of constant-phase rows must cost about what a frame of distinct phases costs (not W / 8
    times more), and equal the oracle bit for bit on sampled rows."""
    W, H = 8192, 1024
    calib, _ = synth.make_calibration(W, H, with_T=False)
    ctx.set_calibration(calib)
    camL, camR, Q, T = calib_parts(oracle, calib)
    ramp = (np.arange(W, dtype=np.float32) * np.float32(0.03) + np.float32(1.0))
    distinct = [torch.from_numpy(np.tile(ramp, (H, 1))).cuda(), torch.from_numpy(np.tile(ramp + np.float32(0.01), (H, 1))).cuda()]
    flatL = np.full((H, W), 12.55, np.float32); flatR = np.full((H, W), 12.5, np.float32)
    flatR[1::2, ::2] = np.float32(77.0)                  # odd rows: two values, alternating columns (no equal neighbours inside a thread)
    flatL[1::2, 1::3] = np.float32(77.05)
    flat = [torch.from_numpy(flatL).cuda(), torch.from_numpy(flatR).cuda()]
    ones = torch.ones((H, W), dtype=torch.uint8, device="cuda")

    def timed(ph):
        for _ in range(2):
            out = ctx.mf_triangulate(ph[0], ones, ph[1], ones, want_match=True)
        ctx.synchronize()
        ctx.timer_begin()
        for _ in range(3):
            out = ctx.mf_triangulate(ph[0], ones, ph[1], ones, want_match=True)
        return ctx.timer_end() / 3, out

    t_distinct, _ = timed(distinct)
    t_flat, (xyz, has, mk) = timed(flat)
    print("wide K4, %d rows of %d: distinct phases %.3f ms, flat rows %.3f ms" % (H, W, t_distinct, t_flat))
    assert t_flat < 6 * t_distinct + 0.5, (t_flat, t_distinct)
    rows = [0, 1, 2, 511, 1023]
    exyz, ehas, emk = oracle.mf_triangulate(flatL[rows], np.ones((len(rows), W), np.uint8), flatR[rows], np.ones((len(rows), W), np.uint8),
                                            camL, camR, Q, T)
    # (the oracle's rows are image rows 0..4 of its own frame: compare the match columns and masks; XYZ on row 0, which is row 0 in both)
    assert bits_equal(np_of(mk)[rows], emk) and bits_equal(np_of(has)[rows], ehas)
    assert bits_equal(np_of(xyz)[0], exyz[0])
    assert (emk[0] == 0).all() and ehas[1].sum() > 0


# ---------------------------------------------------------------------------------------------------------
# K5 lean
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("W,with_T,color", [(2052, True, False), (4096, False, True), (3000, True, True)])
def test_k5_lean_lists_deferrals_and_kstart(ctx, slr, oracle, synth, W, with_T, color):
    rng = np.random.default_rng(W)
    H = 8
    calib, _ = synth.make_calibration(W, H, with_T=with_T)
    ctx.set_calibration(calib)
    _, _, Q, T = calib_parts(oracle, calib)
    cL = np.empty((H, W), np.int32); cR = np.empty((H, W), np.int32)
    base = np.arange(W, dtype=np.int32)
    cL[0] = base + 100; cR[0] = base + 140                               # the usual row: monotone, one column per code
    cL[1] = rng.integers(0, 3000, W); cR[1] = rng.integers(0, 3000, W)    # random codes: short unordered lists, kstart carries
    cL[2] = np.sort(rng.integers(0, 1500, W)); cR[2] = np.sort(rng.integers(0, 1500, W))
    cL[3] = rng.integers(0, 40, W); cR[3] = rng.integers(0, 40, W)        # lists of ~100 columns: deferred to the general kernel
    cL[4] = rng.integers(8000, 8400, W); cR[4] = rng.integers(8000, 8400, W)   # codes on both sides of 8192: deferred
    cL[5] = base // 2 + 7; cR[5] = base // 3 + 7                          # lists of 2-3 columns, monotone
    cL[6] = rng.integers(0, 3000, W); cR[6] = cL[6][::-1].copy()
    cL[7] = base; cR[7] = base; cR[7, W // 2] = 70000                      # one code beyond 16 bits ("no code"), rest identity
    vL = (rng.random((H, W)) < 0.9).astype(np.uint8); vR = (rng.random((H, W)) < 0.9).astype(np.uint8)
    vL[7] = 1; vR[7] = 1
    wl = rng.integers(0, 256, (H, W), dtype=np.uint8) if color else None
    wr = rng.integers(0, 256, (H, W), dtype=np.uint8) if color else None
    exyz, ehas, ecol, emk = oracle.ge_triangulate(cL, vL, cR, vR, Q, T, wl, wr)
    try:
        for flags in (0, 4):                                              # lean + deferrals | the general kernel for every row
            ctx.set_option(slr.capi.OPT_DEBUG_FLAGS, flags)
            xyz, has, col, mk = ctx.ge_triangulate(cL, vL, cR, vR, wl, wr)
            assert bits_equal(mk, emk), flags
            assert bits_equal(has, ehas) and bits_equal(xyz, exyz), flags
            if color:
                assert bits_equal(col, ecol), flags
    finally:
        ctx.set_option(slr.capi.OPT_DEBUG_FLAGS, 0)
    assert emk[7, W // 2] == -1 or cL[7, W // 2] != 70000
    assert all(ehas[r].sum() > 0 for r in (0, 1, 2, 3, 4, 5, 7))


# ---------------------------------------------------------------------------------------------------------
# fused rectify + Gray decode, LDS-DMA form
# ---------------------------------------------------------------------------------------------------------
@pytest.fixture
def dma_ctx(ctx, slr):
    yield ctx
    ctx.set_option(slr.capi.OPT_DEBUG_RECT_RESIDENT, 0)
    ctx.set_option(slr.capi.OPT_RECT_DMA_SHAPE, 3)
    ctx.set_option(slr.capi.OPT_RECT_DECODE_ALGO, 0)


def _gray_expect(oracle, raw, mx, mf, nc, nr, wt, sw, sh):
    rect = np.stack([oracle.remap_u8(raw[p], mx, mf) for p in range(raw.shape[0])])
    return oracle.gray_decode(rect, nc, nr, BLACK, wt, sw, sh)


@pytest.mark.parametrize("W,H,sw,sh,rows,resident", [(640, 200, 600, 0, False, 0),      # 10 bits: 11 pairs
                                                     (640, 120, 1280, 0, False, 8),     # 11 bits: 12 pairs
                                                     (320, 96, 4096, 0, False, 8),      # 12 bits: 13 pairs (the bench's stack)
                                                     (256, 80, 100, 90, True, 8),       # 7 + 7 bits: 15 pairs
                                                     (512, 64, 1280, 1024, True, 16),   # 11 + 10: 22 pairs
                                                     (272, 33, 500, 3, True, 0)])       # 9 + 2: 12 pairs, ragged height
def test_gray_dma_form(dma_ctx, slr, oracle, synth, W, H, sw, sh, rows, resident):
    ctx = dma_ctx
    ctx.set_option(slr.capi.OPT_DEBUG_RECT_RESIDENT, resident)
    st = synth.render_gray_stack(W, H, sw, sh if rows else None, seed=W + sw, noise=3, rows=rows)
    nc = synth.gray_num_bits(sw); nr = synth.gray_num_bits(sh) if rows else 0
    for cam in range(2):
        mx, mf = synth.make_rectify_maps(W, H, cam, strength=1.0)
        mxn, mfn = mx.numpy(), mf.numpy()
        raw = st[cam].numpy()
        ex, ey, ev = _gray_expect(oracle, raw, mxn, mfn, nc, nr, 3, sw, sh)
        dev = st[cam].cuda()
        ran = 0
        for shape in forms(ctx, slr, slr.capi.OPT_RECT_DMA_SHAPE, (1, 3, 4, 5), default=3, required=(1, 3)):
            ctx.set_option(slr.capi.OPT_RECT_DMA_SHAPE, shape)
            ctx.set_option(slr.capi.OPT_RECT_DECODE_ALGO, 7)                    # strict: an error if the form does not run
            ctx.set_rectify_maps(cam, mxn, mfn)
            try:
                cx, cy, v = ctx.gray_decode(dev, nc, nr, BLACK, 3, sw, sh, rectify_cam=cam)
                ctx.synchronize()
            except slr.capi.SlrError as e:                                      # a corner tile's box exceeds a 256-thread shape
                assert e.status == slr.capi.ERR_UNSUPPORTED and shape in (4, 5), (shape, str(e))
                continue
            assert bits_equal(np_of(cx), ex) and bits_equal(np_of(v), ev), (cam, shape)
            if rows:
                assert bits_equal(np_of(cy), ey), (cam, shape)
            ran += 1
        assert ran >= 2
    assert ev.mean() > 0.2


def test_gray_dma_form_borders_and_strictness(dma_ctx, slr, oracle, synth):
    ctx = dma_ctx
    W, H, sw = 512, 64, 700
    st = synth.render_gray_stack(W, H, sw, seed=5, noise=3)
    nc = synth.gray_num_bits(sw)
    raw = st[0].numpy() | 1
    dev = torch.from_numpy(raw).cuda()
    ctx.set_option(slr.capi.OPT_RECT_DECODE_ALGO, 7)
    for dx, dy, fx, fy in [(-37, -9, 7, 19), (41, 11, 31, 31), (700, 0, 3, 3), (0, 90, 1, 1), (-16, -16, 0, 0)]:
        mx, mf = synth.identity_maps(W, H, dx=dx, dy=dy, fx=fx, fy=fy)
        mxn, mfn = mx.numpy(), mf.numpy()
        ex, _, ev = _gray_expect(oracle, raw, mxn, mfn, nc, 0, 0, sw, 0)
        ctx.set_rectify_maps(0, mxn, mfn)
        cx, _, v = ctx.gray_decode(dev, nc, 0, BLACK, 0, sw, 0, rectify_cam=0)
        ctx.synchronize()
        assert bits_equal(np_of(cx), ex) and bits_equal(np_of(v), ev), (dx, dy)
    # a one-bit stack has a single phase per tile: not this form (strict 7 says so, auto takes the round-1 kernel)
    with pytest.raises(slr.capi.SlrError) as ei:
        ctx.gray_decode(dev[:4], 1, 0, BLACK, 0, 2, 0, rectify_cam=0)
    assert ei.value.status == slr.capi.ERR_UNSUPPORTED
    ctx.set_option(slr.capi.OPT_RECT_DECODE_ALGO, 0)
    ex, _, ev = _gray_expect(oracle, raw[:4], mxn, mfn, 1, 0, 0, 2, 0)
    cx, _, v = ctx.gray_decode(dev[:4], 1, 0, BLACK, 0, 2, 0, rectify_cam=0)
    ctx.synchronize()
    assert bits_equal(np_of(cx), ex) and bits_equal(np_of(v), ev)


# ---------------------------------------------------------------------------------------------------------
# K4 over a group of frames in one launch (slr_reconstruct_mf_batch, SLR_OPT_MF_BATCH_GROUP)
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("W,H,frames", [(4096, 21, 5), (2052, 9, 3), (1024, 12, 3), (4096, 300, 9)])
def test_mf_batch_frame_groups_equal_frame_by_frame(ctx, slr, synth, W, H, frames):
    """the batch entry with groups of 1 (every frame on its own: rounds 1-3), 2, 4 and 8 frames per match launch (rows that are not a
    multiple of the 8 a group of workgroups takes, a last group that is not full, a row width the lean kernel does not take: those
    fall back to frame-by-frame launches from the group's phase scratch) -- identical clouds, and frame 0 == the single-frame entry"""
    calib, _ = synth.make_calibration(W, H, with_T=True)
    ctx.set_calibration(calib)
    for cam in range(2):
        mx, mf = synth.make_rectify_maps(W, H, cam, strength=2.0)
        ctx.set_rectify_maps(cam, mx.numpy(), mf.numpy())
    stack = torch.stack([synth.render_mf_stack(W, H, seed=300 + f, noise=2) for f in range(frames)]).cuda()
    res = {}
    try:
        for group in (1, 2, 4, 8, 8 + 256, 3, 8 + 512):  # (+ 256: the group's fused decodes frame by frame, + 512: three frames per decode launch)
            ctx.set_option(slr.capi.OPT_MF_BATCH_GROUP, group & 255)
            ctx.set_option(slr.capi.OPT_MF_BATCH_DECODE_GROUP, 1 if group & 256 else 3 if group & 512 else 8)
            x, h = ctx.reconstruct_mf_batch(stack, BLACK, True)
            ctx.synchronize()
            res[group] = (x.clone(), h.clone())
        # SLR_OPT_MF_MATCH_ALGO 7: the grouped match launch as the persistent kernel (round 5, opt-in: measured slower)
        ctx.set_option(slr.capi.OPT_MF_BATCH_GROUP, 8)
        ctx.set_option(slr.capi.OPT_MF_BATCH_DECODE_GROUP, 8)
        if 7 in forms(ctx, slr, slr.capi.OPT_MF_MATCH_ALGO, (7,)):       # (`make FORMS=all` builds only)
            ctx.set_option(slr.capi.OPT_MF_MATCH_ALGO, 7)
            x, h = ctx.reconstruct_mf_batch(stack, BLACK, True)
            ctx.synchronize()
            res["persist"] = (x.clone(), h.clone())
    finally:
        ctx.set_option(slr.capi.OPT_MF_MATCH_ALGO, 0)
        ctx.set_option(slr.capi.OPT_MF_BATCH_GROUP, 8)
        ctx.set_option(slr.capi.OPT_MF_BATCH_DECODE_GROUP, 8)
    for group in [g for g in (2, 4, 8, 8 + 256, 3, 8 + 512, "persist") if g in res]:
        assert torch.equal(res[group][0].view(torch.int32), res[1][0].view(torch.int32)) and torch.equal(res[group][1], res[1][1]), group
    x0, h0 = ctx.reconstruct_mf(stack[0, 0], stack[0, 1], BLACK, True)
    ctx.synchronize()
    assert torch.equal(x0.view(torch.int32), res[8][0][0].view(torch.int32)) and torch.equal(h0, res[8][1][0])
    assert res[1][1].float().mean().item() > 0.05
