"""The LDS-DMA form of the fused rectify + multi-frequency decode (SLR_OPT_RECT_DECODE_ALGO = 7, kernels_rectdma.hip):
every tile shape x DMA depth x (valid bytes | valid folded into the phase) against the oracle's cv::remap + getPhase, bit for
bit -- smooth maps, maps that leave the image on every side (BORDER_CONSTANT), ragged sizes, many tiles per workgroup, the
strictness of an explicit 7 and the fallback of auto."""
import numpy as np
import pytest
import torch

from util import bits_equal, forms, np_of

pytestmark = pytest.mark.gpu
BLACK = 40
SHAPES = tuple(range(7))         # 256x16/512 threads, 256x8/512, 256x8/256, 128x16/512, 128x8/256, 256x4/256, 128x16/256


def _expect(oracle, raw, mx, mf):
    rect = np.stack([oracle.remap_u8(raw[p], mx, mf) for p in range(14)])
    return oracle.mf_decode(rect, BLACK)


def _opts(ctx, slr, algo=7, shape=0, depth=1):
    ctx.set_option(slr.capi.OPT_RECT_DMA_SHAPE, shape)
    ctx.set_option(slr.capi.OPT_RECT_DMA_DEPTH, depth)
    ctx.set_option(slr.capi.OPT_RECT_DECODE_ALGO, algo)


@pytest.fixture
def dma_ctx(ctx, slr):
    yield ctx
    ctx.set_option(slr.capi.OPT_DEBUG_RECT_RESIDENT, 0)
    _opts(ctx, slr, 0, 3, 2)                                 # the defaults: auto, 128x16 tiles, depth 2


@pytest.mark.parametrize("W,H,strength", [(640, 480, 1.0), (1024, 100, 2.0), (4096, 40, 1.0), (400, 77, 2.0), (16, 5, 1.0),
                                          (272, 33, 3.0)])
def test_dma_form_every_shape_and_depth(dma_ctx, slr, oracle, synth, W, H, strength):
    ctx = dma_ctx
    st = synth.render_mf_stack(W, H, seed=W + H, noise=3)
    for cam in range(2):
        mx, mf = synth.make_rectify_maps(W, H, cam, strength=strength)
        mxn, mfn = mx.numpy(), mf.numpy()
        raw = st[cam].numpy()
        eph, ev = _expect(oracle, raw, mxn, mfn)
        dev = st[cam].cuda()
        ran = 0
        shapes = forms(ctx, slr, slr.capi.OPT_RECT_DMA_SHAPE, SHAPES, default=3, required=(0, 1, 3))   # (2, 4, 5, 6: FORMS=all builds)
        for shape in shapes:
            _opts(ctx, slr, 7, shape, 1)
            ctx.set_rectify_maps(cam, mxn, mfn)
            try:
                for depth in (1, 2):
                    ctx.set_option(slr.capi.OPT_RECT_DMA_DEPTH, depth)
                    ph, v = ctx.mf_decode(dev, BLACK, rectify_cam=cam)          # device stack, separate valid bytes
                    ctx.synchronize()
                    assert bits_equal(np_of(v), ev) and bits_equal(np_of(ph), eph), (shape, depth)
                ph, v = ctx.mf_decode(raw, BLACK, rectify_cam=cam)              # host planes staged by the library
                assert bits_equal(v, ev) and bits_equal(ph, eph), shape
                ran += 1
            except slr.capi.SlrError as e:                                      # a strong map: some tile's box exceeds this shape
                assert e.status == slr.capi.ERR_UNSUPPORTED and strength > 1.0, (shape, str(e))
        assert ran >= min(4, len(shapes)), ran


def test_dma_form_borders_on_every_side(dma_ctx, slr, oracle, synth):
    """identity maps shifted far enough that footprints leave the source image left, right, above and below: everything
    outside must read as 0 (BORDER_CONSTANT), partially outside footprints blend the zeros in"""
    ctx = dma_ctx
    W, H = 512, 64
    st = synth.render_mf_stack(W, H, seed=5, noise=3)
    raw = st[0].numpy() | 1                                                      # no zero samples: zeros can only come from the border
    dev = torch.from_numpy(raw).cuda()
    for dx, dy, fx, fy in [(-37, -9, 7, 19), (41, 11, 31, 31), (-1, 0, 0, 13), (0, -1, 30, 0), (700, 0, 3, 3), (0, 90, 1, 1),
                           (-16, -16, 0, 0), (15, 3, 16, 16)]:
        mx, mf = synth.identity_maps(W, H, dx=dx, dy=dy, fx=fx, fy=fy)
        mxn, mfn = mx.numpy(), mf.numpy()
        eph, ev = _expect(oracle, raw, mxn, mfn)
        for shape in forms(ctx, slr, slr.capi.OPT_RECT_DMA_SHAPE, SHAPES, default=3, required=(0, 1, 3)):
            _opts(ctx, slr, 7, shape, 1 + shape % 2)
            ctx.set_rectify_maps(0, mxn, mfn)
            ph, v = ctx.mf_decode(dev, BLACK, rectify_cam=0)
            ctx.synchronize()
            assert bits_equal(np_of(v), ev) and bits_equal(np_of(ph), eph), (dx, dy, shape)


@pytest.mark.parametrize("resident", ["8", "16"])
def test_dma_form_many_tiles_per_workgroup_and_whole_path(dma_ctx, slr, oracle, synth, monkeypatch, resident):
    """few resident workgroups -> every workgroup walks many tiles (buffer rotation over tile boundaries, the dummy DMAs
    behind the last tile); then the whole path (both cameras in one launch, valid folded into the phase as a NaN) against
    the step-by-step oracle pipeline"""
    from util import calib_parts
    ctx = dma_ctx
    ctx.set_option(slr.capi.OPT_DEBUG_RECT_RESIDENT, int(resident))
    W, H = 1280, 200
    calib, _ = synth.make_calibration(W, H)
    ctx.set_calibration(calib)
    st = synth.render_mf_stack(W, H, seed=11, noise=2)
    maps = [synth.make_rectify_maps(W, H, cam) for cam in range(2)]
    exp = []
    for cam in range(2):
        exp.append(_expect(oracle, st[cam].numpy(), maps[cam][0].numpy(), maps[cam][1].numpy()))
    camL, camR, Q, T = calib_parts(oracle, calib)
    exyz, ehas, _ = oracle.mf_triangulate(exp[0][0], exp[0][1], exp[1][0], exp[1][1], camL, camR, Q, T)
    dev = st.cuda()
    for shape in forms(ctx, slr, slr.capi.OPT_RECT_DMA_SHAPE, SHAPES, default=3, required=(0, 1, 3)):
        for depth in (1, 2):
            _opts(ctx, slr, 7, shape, depth)
            for cam in range(2):
                ctx.set_rectify_maps(cam, maps[cam][0].numpy(), maps[cam][1].numpy())
            for cam in range(2):
                ph, v = ctx.mf_decode(dev[cam], BLACK, rectify_cam=cam)
                ctx.synchronize()
                assert bits_equal(np_of(v), exp[cam][1]) and bits_equal(np_of(ph), exp[cam][0]), (shape, depth, cam)
            xyz, has = ctx.reconstruct_mf(dev[0], dev[1], BLACK, True)
            ctx.synchronize()
            assert bits_equal(np_of(has), ehas) and bits_equal(np_of(xyz), exyz), (shape, depth)
    assert ehas.mean() > 0.2


def test_dma_form_strict_when_asked_for_and_auto_falls_back(dma_ctx, slr, oracle, synth):
    ctx = dma_ctx
    rng = np.random.default_rng(3)
    W, H = 320, 48
    st = synth.render_mf_stack(W, H, seed=2, noise=3)
    raw = st[0].numpy()
    wild = (np.stack([rng.integers(-9, W + 9, (H, W)), rng.integers(-9, H + 9, (H, W))], -1).astype(np.int16),
            rng.integers(0, 1024, (H, W)).astype(np.uint16))
    eph, ev = _expect(oracle, raw, wild[0], wild[1])
    _opts(ctx, slr, 7, 0, 1)
    ctx.set_rectify_maps(0, wild[0], wild[1])                                    # tile boxes far too large for the form
    with pytest.raises(slr.capi.SlrError) as ei:
        ctx.mf_decode(st[0].cuda(), BLACK, rectify_cam=0)
    assert ei.value.status == slr.capi.ERR_UNSUPPORTED
    _opts(ctx, slr, 0, 0, 1)                                                     # auto: silently the round-1 forms
    ph, v = ctx.mf_decode(st[0].cuda(), BLACK, rectify_cam=0)
    ctx.synchronize()
    assert bits_equal(np_of(v), ev) and bits_equal(np_of(ph), eph)
    # separate plane allocations (no common stride): not this form either
    mx, mf = synth.make_rectify_maps(W, H, 0)
    ctx.set_rectify_maps(0, mx.numpy(), mf.numpy())
    planes = [st[0, p].cuda().clone() for p in range(14)]
    planes[3] = torch.cat([torch.zeros(7, W, dtype=torch.uint8, device="cuda"), planes[3]])[7:]
    _opts(ctx, slr, 7, 0, 1)
    with pytest.raises(slr.capi.SlrError):
        ctx.mf_decode(planes, BLACK, rectify_cam=0)
    _opts(ctx, slr, 0, 0, 1)
    eph, ev = _expect(oracle, raw, mx.numpy(), mf.numpy())
    ph, v = ctx.mf_decode(planes, BLACK, rectify_cam=0)
    ctx.synchronize()
    assert bits_equal(np_of(v), ev) and bits_equal(np_of(ph), eph)


def test_dma_form_fullsize_every_shape(dma_ctx, slr, oracle, synth):
    """4096x3000, both cameras one launch each (the pair launch proper: test_gpu_fullsize.py): every shape at depth 1 and four of
    them at depth 2, the shipped default (3, 2) among them"""
    ctx = dma_ctx
    W, H = 4096, 3000
    dev = torch.device("cuda", 0)
    st = synth.render_mf_stack(W, H, seed=1234, device=dev)
    maps = [synth.make_rectify_maps(W, H, cam, device=dev) for cam in range(2)]
    torch.cuda.synchronize()
    exp = [_expect(oracle, st[cam].cpu().numpy(), maps[cam][0].cpu().numpy(), maps[cam][1].cpu().numpy()) for cam in range(2)]
    ran = []
    have = forms(ctx, slr, slr.capi.OPT_RECT_DMA_SHAPE, SHAPES, default=3, required=(0, 1, 3))
    for shape, depth in [(0, 1), (1, 1), (2, 1), (3, 1), (4, 1), (5, 1), (6, 1), (0, 2), (1, 2), (3, 2), (4, 2)]:
        if shape not in have: continue
        _opts(ctx, slr, 7, shape, depth)
        for cam in range(2):
            ctx.set_rectify_maps(cam, maps[cam][0], maps[cam][1])
        try:
            for cam in range(2):
                ph, v = ctx.mf_decode(st[cam], BLACK, rectify_cam=cam)
                ctx.synchronize()
                assert bits_equal(np_of(v), exp[cam][1]) and bits_equal(np_of(ph), exp[cam][0]), (shape, depth, cam)
            ran.append((shape, depth))
        except slr.capi.SlrError as e:       # the 256-thread shapes hold few source rows: these maps' corner tiles may not fit
            assert e.status == slr.capi.ERR_UNSUPPORTED and shape not in (0, 1, 3), (shape, str(e))
    assert (0, 1) in ran and (1, 1) in ran and (3, 2) in ran and len(ran) >= 6, ran


@pytest.mark.parametrize("W,H", [(1040, 524), (528, 260)])
def test_split_tiles_as_a_workgroups_only_entries(dma_ctx, slr, synth, W, H):
    """A small image on the FULL resident set: there are more workgroups than tile-table entries, so every entry -- the parts of
    split corner tiles of a verged rig among them -- is some workgroup's first and only one, and the waves outside such a part
    never decode anything.  Their end-of-kernel flush once stored whatever their registers held to whatever address those
    registers made (wrong values scattered over the image, another set every run; with few resident workgroups, or at 4096 x
    3000, every wave has decoded a tile before and the flush merely repeated it).  Both fused decodes, every compiled tile shape
    that serves them, sorted and own-order digests, three runs each, against the per-pixel gather form."""
    ctx = dma_ctx
    st = synth.render_mf_stack(W, H, seed=7, noise=3, device="cuda")
    g = synth.render_gray_stack(W, H, 1024, seed=9, noise=2, device="cuda")
    ncol = synth.gray_num_bits(1024)
    hy = synth.render_hybrid_stack(W, H, 1024, seed=5, noise=2, device="cuda")
    ctx.set_calibration(synth.make_calibration(W, H)[0])

    def decode():
        ph, vd = ctx.mf_rectify_decode_pair(st[0], st[1], BLACK, want_valid=True)
        ctx.synchronize()
        outs = [ph[0].clone(), ph[1].clone(), vd[0].clone(), vd[1].clone()]
        for cam in range(2):
            cx, _, v = ctx.gray_decode(g[cam], ncol, 0, BLACK, 3, 1024, 0, rectify_cam=cam)
            ctx.synchronize()
            outs += [cx.clone(), v.clone()]
        for one_pass in (0, 1):                                     # the hybrid stack: two fused launches, or the one-pass kernel
            ctx.set_option(slr.capi.OPT_HYBRID_ONE_PASS, one_pass)
            hx, hp = ctx.hybrid_rectify_decode_pair(hy[0], hy[1], ncol, BLACK, 3, 1024)
            ctx.synchronize()
            outs += [hx[0].clone(), hx[1].clone(), hp[0].clone(), hp[1].clone()]
        ctx.set_option(slr.capi.OPT_HYBRID_ONE_PASS, 0)
        return outs

    try:
        split = 0
        for theta, k1 in ((0.15, -0.12), (0.25, 0.1), (0.35, -0.25)):
            rig = synth.make_verged_rig(W, H, theta, k1)
            ctx.set_option(slr.capi.OPT_DEBUG_FLAGS, 0)
            ctx.set_option(slr.capi.OPT_RECT_DECODE_ALGO, 1)
            synth.install_verged_maps(ctx, rig, W, H)
            ref = decode()
            ctx.set_option(slr.capi.OPT_RECT_DECODE_ALGO, 0)
            for shape in (0, 1, 3):
                ctx.set_option(slr.capi.OPT_RECT_DMA_SHAPE, shape)
                for flags in (0, 32):
                    ctx.set_option(slr.capi.OPT_DEBUG_FLAGS, flags)
                    synth.install_verged_maps(ctx, rig, W, H)
                    split += sum(ctx.rectify_info(cam)["dma_extra_entries"] for cam in range(2))
                    for rep in range(3):
                        for k, (a, b) in enumerate(zip(decode(), ref)):
                            same = a.view(torch.uint8) == b.view(torch.uint8)
                            assert bool(same.all()), (theta, shape, flags, rep, k, int((~same).sum()))
        assert split > 50, split
    finally:
        ctx.set_option(slr.capi.OPT_DEBUG_FLAGS, 0)
        ctx.set_option(slr.capi.OPT_RECT_DECODE_ALGO, 0)
        ctx.set_option(slr.capi.OPT_HYBRID_ONE_PASS, 0)


@pytest.mark.parametrize("W,H,scan_w,scan_h,strength,resident", [(640, 200, 1100, 0, 1.0, 0), (1280, 96, 100, 37, 2.0, 8), (512, 130, 5, 0, 1.0, 16),
                                                                 (2048, 64, 4096, 600, 1.5, 0), (400 // 16 * 16, 77, 300, 7, 3.0, 8), (256, 40, 4, 0, 1.0, 0)])
def test_gray_dma_counted_wait_form_equals_the_drained_form_and_the_oracle(dma_ctx, slr, oracle, synth, W, H, scan_w, scan_h, strength, resident):
    """The fused Gray decode's round-5 form (SLR_OPT_RECT_DMA_DEPTH = 2, the default: three LDS buffers of one plane pair, the
    planes of phase k + 2 in flight, counted vmcnt waits) against round 2's form (depth 1: two buffers, the queue drained every
    phase) and the oracle's remap + getProjPixel (reconstruct.cpp:325-407): phase counts 1 + ncol + nrow of every residue mod 3
    (the buffer of a tile's phase 0 rotates by that from tile to tile), too few phases for the form (falls back), many tiles
    per workgroup, both tile shapes, codes + rows + valid bytes."""
    ctx = dma_ctx
    ncol, nrow = synth.gray_num_bits(scan_w), (synth.gray_num_bits(scan_h) if scan_h else 0)
    st = synth.render_gray_stack(W, H, scan_w, scan_h if scan_h else None, seed=W + ncol, noise=2, rows=scan_h > 0)
    ctx.set_option(slr.capi.OPT_DEBUG_RECT_RESIDENT, resident)
    for cam in range(2):
        mx, mf = synth.make_rectify_maps(W, H, cam, strength=strength)
        mxn, mfn = mx.numpy(), mf.numpy()
        raw = st[cam].numpy()
        rect = np.stack([oracle.remap_u8(raw[p], mxn, mfn) for p in range(raw.shape[0])])
        ex, ey, ev = oracle.gray_decode(rect, ncol, nrow, BLACK, 3, scan_w, scan_h)
        dev = st[cam].cuda()
        for shape in (3, 1):
            for depth in (2, 1):
                _opts(ctx, slr, 0, shape, depth)
                ctx.set_rectify_maps(cam, mxn, mfn)
                # the maps of every case admit the LDS-DMA form (a fallback kernel would make the depth comparison vacuous); the
                # counted-wait variant (A = 2) takes stacks of >= 4 phases, the (256, 40, 4) case is the explicit short-stack fallback
                info = ctx.rectify_info(cam)
                assert info["mf_form"] == 7 and info["dma_depth"] == depth, info
                assert (1 + ncol + nrow >= 4) == ((W, H, scan_w) != (256, 40, 4))
                cx, cy, v = ctx.gray_decode(dev, ncol, nrow, BLACK, 3, scan_w, scan_h, rectify_cam=cam)
                ctx.synchronize()
                assert bits_equal(np_of(v), ev) and bits_equal(np_of(cx), ex), (shape, depth, cam)
                if nrow:
                    assert bits_equal(np_of(cy), ey), (shape, depth, cam)
        assert (ev != 0).mean() > 0.2
