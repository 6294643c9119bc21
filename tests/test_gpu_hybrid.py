"""BASELINE config 3 on the GPU: the one-pass hybrid decode (Gray code + multi-frequency phase from one stack sharing white / black)
against BOTH oracles -- each output is exactly what its own reference mode computes (reconstruct.cpp:79-97,381-407 and
mfreconstruct.cpp:210-269) -- at small sizes on random maps, through the two-pass fall-back, and at 4096x3000 with 38 planes per
camera; then the batch entry (hybrid decode + phase match + triangulation) against the oracle chain."""
import numpy as np
import pytest
import torch

from util import bits_equal, calib_parts, np_of

pytestmark = pytest.mark.gpu
BLACK, WHITE = 40, 3


def _expect(oracle, raw, mx, mf, ncol, scan_w):
    rect = np.stack([oracle.remap_u8(raw[p], mx, mf) for p in range(raw.shape[0])])
    ex, _, ev = oracle.gray_decode(np.ascontiguousarray(rect[:2 + 2 * ncol]), ncol, 0, BLACK, WHITE, scan_w, 0)
    eph, evp = oracle.mf_decode(np.ascontiguousarray(np.concatenate([rect[:2], rect[2 + 2 * ncol:]])), BLACK)
    folded = np.where(evp != 0, eph, np.float32(np.nan)).astype(np.float32)
    return ex, ev, folded, eph, evp


@pytest.mark.parametrize("W,H,scan_w,strength,resident", [(256, 48, 200, 1.0, 0), (640, 96, 1024, 2.5, 8), (400 // 16 * 16, 70, 37, 0.3, 16),
                                                          (1024, 40, 5000, 3.5, 0)])
def test_hybrid_decode_small_random_maps(ctx, slr, oracle, synth, W, H, scan_w, strength, resident):
    ncol = synth.gray_num_bits(scan_w)
    maps = [synth.make_rectify_maps(W, H, cam, strength=strength) for cam in range(2)]
    st = synth.render_hybrid_stack(W, H, scan_w, seed=11 + W, noise=3)
    assert st.shape[1] == 2 + 2 * ncol + 12
    ctx.set_option(slr.capi.OPT_DEBUG_RECT_RESIDENT, resident)
    try:
        for cam in range(2):
            ctx.set_rectify_maps(cam, maps[cam][0].numpy(), maps[cam][1].numpy())
        exp = [_expect(oracle, st[cam].numpy(), maps[cam][0].numpy(), maps[cam][1].numpy(), ncol, scan_w) for cam in range(2)]
        for one_pass in (0, 1):                              # the two launches over the one stack (default), the one-pass kernel
            ctx.set_option(slr.capi.OPT_HYBRID_ONE_PASS, one_pass)
            cx, ph = ctx.hybrid_rectify_decode_pair(st[0].cuda(), st[1].cuda(), ncol, BLACK, WHITE, scan_w)
            ctx.synchronize()
            hx, hp = ctx.hybrid_rectify_decode_pair(st[0].numpy(), st[1].numpy(), ncol, BLACK, WHITE, scan_w)     # host buffers
            for cam in range(2):
                ex, ev, folded, _, _ = exp[cam]
                assert bits_equal(np_of(cx[cam]), ex) and bits_equal(np_of(ph[cam]), folded), (one_pass, cam)
                assert bits_equal(hx[cam], ex) and bits_equal(hp[cam], folded), (one_pass, cam)
                assert (ex >= 0).mean() > 0.2 and np.isfinite(folded).mean() > 0.2
    finally:
        ctx.set_option(slr.capi.OPT_DEBUG_RECT_RESIDENT, 0)
        ctx.set_option(slr.capi.OPT_HYBRID_ONE_PASS, 0)


def test_hybrid_two_pass_fallback_when_the_form_does_not_apply(ctx, slr, oracle, synth):
    """W % 16 != 0 (no LDS-DMA tables) and a wild map (its tile boxes fit no tiled form): same results through the two fused
    decodes per camera; an explicit SLR_OPT_RECT_DECODE_ALGO = 7 fails loudly instead"""
    W, H, scan_w = 250, 40, 300
    ncol = synth.gray_num_bits(scan_w)
    maps = [synth.make_rectify_maps(W, H, cam, strength=1.5) for cam in range(2)]
    st = synth.render_hybrid_stack(W, H, scan_w, seed=5, noise=2)
    for cam in range(2):
        ctx.set_rectify_maps(cam, maps[cam][0].numpy(), maps[cam][1].numpy())
    cx, ph = ctx.hybrid_rectify_decode_pair(st[0].cuda(), st[1].cuda(), ncol, BLACK, WHITE, scan_w)
    ctx.synchronize()
    for cam in range(2):
        ex, _, folded, _, _ = _expect(oracle, st[cam].numpy(), maps[cam][0].numpy(), maps[cam][1].numpy(), ncol, scan_w)
        assert bits_equal(np_of(cx[cam]), ex) and bits_equal(np_of(ph[cam]), folded), cam
    ctx.set_option(slr.capi.OPT_RECT_DECODE_ALGO, 7)
    try:
        with pytest.raises(slr.capi.SlrError) as ei:
            ctx.hybrid_rectify_decode_pair(st[0].cuda(), st[1].cuda(), ncol, BLACK, WHITE, scan_w)
        assert ei.value.status == slr.capi.ERR_UNSUPPORTED
    finally:
        ctx.set_option(slr.capi.OPT_RECT_DECODE_ALGO, 0)


def test_hybrid_fullsize_and_batch_entry(ctx, slr, oracle, synth):
    """config 3 at its stated size: 4096x3000, 4096-wide projector (12 column bits): 38 planes per camera, rectification; the
    decode against both oracles, then slr_reconstruct_hybrid_batch (two frames, spare planes in the stack) against
    remap -> decode -> MFReconstruct::triangulation of the oracle"""
    W, H, scan_w = 4096, 3000, 4096
    dev = torch.device("cuda", 0)
    ncol = synth.gray_num_bits(scan_w)
    need = 2 + 2 * ncol + 12
    assert need == 38
    calib, _ = synth.make_calibration(W, H)
    ctx.set_calibration(calib)
    maps = [synth.make_rectify_maps(W, H, cam, device=dev) for cam in range(2)]
    for cam in range(2):
        ctx.set_rectify_maps(cam, maps[cam][0], maps[cam][1])
    stack = torch.full((2, 2, need + 1, H, W), 99, dtype=torch.uint8, device=dev)       # one spare plane per camera
    for f in range(2):
        stack[f, :, :need] = synth.render_hybrid_stack(W, H, scan_w, seed=900 + 31 * f, noise=2, device=dev)
    torch.cuda.synchronize()
    assert ctx.rectify_info(0)["mf_form"] == 7
    exp = [_expect(oracle, stack[0, cam, :need].cpu().numpy(), maps[cam][0].cpu().numpy(), maps[cam][1].cpu().numpy(), ncol, scan_w)
           for cam in range(2)]
    dec = [(e[3], e[4]) for e in exp]
    for one_pass in (1, 0):
        ctx.set_option(slr.capi.OPT_HYBRID_ONE_PASS, one_pass)
        cx, ph = ctx.hybrid_rectify_decode_pair(stack[0, 0, :need], stack[0, 1, :need], ncol, BLACK, WHITE, scan_w)
        ctx.synchronize()
        for cam in range(2):
            ex, ev, folded, eph, evp = exp[cam]
            assert bits_equal(np_of(cx[cam]), ex) and bits_equal(np_of(ph[cam]), folded), (one_pass, cam)
            assert (ex >= 0).mean() > 0.5 and evp.mean() > 0.5
    xyz, has, codes = ctx.reconstruct_hybrid_batch(stack, ncol, BLACK, WHITE, scan_w, want_codes=True)
    ctx.synchronize()
    camL, camR, Q, T = calib_parts(oracle, calib)
    exyz, ehas, _ = oracle.mf_triangulate(dec[0][0], dec[0][1], dec[1][0], dec[1][1], camL, camR, Q, T)
    assert bits_equal(np_of(has[0]), ehas) and bits_equal(np_of(xyz[0]), exyz)
    assert torch.equal(codes[0, 0], cx[0]) and torch.equal(codes[0, 1], cx[1])
    # frame 1 of the batch == the same frame on its own (stride / spare-plane handling), without the codes
    x1, h1, _ = ctx.reconstruct_hybrid_batch(stack[1:2].contiguous(), ncol, BLACK, WHITE, scan_w)
    ctx.synchronize()
    assert torch.equal(h1[0], has[1]) and torch.equal(x1[0], xyz[1]) and not torch.equal(has[0], has[1])
