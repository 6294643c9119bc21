"""Exactly what bench.py times and what the library ships, at BASELINE's size, against the oracle chain:

* `slr_reconstruct_mf_batch` over 8 DISTINCT 4096x3000 frames on the stereo head verged by 0.2 rad (bench.py's default maps), at
  the default group sizes (one fused-decode launch of 16 jobs = 16 ticket pools x 8 XCDs, one match launch of 8 frames) -- three
  frames against remap -> decode -> MFReconstruct::triangulation of the oracle on all 3000 rows, the others against
  frame-by-frame launches (mfreconstruct.cpp:160-187).  The suite's poison hooks (outputs and scratch, tests/conftest.py) are on.
* the same under SLR_OPT_EVAL_MODEL = 1 (the reference's MSVC2010 x87 binary: oracle/slr_oracle_x87.c): both cameras' decode
  and the whole path, all rows; the two models must really differ on this scene.
* `slr_reconstruct_hybrid_batch` (BASELINE config 3) over 4 frames on the same rig, grouped launches, under both models.
"""
import numpy as np
import pytest
import torch

from util import bits_equal, calib_parts, np_of

pytestmark = pytest.mark.gpu
W, H, BLACK, WHITE = 4096, 3000, 40, 3
NAN = np.float32(np.nan)


@pytest.fixture(scope="module")
def rig(synth):
    return synth.make_verged_rig(W, H, 0.2, -0.15)


@pytest.fixture(scope="module")
def frames(synth):
    dev = torch.device("cuda", 0)
    st = torch.stack([synth.render_mf_stack(W, H, seed=1234 + f, noise=2, device=dev) for f in range(8)])
    torch.cuda.synchronize()
    return st


def _ctx_on_rig(slr, synth, rig, x87):
    c = slr.Context(0)
    if x87:
        c.set_option(slr.capi.OPT_EVAL_MODEL, 1)
    c.set_calibration(rig["calib"])
    synth.install_verged_maps(c, rig, W, H)
    return c


def _oracle_chain(oracle, planes, maps, calib, x87):
    """remap -> decode (valid folded in as NaN for the comparison with the pair launch) -> triangulation, one frame, all rows"""
    camL, camR, Q, T = calib_parts(oracle, calib)
    tab = oracle.atan_table(1)
    dec = []
    for cam in range(2):
        raw = planes[cam]
        rect = np.stack([oracle.remap_u8(raw[p], maps[cam][0], maps[cam][1]) for p in range(14)])
        dec.append(oracle.mf_decode_ev(rect, BLACK, tab, 1) if x87 else oracle.mf_decode(rect, BLACK))
    if x87:
        exyz, ehas, _ = oracle.mf_triangulate_ev(dec[0][0], dec[0][1], dec[1][0], dec[1][1], camL, camR, Q, 1, T=T)
    else:
        exyz, ehas, _ = oracle.mf_triangulate(dec[0][0], dec[0][1], dec[1][0], dec[1][1], camL, camR, Q, T)
    folded = [np.where(d[1] != 0, d[0], NAN).astype(np.float32) for d in dec]
    return folded, exyz, ehas


@pytest.mark.parametrize("x87", [0, 1])
def test_timed_batch_of_8_frames_on_the_verged_rig(slr, synth, oracle, rig, frames, x87):
    c = _ctx_on_rig(slr, synth, rig, x87)
    try:
        info = [c.rectify_info(cam) for cam in range(2)]
        assert [i["mf_form"] for i in info] == [7, 7]                 # the LDS-DMA form under either model
        maps = [c.get_rectify_maps(cam, W, H) for cam in range(2)]
        xyz, has = c.reconstruct_mf_batch(frames, BLACK, True)        # default groups: 8 frames per decode launch and per match launch
        c.synchronize()
        vs_oracle = (0, 5, 7) if not x87 else (0, 3)
        other = None
        for f in range(8):
            if f in vs_oracle:
                folded, exyz, ehas = _oracle_chain(oracle, frames[f].cpu().numpy(), maps, rig["calib"], x87)
                assert bits_equal(np_of(has[f]), ehas), f
                assert bits_equal(np_of(xyz[f]), exyz), f
                assert 0.2 < ehas.mean() < 1.0
                if f == 0:                                            # both cameras' decode directly (the pair launch, NaN-folded validity)
                    ph, _ = c.mf_rectify_decode_pair(frames[0, 0], frames[0, 1], BLACK, want_valid=False)
                    c.synchronize()
                    for cam in range(2):
                        assert bits_equal(np_of(ph[cam]), folded[cam]), cam
                    # the other model on the same frame: the scene must tell the two apart (else this test proves nothing about the mode)
                    _, oxyz, ohas = _oracle_chain(oracle, frames[0].cpu().numpy(), maps, rig["calib"], 1 - x87)
                    moved = int(((exyz != oxyz).any(axis=2) & (ehas != 0) & (ohas != 0)).sum())
                    assert moved > 1000, moved
                    other = moved
            else:
                x1, h1 = c.reconstruct_mf(frames[f, 0], frames[f, 1], BLACK, True)   # one pair launch + one match launch
                c.synchronize()
                assert torch.equal(h1, has[f]) and torch.equal(x1.view(torch.int32), xyz[f].view(torch.int32)), f
        assert not torch.equal(has[0], has[1])
        # a ragged batch (5 frames: one group of 5, odd pool occupancy) and match groups of 3 / decode groups of 2
        x5, h5 = c.reconstruct_mf_batch(frames[3:8], BLACK, True)
        c.synchronize()
        assert torch.equal(h5, has[3:8]) and torch.equal(x5.view(torch.int32), xyz[3:8].view(torch.int32))
        c.set_option(slr.capi.OPT_MF_BATCH_GROUP, 3)
        c.set_option(slr.capi.OPT_MF_BATCH_DECODE_GROUP, 2)
        x5, h5 = c.reconstruct_mf_batch(frames[:5], BLACK, True)
        c.synchronize()
        assert torch.equal(h5, has[:5]) and torch.equal(x5.view(torch.int32), xyz[:5].view(torch.int32))
        print("eval model %d: %d matched pixels of frame 0 get another XYZ under the other model" % (x87, other))
    finally:
        c.close()


def test_x87_cloud_and_host_entries_agree_with_the_batch(slr, synth, oracle, rig, frames):
    """SLR_OPT_EVAL_MODEL = 1 through the other whole-path entries a host calls (slr_reconstruct_mf with device and with host
    buffers, slr_reconstruct_mf_cloud): the same bits as the batch entry's frame"""
    c = _ctx_on_rig(slr, synth, rig, 1)
    try:
        xyz, has = c.reconstruct_mf_batch(frames[:2], BLACK, True)
        c.synchronize()
        hl, hr = frames[1, 0].cpu().numpy(), frames[1, 1].cpu().numpy()
        xh, hh = c.reconstruct_mf(hl, hr, BLACK, True)
        assert bits_equal(hh, np_of(has[1])) and bits_equal(xh, np_of(xyz[1]))
        es, ec, _ = oracle.pointcloud_from_grid(np_of(xyz[1]), np_of(has[1]), 1280, 1024, None)
        s, cn = c.reconstruct_mf_cloud(frames[1, 0], frames[1, 1], BLACK, True, 1280, 1024)
        c.synchronize()
        assert bits_equal(np_of(cn), ec) and bits_equal(np_of(s), es)
    finally:
        c.close()


@pytest.mark.parametrize("x87", [0, 1])
def test_hybrid_batch_grouped_on_the_verged_rig(slr, synth, oracle, rig, x87):
    """BASELINE config 3 as bench.py --mode hybrid runs it: 4 frames of 38 planes per camera through slr_reconstruct_hybrid_batch
    (fringe decodes and the match in grouped launches): frame 2 against the oracle chain (Gray codes bit-exact, phase match +
    triangulation on all rows), the others against single-frame calls"""
    scan_w = 4096
    ncol = synth.gray_num_bits(scan_w)
    need = 2 + 2 * ncol + 12
    dev = torch.device("cuda", 0)
    stack = torch.stack([synth.render_hybrid_stack(W, H, scan_w, seed=700 + 13 * f, noise=2, device=dev) for f in range(4)])
    torch.cuda.synchronize()
    assert stack.shape[2] == need
    c = _ctx_on_rig(slr, synth, rig, x87)
    try:
        maps = [c.get_rectify_maps(cam, W, H) for cam in range(2)]
        xyz, has, codes = c.reconstruct_hybrid_batch(stack, ncol, BLACK, WHITE, scan_w, want_codes=True)
        c.synchronize()
        f = 2
        mf_planes, ex = [], []
        for cam in range(2):
            raw = stack[f, cam].cpu().numpy()
            rect = np.stack([oracle.remap_u8(raw[p], maps[cam][0], maps[cam][1]) for p in range(need)])
            gx, _, gv = oracle.gray_decode(np.ascontiguousarray(rect[:2 + 2 * ncol]), ncol, 0, BLACK, WHITE, scan_w, 0)
            ex.append(gx)
            mf_planes.append(np.ascontiguousarray(np.concatenate([rect[:2], rect[2 + 2 * ncol:]])))
        camL, camR, Q, T = calib_parts(oracle, rig["calib"])
        if x87:
            tab = oracle.atan_table(1)
            dec = [oracle.mf_decode_ev(p, BLACK, tab, 1) for p in mf_planes]
            exyz, ehas, _ = oracle.mf_triangulate_ev(dec[0][0], dec[0][1], dec[1][0], dec[1][1], camL, camR, Q, 1, T=T)
        else:
            dec = [oracle.mf_decode(p, BLACK) for p in mf_planes]
            exyz, ehas, _ = oracle.mf_triangulate(dec[0][0], dec[0][1], dec[1][0], dec[1][1], camL, camR, Q, T)
        assert bits_equal(np_of(has[f]), ehas) and bits_equal(np_of(xyz[f]), exyz)
        for cam in range(2):
            assert bits_equal(np_of(codes[f, cam]), ex[cam]), cam
        assert 0.2 < ehas.mean() < 1.0
        for g in (0, 1, 3):
            x1, h1, _ = c.reconstruct_hybrid_batch(stack[g:g + 1], ncol, BLACK, WHITE, scan_w)
            c.synchronize()
            assert torch.equal(h1[0], has[g]) and torch.equal(x1[0].view(torch.int32), xyz[g].view(torch.int32)), g
    finally:
        c.close()


def test_config5_rectified_at_size_on_the_verged_rig(slr, synth, oracle):
    """BASELINE config 5 as bench.py --mode mfn runs it, at its stated size: 8192x6000, 4 frequencies x 8 steps of fp16 planes,
    a stereo head verged by 0.2 rad.  The shipped LDS-DMA rectified decode (mfn_rect_dma_kernel) of both cameras against the
    per-pixel gather form on the WHOLE frame (bit for bit), against the fp64 remap + decode model on sampled rows (1e-4 relative
    + the f32 floor, wrap flips counted), and the 8-GPU split: 8 rectified row bands, each decoded from a COPY of only the
    source rows slr_rectify_source_rows names and matched with absolute rows (rows wider than 4096: the round-5 wide K4),
    concatenated == the whole frame, decode and XYZ; the wide K4 against the oracle's literal search on sampled rows."""
    W5, H5, F, N = 8192, 6000, 4, 8
    np_ = 2 + F * N
    dev = torch.device("cuda", 0)
    rig5 = synth.make_verged_rig(W5, H5, 0.2, -0.15)
    c = slr.Context(0)
    try:
        c.set_calibration(rig5["calib"])
        synth.install_verged_maps(c, rig5, W5, H5)
        st = synth.render_mfn_stack(W5, H5, F, N, noise=0.25, seed=5, device=dev)
        torch.cuda.synchronize()
        full = [c.mfn_rectify_decode(cam, st[cam], F, N, 40.0) for cam in range(2)]           # the shipped form
        c.synchronize()
        c.set_option(slr.capi.OPT_DEBUG_FLAGS, 1)                                             # the per-pixel gather form
        for cam in range(2):
            g = c.mfn_rectify_decode(cam, st[cam], F, N, 40.0)
            c.synchronize()
            assert torch.equal(g[1], full[cam][1]) and torch.equal(g[0].view(torch.int32), full[cam][0].view(torch.int32)), cam
            del g
        c.set_option(slr.capi.OPT_DEBUG_FLAGS, 0)
        assert full[0][1].float().mean().item() > 0.3
        # fp64 model, camera 0, sampled row ranges (top / band edges / middle / bottom: the keystone is strongest at the ends)
        mx, mf = c.get_rectify_maps(0, W5, H5)
        host = st[0].cpu().numpy()
        for r0 in (0, 749, 2999, 5998):
            exp, ev = oracle.mfn_rect_decode_f64(host, F, N, 40.0, mx, mf, rows=(r0, r0 + 2))
            ph, v = np_of(full[0][0][r0:r0 + 2]), np_of(full[0][1][r0:r0 + 2])
            exp, ev = exp[r0:r0 + 2], ev[r0:r0 + 2]
            d = np.abs(ph.astype(np.float64) - exp)
            tol = 2e-3 + 1e-4 * np.abs(exp)                                  # north_star: 1e-4 relative (+ f32 floor)
            per = np.abs(d - 255.0 * np.round(d / 255.0))                    # wrap flips at an a > b decided in f32 vs f64
            vd = v != ev
            assert ((d > tol) & (per > tol) & ~vd).sum() == 0, r0
            assert (d > tol).sum() <= 2e-3 * ph.size and vd.sum() <= 1e-4 * ph.size + 1, (r0, int((d > tol).sum()), int(vd.sum()))
        del host, mx, mf
        # the whole frame's match (wide K4) and sampled rows against the oracle's literal search
        fx, fh, fk = c.mf_triangulate(full[0][0], full[0][1], full[1][0], full[1][1])
        c.synchronize()
        camL, camR, Q, T = calib_parts(oracle, rig5["calib"])
        phL, vL, phR, vR = [np_of(t) for t in (full[0][0], full[0][1], full[1][0], full[1][1])]
        for r in (0, 750, 3001, 5999):
            exyz, ehas, emk = oracle.mf_triangulate(phL, vL, phR, vR, camL, camR, Q, T, rows=(r, r + 1))
            assert bits_equal(np_of(fk[r]), emk[r]) and bits_equal(np_of(fh[r]), ehas[r]) and bits_equal(np_of(fx[r]), exyz[r]), r
        assert fh.float().mean().item() > 0.05
        del phL, vL, phR, vR
        # 8 row bands from source windows
        band = (H5 + 7) // 8
        short = 0
        for b in range(8):
            r0, r1 = b * band, min(H5, (b + 1) * band)
            dec = []
            for cam in range(2):
                s0, sn = c.rectify_source_rows(cam, r0, r1 - r0)
                assert 0 <= s0 and s0 + sn <= H5 and sn >= 1
                short += 1 if sn < H5 // 4 else 0
                window = st[cam][:, s0:s0 + sn].clone()                      # what this GPU would hold of the frame
                dec.append(c.mfn_rectify_decode(cam, window, F, N, 40.0, W=W5, H=H5, row0=r0, rows=r1 - r0, src_row0=s0))
                c.synchronize()
                del window
                assert torch.equal(dec[cam][1], full[cam][1][r0:r1]), (b, cam)
                assert torch.equal(dec[cam][0].view(torch.int32), full[cam][0][r0:r1].view(torch.int32)), (b, cam)
            bx, bh, bk = c.mf_triangulate(dec[0][0], dec[0][1], dec[1][0], dec[1][1], row0=r0, image_h=H5)
            c.synchronize()
            assert torch.equal(bh, fh[r0:r1]) and torch.equal(bk, fk[r0:r1]) and torch.equal(bx.view(torch.int32), fx[r0:r1].view(torch.int32)), b
        assert short == 16                                       # every band's window is a small part of the frame
    finally:
        c.close()
