"""BASELINE.json's full size (4096x3000) on the GPU: direct comparison with the oracle where the oracle finishes in
seconds (decode, rectify, Gray, GE, GRAY_ONLY, the PointCloudImage adaptor; the O(W^2) MF match on all 3000 rows), plus
size-independent properties (indexed match forms == literal sweep on the whole frame, depth from disparity) and one
config-5 sized case (8192x6000, 4 freq x 8 steps fp16) on sampled rows."""
import numpy as np
import pytest
import torch

from util import bits_equal, calib_parts, forms, np_of

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


@pytest.fixture(scope="module")
def oracle_dec(oracle, scene):
    """the oracle's remap + decode of BOTH cameras of the scene (28 full-size remaps: computed once per module)"""
    st, maps, _, _ = scene
    out = []
    for cam in range(2):
        raw = st[cam].cpu().numpy()
        mx, mf = maps[cam][0].cpu().numpy(), maps[cam][1].cpu().numpy()
        rect = np.stack([oracle.remap_u8(raw[p], mx, mf) for p in range(14)])
        out.append(oracle.mf_decode(rect, BLACK))
    return out


def test_fullsize_shipped_pair_launch_both_cameras_vs_oracle(ctx, oracle, scene, oracle_dec, slr):
    """What the product runs at its DEFAULT options -- both cameras' fused rectify + decode in ONE launch with the valid flag
    folded into the phase as a NaN, then the lean K4 -- against a pure-oracle chain (remap -> decode -> triangulation), for
    both cameras directly: the pair launch's phases, then the whole-path XYZ / mask.  Then the same launch with separate valid
    bytes, and the other pair-launch shapes / depths."""
    st, maps, calib, _ = scene
    ctx.set_calibration(calib)
    for cam in range(2):
        ctx.set_rectify_maps(cam, maps[cam][0], maps[cam][1])
    nan = np.float32(np.nan)
    folded = [np.where(oracle_dec[cam][1] != 0, oracle_dec[cam][0], nan).astype(np.float32) for cam in range(2)]
    assert np.isnan(folded[0]).any() and (oracle_dec[0][1] != 0).mean() > 0.5
    ph, _ = ctx.mf_rectify_decode_pair(st[0], st[1], BLACK, want_valid=False)      # defaults: auto -> form 7, shape 3, depth 2
    ctx.synchronize()
    for cam in range(2):
        assert bits_equal(np_of(ph[cam]), folded[cam]), cam
    camL, camR, Q, T = calib_parts(oracle, calib)
    exyz, ehas, _ = oracle.mf_triangulate(oracle_dec[0][0], oracle_dec[0][1], oracle_dec[1][0], oracle_dec[1][1], camL, camR, Q, T)
    xyz, has = ctx.reconstruct_mf(st[0], st[1], BLACK, True)
    ctx.synchronize()
    assert bits_equal(np_of(has), ehas) and bits_equal(np_of(xyz), exyz)
    assert 0.2 < ehas.mean() < 1.0
    cap = slr.capi
    try:
        for shape, depth, want_valid in [(3, 2, True), (3, 1, False), (0, 2, False), (1, 2, True), (1, 1, False)]:
            ctx.set_option(cap.OPT_RECT_DECODE_ALGO, 7)                            # explicit 7: an inapplicable form fails loudly
            ctx.set_option(cap.OPT_RECT_DMA_SHAPE, shape)
            ctx.set_option(cap.OPT_RECT_DMA_DEPTH, depth)
            ph, vd = ctx.mf_rectify_decode_pair(st[0], st[1], BLACK, want_valid=want_valid)
            ctx.synchronize()
            for cam in range(2):
                if want_valid:
                    assert bits_equal(np_of(vd[cam]), oracle_dec[cam][1]) and bits_equal(np_of(ph[cam]), oracle_dec[cam][0]), (shape, depth, cam)
                else:
                    assert bits_equal(np_of(ph[cam]), folded[cam]), (shape, depth, cam)
    finally:
        ctx.set_option(cap.OPT_RECT_DECODE_ALGO, 0)
        ctx.set_option(cap.OPT_RECT_DMA_SHAPE, 3)
        ctx.set_option(cap.OPT_RECT_DMA_DEPTH, 2)


def test_fullsize_mf_cloud_entry_vs_oracle(ctx, oracle, scene, oracle_dec):
    """slr_reconstruct_mf_cloud (the whole MF path + the PointCloudImage adaptor, the XYZ grid never leaving the device) against
    the oracle chain, with device buffers at two scan sizes and with host buffers (staged, SLR_MEM_HOST)"""
    st, maps, calib, _ = scene
    ctx.set_calibration(calib)
    for cam in range(2):
        ctx.set_rectify_maps(cam, maps[cam][0], maps[cam][1])
    camL, camR, Q, T = calib_parts(oracle, calib)
    exyz, ehas, _ = oracle.mf_triangulate(oracle_dec[0][0], oracle_dec[0][1], oracle_dec[1][0], oracle_dec[1][1], camL, camR, Q, T)
    for scan_w, scan_h in ((1280, 1024), (3000, 4096)):
        es, ec, _ = oracle.pointcloud_from_grid(exyz, ehas, scan_w, scan_h, None)
        s, c = ctx.reconstruct_mf_cloud(st[0], st[1], BLACK, True, scan_w, scan_h)
        ctx.synchronize()
        assert bits_equal(np_of(c), ec) and bits_equal(np_of(s), es), (scan_w, scan_h)
        assert (ec > 0).mean() > 0.1
    hl, hr = st[0].cpu().numpy(), st[1].cpu().numpy()
    es, ec, _ = oracle.pointcloud_from_grid(exyz, ehas, 1280, 1024, None)
    s, c = ctx.reconstruct_mf_cloud(hl, hr, BLACK, True, 1280, 1024)
    assert bits_equal(c, ec) and bits_equal(s, es)


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
    algos = forms(ctx, slr, slr.capi.OPT_MF_MATCH_ALGO, (3, 2, 1, 0, 4, 5, 6), required=(3, 1, 0, 4))   # (2, the sorted form: FORMS=all builds; 0 / 4 / 5 / 6: the lean shapes)
    for algo in algos:
        ctx.set_option(slr.capi.OPT_MF_MATCH_ALGO, algo)
        out[algo] = ctx.mf_triangulate(dec[0][0], dec[0][1], dec[1][0], dec[1][1])
    ctx.set_option(slr.capi.OPT_MF_MATCH_ALGO, 0)
    ctx.synchronize()
    for algo in algos:
        for a, b in zip(out[1], out[algo]):
            assert torch.equal(a, b)                   # whole frame: indexed forms == literal sweep, bit for bit
    xyz, has, mk = [np_of(t) for t in out[3]]
    assert 0.05 < has.mean() < 1.0
    camL, camR, Q, T = calib_parts(oracle, calib)
    phL, vL, phR, vR = [np_of(t) for t in (dec[0][0], dec[0][1], dec[1][0], dec[1][1])]
    # every one of the 3000 rows against the oracle's O(W^2) first-match search (~7 s single-threaded)
    exyz, ehas, emk = oracle.mf_triangulate(phL, vL, phR, vR, camL, camR, Q, T)
    assert bits_equal(mk, emk) and bits_equal(has, ehas) and bits_equal(xyz, exyz)
    # whole-path entry point == the three stages
    x2, h2 = ctx.reconstruct_mf(st[0], st[1], BLACK, True)
    ctx.synchronize()
    assert torch.equal(x2, out[3][0]) and torch.equal(h2, out[3][1])


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


def test_fullsize_ge_whole_path_with_rectification_and_colour(ctx, oracle, synth, scene):
    """config 3's Gray half at size: slr_reconstruct_ge (fused rectify + Gray decode of 26 planes per camera, K5, colour)
    and the fused decode on its own against remap -> decode -> triangulation_ge of the oracle; then the
    PointCloudImage adaptor (Q11: transposed + cropped) on the full grid"""
    _, maps, calib, _ = scene
    ctx.set_calibration(calib)
    for cam in range(2):
        ctx.set_rectify_maps(cam, maps[cam][0], maps[cam][1])
    dev = torch.device("cuda", 0)
    g = synth.render_gray_stack(W, H, W, seed=77, noise=2, device=dev)
    ncol = synth.gray_num_bits(W)
    torch.cuda.synchronize()
    edec, white = [], []
    for cam in range(2):
        raw = g[cam].cpu().numpy()
        mx, mf = maps[cam][0].cpu().numpy(), maps[cam][1].cpu().numpy()
        rect = np.stack([oracle.remap_u8(raw[p], mx, mf) for p in range(raw.shape[0])])
        ex, _, ev = oracle.gray_decode(rect, ncol, 0, BLACK, 3, W, 0)
        cx, _, v = ctx.gray_decode(g[cam], ncol, 0, BLACK, 3, W, 0, rectify_cam=cam)      # slr_gray_rectify_decode
        ctx.synchronize()
        assert bits_equal(np_of(cx), ex) and bits_equal(np_of(v), ev), cam
        edec.append((ex, ev))
        white.append(rect[0])
    _, _, Q, T = calib_parts(oracle, calib)
    exyz, ehas, ecol, _ = oracle.ge_triangulate(edec[0][0], edec[0][1], edec[1][0], edec[1][1], Q, T, white[0], white[1])
    xyz, has, col = ctx.reconstruct_ge(g[0], g[1], ncol, BLACK, 3, W, True, True)
    ctx.synchronize()
    assert bits_equal(np_of(has), ehas) and bits_equal(np_of(xyz), exyz) and bits_equal(np_of(col), ecol)
    assert ehas.mean() > 0.3
    for scan_w, scan_h in ((1280, 1024), (3000, 4096), (4096, 4096)):
        es, ec, ek = oracle.pointcloud_from_grid(exyz, ehas, scan_w, scan_h, ecol)
        s, c, k = ctx.pointcloud_from_grid(xyz, has, scan_w, scan_h, col)
        ctx.synchronize()
        assert bits_equal(np_of(c), ec) and bits_equal(np_of(s), es) and bits_equal(np_of(k), ek), (scan_w, scan_h)


def test_fullsize_gray_only_whole_path(ctx, oracle, synth):
    """GRAY_ONLY at size (reconstruct.cpp:230-265): 4096x3000 cameras, 1280x1024 projector, column + row bits (44 planes
    per camera) -> bucket scatter -> ray-ray triangulation, against the oracle's decode + bucket + Reconstruct::triangulation"""
    scan_w, scan_h = 1280, 1024
    calib, _ = synth.make_calibration(W, H, baseline=400.0, theta=0.6)
    ctx.set_calibration(calib)
    camL, camR, _, T = calib_parts(oracle, calib)
    dev = torch.device("cuda", 0)
    st = synth.render_gray_stack(W, H, scan_w, scan_h, seed=52, noise=2, device=dev, rows=True)
    ncol, nrow = synth.gray_num_bits(scan_w), synth.gray_num_bits(scan_h)
    torch.cuda.synchronize()
    dec = [oracle.gray_decode(st[c].cpu().numpy(), ncol, nrow, BLACK, 0, scan_w, scan_h) for c in range(2)]
    offL, itL = oracle.gray_bucket(dec[0][0], dec[0][1], dec[0][2], scan_w, scan_h)
    offR, itR = oracle.gray_bucket(dec[1][0], dec[1][1], dec[1][2], scan_w, scan_h)
    exyz, ecnt = oracle.ray_triangulate(offL, itL, offR, itR, camL, camR, scan_w, scan_h, T)
    xyz, cnt = ctx.reconstruct_gray(st[0], st[1], ncol, nrow, BLACK, 0, scan_w, scan_h)
    ctx.synchronize()
    assert bits_equal(np_of(cnt), ecnt) and bits_equal(np_of(xyz), exyz)
    assert (ecnt > 0).mean() > 0.2 and (ecnt > 1).any()
    got = ctx.pointcloud_get(xyz, cnt)
    ctx.synchronize()
    assert bits_equal(np_of(got), oracle.pointcloud_get(exyz, ecnt))


def test_config5_size_mfn_decode_and_chunked_match(ctx, oracle, synth):
    """BASELINE config 5 at its stated size on one GPU: 8192x6000, 4 frequencies x 8 steps, fp16 planes.  The decode
    (build extension: no reference counterpart) against the fp64 model on sampled rows, the match (rows wider than 4096 run
    the chunked K4) against the oracle's literal search on sampled rows, and the whole frame against itself cut into
    the row bands an 8-GPU run would use (dist.shard_rows)"""
    W5, H5, F, N = 8192, 6000, 4, 8
    dev = torch.device("cuda", 0)
    calib, _ = synth.make_calibration(W5, H5)
    ctx.set_calibration(calib)
    st = synth.render_mfn_stack(W5, H5, F, N, noise=0.25, seed=5, device=dev)
    torch.cuda.synchronize()
    dec = [ctx.mfn_decode(st[cam], F, N, 40.0) for cam in range(2)]
    xyz, has, mk = ctx.mf_triangulate(dec[0][0], dec[0][1], dec[1][0], dec[1][1])
    ctx.synchronize()
    rows = [0, 1, 749, 750, 2999, 3000, 4242, 5999]
    for cam in range(2):
        sub = st[cam][:, rows, :].cpu().numpy()
        exp, ev = oracle.mfn_decode_f64(sub, F, N, 40.0)
        ph, v = np_of(dec[cam][0])[rows], np_of(dec[cam][1])[rows]
        assert np.array_equal(v, ev)
        d = np.abs(ph.astype(np.float64) - exp)
        tol = 2e-3 + 1e-4 * np.abs(exp)                                  # north_star: 1e-4 relative (+ f32 floor)
        per = np.abs(d - 255.0 * np.round(d / 255.0))                    # wrap flips at an a > b decided in f32 vs f64
        assert ((d > tol) & (per > tol)).sum() == 0
        assert (d > tol).sum() <= 1e-3 * ph.size
    camL, camR, Q, T = calib_parts(oracle, calib)
    phL, vL, phR, vR = [np_of(t) for t in (dec[0][0], dec[0][1], dec[1][0], dec[1][1])]
    hx, hh, hk = np_of(xyz), np_of(has), np_of(mk)
    for r in rows:
        exyz, ehas, emk = oracle.mf_triangulate(phL, vL, phR, vR, camL, camR, Q, T, rows=(r, r + 1))
        assert bits_equal(hk[r], emk[r]) and bits_equal(hh[r], ehas[r]) and bits_equal(hx[r], exyz[r]), r
    assert hh.mean() > 0.05
    del hx, hh, hk, phL, phR
    # 8 row bands (what rank r of an 8-GPU run computes) concatenated == the whole frame
    band = (H5 + 7) // 8
    for b in range(8):
        r0, r1 = b * band, min(H5, (b + 1) * band)
        bx, bh, bk = ctx.mf_triangulate(dec[0][0][r0:r1], dec[0][1][r0:r1], dec[1][0][r0:r1], dec[1][1][r0:r1],
                                        row0=r0, image_h=H5)
        ctx.synchronize()
        assert torch.equal(bx, xyz[r0:r1]) and torch.equal(bh, has[r0:r1]) and torch.equal(bk, mk[r0:r1]), b


def test_batch_entry_ge_two_frames_with_spare_planes(ctx, oracle, synth, scene, slr):
    """slr_reconstruct_batch in SLR_MODE_GE (what `bench.py --mode ge` times; Reconstruct::runReconstruction_GE,
    reconstruct.cpp:271-307): two different 4096x3000 frames in one stack whose planes_per_cam (28) exceeds what the mode needs
    (26), rectification + colour, every frame against the oracle chain remap -> decode -> triangulation_ge"""
    _, maps, calib, _ = scene
    ctx.set_calibration(calib)
    for cam in range(2):
        ctx.set_rectify_maps(cam, maps[cam][0], maps[cam][1])
    dev = torch.device("cuda", 0)
    ncol = synth.gray_num_bits(W)
    ppc = 2 + 2 * ncol + 2
    stack = torch.full((2, 2, ppc, H, W), 123, dtype=torch.uint8, device=dev)         # spare planes: junk that must not be read
    for f in range(2):
        stack[f, :, :2 + 2 * ncol] = synth.render_gray_stack(W, H, W, seed=300 + f, noise=2, device=dev)
    torch.cuda.synchronize()
    xyz, has, col = ctx.reconstruct_batch(slr.capi.MODE_GE, stack, BLACK, 3, n_col_bits=ncol, scan_w=W, rectify=True, have_color=True)
    ctx.synchronize()
    _, _, Q, T = calib_parts(oracle, calib)
    for f in range(2):
        edec, white = [], []
        for cam in range(2):
            raw = stack[f, cam, :2 + 2 * ncol].cpu().numpy()
            mx, mf = maps[cam][0].cpu().numpy(), maps[cam][1].cpu().numpy()
            rect = np.stack([oracle.remap_u8(raw[p], mx, mf) for p in range(raw.shape[0])])
            ex, _, ev = oracle.gray_decode(rect, ncol, 0, BLACK, 3, W, 0)
            edec.append((ex, ev))
            white.append(rect[0])
        exyz, ehas, ecol, _ = oracle.ge_triangulate(edec[0][0], edec[0][1], edec[1][0], edec[1][1], Q, T, white[0], white[1])
        assert bits_equal(np_of(has[f]), ehas) and bits_equal(np_of(xyz[f]), exyz) and bits_equal(np_of(col[f]), ecol), f
        assert ehas.mean() > 0.3
    assert not torch.equal(has[0], has[1])


def test_batch_entry_gray_only_two_frames_with_spare_planes(ctx, oracle, synth, slr):
    """slr_reconstruct_batch in SLR_MODE_GRAY (what `bench.py --mode gray` times; Reconstruct::runReconstruction,
    reconstruct.cpp:230-265): two different frames, 46 planes per camera in the stack for a mode that needs 44"""
    scan_w, scan_h = 1280, 1024
    calib, _ = synth.make_calibration(W, H, baseline=400.0, theta=0.6)
    ctx.set_calibration(calib)
    camL, camR, _, T = calib_parts(oracle, calib)
    dev = torch.device("cuda", 0)
    ncol, nrow = synth.gray_num_bits(scan_w), synth.gray_num_bits(scan_h)
    need = 2 + 2 * ncol + 2 * nrow
    stack = torch.full((2, 2, need + 2, H, W), 77, dtype=torch.uint8, device=dev)
    for f in range(2):
        g = synth.render_gray_stack(W, H, scan_w, scan_h, seed=60 + f, noise=2, device=dev, rows=True)
        stack[f, :, :need] = g if f == 0 else torch.flip(g, dims=[0])          # frame 1: the cameras swapped
        del g
    torch.cuda.synchronize()
    xyz, cnt, _ = ctx.reconstruct_batch(slr.capi.MODE_GRAY, stack, BLACK, 0, n_col_bits=ncol, n_row_bits=nrow, scan_w=scan_w,
                                        scan_h=scan_h, rectify=False)
    ctx.synchronize()
    for f in range(2):
        dec = [oracle.gray_decode(stack[f, c, :need].cpu().numpy(), ncol, nrow, BLACK, 0, scan_w, scan_h) for c in range(2)]
        offL, itL = oracle.gray_bucket(dec[0][0], dec[0][1], dec[0][2], scan_w, scan_h)
        offR, itR = oracle.gray_bucket(dec[1][0], dec[1][1], dec[1][2], scan_w, scan_h)
        exyz, ecnt = oracle.ray_triangulate(offL, itL, offR, itR, camL, camR, scan_w, scan_h, T)
        assert bits_equal(np_of(cnt[f]), ecnt) and bits_equal(np_of(xyz[f]), exyz), f
        assert (ecnt > 0).mean() > 0.2
    assert not torch.equal(xyz[0], xyz[1])                    # (the pair COUNTS are symmetric in the cameras, the points are not)


def test_batch_entry_gray_only_frames_pipelined_over_two_streams(ctx, synth, slr):
    """slr_reconstruct_batch in SLR_MODE_GRAY alternates its frames between two streams and two scratch sets: 7 different frames in
    one call (each set reused by four / three frames, different sizes before and after so that the scratch is reallocated) ==
    the same frames through slr_reconstruct_gray one at a time, twice in a row, and the context is usable right after"""
    dev = torch.device("cuda", 0)
    for (w, h, sw, sh, nf) in ((640, 96, 200, 60, 3), (1024, 768, 320, 256, 7), (512, 64, 128, 40, 2)):
        calib, _ = synth.make_calibration(w, h, baseline=400.0, theta=0.6)
        ctx.set_calibration(calib)
        ncol, nrow = synth.gray_num_bits(sw), synth.gray_num_bits(sh)
        stack = torch.stack([synth.render_gray_stack(w, h, sw, sh, seed=300 + f, noise=2 + f % 3, device=dev, rows=True) for f in range(nf)]).contiguous()
        torch.cuda.synchronize()
        one = [ctx.reconstruct_gray(stack[f, 0], stack[f, 1], ncol, nrow, BLACK, 0, sw, sh) for f in range(nf)]
        ctx.synchronize()
        for rep in range(3):
            ctx.set_option(slr.capi.OPT_BATCH_STREAMS, 1 if rep == 2 else 2)
            xyz, cnt, _ = ctx.reconstruct_batch(slr.capi.MODE_GRAY, stack, BLACK, 0, n_col_bits=ncol, n_row_bits=nrow, scan_w=sw, scan_h=sh,
                                                rectify=False)
            ctx.set_option(slr.capi.OPT_BATCH_STREAMS, 2)
            again = ctx.reconstruct_gray(stack[0, 0], stack[0, 1], ncol, nrow, BLACK, 0, sw, sh)     # queued right behind the batch
            ctx.synchronize()
            for f in range(nf):
                assert torch.equal(cnt[f], one[f][1]) and torch.equal(xyz[f], one[f][0]), (w, h, f, rep)
            assert torch.equal(again[1], one[0][1]) and torch.equal(again[0], one[0][0])
            assert (cnt[0] > 0).float().mean().item() > 0.2


@pytest.mark.parametrize("theta,k1", [(0.2, -0.15), (0.3, -0.2)])
def test_fullsize_verged_rig_maps_from_stereo_rectify(ctx, oracle, synth, scene, theta, k1):
    """Maps of a VERGED rig (keystone: rows tilt towards the image sides) built the way the reference builds them -- the host
    mirror's stereoRectify, then initUndistortRectifyMap on the device (stereorect.cpp:36-44) -- instead of the near-identity maps
    of the other tests: the device maps equal the oracle's map builder, and the fused pair decode (whatever form auto selects
    for these maps: the LDS-DMA form when every tile box fits, else round 1's forms with their per-tile gather) equals
    remap -> decode of the oracle for both cameras."""
    st, _, _, _ = scene
    rig = synth.make_verged_rig(W, H, theta, k1)
    synth.install_verged_maps(ctx, rig, W, H)
    info = [ctx.rectify_info(cam) for cam in range(2)]
    print("verged rig theta %.1f k1 %.2f:" % (theta, k1), info)
    ph, vd = ctx.mf_rectify_decode_pair(st[0], st[1], BLACK, want_valid=True)
    ctx.synchronize()
    for cam in range(2):
        mx, mf = ctx.get_rectify_maps(cam, W, H)
        R, P = (rig["R1"], rig["P1"]) if cam == 0 else (rig["R2"], rig["P2"])
        ex, ef = oracle.init_undistort_rectify_map(rig["M"][cam], rig["D"][cam], R, P, W, H)
        assert np.array_equal(mx, ex) and np.array_equal(mf, ef), cam
        raw = st[cam].cpu().numpy()
        rect = np.stack([oracle.remap_u8(raw[p], mx, mf) for p in range(14)])
        eph, ev = oracle.mf_decode(rect, BLACK)
        assert bits_equal(np_of(vd[cam]), ev) and bits_equal(np_of(ph[cam]), eph), cam
        assert 0.3 < ev.mean() < 1.0
    # no silent drop to round 1's forms: the LDS-DMA form keeps these maps, the few corner tiles whose boxes it does not hold
    # (keystone) go through its gather fix-up pass
    assert all(i["mf_form"] == 7 and 4 * i["dma_nofit_tiles"] <= i["dma_tiles"] for i in info), info
    assert sum(i["dma_extra_entries"] for i in info) > 0                                     # (tiles were decoded in parts)
    # the Gray fused decode on the same maps (its own fix-up kernel)
    g = synth.render_gray_stack(W, H, 1024, seed=9, noise=2, device=st.device)
    ncol = synth.gray_num_bits(1024)
    torch.cuda.synchronize()
    for cam in range(2):
        mx, mf = ctx.get_rectify_maps(cam, W, H)
        raw = g[cam].cpu().numpy()
        rect = np.stack([oracle.remap_u8(raw[p], mx, mf) for p in range(raw.shape[0])])
        ex, _, ev = oracle.gray_decode(rect, ncol, 0, BLACK, 3, 1024, 0)
        cx, _, v = ctx.gray_decode(g[cam], ncol, 0, BLACK, 3, 1024, 0, rectify_cam=cam)
        ctx.synchronize()
        assert bits_equal(np_of(cx), ex) and bits_equal(np_of(v), ev), cam


@pytest.mark.parametrize("theta,k1", [(0.1, -0.10), (0.3, -0.2)])
def test_verged_rig_quads_sorted_by_class(ctx, synth, scene, slr, theta, k1):
    """A tile's straddling quads are handed to as few of its waves as hold them when the maps are installed (dma_tiles_kernel,
    "quads sorted by class"; every quad carries its slot in the tile): the fused multi-frequency pair decode and the fused Gray
    decode give bit for bit what they give with every wave on the quads of its own block (SLR_OPT_DEBUG_FLAGS bit 5), and far
    fewer waves take the three-row / per-pixel read modes."""
    st = scene[0]
    rig = synth.make_verged_rig(W, H, theta, k1)
    g = synth.render_gray_stack(W, H, 1024, seed=11, noise=2, device=st.device)
    ncol = synth.gray_num_bits(1024)
    got = {}
    try:
        for flags in (32, 0):
            ctx.set_option(slr.capi.OPT_DEBUG_FLAGS, flags)
            synth.install_verged_maps(ctx, rig, W, H)
            info = [ctx.rectify_info(cam) for cam in range(2)]
            ph, vd = ctx.mf_rectify_decode_pair(st[0], st[1], BLACK, want_valid=True)
            ctx.synchronize()
            outs = [ph[0].clone(), ph[1].clone(), vd[0].clone(), vd[1].clone()]
            for cam in range(2):
                cx, _, v = ctx.gray_decode(g[cam], ncol, 0, BLACK, 3, 1024, 0, rectify_cam=cam)
                ctx.synchronize()
                outs += [cx.clone(), v.clone()]
            got[flags] = (info, outs)
    finally:
        ctx.set_option(slr.capi.OPT_DEBUG_FLAGS, 0)
    print("verged rig theta %.1f: own blocks %s, sorted %s" % (theta, [i["waves_by_mode"] for i in got[32][0]],
                                                               [i["waves_by_mode"] for i in got[0][0]]))
    for a, b in zip(got[32][1], got[0][1]):
        assert torch.equal(a.view(torch.uint8) if a.dtype != torch.uint8 else a, b.view(torch.uint8) if b.dtype != torch.uint8 else b)
    for cam in range(2):
        own, srt = got[32][0][cam], got[0][0][cam]
        assert own["mf_form"] == 7 and srt["mf_form"] == 7
        assert own["quads_by_class"][0] + own["quads_by_class"][1] + own["quads_by_class"][2] == sum(srt["quads_by_class"])
        assert 2 * (srt["waves_by_mode"][1] + srt["waves_by_mode"][2]) < own["waves_by_mode"][1] + own["waves_by_mode"][2], (own, srt)


def test_split_tiles_every_part_is_decoded(ctx, synth, scene, slr):
    """Entries of split tiles (a verged rig's corner tiles are decoded in wave-aligned parts, dma_tiles_kernel) sit at the end of
    every pool of the tile schedule, one after the other, with wave 0 idle in most of them -- the path on which the next tile's
    ticket was once read before the scalar atomic had returned it (fused Gray decode, 256 x 8 tiles: the workgroup decoded entry
    first + 1 again and the entry the ticket named never; ~12 % of the parts, another set every run, invisible wherever the output
    buffer still held an earlier run's values).  Outputs are therefore allocated over poisoned blocks: a pixel nobody writes
    cannot equal the gather form's result.  Both fused decodes, both tile shapes that serve them, few and many workgroups."""
    st = scene[0]
    rig = synth.make_verged_rig(W, H, 0.15, -0.12)
    g = synth.render_gray_stack(W, H, 1024, seed=9, noise=2, device=st.device)
    ncol = synth.gray_num_bits(1024)

    def poison():
        t = [torch.full((H, W), 0x7B7B7B7B, dtype=torch.int32, device=st.device) for _ in range(6)]
        u = [torch.full((H, W), 0x7B, dtype=torch.uint8, device=st.device) for _ in range(6)]
        torch.cuda.synchronize()
        del t, u

    def decode():
        poison()
        ph, vd = ctx.mf_rectify_decode_pair(st[0], st[1], BLACK, want_valid=True)
        ctx.synchronize()
        outs = [ph[0].clone(), ph[1].clone(), vd[0].clone(), vd[1].clone()]
        del ph, vd
        for cam in range(2):
            poison()
            cx, _, v = ctx.gray_decode(g[cam], ncol, 0, BLACK, 3, 1024, 0, rectify_cam=cam)
            ctx.synchronize()
            outs += [cx.clone(), v.clone()]
            del cx, v
        return outs

    try:
        ctx.set_option(slr.capi.OPT_RECT_DECODE_ALGO, 1)                    # the per-pixel gather form: no tiles, no schedule
        synth.install_verged_maps(ctx, rig, W, H)
        ref = decode()
        ctx.set_option(slr.capi.OPT_RECT_DECODE_ALGO, 0)
        for shape in (1, 3):
            ctx.set_option(slr.capi.OPT_RECT_DMA_SHAPE, shape)
            synth.install_verged_maps(ctx, rig, W, H)
            info = [ctx.rectify_info(cam) for cam in range(2)]
            assert all(i["mf_form"] == 7 for i in info) and sum(i["dma_extra_entries"] for i in info) > (100 if shape == 1 else 0), info
            for resident in (0, 8):
                ctx.set_option(slr.capi.OPT_DEBUG_RECT_RESIDENT, resident)
                for rep in range(2):
                    got = decode()
                    for k, (a, b) in enumerate(zip(got, ref)):
                        same = a.view(torch.uint8) == b.view(torch.uint8)
                        assert bool(same.all()), (shape, resident, rep, k, int((~same).sum()))
    finally:
        ctx.set_option(slr.capi.OPT_DEBUG_RECT_RESIDENT, 0)
        ctx.set_option(slr.capi.OPT_RECT_DECODE_ALGO, 0)
        ctx.set_option(slr.capi.OPT_RECT_DMA_SHAPE, 3)
