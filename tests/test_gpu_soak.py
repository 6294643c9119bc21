"""Repetition test: many different frames through the production kernels (hash-binned K4 with LDS atomics, persistent
pipelined fused decode in pair launches) against their independent device forms (radix-sorted K4, direct-gather decode).
A missing barrier or an atomics race would show up as a rare mismatch; every frame must agree bit for bit."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
N_FRAMES = int(os.environ.get("SLR_SOAK_FRAMES", "40"))


def test_soak_production_vs_independent_device_forms(ctx, synth, slr):
    W, H = int(os.environ.get("SLR_SOAK_W", "2048")), int(os.environ.get("SLR_SOAK_H", "256"))
    dev = torch.device("cuda", 0)
    calib, _ = synth.make_calibration(W, H, with_T=True)
    ctx.set_calibration(calib)
    maps = [synth.make_rectify_maps(W, H, cam, device=dev, strength=2.0) for cam in range(2)]
    torch.cuda.synchronize()
    for cam in range(2):
        ctx.set_rectify_maps(cam, maps[cam][0], maps[cam][1])
    cap = slr.capi
    for f in range(N_FRAMES):
        st = synth.render_mf_stack(W, H, seed=5000 + f, noise=f % 4, device=dev).unsqueeze(0).contiguous()
        ctx.set_option(cap.OPT_RECT_DECODE_ALGO, 0); ctx.set_option(cap.OPT_MF_MATCH_ALGO, 0)
        xyz, has = ctx.reconstruct_mf_batch(st, 40, True)                      # pair launch + binned K4
        ctx.set_option(cap.OPT_RECT_DECODE_ALGO, 1 if f % 2 else 6)            # gather / 64 x 8 LDS tiles, one camera per launch
        dec = [ctx.mf_decode(st[0, cam], 40, rectify_cam=cam) for cam in range(2)]
        ctx.set_option(cap.OPT_MF_MATCH_ALGO, 1 if f % 4 == 0 else 3)          # literal sweep / general binned form
        ex, eh, _ = ctx.mf_triangulate(dec[0][0], dec[0][1], dec[1][0], dec[1][1])
        ctx.synchronize()
        assert torch.equal(has[0], eh) and torch.equal(xyz[0], ex), f
        assert eh.float().mean().item() > 0.3
    ctx.set_option(cap.OPT_RECT_DECODE_ALGO, 0); ctx.set_option(cap.OPT_MF_MATCH_ALGO, 0)


def test_soak_randomized_slice_against_the_oracle(ctx, synth, slr, oracle):
    """A seeded, time-bounded slice (SLR_SOAK_SECONDS, default 25 s) of the round-2 randomized soak (profiles/exp/soak/lean_soak.py:
    900 cases there) -- this one compares with the ORACLE, not with another device form: lean K4 / K5 at random widths of their
    ranges on random, sorted and quantised rows, the fused MF and Gray LDS-DMA decodes on random smooth maps of random strength
    with few resident workgroups (many tiles per workgroup)."""
    import time

    import numpy as np

    from util import bits_equal, calib_parts, np_of
    cap = slr.capi
    rng = np.random.default_rng(20260929)
    t_end = time.time() + float(os.environ.get("SLR_SOAK_SECONDS", "25"))
    n = {"k4": 0, "k5": 0, "mf": 0, "gray": 0}
    try:
        while time.time() < t_end or min(n.values()) < 2:
            W = int(rng.choice([516, 1000, 1024, 2052, 3000, 3584, 4096])); H = int(rng.integers(1, 4))
            calib, _ = synth.make_calibration(max(W, 8), max(H, 8), with_T=bool(rng.integers(0, 2)))
            ctx.set_calibration(calib)
            camL, camR, Q, T = calib_parts(oracle, calib)
            q = float(rng.choice([0.03, 0.07, 0.25, 1.0]))
            phL = (rng.integers(-60, 500, (H, W)) * q).astype(np.float32); phR = (rng.integers(-60, 500, (H, W)) * q).astype(np.float32)
            if rng.integers(0, 2):
                phL.sort(axis=1); phR.sort(axis=1)
            vL = (rng.random((H, W)) < 0.9).astype(np.uint8); vR = (rng.random((H, W)) < 0.9).astype(np.uint8)
            e = oracle.mf_triangulate(phL, vL, phR, vR, camL, camR, Q, T)
            g = ctx.mf_triangulate(phL, vL, phR, vR)
            assert bits_equal(g[2], e[2]) and bits_equal(g[1], e[1]) and bits_equal(g[0], e[0]), ("k4", W, H, q, n)
            n["k4"] += 1
            if W > 2048:
                nc = int(rng.choice([5, 300, 3000, 9000]))
                cL = rng.integers(0, nc, (H, W)).astype(np.int32); cR = rng.integers(0, nc, (H, W)).astype(np.int32)
                if rng.integers(0, 2):
                    cL.sort(axis=1); cR.sort(axis=1)
                e = oracle.ge_triangulate(cL, vL, cR, vR, Q, T)
                g = ctx.ge_triangulate(cL, vL, cR, vR)
                assert bits_equal(g[3], e[3]) and bits_equal(g[1], e[1]) and bits_equal(g[0], e[0]), ("k5", W, H, nc, n)
                n["k5"] += 1
            W = int(rng.choice([256, 400 // 16 * 16, 640, 1024])); H = int(rng.integers(20, 120)); cam = int(rng.integers(0, 2))
            strength = float(rng.choice([0.3, 1.0, 2.0, 3.5]))
            mx, mf = synth.make_rectify_maps(W, H, cam, strength=strength)
            mxn, mfn = mx.numpy(), mf.numpy()
            ctx.set_option(cap.OPT_DEBUG_RECT_RESIDENT, int(rng.choice([0, 8, 16])))
            ctx.set_rectify_maps(cam, mxn, mfn)
            st = synth.render_mf_stack(W, H, seed=int(rng.integers(1, 1 << 30)), noise=3)
            raw = st[cam].numpy()
            rect = np.stack([oracle.remap_u8(raw[p], mxn, mfn) for p in range(14)])
            eph, ev = oracle.mf_decode(rect, 40)
            ph, v = ctx.mf_decode(st[cam].cuda(), 40, rectify_cam=cam)
            ctx.synchronize()
            assert bits_equal(np_of(ph), eph) and bits_equal(np_of(v), ev), ("mf", W, H, strength, n)
            n["mf"] += 1
            sw = int(rng.choice([100, 600, 1280, 4096])); rows = bool(rng.integers(0, 2)); sh = int(rng.choice([90, 1024]))
            gs = synth.render_gray_stack(W, H, sw, sh if rows else None, seed=int(rng.integers(1, 1 << 30)), noise=3, rows=rows)
            nc_, nr_ = synth.gray_num_bits(sw), (synth.gray_num_bits(sh) if rows else 0)
            raw = gs[cam].numpy()
            rect = np.stack([oracle.remap_u8(raw[p], mxn, mfn) for p in range(raw.shape[0])])
            ex, ey, ev = oracle.gray_decode(rect, nc_, nr_, 40, 3, sw, sh if rows else 0)
            cx, cy, v = ctx.gray_decode(gs[cam].cuda(), nc_, nr_, 40, 3, sw, sh if rows else 0, rectify_cam=cam)
            ctx.synchronize()
            assert bits_equal(np_of(cx), ex) and bits_equal(np_of(v), ev) and (not rows or bits_equal(np_of(cy), ey)), ("gray", W, H, sw, rows, strength, n)
            n["gray"] += 1
    finally:
        ctx.set_option(cap.OPT_DEBUG_RECT_RESIDENT, 0)
    assert min(n.values()) >= 2, n
