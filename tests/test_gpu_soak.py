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
        ctx.set_option(cap.OPT_RECT_DECODE_ALGO, 1 if f % 2 else 3)            # gather / ring forms, one camera per launch
        dec = [ctx.mf_decode(st[0, cam], 40, rectify_cam=cam) for cam in range(2)]
        ctx.set_option(cap.OPT_MF_MATCH_ALGO, 2)                               # radix-sorted form
        ex, eh, _ = ctx.mf_triangulate(dec[0][0], dec[0][1], dec[1][0], dec[1][1])
        ctx.synchronize()
        assert torch.equal(has[0], eh) and torch.equal(xyz[0], ex), f
        assert eh.float().mean().item() > 0.3
    ctx.set_option(cap.OPT_RECT_DECODE_ALGO, 0); ctx.set_option(cap.OPT_MF_MATCH_ALGO, 0)
