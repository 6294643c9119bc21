"""Small driver for rocprofv3 runs: launches each kernel of the MF / Gray paths a few times on one 4096x3000 frame.
Used by profiles/run_profile.sh (kernel-trace stats and separate --pmc passes); not part of the product."""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
slr = importlib.import_module("structure-light-reconstructor_amd")
synth = importlib.import_module("structure-light-reconstructor_amd.synth")

W = int(os.environ.get("SLR_W", "4096"))
H = int(os.environ.get("SLR_H", "3000"))
REPS = int(os.environ.get("SLR_REPS", "3"))
WHAT = os.environ.get("SLR_WHAT", "mf").split(",")
dev = torch.device("cuda", 0)
ctx = slr.Context(0)
calib, _ = synth.make_calibration(W, H)
ctx.set_calibration(calib)
maps = [synth.make_rectify_maps(W, H, cam, device=dev) for cam in range(2)]
torch.cuda.synchronize()
for cam in range(2):
    ctx.set_rectify_maps(cam, maps[cam][0], maps[cam][1])

if "mf" in WHAT:
    st = synth.render_mf_stack(W, H, seed=1234, device=dev)
    torch.cuda.synchronize()
    ph = [torch.empty((H, W), dtype=torch.float32, device=dev) for _ in range(2)]
    vd = [torch.empty((H, W), dtype=torch.uint8, device=dev) for _ in range(2)]
    for _ in range(REPS):
        ctx.mf_decode(st[0], 40, phase=ph[0], valid=vd[0])                     # K2 unfused
    for _ in range(REPS):
        for cam in range(2):
            ctx.mf_decode(st[cam], 40, rectify_cam=cam, phase=ph[cam], valid=vd[cam])   # fused K1+K2 (auto: the LDS-DMA form)
    ctx.set_option(slr.capi.OPT_RECT_DECODE_ALGO, 5)
    for _ in range(REPS):
        for cam in range(2):
            ctx.mf_decode(st[cam], 40, rectify_cam=cam, phase=ph[cam], valid=vd[cam])   # round 1's 128x8 register-staged form
    ctx.set_option(slr.capi.OPT_RECT_DECODE_ALGO, 0)
    x2 = torch.empty((2, H, W, 3), dtype=torch.float32, device=dev); h2 = torch.empty((2, H, W), dtype=torch.uint8, device=dev)
    st2 = torch.stack([st, torch.flip(st, dims=[0])]).contiguous()
    for _ in range(REPS):
        ctx.reconstruct_mf_batch(st2, 40, True, xyz=x2, has=h2)                # the pair launch (valid folded) + K4, as bench.py
    for _ in range(REPS):
        ctx.mf_triangulate(ph[0], vd[0], ph[1], vd[1], want_match=False)       # K4 indexed (lean form)
    ctx.set_option(slr.capi.OPT_MF_MATCH_ALGO, 3)
    for _ in range(REPS):
        ctx.mf_triangulate(ph[0], vd[0], ph[1], vd[1], want_match=False)       # K4, round 1's general binned form
    ctx.set_option(slr.capi.OPT_MF_MATCH_ALGO, 0)
    tmp = torch.empty((H, W), dtype=torch.uint8, device=dev)
    for _ in range(REPS):
        ctx.remap_u8(0, st[0, 3], out=tmp)                                     # K1
    if "sweep" in WHAT:
        ctx.set_option(slr.capi.OPT_MF_MATCH_ALGO, 1)
        ctx.mf_triangulate(ph[0], vd[0], ph[1], vd[1], want_match=False)       # K4 linear sweep
        ctx.set_option(slr.capi.OPT_MF_MATCH_ALGO, 0)
if "gray" in WHAT:
    g = synth.render_gray_stack(W, H, W, seed=1234, device=dev)
    ncol = synth.gray_num_bits(W)
    torch.cuda.synchronize()
    for _ in range(REPS):
        dec = [ctx.gray_decode(g[cam], ncol, 0, 40, 0, W, 0) for cam in range(2)]          # K3
    for _ in range(REPS):
        dec = [ctx.gray_decode(g[cam], ncol, 0, 40, 0, W, 0, rectify_cam=cam) for cam in range(2)]   # fused K1+K3
    for _ in range(REPS):
        ctx.ge_triangulate(dec[0][0], dec[0][2], dec[1][0], dec[1][2], want_match=False)   # K5 (lean form)
    ctx.set_option(slr.capi.OPT_RECT_DECODE_ALGO, 5)
    for _ in range(REPS):
        ctx.gray_decode(g[0], ncol, 0, 40, 0, W, 0, rectify_cam=0)                          # round 1's register-staged fused form
    ctx.set_option(slr.capi.OPT_RECT_DECODE_ALGO, 0)
if "ge" in WHAT:
    g = synth.render_gray_stack(W, H, W, seed=1234, device=dev)
    ncol = synth.gray_num_bits(W)
    torch.cuda.synchronize()
    for _ in range(REPS):
        ctx.reconstruct_ge(g[0], g[1], ncol, 40, 0, W, True, False)                                # GRAY_EPI as bench.py --mode ge: pair launch + K5
if "ray" in WHAT:
    # GRAY_ONLY as bench.py --mode gray: column + row bits of a 1280 x 1024 projector under the 4096 x 3000 cameras (~10 camera
    # pixels per projector cell and camera); decode fused into the bucket histogram, scan, scatter beside K6's list kernels, K6
    calib2, _ = synth.make_calibration(W, H, baseline=400.0, theta=0.6)
    ctx.set_calibration(calib2)
    SW, SH = int(os.environ.get("SLR_SCAN_W", "1280")), int(os.environ.get("SLR_SCAN_H", "1024"))
    g2 = synth.render_gray_stack(W, H, SW, SH, seed=1234, device=dev, rows=True)
    nc, nr = synth.gray_num_bits(SW), synth.gray_num_bits(SH)
    torch.cuda.synchronize()
    for _ in range(REPS):
        ctx.reconstruct_gray(g2[0], g2[1], nc, nr, 40, 0, SW, SH)
ctx.synchronize()
ctx.close()
print("prof_driver done")
