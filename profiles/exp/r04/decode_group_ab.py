#!/usr/bin/env python
"""the fused MF decode inside slr_reconstruct_mf_batch (8 frames) with 1 / 2 / 4 / 8 frames per decode launch, on four rigs: us per frame
of the pair decode and of K4 (library profiler, 3 batches each, 3 alternations)"""
import importlib, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
os.environ.pop("SLR_POISON_OUTPUTS", None); os.environ.pop("SLR_POISON_SCRATCH", None)
import torch
slr = importlib.import_module("structure-light-reconstructor_amd"); synth = importlib.import_module("structure-light-reconstructor_amd.synth")
capi = slr.capi
W, H, F = 4096, 3000, 8
dev = torch.device("cuda", 0)
stack = torch.stack([synth.render_mf_stack(W, H, seed=1234 + f, noise=2, device=dev) for f in range(F)])
xyz = torch.empty((F, H, W, 3), dtype=torch.float32, device=dev); has = torch.empty((F, H, W), dtype=torch.uint8, device=dev)
ctx = slr.Context(0)
rigs = [("near-identity", None), ("verged 0.1", (0.1, -0.10)), ("verged 0.2", (0.2, -0.15)), ("verged 0.3", (0.3, -0.20))]
groups = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1,2,4,8").split(",")]
ctx.set_option(capi.OPT_PROFILE_STRIDE, 1)
for name, rg in rigs:
    if rg:
        rig = synth.make_verged_rig(W, H, rg[0], rg[1]); ctx.set_calibration(rig["calib"]); synth.install_verged_maps(ctx, rig, W, H)
    else:
        calib, _ = synth.make_calibration(W, H); ctx.set_calibration(calib)
        for cam in range(2):
            mx, mf = synth.make_rectify_maps(W, H, cam, device=dev); ctx.set_rectify_maps(cam, mx, mf)
    res = {g: [] for g in groups}
    for rep in range(3):
        for g in groups:
            ctx.set_option(capi.OPT_MF_BATCH_DECODE_GROUP, g)
            ctx.reconstruct_mf_batch(stack, 40, True, W=W, xyz=xyz, has=has); ctx.synchronize()
            ctx.profile_enable(True); ctx.profile_reset()
            for _ in range(3):
                ctx.reconstruct_mf_batch(stack, 40, True, W=W, xyz=xyz, has=has)
            p = ctx.profile(); ctx.profile_enable(False)
            d, k = p["slr_mf_rectify_decode_pair"], p["slr_mf_match_triangulate"]
            res[g].append((d[0] / d[1] * 1e3, k[0] / k[1] * 1e3))
    print("%-14s " % name + "   ".join("dg %d: decode %.1f  K4 %.1f" % (g, statistics.median(x[0] for x in res[g]), statistics.median(x[1] for x in res[g])) for g in groups), flush=True)
