"""cost asymmetry of the two cameras' maps: the fused decode launched per camera (HASVALID and folded), event-timed;
plus the map statistics (quad classes, wave modes)"""
import importlib, os, sys, torch, numpy as np
sys.path.insert(0, os.getcwd())
slr = importlib.import_module("structure-light-reconstructor_amd"); synth = importlib.import_module("structure-light-reconstructor_amd.synth")
W, H = 4096, 3000; dev = torch.device("cuda", 0); ctx = slr.Context(0)
maps = [synth.make_rectify_maps(W, H, cam, device=dev) for cam in range(2)]
sts = [synth.render_mf_stack(W, H, seed=1234 + i, device=dev) for i in range(4)]; torch.cuda.synchronize()
for swap in (0, 1):
    for cam in range(2):
        m = maps[cam ^ swap]
        ctx.set_rectify_maps(cam, m[0], m[1])
    print("swap", swap, [ctx.rectify_info(c) for c in range(2)])
    for cam in range(2):
        for _ in range(3): ctx.mf_decode(sts[0][cam], 40, rectify_cam=cam)
        ctx.profile_enable(True); ctx.profile_reset()
        for i in range(12): ctx.mf_decode(sts[i % 4][cam], 40, rectify_cam=cam)
        p = ctx.profile(); ctx.profile_enable(False)
        print("  cam %d alone:" % cam, {k: round(v[0] / v[1] * 1e3, 1) for k, v in p.items()})
    for _ in range(3): ctx.mf_rectify_decode_pair(sts[0][0], sts[0][1], 40, want_valid=False)
    ctx.profile_enable(True); ctx.profile_reset()
    for i in range(12): ctx.mf_rectify_decode_pair(sts[i % 4][0], sts[i % 4][1], 40, want_valid=False)
    p = ctx.profile(); ctx.profile_enable(False)
    print("  pair:", {k: round(v[0] / v[1] * 1e3, 1) for k, v in p.items()})
