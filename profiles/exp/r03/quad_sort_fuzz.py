"""One-off stress of the sorted map digest (dma_tiles_kernel, "quads sorted by class"): for verged rigs and rolled / distorted
synthetic maps of several strengths, at 4096x3000 and at a ragged small size, on every compiled tile shape, the fused MF pair decode
and the fused Gray decode must give bit for bit what the unsorted digest (SLR_OPT_DEBUG_FLAGS bit 5) gives."""
import sys, os, importlib
import numpy as np, torch
sys.path.insert(0, os.getcwd())
slr = importlib.import_module("structure-light-reconstructor_amd")
synth = importlib.import_module("structure-light-reconstructor_amd.synth")
capi = slr.capi
BLACK = 40
ctx = slr.Context(0)
n = 0
def run(W, H, install, tag):
    global n
    st = synth.render_mf_stack(W, H, seed=7, noise=3, device="cuda")
    g = synth.render_gray_stack(W, H, 1024, seed=9, noise=2, device="cuda")
    ncol = synth.gray_num_bits(1024)
    ctx.set_calibration(synth.make_calibration(W, H)[0])
    for shape in (0, 1, 3):
        ctx.set_option(capi.OPT_RECT_DECODE_ALGO, 0)
        ctx.set_option(capi.OPT_RECT_DMA_SHAPE, shape)
        got = {}
        for flags in (32, 0):
            ctx.set_option(capi.OPT_DEBUG_FLAGS, flags)
            install()
            info = [ctx.rectify_info(c) for c in range(2)]
            ph, vd = ctx.mf_rectify_decode_pair(st[0], st[1], BLACK, want_valid=True)
            ctx.synchronize()
            outs = [ph[0].clone(), ph[1].clone(), vd[0].clone(), vd[1].clone()]
            for cam in range(2):
                cx, _, v = ctx.gray_decode(g[cam], ncol, 0, BLACK, 3, 1024, 0, rectify_cam=cam)
                ctx.synchronize()
                outs += [cx.clone(), v.clone()]
            got[flags] = (info, outs)
        ctx.set_option(capi.OPT_DEBUG_FLAGS, 0)
        for a, b in zip(got[32][1], got[0][1]):
            assert torch.equal(a.view(torch.uint8), b.view(torch.uint8)), (tag, shape)
        n += 1
        print(tag, "shape", shape, "form", [i["mf_form"] for i in got[0][0]], "waves own", [i["waves_by_mode"] for i in got[32][0]],
              "sorted", [i["waves_by_mode"] for i in got[0][0]], flush=True)
for (W, H) in ((4096, 3000), (1040, 524)):
    for theta, k1 in ((0.05, -0.05), (0.15, -0.12), (0.25, 0.1), (0.35, -0.25)):
        rig = synth.make_verged_rig(W, H, theta, k1)
        run(W, H, lambda: synth.install_verged_maps(ctx, rig, W, H), "rig %dx%d theta %.2f k1 %.2f" % (W, H, theta, k1))
    for strength in (0.5, 1.0, 2.0, 4.0):
        maps = [synth.make_rectify_maps(W, H, cam, strength=strength) for cam in range(2)]
        def inst():
            for cam in range(2):
                ctx.set_rectify_maps(cam, maps[cam][0].numpy(), maps[cam][1].numpy())
        run(W, H, inst, "maps %dx%d strength %.1f" % (W, H, strength))
print("quad sort fuzz: %d configurations bit-equal" % n)
