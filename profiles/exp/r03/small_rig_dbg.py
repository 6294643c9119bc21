"""the quad-sort fuzz under poisoned outputs failed on 'rig 1040x524 theta 0.15 k1 -0.12' shape 1: which output, which pixels"""
import sys, os, importlib
os.environ["SLR_POISON_OUTPUTS"] = "1"
import numpy as np, torch
sys.path.insert(0, os.getcwd())
slr = importlib.import_module("structure-light-reconstructor_amd")
synth = importlib.import_module("structure-light-reconstructor_amd.synth")
capi = slr.capi
BLACK = 40
ctx = slr.Context(0)
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1040, 524)
st = synth.render_mf_stack(W, H, seed=7, noise=3, device="cuda")
g = synth.render_gray_stack(W, H, 1024, seed=9, noise=2, device="cuda")
ncol = synth.gray_num_bits(1024)
ctx.set_calibration(synth.make_calibration(W, H)[0])
names = ["ph0", "ph1", "vd0", "vd1", "cx0", "v0", "cx1", "v1"]
def unwritten(x):
    if x.dtype == torch.uint8: return x == 0x7B
    return x.view(torch.int32) == 0x7B7B7B7B
def decode():
    ph, vd = ctx.mf_rectify_decode_pair(st[0], st[1], BLACK, want_valid=True)
    ctx.synchronize()
    outs = [ph[0].clone(), ph[1].clone(), vd[0].clone(), vd[1].clone()]
    for cam in range(2):
        cx, _, v = ctx.gray_decode(g[cam], ncol, 0, BLACK, 3, 1024, 0, rectify_cam=cam)
        ctx.synchronize()
        outs += [cx.clone(), v.clone()]
    return outs
SH = {0: (256, 16), 1: (256, 8), 3: (128, 16)}
for theta, k1 in ((0.15, -0.12), (0.25, 0.1)):
    rig = synth.make_verged_rig(W, H, theta, k1)
    ctx.set_option(capi.OPT_DEBUG_FLAGS, 0)
    ctx.set_option(capi.OPT_RECT_DECODE_ALGO, 1)
    synth.install_verged_maps(ctx, rig, W, H)
    ref = decode()
    print("theta", theta, "ref unwritten", [int(unwritten(x).sum()) for x in ref], flush=True)
    ctx.set_option(capi.OPT_RECT_DECODE_ALGO, 0)
    for shape in (0, 1, 3):
        ctx.set_option(capi.OPT_RECT_DMA_SHAPE, shape)
        for flags, resident in ((0, 0), (0, 1), (0, 3), (32, 0)):
            ctx.set_option(capi.OPT_DEBUG_FLAGS, flags)
            ctx.set_option(capi.OPT_DEBUG_RECT_RESIDENT, resident)
            synth.install_verged_maps(ctx, rig, W, H)
            info = [ctx.rectify_info(c) for c in range(2)]
            print(" theta", theta, "shape", shape, "flags", flags, "resident", resident, [(i["dma_tiles"], i["dma_nofit_tiles"], i["dma_extra_entries"], list(i["lds_nofit_tiles"])) for i in info], flush=True)
            for rep in range(3):
                got = decode()
                for nm, a, b in zip(names, got, ref):
                    un = unwritten(a) & ~unwritten(b)
                    wrong = (a != b) if a.dtype != torch.float32 else (a.view(torch.int32) != b.view(torch.int32))
                    wrong = wrong & ~un
                    if int(un.sum()) or int(wrong.sum()):
                        bad = un | wrong
                        idx = bad.nonzero(); r, c = idx[:, 0], idx[:, 1]
                        TW, TH = SH[shape]
                        tiles = torch.unique((r // TH) * 1000 + (c // TW)).tolist()
                        print("  theta", theta, "shape", shape, "flags", flags, "res", resident, "rep", rep, nm, "unwritten", int(un.sum()), "wrong", int(wrong.sum()),
                              "tiles(ty*1000+tx)", tiles[:12], "rows", int(r.min()), int(r.max()), "cols", int(c.min()), int(c.max()),
                              "sample", [(int(r[j]), int(c[j]), a[r[j], c[j]].item(), b[r[j], c[j]].item()) for j in range(min(3, len(r)))], flush=True)
    ctx.set_option(capi.OPT_DEBUG_FLAGS, 0); ctx.set_option(capi.OPT_DEBUG_RECT_RESIDENT, 0)
print("done")
