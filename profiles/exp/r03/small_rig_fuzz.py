"""Randomized companion of test_split_tiles_as_a_workgroups_only_entries: random SMALL image sizes (so that the resident set of the
persistent fused decodes exceeds the tile-table entries in most cases), random verged rigs, every compiled tile shape, random
resident-set override, outputs poisoned -- MF pair / Gray / hybrid fused decodes against the per-pixel gather form.
  python profiles/exp/r03/small_rig_fuzz.py <seed> <seconds>"""
import sys, os, importlib, time
os.environ["SLR_POISON_OUTPUTS"] = "1"
import numpy as np, torch
sys.path.insert(0, os.getcwd())
slr = importlib.import_module("structure-light-reconstructor_amd")
synth = importlib.import_module("structure-light-reconstructor_amd.synth")
capi = slr.capi
BLACK = 40
ctx = slr.Context(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
T_END = time.time() + (float(sys.argv[2]) if len(sys.argv) > 2 else 60.0)
n = split = 0
while time.time() < T_END:
    W = 16 * int(rng.integers(4, 110)); H = int(rng.integers(9, 700))
    theta = float(rng.uniform(0.02, 0.4)); k1 = float(rng.uniform(-0.25, 0.12))
    st = synth.render_mf_stack(W, H, seed=int(rng.integers(1, 1000)), noise=3, device="cuda")
    g = synth.render_gray_stack(W, H, 1024, seed=9, noise=2, device="cuda")
    hy = synth.render_hybrid_stack(W, H, 1024, seed=5, noise=2, device="cuda")
    ncol = synth.gray_num_bits(1024)
    ctx.set_calibration(synth.make_calibration(W, H)[0])
    rig = synth.make_verged_rig(W, H, theta, k1)
    def decode():
        ph, vd = ctx.mf_rectify_decode_pair(st[0], st[1], BLACK, want_valid=True)
        ctx.synchronize()
        outs = [ph[0].clone(), ph[1].clone(), vd[0].clone(), vd[1].clone()]
        ph2, _ = ctx.mf_rectify_decode_pair(st[0], st[1], BLACK, want_valid=False)      # the shipped form: valid folded into the phase
        ctx.synchronize()
        outs += [ph2[0].clone(), ph2[1].clone()]
        for cam in range(2):
            cx, _, v = ctx.gray_decode(g[cam], ncol, 0, BLACK, 3, 1024, 0, rectify_cam=cam)
            ctx.synchronize()
            outs += [cx.clone(), v.clone()]
        for one_pass in (0, 1):
            ctx.set_option(capi.OPT_HYBRID_ONE_PASS, one_pass)
            hx, hp = ctx.hybrid_rectify_decode_pair(hy[0], hy[1], ncol, BLACK, 3, 1024)
            ctx.synchronize()
            outs += [hx[0].clone(), hx[1].clone(), hp[0].clone(), hp[1].clone()]
        ctx.set_option(capi.OPT_HYBRID_ONE_PASS, 0)
        return outs
    ctx.set_option(capi.OPT_DEBUG_RECT_RESIDENT, 0)
    ctx.set_option(capi.OPT_RECT_DECODE_ALGO, 1)
    synth.install_verged_maps(ctx, rig, W, H)
    ref = decode()
    ctx.set_option(capi.OPT_RECT_DECODE_ALGO, 0)
    for shape in (0, 1, 3):
        ctx.set_option(capi.OPT_RECT_DMA_SHAPE, shape)
        res = int(rng.choice([0, 0, 0, 2, 24, 200]))
        ctx.set_option(capi.OPT_DEBUG_RECT_RESIDENT, res)
        synth.install_verged_maps(ctx, rig, W, H)
        info = [ctx.rectify_info(c) for c in range(2)]
        split += sum(i["dma_extra_entries"] for i in info)
        for rep in range(2):
            for k, (a, b) in enumerate(zip(decode(), ref)):
                same = a.view(torch.uint8) == b.view(torch.uint8)
                assert not bool((a.view(torch.uint8) == 0x7B).all()), ("nothing written", k)
                assert bool(same.all()), (W, H, theta, k1, shape, res, rep, k, int((~same).sum()), [i["mf_form"] for i in info])
        n += 1
    ctx.set_option(capi.OPT_RECT_DMA_SHAPE, 3)
print("small rig fuzz ok:", n, "configurations,", split, "extra entries in all")
