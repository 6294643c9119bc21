"""Whole-path companion of small_rig_fuzz.py: the BATCH entries (slr_reconstruct_mf_batch, slr_reconstruct_batch in GE and GRAY
modes -- GRAY with and without rectification --, slr_reconstruct_hybrid_batch) on 2-3 distinct small frames, verged rigs, random tile shape / resident set, outputs AND
scratch poisoned -- against the same entry on the per-pixel gather forms (SLR_OPT_RECT_DECODE_ALGO = 1), frame by frame.
  python profiles/exp/r03/small_batch_fuzz.py <seed> <seconds>"""
import sys, os, importlib, time
os.environ["SLR_POISON_OUTPUTS"] = "1"; os.environ["SLR_POISON_SCRATCH"] = "1"
import numpy as np, torch
sys.path.insert(0, os.getcwd())
slr = importlib.import_module("structure-light-reconstructor_amd")
synth = importlib.import_module("structure-light-reconstructor_amd.synth")
capi = slr.capi
BLACK = 40
ctx = slr.Context(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
T_END = time.time() + (float(sys.argv[2]) if len(sys.argv) > 2 else 60.0)
n = 0
while time.time() < T_END:
    W = 16 * int(rng.integers(4, 90)); H = int(rng.integers(9, 500)); nf = int(rng.integers(2, 4))
    theta = float(rng.uniform(0.02, 0.38)); k1 = float(rng.uniform(-0.25, 0.12))
    ncol = synth.gray_num_bits(1024)
    mf = torch.stack([synth.render_mf_stack(W, H, seed=int(rng.integers(1, 9999)), noise=3, device="cuda") for _ in range(nf)]).contiguous()
    gr = torch.stack([synth.render_gray_stack(W, H, 1024, seed=int(rng.integers(1, 9999)), noise=2, device="cuda") for _ in range(nf)]).contiguous()
    hy = torch.stack([synth.render_hybrid_stack(W, H, 1024, seed=int(rng.integers(1, 9999)), noise=2, device="cuda") for _ in range(nf)]).contiguous()
    sw, sh = 256, 128
    gc, gw = synth.gray_num_bits(sw), synth.gray_num_bits(sh)
    go = torch.stack([synth.render_gray_stack(W, H, sw, sh, seed=int(rng.integers(1, 9999)), noise=2, device="cuda", rows=True) for _ in range(nf)]).contiguous()
    ctx.set_calibration(synth.make_calibration(W, H, with_T=bool(rng.integers(0, 2)))[0])
    rig = synth.make_verged_rig(W, H, theta, k1)
    def run():
        outs = []
        x, h = ctx.reconstruct_mf_batch(mf, BLACK, True); ctx.synchronize(); outs += [x.clone(), h.clone()]
        x, h, c = ctx.reconstruct_batch(capi.MODE_GE, gr, BLACK, 3, ncol, 0, 1024, 0, rectify=True, have_color=True); ctx.synchronize()
        outs += [x.clone(), h.clone(), c.clone()]
        x, h, cx = ctx.reconstruct_hybrid_batch(hy, ncol, BLACK, 3, 1024, want_codes=True); ctx.synchronize()
        outs += [x.clone(), h.clone(), cx.clone()]
        for rect in (True, False):
            x, h, _ = ctx.reconstruct_batch(capi.MODE_GRAY, go, BLACK, 0, n_col_bits=gc, n_row_bits=gw, scan_w=sw, scan_h=sh, rectify=rect); ctx.synchronize()
            outs += [x.clone(), h.clone()]
        return outs
    ctx.set_option(capi.OPT_DEBUG_RECT_RESIDENT, 0)
    ctx.set_option(capi.OPT_RECT_DECODE_ALGO, 1)
    synth.install_verged_maps(ctx, rig, W, H)
    ref = run()
    ctx.set_option(capi.OPT_RECT_DECODE_ALGO, 0)
    for shape in (0, 1, 3):
        ctx.set_option(capi.OPT_RECT_DMA_SHAPE, shape)
        res = int(rng.choice([0, 0, 0, 2, 24, 200]))
        ctx.set_option(capi.OPT_DEBUG_RECT_RESIDENT, res)
        synth.install_verged_maps(ctx, rig, W, H)
        for rep in range(2):
            for k, (a, b) in enumerate(zip(run(), ref)):
                same = a.view(torch.uint8) == b.view(torch.uint8)
                assert not bool((a.view(torch.uint8) == 0x7B).all()), ("nothing written", k)
                assert bool(same.all()), (W, H, nf, theta, k1, shape, res, rep, k, int((~same).sum()))
        n += 1
    ctx.set_option(capi.OPT_RECT_DMA_SHAPE, 3)
print("small batch fuzz ok:", n, "configurations")
