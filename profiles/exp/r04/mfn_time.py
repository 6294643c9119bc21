#!/usr/bin/env python
"""times slr_mfn_rectify_decode (and the unrectified slr_mfn_decode beside it) at BASELINE config 5's size for library variants
loaded in ONE process; the variants' outputs must be bit-identical.  python profiles/exp/r04/mfn_time.py name=path,..."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import torch
slr = importlib.import_module("structure-light-reconstructor_amd")
synth = importlib.import_module("structure-light-reconstructor_amd.synth")
capi = slr.capi
base = capi.LIB_PATH
variants = [("base", base)]
for v in (sys.argv[1].split(",") if len(sys.argv) > 1 and sys.argv[1] else []):
    n, _, p = v.partition("=")
    variants.append((n, os.path.join(ROOT, p)))
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (8192, 6000)
dev = torch.device("cuda", 0)
st = synth.render_mfn_stack(W, H, 4, 8, seed=1234, noise=0.5, device=dev)
rig = synth.make_verged_rig(W, H, 0.2, -0.15)
ph = torch.empty((H, W), dtype=torch.float32, device=dev); vd = torch.empty((H, W), dtype=torch.uint8, device=dev)
ref = None
for name, path in variants:
    capi._lib = None; capi.LIB_PATH = path
    capi.load_library()
    ctx = slr.Context(0)
    ctx.set_calibration(rig["calib"]); synth.install_verged_maps(ctx, rig, W, H)
    forms = [int(x) for x in os.environ.get("MFN_FORMS", "0,16,2,1").split(",")]
    for rep in range(int(os.environ.get("MFN_REPS", "2")) * len(forms)):
        fl = forms[rep % len(forms)]                           # 0 / 16: LDS-DMA ring of 3 / of 4, 2: register-staged tiles, 1: per-pixel gather
        ctx.set_option(capi.OPT_DEBUG_FLAGS, fl)
        line = [{0: "dma3/256x4", 16: "dma4/256x4", 8: "dma3/512x2", 24: "dma4/512x2", 2: "tiled ", 1: "gather"}[fl]]
        for cam in range(2):
            ctx.mfn_rectify_decode(cam, st[cam], 4, 8, 40.0, phase=ph, valid=vd)
            ctx.synchronize(); ctx.timer_begin()
            for _ in range(3):
                ctx.mfn_rectify_decode(cam, st[cam], 4, 8, 40.0, phase=ph, valid=vd)
            us = ctx.timer_end() / 3 * 1e3
            line.append("cam%d %.0f us (%.3f of 8 TB/s at 79 B/px)" % (cam, us, 79.0 * W * H / (us * 1e-6) / 8e12))
        print(name, "rep", rep, "  ".join(line), flush=True)
    ctx.set_option(capi.OPT_DEBUG_FLAGS, 0)
    cur = (ph.clone(), vd.clone())
    if ref is None:
        ref = cur
    else:
        print(name, "identical to base:", bool(torch.equal(ref[0].view(torch.int32), cur[0].view(torch.int32)) and torch.equal(ref[1], cur[1])))
    ctx.mfn_decode(st[1], 4, 8, 40.0, phase=ph, valid=vd); ctx.synchronize(); ctx.timer_begin()
    for _ in range(3):
        ctx.mfn_decode(st[1], 4, 8, 40.0, phase=ph, valid=vd)
    print(name, "unrectified decode %.0f us" % (ctx.timer_end() / 3 * 1e3))
    ctx.close()
