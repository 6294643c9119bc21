"""debug helper: where does the LDS-DMA form differ from the oracle?  (run on the GPU box from the repo root)"""
import importlib, sys, os
import numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
slr = importlib.import_module("structure-light-reconstructor_amd")
synth = importlib.import_module("structure-light-reconstructor_amd.synth")
import oracle as O
O.build()
W, H = int(sys.argv[1]) if len(sys.argv) > 1 else 640, int(sys.argv[2]) if len(sys.argv) > 2 else 480
ctx = slr.Context(0)
st = synth.render_mf_stack(W, H, seed=W + H, noise=3)
cam = 0
mx, mf = synth.make_rectify_maps(W, H, cam, strength=1.0)
mxn, mfn = mx.numpy(), mf.numpy()
raw = st[cam].numpy()
rect = np.stack([O.remap_u8(raw[p], mxn, mfn) for p in range(14)])
eph, ev = O.mf_decode(rect, 40)
dev = st[cam].cuda()
for shape in range(7):
    for depth in (1,):
        ctx.set_option(slr.capi.OPT_RECT_DMA_SHAPE, shape)
        ctx.set_option(slr.capi.OPT_RECT_DMA_DEPTH, depth)
        ctx.set_option(slr.capi.OPT_RECT_DECODE_ALGO, 7)
        ctx.set_rectify_maps(cam, mxn, mfn)
        ph = torch.full((H, W), -7.0, device="cuda")
        v = torch.full((H, W), 9, dtype=torch.uint8, device="cuda")
        try:
            ctx.mf_decode(dev, 40, rectify_cam=cam, phase=ph, valid=v)
            ctx.synchronize()
        except Exception as ex:
            print("shape", shape, "depth", depth, "EXC", str(ex)[:300]); continue
        ph, v = ph.cpu().numpy(), v.cpu().numpy()
        bad = (ph.view(np.uint32) != eph.view(np.uint32)) | (v != ev)
        print("shape", shape, "depth", depth, "bad px", int(bad.sum()), "unwritten", int((v == 9).sum()))
        if bad.any():
            rows = np.flatnonzero(bad.any(axis=1)); cols = np.flatnonzero(bad.any(axis=0))
            print("  bad rows", rows[:24], "... n", len(rows)); print("  bad cols", cols[:12], "...", cols[-4:], "n", len(cols))
            r, c = np.argwhere(bad)[0]
            print("  first", r, c, "got", ph[r, c], v[r, c], "exp", eph[r, c], ev[r, c], "map", mxn[r, c], mfn[r, c])
            print("  row histogram mod 16:", np.bincount(rows % 16, minlength=16))
