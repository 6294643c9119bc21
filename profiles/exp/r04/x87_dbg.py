import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["SLR_POISON_OUTPUTS"] = "1"; os.environ["SLR_POISON_SCRATCH"] = "1"
slr = importlib.import_module("structure-light-reconstructor_amd")
synth = importlib.import_module("structure-light-reconstructor_amd.synth")
import oracle as O
W, H = 4096, 3000
tab = O.atan_table(1)
dev = torch.device("cuda", 0)
rig = synth.make_verged_rig(W, H, 0.2, -0.15)
st = synth.render_mf_stack(W, H, seed=1234, noise=2, device=dev)
ref = None
for trial in range(4):
    c = slr.Context(0)
    if trial >= 2:                       # first use the context in strict mode with the DMA forms, then switch
        c.set_calibration(rig["calib"]); synth.install_verged_maps(c, rig, W, H)
        c.mf_decode(st[0], 40, rectify_cam=0)
    c.set_option(slr.capi.OPT_EVAL_MODEL, 1)
    c.set_calibration(rig["calib"]); synth.install_verged_maps(c, rig, W, H)
    p, v = c.mf_decode(st[0], 40, rectify_cam=0)
    p = p.cpu().numpy(); v = v.cpu().numpy()
    if ref is None:
        mx, mf = c.get_rectify_maps(0, W, H)
        pl = np.stack([O.remap_u8(st[0, q].cpu().numpy(), mx, mf) for q in range(14)])
        ref = O.mf_decode_ev(pl, 40, tab, 1)
        strict = O.mf_decode(pl, 40)
    bad = p.view(np.int32) != ref[0].view(np.int32)
    print("trial", trial, "valid equal", np.array_equal(v, ref[1]), "phase mismatches", int(bad.sum()), "vs strict", int((p.view(np.int32) != strict[0].view(np.int32)).sum()))
    if bad.any():
        ys, xs = np.nonzero(bad)
        print("  first", [(int(y), int(x), float(p[y, x]), float(ref[0][y, x]), int(v[y, x])) for y, x in list(zip(ys, xs))[:6]], "rows", int(ys.min()), int(ys.max()), "cols", int(xs.min()), int(xs.max()))
    c.close()
