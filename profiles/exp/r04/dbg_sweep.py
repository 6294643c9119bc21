import importlib, os, sys, torch
sys.path.insert(0, "/root/repo")
os.environ.pop("SLR_POISON_OUTPUTS", None); os.environ.pop("SLR_POISON_SCRATCH", None)
slr = importlib.import_module("structure-light-reconstructor_amd"); synth = importlib.import_module("structure-light-reconstructor_amd.synth")
capi = slr.capi
W, H, F = 4096, 3000, 8
dev = torch.device("cuda", 0)
stack = torch.stack([synth.render_mf_stack(W, H, seed=1234 + f, noise=2, device=dev) for f in range(F)])
xyz = torch.empty((F, H, W, 3), dtype=torch.float32, device=dev); has = torch.empty((F, H, W), dtype=torch.uint8, device=dev)
ctx = slr.Context(0)
rig = synth.make_verged_rig(W, H, 0.2, -0.15); ctx.set_calibration(rig["calib"]); synth.install_verged_maps(ctx, rig, W, H)
def batch_us(tag):
    ctx.set_option(capi.OPT_PROFILE_STRIDE, 1)
    ctx.profile_enable(True); ctx.profile_reset()
    for _ in range(3):
        ctx.reconstruct_mf_batch(stack, 40, True, W=W, xyz=xyz, has=has)
    p = ctx.profile(); ctx.profile_enable(False)
    d = p["slr_mf_rectify_decode_pair"]
    print(tag, "decode %.1f us/frame (n=%d)" % (d[0] / d[1] * 1e3, d[1]), flush=True)
ctx.reconstruct_mf_batch(stack, 40, True, W=W, xyz=xyz, has=has); ctx.synchronize()
batch_us("verged, fresh")
for cam in range(2):
    mx, mf = synth.make_rectify_maps(W, H, cam, device=dev); ctx.set_rectify_maps(cam, mx, mf)
ctx.reconstruct_mf_batch(stack, 40, True, W=W, xyz=xyz, has=has); ctx.synchronize()
batch_us("near-identity after map change")
ph = [torch.empty((H, W), dtype=torch.float32, device=dev) for _ in range(2)]
for f in range(F):
    ctx.mf_rectify_decode_pair(stack[f, 0], stack[f, 1], 40, W=W, want_valid=False, phase=ph)
ctx.synchronize()
batch_us("near-identity after single-frame pair calls")
ctx.reconstruct_mf_batch(stack, 40, True, W=W, xyz=xyz, has=has); ctx.synchronize()
batch_us("near-identity after one more warm batch")
info = [ctx.rectify_info(cam) for cam in range(2)]
batch_us("after rectify_info")
