"""K4 inside slr_reconstruct_mf_batch with SLR_OPT_MF_BATCH_GROUP = $K4_GROUP (default 8): one batch of 8 frames, for rocprofv3 --pmc"""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
os.environ.pop("SLR_POISON_OUTPUTS", None); os.environ.pop("SLR_POISON_SCRATCH", None)
slr = importlib.import_module("structure-light-reconstructor_amd"); synth = importlib.import_module("structure-light-reconstructor_amd.synth")
W, H, F = 4096, 3000, 8
dev = torch.device("cuda", 0)
stack = torch.stack([synth.render_mf_stack(W, H, seed=1234 + f, noise=2, device=dev) for f in range(F)])
rig = synth.make_verged_rig(W, H, 0.2, -0.15)
ctx = slr.Context(0); ctx.set_calibration(rig["calib"]); synth.install_verged_maps(ctx, rig, W, H)
ctx.set_option(slr.capi.OPT_MF_BATCH_GROUP, int(os.environ.get("K4_GROUP", "8")))
xyz = torch.empty((F, H, W, 3), dtype=torch.float32, device=dev); has = torch.empty((F, H, W), dtype=torch.uint8, device=dev)
for _ in range(3):
    ctx.reconstruct_mf_batch(stack, 40, True, W=W, xyz=xyz, has=has)
ctx.synchronize()
