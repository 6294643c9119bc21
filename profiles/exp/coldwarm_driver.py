"""warm (same camera stack, resident in the Infinity Cache) vs cold (alternating cameras) launches of the fused kernel"""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
slr = importlib.import_module("structure-light-reconstructor_amd")
synth = importlib.import_module("structure-light-reconstructor_amd.synth")
W, H = 4096, 3000
dev = torch.device("cuda", 0)
ctx = slr.Context(0)
calib, _ = synth.make_calibration(W, H)
ctx.set_calibration(calib)
maps = [synth.make_rectify_maps(W, H, cam, device=dev) for cam in range(2)]
torch.cuda.synchronize()
for cam in range(2):
    ctx.set_rectify_maps(cam, maps[cam][0], maps[cam][1])
st = synth.render_mf_stack(W, H, seed=1234, device=dev)
torch.cuda.synchronize()
ctx.set_option(slr.capi.OPT_RECT_DECODE_ALGO, int(os.environ.get('SLR_RECT_ALGO', '0')))
ph = [torch.empty((H, W), dtype=torch.float32, device=dev) for _ in range(2)]
vd = [torch.empty((H, W), dtype=torch.uint8, device=dev) for _ in range(2)]
for _ in range(8):                       # dispatches 0..7: warm
    ctx.mf_decode(st[0], 40, rectify_cam=0, phase=ph[0], valid=vd[0])
ctx.synchronize()
for _ in range(4):                       # dispatches 8..15: cold
    for cam in range(2):
        ctx.mf_decode(st[cam], 40, rectify_cam=cam, phase=ph[cam], valid=vd[cam])
ctx.synchronize()
ctx.close()
