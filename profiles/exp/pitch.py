"""does a power-of-two row pitch (4096 B = the image width) hurt the fused rectify+decode (channel / page aliasing)?"""
import importlib, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
slr = importlib.import_module("structure-light-reconstructor_amd"); synth = importlib.import_module("structure-light-reconstructor_amd.synth")
W, H = 4096, 3000; dev = torch.device("cuda", 0); ctx = slr.Context(0)
calib, _ = synth.make_calibration(W, H); ctx.set_calibration(calib)
maps = [synth.make_rectify_maps(W, H, cam, device=dev) for cam in range(2)]
for cam in range(2): ctx.set_rectify_maps(cam, maps[cam][0], maps[cam][1])
st = synth.render_mf_stack(W, H, seed=1234, device=dev)
ph = [torch.empty((H, W), dtype=torch.float32, device=dev) for _ in range(2)]
vd = [torch.empty((H, W), dtype=torch.uint8, device=dev) for _ in range(2)]
for pitch in (4096, 4096 + 64, 4096 + 256, 4096 + 1024 + 128):
    buf = torch.zeros((2, 14, H, pitch), dtype=torch.uint8, device=dev)
    buf[..., :W] = st
    torch.cuda.synchronize()
    def both():
        for cam in range(2):
            ctx.mf_decode(buf[cam], 40, W=W, rectify_cam=cam, phase=ph[cam], valid=vd[cam])
    for _ in range(3): both()
    ctx.profile_enable(True); ctx.profile_reset()
    for _ in range(10): both()
    prof = ctx.profile(); ctx.profile_enable(False)
    for name, (ms, n) in prof.items():
        print("pitch %5d  %-26s %7.1f us" % (pitch, name, ms / n * 1e3))
