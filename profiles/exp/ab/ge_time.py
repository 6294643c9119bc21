"""times slr_reconstruct_ge (GRAY_EPI: 2 x fused Gray decode + K5) on device-resident 4096x3000 stacks"""
import importlib, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))
import torch
slr = importlib.import_module("structure-light-reconstructor_amd")
synth = importlib.import_module("structure-light-reconstructor_amd.synth")
W, H = 4096, 3000
ctx = slr.Context(0)
calib, _ = synth.make_calibration(W, H)
ctx.set_calibration(calib)
for cam in range(2):
    mx, mf = synth.make_rectify_maps(W, H, cam)
    ctx.set_rectify_maps(cam, mx.numpy(), mf.numpy())
st = synth.render_gray_stack(W, H, W, seed=5, noise=2, device="cuda")
ncol = synth.gray_num_bits(W)
for it in range(3):
    ctx.reconstruct_ge(st[0], st[1], ncol, 40, 4, W, True, False)
ctx.synchronize()
t0 = time.perf_counter()
N = 20
for it in range(N):
    ctx.reconstruct_ge(st[0], st[1], ncol, 40, 4, W, True, False)
ctx.synchronize()
print("reconstruct_ge %.1f us per frame" % ((time.perf_counter() - t0) / N * 1e6))
