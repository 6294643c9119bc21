"""how much do the fused decode (VALU-heavy) and K4 (latency/LDS-heavy) gain from running concurrently on two streams?"""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
slr = importlib.import_module("structure-light-reconstructor_amd")
synth = importlib.import_module("structure-light-reconstructor_amd.synth")
W, H = 4096, 3000
dev = torch.device("cuda", 0)
sA, sB = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
cA, cB = slr.Context(0, stream=sA), slr.Context(0, stream=sB)
calib, _ = synth.make_calibration(W, H)
maps = [synth.make_rectify_maps(W, H, cam, device=dev) for cam in range(2)]
torch.cuda.synchronize()
for c in (cA, cB):
    c.set_calibration(calib)
    for cam in range(2):
        c.set_rectify_maps(cam, maps[cam][0], maps[cam][1])
st = synth.render_mf_stack(W, H, seed=1234, device=dev).unsqueeze(0).contiguous()
torch.cuda.synchronize()
ph = [torch.empty((H, W), dtype=torch.float32, device=dev) for _ in range(2)]
vd = [torch.empty((H, W), dtype=torch.uint8, device=dev) for _ in range(2)]
for cam in range(2):
    cA.mf_decode(st[0, cam], 40, rectify_cam=cam, phase=ph[cam], valid=vd[cam])
torch.cuda.synchronize()
xyzA = torch.empty((1, H, W, 3), dtype=torch.float32, device=dev); hasA = torch.empty((1, H, W), dtype=torch.uint8, device=dev)
N = 20
def dec(): cA.reconstruct_mf_batch(st, 40, True, xyz=xyzA, has=hasA)       # decode pair + K4 on stream A (full frame)
def k4(): cB.mf_triangulate(ph[0], vd[0], ph[1], vd[1], want_match=False)   # K4 only on stream B
for f in (dec, k4):
    for _ in range(3): f()
torch.cuda.synchronize()
def timed(fs):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(N):
        for f in fs: f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / N * 1e3
print("full frame on A alone        %.3f ms" % timed([dec]))
print("K4 on B alone                %.3f ms" % timed([k4]))
print("full frame on A + K4 on B    %.3f ms (sum of the two alone above if no overlap)" % timed([dec, k4]))
xyzB = torch.empty((1, H, W, 3), dtype=torch.float32, device=dev); hasB = torch.empty((1, H, W), dtype=torch.uint8, device=dev)
def decB(): cB.reconstruct_mf_batch(st, 40, True, xyz=xyzB, has=hasB)
for _ in range(3): decB()
print("full frames alternating A,B  %.3f ms per frame" % (timed([dec, decB]) / 2))
