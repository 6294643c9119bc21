"""shader clock under the fused decode's load: a -DSLR_DMA_CLOCKPROBE build writes workgroup 0's s_memtime and s_memrealtime
(100 MHz) spans into phase[0..3]; also the wall time of the launch"""
import importlib, os, sys, torch, numpy as np
sys.path.insert(0, os.getcwd())
slr = importlib.import_module("structure-light-reconstructor_amd"); synth = importlib.import_module("structure-light-reconstructor_amd.synth")
W, H = 4096, 3000; dev = torch.device("cuda", 0); ctx = slr.Context(0)
maps = [synth.make_rectify_maps(W, H, cam, device=dev) for cam in range(2)]
for cam in range(2): ctx.set_rectify_maps(cam, maps[cam][0], maps[cam][1])
sts = [synth.render_mf_stack(W, H, seed=1234 + i, device=dev) for i in range(4)]; torch.cuda.synchronize()
for it in range(6):
    st = sts[it % 4]
    ph, _ = ctx.mf_rectify_decode_pair(st[0], st[1], 40, want_valid=False); ctx.synchronize()
    o = ph[0].view(-1)[:4].cpu().numpy().view(np.uint64)
    print("iter %d: shader cycles %d  realtime ticks %d (100 MHz) -> %.1f us, %.0f MHz" % (it, o[0], o[1], o[1] / 100.0, o[0] / (o[1] / 100.0)))
