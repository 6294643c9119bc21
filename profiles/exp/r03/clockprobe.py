"""per-workgroup timeline of the fused decode's pair launch: a -DSLR_DMA_CLOCKPROBE build writes, per workgroup, its entry and
exit on the 100 MHz constant clock, its shader cycles and (tiles done, XCC id, band) into the left phase buffer"""
import importlib, os, sys, torch, numpy as np
sys.path.insert(0, os.getcwd())
slr = importlib.import_module("structure-light-reconstructor_amd"); synth = importlib.import_module("structure-light-reconstructor_amd.synth")
W, H = 4096, 3000; dev = torch.device("cuda", 0); ctx = slr.Context(0)
maps = [synth.make_rectify_maps(W, H, cam, device=dev) for cam in range(2)]
for cam in range(2): ctx.set_rectify_maps(cam, maps[cam][0], maps[cam][1])
sts = [synth.render_mf_stack(W, H, seed=1234 + i, device=dev) for i in range(4)]; torch.cuda.synchronize()
NB = int(os.environ.get("NBLK", "768"))
for it in range(6):
    st = sts[it % 4]
    ph, _ = ctx.mf_rectify_decode_pair(st[0], st[1], 40, want_valid=False); ctx.synchronize()
    o = ph[0].view(-1)[:8 * NB].cpu().numpy().view(np.uint64).reshape(NB, 4)
    ok = o[:, 1] > o[:, 0]
    o = o[ok]
    t0 = o[:, 0].min()
    st_, en = (o[:, 0] - t0) / 100.0, (o[:, 1] - t0) / 100.0
    life = en - st_
    tiles = (o[:, 3] >> np.uint64(32)).astype(int); job = ((o[:, 3] >> np.uint64(16)) & np.uint64(255)).astype(int); xcc = ((o[:, 3] >> np.uint64(8)) & np.uint64(15)).astype(int); band = (o[:, 3] & np.uint64(255)).astype(int)
    mhz = o[:, 2] / np.maximum(life, 1e-3)
    print("iter %d: %d wgs; start p50 %.1f p90 %.1f max %.1f us | end min %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f us | life p50 %.1f min %.1f max %.1f | clk %.0f MHz"
          % (it, len(o), np.percentile(st_, 50), np.percentile(st_, 90), st_.max(), en.min(), np.percentile(en, 10), np.percentile(en, 50),
             np.percentile(en, 90), en.max(), np.percentile(life, 50), life.min(), life.max(), np.median(mhz)))
    if it == 5:
        for x in range(8):
            m = xcc == x
            if m.any():
                print("  xcc %d: %d wgs bands %s start %.1f..%.1f end %.1f..%.1f (p50 %.1f) tiles %d..%d" % (x, m.sum(), sorted(set(band[m].tolist())), st_[m].min(), st_[m].max(), en[m].min(), en[m].max(), np.median(en[m]), tiles[m].min(), tiles[m].max()))
        for j in range(2):
            m = job == j
            if m.any():
                print("  camera %d: %d wgs, end p50 %.1f max %.1f, tiles/wg p50 %.0f" % (j, m.sum(), np.median(en[m]), en[m].max(), np.median(tiles[m])))
        third = len(o) // 3
        for k in range(3):
            sl = slice(k * third, (k + 1) * third)
            print("  blocks %d..%d (launch order): life p50 %.1f, tiles p50 %.0f, us/tile %.2f" % (k * third, (k + 1) * third - 1, np.median(life[sl]), np.median(tiles[sl]), np.median(life[sl] / np.maximum(tiles[sl], 1))))
        for tcount in sorted(set(tiles.tolist())):
            m = tiles == tcount
            print("  wgs with %d tiles: %d, life p50 %.1f, end p50 %.1f max %.1f" % (tcount, m.sum(), np.median(life[m]), np.median(en[m]), en[m].max()))
