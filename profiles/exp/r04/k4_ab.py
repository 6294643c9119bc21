#!/usr/bin/env python
"""Same-process A/B of the K4 shapes (SLR_OPT_MF_MATCH_ALGO 4 = 1024 x 4, 5 = 512 x 8 two rows per CU, 6 = 512 x 8 three rows per CU,
0 = auto) on HBM-resident phases of the bench's scene: K4 alone (stream timer over --frames calls) and inside the batch entry (the
library's per-kernel profiler).   python profiles/exp/r04/k4_ab.py [--algos 4,5,6] [--reps 5]"""
import argparse, importlib, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
os.environ.pop("SLR_POISON_OUTPUTS", None); os.environ.pop("SLR_POISON_SCRATCH", None)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--algos", default="4,5,6")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--batch", type=int, default=1)
    args = ap.parse_args()
    import torch
    slr = importlib.import_module("structure-light-reconstructor_amd")
    synth = importlib.import_module("structure-light-reconstructor_amd.synth")
    capi = slr.capi
    W, H, F = 4096, 3000, args.frames
    dev = torch.device("cuda", 0)
    stack = torch.stack([synth.render_mf_stack(W, H, seed=1234 + f, noise=2, device=dev) for f in range(F)])
    rig = synth.make_verged_rig(W, H, 0.2, -0.15)
    ctx = slr.Context(0)
    ctx.set_calibration(rig["calib"])
    synth.install_verged_maps(ctx, rig, W, H)
    phases = []
    for f in range(F):
        ph = [torch.empty((H, W), dtype=torch.float32, device=dev) for _ in range(2)]
        ctx.mf_rectify_decode_pair(stack[f, 0], stack[f, 1], 40, W=W, want_valid=False, phase=ph)
        phases.append(ph)
    ctx.synchronize()
    xyz = torch.empty((F, H, W, 3), dtype=torch.float32, device=dev)
    has = torch.empty((F, H, W), dtype=torch.uint8, device=dev)
    algos = [int(a) for a in args.algos.split(",")]
    ones = torch.ones((H, W), dtype=torch.uint8, device=dev)   # the public entry wants valid bytes (the batch entry folds them into NaNs)
    ref = None
    res = {a: [] for a in algos}
    resb = {a: [] for a in algos}
    for rep in range(args.reps):
        for a in algos:
            ctx.set_option(capi.OPT_MF_MATCH_ALGO, a)
            for f in range(2):
                ctx.mf_triangulate(phases[f][0], ones, phases[f][1], ones, want_match=False, xyz=xyz[f], has=has[f])
            ctx.synchronize()
            ctx.timer_begin()
            for f in range(F):
                ctx.mf_triangulate(phases[f][0], ones, phases[f][1], ones, want_match=False, xyz=xyz[f], has=has[f])
            us = ctx.timer_end() / F * 1e3
            res[a].append(us)
            if rep == 0:
                cs = (xyz.view(torch.int32).to(torch.int64).sum().item(), has.to(torch.int64).sum().item())
                if ref is None:
                    ref = cs
                print("algo %d checksum %s %s" % (a, cs, "== first" if cs == ref else "DIFFERENT"), flush=True)
            line = "rep %d algo %d  K4 alone %.1f us" % (rep, a, us)
            if args.batch:
                ctx.reconstruct_mf_batch(stack, 40, True, W=W, xyz=xyz, has=has)
                ctx.synchronize()
                ctx.set_option(capi.OPT_PROFILE_STRIDE, 1)
                ctx.profile_enable(True); ctx.profile_reset()
                ctx.timer_begin()
                for _ in range(3):
                    ctx.reconstruct_mf_batch(stack, 40, True, W=W, xyz=xyz, has=has)
                ms = ctx.timer_end() / (3 * F)
                prof = ctx.profile(); ctx.profile_enable(False)
                k4 = prof.get("slr_mf_match_triangulate")
                resb[a].append((ms, k4[0] / k4[1] * 1e3))
                line += "   batch %.4f ms/frame, K4 inside %.1f us" % (ms, k4[0] / k4[1] * 1e3)
            print(line, flush=True)
    ctx.set_option(capi.OPT_MF_MATCH_ALGO, 0)
    print("---- medians")
    for a in algos:
        print("algo %d: K4 alone %.1f us" % (a, statistics.median(res[a])) +
              ("   batch %.4f ms/frame, K4 inside %.1f us" % (statistics.median(x[0] for x in resb[a]), statistics.median(x[1] for x in resb[a])) if resb[a] else ""))


if __name__ == "__main__":
    main()
