#!/usr/bin/env python
"""Same-box, same-PROCESS A/B of library builds (a fresh python process costs minutes on a cold GPU box: the image pages in
over the network, so round 3's one-process-per-variant scripts no longer fit a gpurun call).

  python profiles/exp/r04/ab_inproc.py --variants base,abl1=profiles/exp/ab/so/var_abl1.so --maps near-identity,verged \
         --reps 3 [--flags 0,32] [--shapes -1,0,1] [--k4 1] [--frames 8]

Every variant is dlopen'ed once (ctypes handles of different files are independent); per (rep, variant, map, flags, shape) the
fused pair decode is timed by stream events over --frames back-to-back calls on distinct HBM-resident frames (as bench.py's map
sweep does), and with --k4 1 the whole batch entry with the library's per-kernel profiler.  Prints one line per measurement and
a median table at the end.
"""
import argparse
import time
import ctypes as C
import importlib
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
os.environ.pop("SLR_POISON_OUTPUTS", None)
os.environ.pop("SLR_POISON_SCRATCH", None)


_T0 = time.perf_counter()


def trace(what):
    if os.environ.get("SLR_AB_TRACE"):
        print("[ab %7.1f s] %s" % (time.perf_counter() - _T0, what), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", default="base")
    ap.add_argument("--maps", default="near-identity,verged")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--flags", default="0")
    ap.add_argument("--shapes", default="-1")
    ap.add_argument("--depths", default="-1")
    ap.add_argument("--k4", type=int, default=0)
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--width", type=int, default=4096)
    ap.add_argument("--height", type=int, default=3000)
    ap.add_argument("--mode", default="mf", choices=["mf", "ge"])
    args = ap.parse_args()
    import torch
    trace("torch imported")
    slr = importlib.import_module("structure-light-reconstructor_amd")
    synth = importlib.import_module("structure-light-reconstructor_amd.synth")
    capi = slr.capi
    base_path = capi.LIB_PATH
    W, H, F = args.width, args.height, args.frames
    dev = torch.device("cuda", 0)
    variants = []
    for v in args.variants.split(","):
        name, _, path = v.partition("=")
        variants.append((name, os.path.join(ROOT, path) if path else base_path))
    libs = {}
    for name, path in variants:
        capi._lib = None
        capi.LIB_PATH = path
        libs[name] = capi.load_library()
    trace("libraries loaded")
    scan_w = W
    ncol = synth.gray_num_bits(scan_w)
    if args.mode == "mf":
        stack = torch.stack([synth.render_mf_stack(W, H, seed=1234 + f, noise=2, device=dev) for f in range(F)])
    else:
        stack = torch.stack([synth.render_gray_stack(W, H, scan_w, seed=1234 + f, noise=2, device=dev) for f in range(F)])
    torch.cuda.synchronize()
    trace("frames rendered")
    rigs = {}
    for m in args.maps.split(","):
        if m.startswith("verged"):
            parts = m.split(":")
            rigs[m] = synth.make_verged_rig(W, H, float(parts[1]) if len(parts) > 1 else 0.2, float(parts[2]) if len(parts) > 2 else -0.15)
        else:
            rigs[m] = None
    trace("rigs made")
    near = [synth.make_rectify_maps(W, H, cam, device=dev) for cam in range(2)]
    calib0, _ = synth.make_calibration(W, H)
    ph = [torch.empty((H, W), dtype=torch.float32, device=dev) for _ in range(2)]
    xyz = torch.empty((F, H, W, 3), dtype=torch.float32, device=dev)
    has = torch.empty((F, H, W), dtype=torch.uint8, device=dev)
    results = {}
    for rep in range(args.reps):
        for name, _ in variants:
            capi._lib = libs[name]
            ctx = slr.Context(0)
            trace("context of %s" % name)
            for m in args.maps.split(","):
                rig = rigs[m]
                ctx.set_calibration(rig["calib"] if rig else calib0)
                for shape in [int(x) for x in args.shapes.split(",")]:
                    if shape >= 0:
                        ctx.set_option(capi.OPT_RECT_DMA_SHAPE, shape)
                    for depth in [int(x) for x in args.depths.split(",")]:
                        if depth >= 0:
                            ctx.set_option(capi.OPT_RECT_DMA_DEPTH, depth)
                        for fl in [int(x) for x in args.flags.split(",")]:
                            ctx.set_option(capi.OPT_DEBUG_FLAGS, fl)
                            if rig:
                                synth.install_verged_maps(ctx, rig, W, H)
                            else:
                                for cam in range(2):
                                    ctx.set_rectify_maps(cam, near[cam][0], near[cam][1])
                            trace("maps installed")
                            key = (name, m, shape, depth, fl)
                            if args.mode == "mf":
                                for f in range(2):
                                    ctx.mf_rectify_decode_pair(stack[f % F, 0], stack[f % F, 1], 40, W=W, want_valid=False, phase=ph)
                                ctx.synchronize()
                                ctx.timer_begin()
                                for f in range(F):
                                    ctx.mf_rectify_decode_pair(stack[f, 0], stack[f, 1], 40, W=W, want_valid=False, phase=ph)
                                us = ctx.timer_end() / F * 1e3
                                line = "decode_pair %.1f us" % us
                                results.setdefault(key, {}).setdefault("decode", []).append(us)
                            else:
                                line = ""
                            if args.k4 or args.mode == "ge":
                                def batch():
                                    if args.mode == "mf":
                                        ctx.reconstruct_mf_batch(stack, 40, True, W=W, xyz=xyz, has=has)
                                    else:
                                        ctx.reconstruct_batch(capi.MODE_GE, stack, 40, 0, n_col_bits=ncol, scan_w=scan_w, rectify=True, W=W, xyz=xyz, has=has)
                                batch()
                                ctx.synchronize()
                                ctx.set_option(capi.OPT_PROFILE_STRIDE, 1)
                                ctx.profile_enable(True)
                                ctx.profile_reset()
                                ctx.timer_begin()
                                for _ in range(3):
                                    batch()
                                ms = ctx.timer_end() / (3 * F)
                                prof = ctx.profile()
                                ctx.profile_enable(False)
                                line += "  batch %.4f ms/frame  " % ms + "  ".join("%s %.1f" % (k.replace("slr_", ""), v[0] / v[1] * 1e3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0]))
                                results.setdefault(key, {}).setdefault("batch_ms", []).append(ms)
                                for k, v in prof.items():
                                    results[key].setdefault(k, []).append(v[0] / v[1] * 1e3)
                            print("rep %d %-10s maps=%-18s shape=%d depth=%d flags=%d : %s" % (rep, name, m, shape, depth, fl, line), flush=True)
            ctx.close()
    print("---- medians")
    for key in sorted(results):
        print("%-10s maps=%-18s shape=%d depth=%d flags=%-2d : " % key + "  ".join("%s %.1f" % (k.replace("slr_", ""), statistics.median(v)) if k != "batch_ms" else "batch %.4f" % statistics.median(v) for k, v in results[key].items()))


if __name__ == "__main__":
    main()
