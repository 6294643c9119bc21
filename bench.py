#!/usr/bin/env python
"""bench.py -- throughput of the structured-light hot path on MI355X.

Metric (BASELINE.json): Mpixels/s of decode + unwrap + triangulate on synthetic 4096x3000 stereo x 14-image
multi-frequency stacks (configs[1]).  One "step" = one batch of --frames (default 8: config 4's 64 frames over 8 GPUs)
DISTINCT stereo frames per GPU through the whole MF path (MFReconstruct::runReconstruction between imread and
MeshCreator): fused rectify+decode of both cameras, then row-wise phase match + Q-matrix triangulation -> XYZ [H][W][3] +
mask.  1 pixel = one (row, col) of one stereo frame.  The frames are resident in HBM when the timed region starts (8 x 344 MB:
nothing of a frame survives in the 256 MB Infinity Cache until its next turn).  --mode ge / gray time the two Gray-code
modes of the reference (GRAY_EPI: Reconstruct::runReconstruction_GE, GRAY_ONLY: Reconstruct::runReconstruction) the same way.

N > 1 (launched by torch.distributed.run, one rank per GPU, RCCL): frames shard across ranks (weak scaling, no
data-path collective inside the kernels); the per-step point cloud is assembled on every rank with one RCCL
all-gather (north_star: "RCCL all-gather over xGMI only to assemble the final point cloud").  The metric is Mpix/s of
decode + unwrap + triangulate, so by default that one assembly runs right AFTER the K timed steps, bracketed and timed on
its own and reported as "final_allgather_ms" (--gather after); --gather final puts it inside the timed region; --gather
step gathers after every step on a side stream, double-buffered so it overlaps the next step's compute (160 MB per frame
per peer: that mode measures xGMI, not the kernels); --gather off
measures the sharded path alone.

Prints ONE JSON line (rank 0).  Extra objects: "roofline" (dominant kernel, HIP-event timed inside the timed
region on the kernels' own stream; "traffic" = HBM bytes per launch from rocprofv3 PMC passes run by this script when
rocprofv3 is on PATH, else from profiles/pmc_traffic.json -- "traffic_source" says which), "kernels" (all kernels),
"cpu_baseline" (the CPU oracle on a bounded sample).
"""
import argparse
import csv
import glob
import importlib
import json
import math
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
# the test suite's poison hooks (tests/conftest.py) must never reach a measurement, e.g. when a test launches this file
os.environ.pop("SLR_POISON_OUTPUTS", None)
os.environ.pop("SLR_POISON_SCRATCH", None)
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
HBM_ACHIEVABLE_GBS = 6300.0    # what the same guide gives as achievable; both fractions are reported
BLACK_THR = 40                 # Duke/Set.ui:429-431 default

# algorithmic bytes per camera-pixel (or stereo-pixel for the match kernel) -- SURVEY.md 8(d), DESIGN.md
ALG_BYTES = {
    "slr_mf_rectify_decode": 25.0,     # 14 src + 6 map + 4 phase + 1 valid
    # inside slr_reconstruct_mf* the valid flag travels in the phase (NaN): no separate valid bytes on either side
    "slr_mf_rectify_decode_pair": 48.0,  # both cameras of the frame in one launch, per st-px: 2 x (14 src + 6 map + 4 phase)
    "slr_mf_decode": 19.0,             # 12 fringe + 2 white/black + 4 phase + 1 valid
    "slr_mf_match_triangulate": 21.0,  # 2 x 4 phase read + 12 xyz + 1 mask write (23 with separate valid bytes)
    "slr_remap_u8": 8.0,
    # Gray modes (per cam-px / per st-px), 4096-wide projector: 2 + 2 x 12 planes
    "slr_gray_rectify_decode": 36.0,   # 26 src + 6 map + 4 code (inside slr_reconstruct_ge the valid flag is code -1)
    "slr_gray_rectify_decode_pair": 72.0,  # both cameras of the frame in one launch, per st-px
    "slr_ge_match_triangulate": 21.0,  # 2 x 4 code read + 12 xyz + 1 mask write
    # BASELINE config 3, one pass over the hybrid stack, both cameras per launch: per cam-px 38 source planes + 4 map digest +
    # 4 code + 4 phase = 50 B
    "slr_hybrid_rectify_decode_pair": 100.0,
    # GRAY_ONLY, per st-px of the CAMERA image (the projector's 1280x1024 cells add 29 B each = 3 B per camera pixel here)
    "slr_ray_triangulate": 35.0,       # 2 x (4 item + 12 ray) read + per cell 16 offsets read + 13 sum/count write
    # BASELINE config 5 through the rectification, per cam-px: 34 fp16 planes + 6 map bytes read, 4 phase + 1 valid written
    "slr_mfn_rectify_decode": 79.0,
    "slr_ray_count": 34.0,             # both cameras: 2 x (4 + 4 code + 1 valid read, 4 cell + 4 rank write); inside
                                       # slr_reconstruct_gray it is part of the decode kernel since round 3 (no such launch)
}

# rocprofv3 kernel-name prefixes of the profiler names above (roofline.traffic)
DEVICE_KERNEL = {
    "slr_mf_rectify_decode_pair": ("mf_rect_decode_dma_kernel", "mf_rect_decode_lds_kernel"),
    "slr_mf_rectify_decode": ("mf_rect_decode_dma_kernel", "mf_rect_decode_lds_kernel"),
    "slr_mf_match_triangulate": ("mf_match_lean_kernel", "mf_match_binned_kernel", "mf_match_chunked_kernel"),
    "slr_mf_decode": ("mf_decode_kernel",),
    "slr_gray_rectify_decode": ("gray_rect_decode_dma_kernel", "gray_rect_decode_lds_kernel"),
    "slr_gray_rectify_decode_pair": ("gray_rect_decode_dma_kernel", "gray_rect_decode_lds_kernel"),
    "slr_hybrid_rectify_decode_pair": ("gray_rect_decode_dma_kernel",),
    "slr_gray_decode": ("gray_decode_kernel", "gray_decode_count_kernel"),
    "slr_ge_match_triangulate": ("ge_match_lean_kernel", "ge_match_kernel"),
    "slr_ray_triangulate": ("ray_triangulate_small_kernel", "ray_triangulate_staged_kernel"),
    "slr_ray_count": ("ray_count_kernel",),
    "slr_mfn_rectify_decode": ("mfn_rect_decode_kernel",),
}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--mode", choices=["mf", "ge", "gray", "hybrid", "mfn"], default="mf",
                    help="mf: the metric's 3-freq x 4-step path (default); ge: GRAY_EPI (Gray columns + rectification); gray: GRAY_ONLY "
                         "(Gray columns + rows, ray-ray triangulation, 1280x1024 projector); hybrid: BASELINE config 3 -- Gray columns + "
                         "3-freq x 4-step fringes in one stack (38 planes per camera), decoded in ONE pass, then phase match + triangulation; "
                         "mfn: BASELINE config 5 -- one 8192x6000 stereo frame, 4 frequencies x 8 steps of fp16 planes, f32 accumulation, "
                         "rectified (a build extension: the reference is hard-wired to 3 x 4 steps of u8); N > 1 shards ROW BANDS")
    ap.add_argument("--frames", type=int, default=0, help="distinct HBM-resident stereo frames per GPU per step (0 = 8; hybrid: 4)")
    ap.add_argument("--traffic", choices=["auto", "live", "file", "off"], default="auto",
                    help="roofline.traffic: live = two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of a 3-step child run, "
                         "file = profiles/pmc_traffic.json, auto = live when rocprofv3 is on PATH and N == 1")
    ap.add_argument("--pmc-child", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--width", type=int, default=4096)
    ap.add_argument("--height", type=int, default=3000)
    ap.add_argument("--rectify", type=int, default=1, help="1: raw planes + fused rectification (the reference path)")
    ap.add_argument("--gather", choices=["after", "final", "step", "off"], default="after",
                    help="N>1: RCCL all-gather of the point cloud: once per job right after the timed steps, timed separately "
                         "(after, default), once per job inside the timed region (final), "
                         "after every step (overlapped with the next step's compute), or never")
    ap.add_argument("--impl", choices=["ranks", "one-process"], default="ranks",
                    help="N > 1: ranks = one process per GPU, torch.distributed over RCCL (this script spawns its own ranks under "
                         "torch.distributed.run when it is started bare); one-process = N contexts on N devices driven by ONE host "
                         "thread through the C ABI's multi-GPU entries (slr_reconstruct_mf_multi + slr_allgather_clouds / "
                         "slr_reconstruct_mf_allgather_ex: direct one-hop peer copies over the xGMI mesh) -- what a Qt/C++ host calls")
    ap.add_argument("--profile", type=int, default=1, help="bracket kernels with HIP events (roofline)")
    ap.add_argument("--profile-stride", type=int, default=4,
                    help="bracket every n-th launch of a kernel inside the timed region (two event records per launch cost ~2.5 %% "
                         "of a 0.35 ms step each; 1 = every launch)")
    ap.add_argument("--cpu-baseline", type=int, default=1)
    ap.add_argument("--cpu-rows", type=int, default=0, help="rows of the CPU triangulation sample (0 = auto)")
    ap.add_argument("--streams", type=int, default=1,
                    help="frames in flight per GPU: S contexts (one HIP stream each) take the steps in turn, so the match of frame i "
                         "overlaps the decode of frame i+1 (application-level double buffering).  Default 1: kernels run back to "
                         "back and the per-kernel roofline is undisturbed")
    ap.add_argument("--pitch-pad", type=int, default=0,
                    help="bytes of row padding of the HBM-resident stacks (pitch = width + pad).  A pitch that is a power of two "
                         "(4096) puts the ~10 source rows of every tile of the rectifying decode on the same HBM channels: the "
                         "fused kernel is 3-6 %% faster with 64..1152 bytes of padding (the unfused one 1-3 %% slower)")
    ap.add_argument("--rect-algo", type=int, default=0, help="SLR_OPT_RECT_DECODE_ALGO (tuning: 0 auto, 1 gather, 2 64x16 tiles, 3 ring, 4 128x8/256thr, 5 128x8/512thr, 6 64x8, 7 LDS-DMA form)")
    ap.add_argument("--match-group", type=int, default=0, help="SLR_OPT_MF_BATCH_GROUP (0 = the library's default, 8; 1 = one match launch per frame)")
    ap.add_argument("--passes", type=int, default=4,
                    help="batch calls per step: a step is this many passes over the --frames distinct HBM-resident frames (default 4 x 8 = 32 "
                         "frames, 2.75 GB of input re-read from HBM every pass).  The GPU's clocks ramp for ~60 ms after an idle gap "
                         "(profiles/exp/r04/ramp.py: 225-250 us per frame in the first batches, 190 after 40): with the driver's --warmup 5 a "
                         "step of ONE batch starts the timed region 8 ms into that ramp")
    ap.add_argument("--decode-group", type=int, default=0, help="SLR_OPT_MF_BATCH_DECODE_GROUP (0 = the library's default, 8; 1 = one fused-decode launch per frame)")
    ap.add_argument("--match-algo", type=int, default=0, help="SLR_OPT_MF_MATCH_ALGO (tuning: 0 auto, 4 lean K4 with per-thread stores, 5 / 6 512 x 8 shapes, 7 persistent grouped K4: FORMS=all builds, 8 lean K4 with the hash dedup = round 5's kernel)")
    ap.add_argument("--dma-shape", type=int, default=-1, help="SLR_OPT_RECT_DMA_SHAPE (tuning: tile of the LDS-DMA form 7: 0 256x16/512thr, 1 256x8/512, 2 256x8/256, 3 128x16/512, 4 128x8/256, 5 256x4/256, 6 128x16/256)")
    ap.add_argument("--dma-depth", type=int, default=-1, help="SLR_OPT_RECT_DMA_DEPTH (tuning: 1 or 2 phases of LDS-DMA in flight)")
    ap.add_argument("--rect-resident", type=int, default=0, help="SLR_OPT_DEBUG_RECT_RESIDENT (experiments: workgroups of the persistent fused decodes; 0 = as many as are resident)")
    ap.add_argument("--debug-flags", type=int, default=0,
                    help="SLR_OPT_DEBUG_FLAGS (A/B runs of forms with identical results, e.g. 32 = map digests without the quad sort)")
    ap.add_argument("--batch-streams", type=int, default=0,
                    help="SLR_OPT_BATCH_STREAMS (0 = the library's default 2: GRAY_ONLY batches pipeline their frames over two streams; 1 = never)")
    ap.add_argument("--hybrid-one-pass", type=int, default=0,
                    help="--mode hybrid: 1 = SLR_OPT_HYBRID_ONE_PASS (one kernel over all 38 planes of a tile) instead of the default two "
                         "fused launches over the one stack")
    ap.add_argument("--maps", default="verged",
                    help="rectification maps of the timed region: 'verged' (default) = a stereo head verged by 0.2 rad in total with "
                         "k1 = -0.15, rectified by the repo's stereoRectify + initUndistortRectifyMap restatements (keystone maps: "
                         "tilted rows, taller tile boxes); 'verged:THETA:K1' = another such rig; 'near-identity' = "
                         "synth.make_rectify_maps (0.2 deg of roll, k1 -0.08: the maps of rounds 1-3)")
    ap.add_argument("--map-sweep", type=int, default=1,
                    help="also time the fused decode on the maps of three verged rigs (stereoRectify + initUndistortRectifyMap), outside "
                         "the timed region: form selected, tiles that do not fit, read-mode histogram (realistic_maps)")
    ap.add_argument("--eval-model", choices=["strict", "x87"], default="strict",
                    help="SLR_OPT_EVAL_MODEL: strict IEEE (the default) or the reference's own MSVC2010 x87 / fp:precise evaluation "
                         "(same kernels, x87 heterodyne tail / match predicate / disparity; DESIGN.md section 2)")
    ap.add_argument("--both-models", type=int, default=1,
                    help="1: after the timed region, a short leg of the same steps under the OTHER evaluation model; the line's `models` carries both")
    ap.add_argument("--self-check", type=int, default=1,
                    help="after the timed region: the batch output of every distinct frame against the single-frame entry (checksums)")
    ap.add_argument("--host-io", type=int, default=1,
                    help="also time the SLR_MEM_HOST entry point (PCIe-inclusive, reported beside the result, never `value`)")
    return ap.parse_args()


def _device_info(torch, index):
    """what the step ran on (box-to-box variation in the pool is +-5 %, one box was 25 % slower on every LDS-tiled kernel)"""
    try:
        p = torch.cuda.get_device_properties(index)
        return {"name": p.name, "compute_units": p.multi_processor_count, "hbm_gib": round(p.total_memory / 2 ** 30, 1),
                "gcn_arch": getattr(p, "gcnArchName", None), "clock_mhz": round(getattr(p, "clock_rate", 0) / 1000)}
    except Exception as e:                                   # pragma: no cover
        return {"error": str(e)}


def cpu_baseline(synth, W, H, stack_cpu, maps_cpu, calib, rows):
    """The CPU oracle (a port of the reference loops, 1 thread like the reference) on a bounded sample of the same
    workload: full-frame remap + decode for both cameras, triangulation on `rows` image rows (it is O(W^2) per
    row), scaled to a per-frame time."""
    import numpy as np
    import oracle as O                      # checker / baseline only, never the product path
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from util import calib_parts
    O.build()
    camL, camR, Q, T = calib_parts(O, calib)
    t0 = time.perf_counter()
    dec = []
    t_remap = 0.0
    for cam in range(2):
        planes = stack_cpu[cam]
        if maps_cpu is not None:
            tr = time.perf_counter()
            planes = np.stack([O.remap_u8(planes[p], maps_cpu[cam][0], maps_cpu[cam][1]) for p in range(14)])
            t_remap += time.perf_counter() - tr
        dec.append(O.mf_decode(planes, BLACK_THR))
    t_dec = time.perf_counter() - t0
    r0 = H // 2 - rows // 2
    t0 = time.perf_counter()
    O.mf_triangulate(dec[0][0], dec[0][1], dec[1][0], dec[1][1], camL, camR, Q, T, rows=(r0, r0 + rows))
    t_tri = time.perf_counter() - t0
    per_frame = t_dec + t_tri * (H / float(rows))
    # extra, explicitly NOT the reference (which is single-threaded): the same port row-parallel on every host core
    # (ctypes releases the GIL, so plain threads run the C loops concurrently)
    extra = None
    try:
        from concurrent.futures import ThreadPoolExecutor
        nthr = os.cpu_count() or 1
        t0 = time.perf_counter()
        with ThreadPoolExecutor(nthr) as ex:
            jobs = [(cam, p) for cam in range(2) for p in range(14)]
            if maps_cpu is not None:
                rect = list(ex.map(lambda cp: O.remap_u8(stack_cpu[cp[0]][cp[1]], maps_cpu[cp[0]][0], maps_cpu[cp[0]][1]), jobs))
                planes2 = [np.stack(rect[0:14]), np.stack(rect[14:28])]
            else:
                planes2 = [stack_cpu[0], stack_cpu[1]]
            band = max(1, (H + nthr - 1) // nthr)
            bands = [(r, min(H, r + band)) for r in range(0, H, band)]
            decs = []
            for cam in range(2):
                parts = list(ex.map(lambda b: O.mf_decode(np.ascontiguousarray(planes2[cam][:, b[0]:b[1]]), BLACK_THR), bands))
                decs.append((np.concatenate([q[0] for q in parts]), np.concatenate([q[1] for q in parts])))
            list(ex.map(lambda b: O.mf_triangulate(decs[0][0], decs[0][1], decs[1][0], decs[1][1], camL, camR, Q, T, rows=b), bands))
        t_all = time.perf_counter() - t0
        extra = {"value": round(W * H / t_all / 1e6, 3), "unit": "Mpix/s", "cores": nthr,
                 "note": "NOT the reference (single-threaded): the same C port, row-parallel over all host cores; whole frame, %.2f s" % t_all}
    except Exception as e:                               # the baseline must never break the bench line
        extra = {"error": repr(e)}
    try:
        literal = literal_cost_baseline(synth, O, W, H, t_remap)
    except Exception as e:                                   # the baseline must never break the bench line
        literal = {"error": repr(e)}
    return {
        "literal_cost": literal,
        "value": round(W * H / per_frame / 1e6, 4), "unit": "Mpix/s", "cores": 1, "kind": "port",
        "sample": "1 stereo frame %dx%d: full-frame remap+decode of both cameras (%.1f s) + match/triangulate on %d of %d "
                  "rows (%.1f s), scaled to the frame; single thread, gcc -O2" % (W, H, t_dec, rows, H, t_tri),
        "host_cpus": os.cpu_count(), "all_cores_extra": extra,
    }


def literal_cost_baseline(synth, O, W, H, t_remap_frame):
    """the reference's REAL cost model (oracle/slr_literal.cpp: a heap vector per pixel, by-value matrix headers, a
    vector copy per comparison -- mfreconstruct.cpp:165, :286-291, utilities.cpp:125) timed at 640x480 (whole frame) and
    1280x1024 (whole-frame decode, match on every 16th row) and extrapolated to W x H: the decode linearly in pixels, the match
    as rows x W^(1 + e) with the exponent e fitted between the two sizes (a left pixel's search walks ~half the right row, so
    e ~ 1).  A reported baseline only, like the flat port beside it."""
    import math
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from util import calib_parts
    pts = []
    for (w, h, step) in ((640, 480, 1), (1280, 1024, 16)):
        calib, _ = synth.make_calibration(w, h)
        camL, camR, Q, T = calib_parts(O, calib)
        st = synth.render_mf_stack(w, h, seed=1234, noise=2).numpy()
        rows = len(range(0, h, step))
        _, has, t_dec, t_tri = O.literal_mf(st[0], st[1], BLACK_THR, camL, camR, Q, T, rows=(0, h), row_step=step)
        pts.append({"size": "%dx%d" % (w, h), "decode_s": round(t_dec, 3), "match_rows": rows, "match_s": round(t_tri, 3),
                    "decode_ns_per_cam_px": round(t_dec / (2.0 * w * h) * 1e9, 1),
                    "match_ns_per_left_px": round(t_tri / (rows * float(w)) * 1e9, 1), "matched_frac": round(float(has[::step].mean()), 3),
                    "_w": w})
    a, b = pts
    e = math.log(b["match_ns_per_left_px"] / a["match_ns_per_left_px"]) / math.log(b["_w"] / float(a["_w"]))
    t_dec = b["decode_ns_per_cam_px"] * 1e-9 * 2.0 * W * H
    t_match = b["match_ns_per_left_px"] * 1e-9 * (W / float(b["_w"])) ** e * float(W) * H
    total = t_dec + t_match + t_remap_frame
    for p_ in pts:
        del p_["_w"]
    return {"value": round(W * H / total / 1e6, 5), "unit": "Mpix/s", "cores": 1, "kind": "port (literal cost model)",
            "extrapolated_s_per_frame": {"decode": round(t_dec, 1), "match_triangulate": round(t_match, 1),
                                         "remap_flat_port": round(t_remap_frame, 2), "total": round(total, 1)},
            "match_width_exponent": round(e, 3), "measured": pts,
            "sample": "oracle/slr_literal.cpp at 640x480 (whole frame) and 1280x1024 (decode whole, match on every 16th row), extrapolated to "
                      "%dx%d; single thread, g++ -O2" % (W, H)}


def frames_per_launch(args, mode, rectify, F, kernel_name):
    """frames one launch of `kernel_name` serves inside the timed region: slr_reconstruct_mf_batch hands groups of frames
    (SLR_OPT_MF_BATCH_GROUP, default 8) to ONE fused-decode launch and ONE match launch (round 4); everything else: 1"""
    if mode != "mf":
        return 1.0
    g = min(args.match_group or 8, F)
    groups = [g] * (F // g) + ([F % g] if F % g else [])
    if kernel_name == "slr_mf_match_triangulate" and g >= 2 and args.match_algo in (0, 4, 8):
        return F / float(len(groups))
    dg = args.decode_group or 8
    if kernel_name == "slr_mf_rectify_decode_pair" and g >= 2 and dg >= 2 and rectify and args.rect_algo in (0, 7):
        launches = sum(x // dg + (1 if x % dg else 0) for x in groups if x > 1) + sum(1 for x in groups if x == 1)
        return F / float(launches)          # (a last sub-group of one frame is a single-frame launch: mixed, averaged)
    return 1.0


def live_traffic(args, kernel_name, fpl=1.0):
    """HBM bytes per launch of `kernel_name` (a profiler name) from two rocprofv3 --pmc passes over a short child run of this
    script (the timed region's batch per step), collected and corrected as MI355X_MICROARCH.md's HBM section prescribes: separate
    passes for FETCH_SIZE and WRITE_SIZE (KiB), gfx950's FETCH_SIZE doubled.  Returns (bytes, detail) or (None, reason)."""
    exe = shutil.which("rocprofv3")
    prefixes = DEVICE_KERNEL.get(kernel_name)
    if not exe or not prefixes:
        return None, "rocprofv3 not on PATH" if not exe else "no device kernel name known for " + kernel_name
    vals = {}
    env = dict(os.environ, TMPDIR="/tmp")
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="slr_pmc_", dir="/tmp")
        cmd = [exe, "--kernel-trace", "--pmc", counter, "-f", "csv", "-d", d, "-o", "pmc", "--", sys.executable,
               os.path.abspath(__file__), "--pmc-child", "1", "--mode", args.mode, "--width", str(args.width), "--height", str(args.height),
               "--rectify", str(args.rectify), "--rect-algo", str(args.rect_algo), "--match-algo", str(args.match_algo), "--match-group", str(args.match_group), "--decode-group", str(args.decode_group), "--dma-shape", str(args.dma_shape),
               "--dma-depth", str(args.dma_depth), "--pitch-pad", str(args.pitch_pad), "--debug-flags", str(args.debug_flags),
               "--maps", args.maps, "--frames", str(args.frames), "--eval-model", args.eval_model]
        try:
            subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=100, check=False)   # (a pass takes 15-25 s; a stuck profiler must not cost the bench line minutes)
            got = []
            for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                with open(path) as f:
                    for row in csv.DictReader(f):
                        kn = row.get("Kernel_Name", "")
                        if row.get("Counter_Name") == counter and any(px in kn for px in prefixes):
                            got.append(float(row["Counter_Value"]))
            if got:
                vals[counter] = sum(got) / len(got)
        except Exception as e:                              # the bench line must survive a profiler hiccup
            vals[counter + "_error"] = repr(e)
        finally:
            shutil.rmtree(d, ignore_errors=True)
    if "FETCH_SIZE" not in vals or "WRITE_SIZE" not in vals:
        return None, "rocprofv3 produced no counters for %s (%s)" % (kernel_name, vals)
    rd, wr = 2.0 * vals["FETCH_SIZE"] * 1024.0, vals["WRITE_SIZE"] * 1024.0
    return int(rd + wr), {"hbm_read_bytes": int(rd), "hbm_write_bytes": int(wr), "frames_per_launch": fpl,
                          "hbm_bytes_per_frame": int((rd + wr) / fpl),
                          "how": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, one pass each, mean per dispatch of a short child run of the timed "
                                 "region's batch; FETCH_SIZE KiB x 2 (gfx950 counts 128-B requests at 64 B), WRITE_SIZE KiB x 1"}


def copy_ceiling(torch, dev, ctx):
    """The box's streaming rates as MI355X_MICROARCH.md measures them: float4 non-temporal kernels, ONE 16-byte word per
    thread (the launch shape profiles/exp/r05/bw.txt found fastest), settled clocks, read + written bytes / time:
      copy      slr_stream_copy over 1 GiB (1 : 1);
      read mix  slr_stream_mix with the fused MF decode's 20 : 4 read : write ratio (5 words read from 5 streams per word written,
                256 MiB out / 1.25 GiB in) -- the ceiling a kernel with the decode's traffic shape has on this box.
    Returns (copy GB/s, read-mix GB/s, GB/s of torch's copy_ for the record: a library memcpy is not a ceiling)."""
    n = 1 << 30
    a = torch.zeros(n, dtype=torch.uint8, device=dev)
    b = torch.empty(n, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    for _ in range(30):                                      # ~10 ms: the clocks settle
        ctx.stream_copy(b, a)
    ctx.timer_begin()
    for _ in range(10):
        ctx.stream_copy(b, a)
    kern = 2.0 * n * 10 / (ctx.timer_end() * 1e-3) / 1e9
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    b.copy_(a)
    e0.record()
    for _ in range(5):
        b.copy_(a)
    e1.record()
    e1.synchronize()
    memcpy = 2.0 * n * 5 / (e0.elapsed_time(e1) * 1e-3) / 1e9
    mix = None
    try:
        m = 1 << 28
        src = torch.zeros(5 * m + 0, dtype=torch.uint8, device=dev)
        dst = b[:m]
        torch.cuda.synchronize()
        for _ in range(20):
            ctx.stream_mix(dst, src, 5)
        ctx.timer_begin()
        for _ in range(10):
            ctx.stream_mix(dst, src, 5)
        mix = 6.0 * m * 10 / (ctx.timer_end() * 1e-3) / 1e9
    except Exception:                                        # never break the bench line
        mix = None
    return kern, mix, memcpy


def host_io_rate(np, torch, ctx, stack, W, H, rectify, slr_mod, calib_obj):
    """The drop-in boundary with host buffers (what a cv::Mat caller hands over): H2D of 2x14 planes, the path, D2H of
    XYZ + mask, synchronous.  Pinned host memory, 3 frames after one warm-up."""
    host = stack[0].cpu().pin_memory()
    L, R = host[0].numpy(), host[1].numpy()
    xyz = torch.empty((H, W, 3), dtype=torch.float32).pin_memory().numpy()
    has = torch.empty((H, W), dtype=torch.uint8).pin_memory().numpy()
    ctx.reconstruct_mf(L, R, BLACK_THR, rectify, W=W, xyz=xyz, has=has)
    t0 = time.perf_counter()
    for _ in range(3):
        ctx.reconstruct_mf(L, R, BLACK_THR, rectify, W=W, xyz=xyz, has=has)
    dt = (time.perf_counter() - t0) / 3
    out = {"value": round(W * H / dt / 1e6, 1), "unit": "Mpix/s", "ms_per_frame": round(dt * 1e3, 3),
           "bytes_in": int(host.numel()), "bytes_out": int(xyz.nbytes + has.nbytes),
           "note": "slr_reconstruct_mf with SLR_MEM_HOST pinned buffers: H2D + kernels + D2H, synchronous, no overlap"}
    # double-buffered: two contexts fed alternately with SLR_OPT_ASYNC_HOST -- the upload of frame i+1, the kernels of
    # frame i and the download of frame i-1 overlap
    try:
        ctxs, outs = [], []
        for k in range(2):
            c2 = slr_mod.Context(ctx.device_id)
            c2.set_calibration(calib_obj)
            if rectify:
                for cam in range(2):
                    mx, mf = ctx.get_rectify_maps(cam, W, H)
                    c2.set_rectify_maps(cam, mx, mf)
            c2.set_option(slr_mod.capi.OPT_ASYNC_HOST, 1)
            ctxs.append(c2)
            outs.append((torch.empty((H, W, 3), dtype=torch.float32).pin_memory().numpy(),
                         torch.empty((H, W), dtype=torch.uint8).pin_memory().numpy()))
        for k in range(2):
            ctxs[k].reconstruct_mf(L, R, BLACK_THR, rectify, W=W, xyz=outs[k][0], has=outs[k][1])
        for k in range(2):
            ctxs[k].synchronize()
        n = 8
        t0 = time.perf_counter()
        for i in range(n):
            k = i % 2
            ctxs[k].synchronize()                           # frame i-2 of this slot is complete: its buffers are free again
            ctxs[k].reconstruct_mf(L, R, BLACK_THR, rectify, W=W, xyz=outs[k][0], has=outs[k][1])
        for k in range(2):
            ctxs[k].synchronize()
        dt2 = (time.perf_counter() - t0) / n
        same = bool((outs[0][1] == has).all() and (outs[1][1] == has).all() and (outs[0][0] == xyz).all())
        out["double_buffered"] = {"value": round(W * H / dt2 / 1e6, 1), "unit": "Mpix/s", "ms_per_frame": round(dt2 * 1e3, 3),
                                  "identical_to_synchronous": same,
                                  "note": "two contexts, SLR_OPT_ASYNC_HOST: upload / kernels / download of consecutive frames overlap"}
        for c2 in ctxs:
            c2.close()
    except Exception as e:                                  # never break the bench line
        out["double_buffered"] = {"error": repr(e)}
    return out


def self_spawn(args):
    """`python bench.py --gpus N` started WITHOUT a launcher: run N ranks of this script under torch.distributed.run (one process per
    GPU, rendezvous on 127.0.0.1 at a free port) and pass their output and exit status through.  On a box with fewer than N GPUs the
    ranks are refused unless SLR_BENCH_ONE_DEVICE=1 (all ranks on device 0: a dry run of the N > 1 code path)."""
    import socket
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < args.gpus and not os.environ.get("SLR_BENCH_ONE_DEVICE"):
        raise SystemExit("bench.py --gpus %d: this box has %d GPU(s) (SLR_BENCH_ONE_DEVICE=1 puts every rank on device 0: a dry run)" %
                         (args.gpus, have))
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL between processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    _trace("no launcher: spawning %d ranks under torch.distributed.run (port %d)" % (args.gpus, port))
    raise SystemExit(subprocess.call(cmd, env=env))


def hbm_footprint(torch, dev, items):
    """what one GPU holds for the job, by part (bytes), beside what the allocator and the driver report"""
    out = {k: int(v) for k, v in items.items()}
    out["sum_of_parts"] = int(sum(items.values()))
    try:
        free, total = torch.cuda.mem_get_info(dev)
        out.update({"torch_max_allocated": int(torch.cuda.max_memory_allocated(dev)), "device_in_use_now": int(total - free),
                    "device_total": int(total)})
    except Exception as e:                                   # pragma: no cover
        out["error"] = repr(e)
    return out


def main_one_process(args):
    """--impl one-process: the multi-GPU leg as a C++ host drives it -- ONE process, one slr_ctx per device, frames sharded "blocked"
    (context k owns frames [k F, (k + 1) F) of the job) and computed straight into their slots of every context's own assembled
    arrays by slr_reconstruct_mf_multi (no assembly inside the timed steps), then ONE exchange: slr_allgather_clouds -- every
    source pushes its shard to its n - 1 peers by hipMemcpyPeerAsync on per-destination streams, one hop over the point-to-point
    xGMI mesh, no ring, no staging -- timed on its own as final_allgather_ms (--gather after), or inside the timed
    region as the last step's slr_reconstruct_mf_allgather_ex (--gather final).  The exchange proves itself: every context's
    checksums of its LOCAL frames taken before the exchange must be reproduced by every device's assembled copy
    (slr_cloud_checksums on each device; slr_verify_assembled compares the devices with each other)."""
    import numpy as np  # noqa: F401
    import torch
    if args.mode != "mf":
        raise SystemExit("--impl one-process runs the MF path (the C ABI's multi-GPU entries are slr_reconstruct_mf_*)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback for the product path)")
    N = max(1, args.gpus)
    one_dev = bool(os.environ.get("SLR_BENCH_ONE_DEVICE"))
    have = torch.cuda.device_count()
    if have < N and not one_dev:
        raise SystemExit("bench.py --impl one-process --gpus %d: this box has %d GPU(s) (SLR_BENCH_ONE_DEVICE=1: every context on device 0)" % (N, have))
    devs = [0 if one_dev else k for k in range(N)]
    slr = importlib.import_module("structure-light-reconstructor_amd")
    synth = importlib.import_module("structure-light-reconstructor_amd.synth")
    W, H = args.width, args.height
    F = args.frames if args.frames > 0 else 8
    PASSES = max(1, args.passes)
    rectify = bool(args.rectify)
    ctxs = [slr.Context(d) for d in devs]
    rig = rig_desc = None
    calib, _ = synth.make_calibration(W, H)
    if rectify and args.maps.startswith("verged"):
        parts = args.maps.split(":")
        rig_theta = float(parts[1]) if len(parts) > 1 else 0.2
        rig_k1 = float(parts[2]) if len(parts) > 2 else -0.15
        rig = synth.make_verged_rig(W, H, rig_theta, rig_k1)
        calib = rig["calib"]
        rig_desc = "verged stereo head: %.2f rad of toe-in in total, k1 %.2f (stereoRectify + slr_init_rectify_maps)" % (rig_theta, rig_k1)
    elif rectify and args.maps != "near-identity":
        raise SystemExit("--maps: verged[:theta:k1] or near-identity")
    pitch = W + max(0, args.pitch_pad)
    stacks, xyz_all, has_all = [], [], []
    for k, c_ in enumerate(ctxs):
        dev = torch.device("cuda", devs[k])
        with torch.cuda.device(dev):
            c_.set_calibration(calib)
            if args.match_group:
                c_.set_option(slr.capi.OPT_MF_BATCH_GROUP, args.match_group)
            if args.decode_group:
                c_.set_option(slr.capi.OPT_MF_BATCH_DECODE_GROUP, args.decode_group)
            if args.eval_model == "x87":
                c_.set_option(slr.capi.OPT_EVAL_MODEL, 1)
            if rectify:
                if rig is not None:
                    synth.install_verged_maps(c_, rig, W, H)
                else:
                    for cam in range(2):
                        mxy, mfr = synth.make_rectify_maps(W, H, cam, device=dev)
                        c_.set_rectify_maps(cam, mxy, mfr)
            st_ = torch.zeros((F, 2, 14, H, pitch), dtype=torch.uint8, device=dev)
            for f in range(F):                              # job frame k F + f: the seeds of the ranks path (1234 + F rank + f)
                st_[f, :, :, :, :W] = synth.render_mf_stack(W, H, seed=1234 + F * k + f, noise=2, device=dev)
            stacks.append(st_)
            xyz_all.append(torch.empty((N * F, H, W, 3), dtype=torch.float32, device=dev))
            has_all.append(torch.empty((N * F, H, W), dtype=torch.uint8, device=dev))
    for d in set(devs):
        torch.cuda.synchronize(d)
    _trace("one-process: %d contexts, frames rendered" % N)
    # context k's shard, in place in ITS assembled arrays (blocked); slr_reconstruct_mf_multi's own labelling of the frames is
    # cyclic, which only matters when it assembles -- it does not here (gather_ctx = -1)
    xyz_loc = [xyz_all[k][k * F:(k + 1) * F] for k in range(N)]
    has_loc = [has_all[k][k * F:(k + 1) * F] for k in range(N)]
    final_gather = args.gather == "final" and N > 1
    after_gather = args.gather in ("after", "step") and N > 1

    def sync_all():
        for d in set(devs):
            torch.cuda.synchronize(d)

    def compute():
        # (frame f of multi's job = stacks[f % N][f // N]: with N * F frames every context gets its F)
        slr.capi.reconstruct_mf_multi(ctxs, stacks, BLACK_THR, rectify, W=W, gather_ctx=-1, xyz=xyz_loc, has=has_loc)

    def step(i, last=False):
        for p_ in range(PASSES):
            if last and final_gather and p_ == PASSES - 1:
                slr.capi.reconstruct_mf_allgather(ctxs, stacks, BLACK_THR, rectify, W=W, assignment=slr.capi.ASSIGN_BLOCKED,
                                                  out=(xyz_all, has_all))
            else:
                compute()

    for i in range(args.warmup):
        step(i)
    sync_all()
    if args.profile:
        for c_ in ctxs:
            c_.set_option(slr.capi.OPT_PROFILE_STRIDE, max(1, args.profile_stride))
            c_.profile_enable(True)
            c_.profile_reset()
    sync_all()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i, last=(i == args.steps - 1))
    sync_all()
    elapsed = time.perf_counter() - t0
    _trace("one-process: timed region done")
    prof = {}
    if args.profile:
        for c_ in ctxs:
            for name, (ms, n) in c_.profile().items():
                a = prof.get(name, (0.0, 0))
                prof[name] = (a[0] + ms, a[1] + n)
            c_.profile_enable(False)
    gather_ms, proof, peer_direct = None, None, None
    if N > 1:
        # the owners' words: every context's LOCAL slots (before the exchange; with --gather final the pushes have already run, but
        # nothing ever writes a context's own slots except its own kernels)
        mine = [ctxs[k].cloud_checksums(xyz_loc[k], has_loc[k]) for k in range(N)]
        if after_gather:
            sync_all()
            tg = time.perf_counter()
            peer_direct = slr.capi.allgather_clouds(ctxs, xyz_all, has_all, assignment=slr.capi.ASSIGN_BLOCKED)
            sync_all()
            gather_ms = (time.perf_counter() - tg) * 1e3
        owner = [int(w) for k in range(N) for w in mine[k]]
        bad = []
        for k in range(N):                                  # every device's whole assembled copy against the owners' words
            got = ctxs[k].cloud_checksums(xyz_all[k], has_all[k])
            bad += [(k, f) for f in range(N * F) if int(got[f]) != owner[f]]
        mism = slr.capi.verify_assembled(ctxs, xyz_all, has_all)
        proof = {"frames_verified_on_every_rank": N * F if not bad else 0, "all_ranks_ok": not bad and mism == 0,
                 "error": None if not bad else "context/frame pairs that differ from their owner's word: %s" % bad[:8],
                 "slr_verify_assembled_mismatches": mism,
                 "how": "slr_cloud_checksums of every context's LOCAL frames before the exchange vs the same words of every device's "
                        "assembled copy afterwards, + slr_verify_assembled (devices against each other)"}
        if not proof["all_ranks_ok"]:
            raise SystemExit("bench.py --impl one-process: the assembled point cloud does not match its owners' checksums: %s" % (proof["error"],))
    npix = float(W) * H
    value = N * npix * F * PASSES * args.steps / elapsed / 1e6
    kernels, roofline = [], None
    for name, (ms, n) in sorted(prof.items(), key=lambda kv: -kv[1][0]):
        avg_ms = ms / n
        fpl = frames_per_launch(args, "mf", rectify, F, name)
        entry = {"name": name, "launches": int(round(n / fpl)), "frames_per_launch": fpl, "avg_launch_us": round(avg_ms * fpl * 1e3, 2),
                 "avg_us": round(avg_ms * 1e3, 2), "total_ms": round(ms, 3)}
        if name in ALG_BYTES:
            gbs = ALG_BYTES[name] * npix / (avg_ms * 1e-3) / 1e9
            entry.update({"alg_bytes_per_px": ALG_BYTES[name], "achieved_GBs": round(gbs, 1), "frac_hbm_peak": round(gbs / HBM_PEAK_GBS, 4)})
        kernels.append(entry)
    if kernels and kernels[0].get("achieved_GBs"):
        k0 = kernels[0]
        roofline = {"kernel": k0["name"], "bound": "hbm", "achieved": k0["achieved_GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": k0["frac_hbm_peak"], "traffic": None, "traffic_source": "not collected in --impl one-process (see the N = 1 line)",
                    "frames_per_launch": k0["frames_per_launch"], "us_per_frame": k0["avg_us"], "avg_launch_us": k0["avg_launch_us"],
                    "alg_bytes_per_launch": ALG_BYTES[k0["name"]] * npix * k0["frames_per_launch"],
                    "note": "mean over all %d contexts' launches (HIP events on each context's own stream)" % N}
    uu = []
    for d in devs:
        try:
            uu.append(str(torch.cuda.get_device_properties(d).uuid))
        except Exception:
            uu.append("device%d" % d)
    print(json.dumps({
        "metric": "Mpixels/s decode+unwrap+triangulate, 4096x3000 stereo, 1/2/4/8 GPU",
        "value": round(value, 2), "unit": "Mpix/s", "n_gpus": N, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u8 in, f32 phase/XYZ (f64 undistort + Q reprojection)", "data": "synthetic",
        "ms_per_frame": round(elapsed / args.steps / (F * PASSES) * 1e3, 4),
        "config": {"workload": "%dx%d stereo, 3-freq x 4-step (14 planes/camera): rectify+decode+unwrap+match+triangulate; a step = %d passes "
                               "over %d distinct HBM-resident frames per GPU (%d frames; every pass re-reads its input from HBM)" % (W, H, PASSES, F, PASSES * F),
                   "maps": rig_desc or ("synthetic near-identity rectification maps" if rectify else None), "mode": "mf", "impl": "one-process",
                   "frames_per_gpu_per_step": F * PASSES, "distinct_frames": F, "passes_per_step": PASSES, "rectify": rectify,
                   "parallelism": "ONE process, %d slr_ctx on %d device(s), frames sharded blocked over the contexts (slr_reconstruct_mf_multi, "
                                  "no assembly inside the steps)%s" % (N, len(set(devs)), "" if N == 1 else (
                                      ", the last step's last pass is slr_reconstruct_mf_allgather_ex (compute + peer pushes, inside the timed region)"
                                      if final_gather else ", one slr_allgather_clouds right after the timed steps (final_allgather_ms)"))},
        "device": _device_info(torch, devs[0]),
        "collective_backend": None if N == 1 else "peer", "collective_ranks": None if N == 1 else N,
        "collective": None if N == 1 else {"backend": "peer", "ranks": N, "devices": [{"rank": k, "local_device": devs[k], "uuid": uu[k]} for k in range(N)],
                                           "distinct_devices": len(set(uu)), "peer_direct": peer_direct,
                                           "how": "hipMemcpyPeerAsync on per-destination streams of each source context (slr_allgather_clouds)"},
        "final_allgather_ms": None if gather_ms is None else round(gather_ms, 3),
        "gather_inclusive_value": (round(N * npix * F * PASSES * args.steps / (elapsed + gather_ms * 1e-3) / 1e6, 2) if gather_ms is not None
                                   else (round(value, 2) if final_gather else None)),
        "gather_proof": proof,
        "final_allgather_bytes_per_rank_out": None if N == 1 else int((N - 1) * F * H * W * 13),
        "hbm_footprint_per_gpu": hbm_footprint(torch, devs[0], {
            "input_stack": stacks[0].numel(), "assembled_xyz_and_mask": xyz_all[0].numel() * 4 + has_all[0].numel(),
            "phase_scratch_of_a_group": 8 * W * H * min(F, args.match_group or 8), "undistortion_tables": 12 * W * H,
            "maps_and_tile_tables_both_cameras": 2 * (6 + 4) * W * H if rectify else 0}),
        "eval_model": args.eval_model, "roofline": roofline, "kernels": kernels, "cpu_baseline": None}))
    for c_ in ctxs:
        c_.close()


def main_mfn(args):
    """BASELINE config 5: one 8192x6000 stereo frame per step unit, 4 x 8 fp16 planes per camera (6.7 GB per frame), raw camera
    images through the rectification (slr_mfn_rectify_decode), then the match + triangulation of the 8192-pixel rows (chunked K4).
    N > 1: the FRAME is split into N row bands: every rank holds only the source rows its bands' maps point into
    (slr_rectify_source_rows), decodes and matches its band, and one all-gather assembles the cloud -- total work is fixed
    ("scaling": "strong")."""
    import numpy as np
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = 0 if os.environ.get("SLR_BENCH_ONE_DEVICE") else int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback for the product path)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        backend = os.environ.get("SLR_BENCH_BACKEND", "nccl")
        dist.init_process_group(backend, device_id=dev) if backend == "nccl" else dist.init_process_group(backend)
    slr = importlib.import_module("structure-light-reconstructor_amd")
    synth = importlib.import_module("structure-light-reconstructor_amd.synth")
    sdist = importlib.import_module("structure-light-reconstructor_amd.dist")
    W, H = (8192, 6000) if (args.width, args.height) == (4096, 3000) else (args.width, args.height)
    F = args.frames if args.frames > 0 else 1
    NF, NS = 4, 8
    NP = 2 + NF * NS
    ctx = slr.Context(local)
    rig = synth.make_verged_rig(W, H, 0.2, -0.15)
    ctx.set_calibration(rig["calib"])
    synth.install_verged_maps(ctx, rig, W, H)
    r0, r1 = sdist.shard_rows(H, rank, world)
    rows = r1 - r0
    win = [ctx.rectify_source_rows(cam, r0, rows) for cam in range(2)]           # (src_row0, src_rows) per camera
    # this rank's share of the frames: ONLY the source rows of its band (rendered whole, the window kept)
    stacks = []
    for f in range(F):
        full = synth.render_mfn_stack(W, H, NF, NS, seed=1234 + f, noise=0.5, device=dev)
        stacks.append([full[cam, :, win[cam][0]:win[cam][0] + win[cam][1]].contiguous() for cam in range(2)])
        del full
    torch.cuda.synchronize()
    band = (H + world - 1) // world
    g_xyz = torch.zeros((F, world * band, W, 3), dtype=torch.float32, device=dev)
    g_has = torch.zeros((F, world * band, W), dtype=torch.uint8, device=dev)
    ph = [torch.empty((max(rows, 1), W), dtype=torch.float32, device=dev) for _ in range(2)]
    vd = [torch.empty((max(rows, 1), W), dtype=torch.uint8, device=dev) for _ in range(2)]

    def step(i):
        for f in range(F):
            if rows <= 0:
                continue
            for cam in range(2):
                ctx.mfn_rectify_decode(cam, stacks[f][cam], NF, NS, float(BLACK_THR), W=W, H=H, row0=r0, rows=rows,
                                       src_row0=win[cam][0], phase=ph[cam][:rows], valid=vd[cam][:rows])
            x, h, _ = ctx.mf_triangulate(ph[0][:rows], vd[0][:rows], ph[1][:rows], vd[1][:rows], want_match=False, row0=r0, image_h=H)
            if world > 1:                                   # (the band's place in the assembled frame; K4 has no output-view entry)
                with torch.cuda.stream(ctx.stream):
                    g_xyz[f, rank * band:rank * band + rows].copy_(x, non_blocking=True)
                    g_has[f, rank * band:rank * band + rows].copy_(h, non_blocking=True)

    def sync_all():
        ctx.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    sync_all()
    ctx.set_option(slr.capi.OPT_PROFILE_STRIDE, 1)
    ctx.profile_enable(True)
    ctx.profile_reset()
    sync_all()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    sync_all()
    elapsed = time.perf_counter() - t0
    prof = ctx.profile()
    ctx.profile_enable(False)
    gather_ms, proof = None, None
    if world > 1:                                           # the bands of every frame -> the whole cloud on every rank
        tg = time.perf_counter()
        for f in range(F):
            loc_x, loc_h = g_xyz[f, rank * band:(rank + 1) * band], g_has[f, rank * band:(rank + 1) * band]
            if dist.get_backend() == "nccl":
                dist.all_gather_into_tensor(g_xyz[f], loc_x)
                dist.all_gather_into_tensor(g_has[f], loc_h)
            else:
                cx, ch = torch.empty(g_xyz[f].shape), torch.empty(g_has[f].shape, dtype=torch.uint8)
                dist.all_gather_into_tensor(cx, loc_x.cpu()); dist.all_gather_into_tensor(ch, loc_h.cpu())
                g_xyz[f].copy_(cx); g_has[f].copy_(ch)
        sync_all()
        tt = torch.tensor([time.perf_counter() - tg], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        gather_ms = float(tt.item()) * 1e3
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # every rank's assembled frames must carry the same checksums (each band was computed by exactly one rank)
        words = sdist.frame_checksums(g_xyz, g_has)
        if dist.get_backend() != "nccl":
            words = words.cpu()
        allw = [torch.empty_like(words) for _ in range(world)]
        dist.all_gather(allw, words)
        proof = {"frames": F, "all_ranks_agree": bool(all(torch.equal(w, words) for w in allw))}
        if not proof["all_ranks_agree"]:
            raise SystemExit("bench.py --mode mfn: the ranks' assembled clouds differ")
    npix = float(W) * H
    value = npix * F * args.steps / elapsed / 1e6
    kernels = []
    for name, (ms, n) in sorted(prof.items(), key=lambda kv: -kv[1][0]):
        ent = {"name": name, "launches": n, "avg_us": round(ms / n * 1e3, 2), "total_ms": round(ms, 3)}
        if name in ALG_BYTES:
            per_launch = ALG_BYTES[name] * W * rows                 # per launch = one camera's band
            gbs = per_launch / (ms / n * 1e-3) / 1e9
            ent.update({"alg_bytes_per_px": ALG_BYTES[name], "achieved_GBs": round(gbs, 1), "frac_hbm_peak": round(gbs / HBM_PEAK_GBS, 4)})
        kernels.append(ent)
    k0 = kernels[0] if kernels else None
    roofline = None
    if k0 and k0.get("achieved_GBs"):
        roofline = {"kernel": k0["name"], "bound": "hbm", "achieved": k0["achieved_GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": k0["frac_hbm_peak"], "traffic": None, "avg_launch_us": k0["avg_us"],
                    "alg_bytes_per_launch": ALG_BYTES[k0["name"]] * W * rows,
                    "form": "LDS-DMA form (round 4): 64 x 16 tiles, source boxes by buffer_load ... lds into a ring of 3 four-plane groups streamed across tiles, counted vmcnt waits; bit-identical to the per-pixel gather form"}
    if rank == 0:
        print(json.dumps({
            "metric": "Mpixels/s decode+unwrap+triangulate, 4096x3000 stereo, 1/2/4/8 GPU", "value": round(value, 2), "unit": "Mpix/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "fp16 in, f32 accumulate / phase / XYZ (f64 undistort + Q reprojection)", "data": "synthetic",
            "ms_per_frame": round(elapsed / args.steps / F * 1e3, 4),
            "config": {"workload": "BASELINE config 5 (NOT the headline config): %dx%d stereo, 4 frequencies x 8 steps of fp16 planes (34 per "
                                   "camera), rectify+decode+unwrap (build extension, no reference counterpart) + match+triangulate; a step "
                                   "= %d frame(s), each split into %d row band(s)" % (W, H, F, world),
                       "maps": "verged stereo head, 0.2 rad, k1 -0.15 (stereoRectify + slr_init_rectify_maps)", "mode": "mfn",
                       "rows_of_this_rank": [r0, r1], "source_rows_held": [list(w) for w in win],
                       "parallelism": "one frame's rows split over %d GPU(s): row bands with read-only source windows, one all-gather" % world},
            "final_allgather_ms": None if gather_ms is None else round(gather_ms, 3), "gather_proof": proof,
            "roofline": roofline, "kernels": kernels, "cpu_baseline": None}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


_T0 = time.perf_counter()


def _trace(what):
    """SLR_BENCH_TRACE=1: wall-clock stage marks on stderr (where a run's minutes go outside the timed region)"""
    if os.environ.get("SLR_BENCH_TRACE"):
        print("[bench %7.1f s] %s" % (time.perf_counter() - _T0, what), file=sys.stderr, flush=True)


def main():
    args = parse_args()
    if args.impl == "one-process":
        return main_one_process(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:     # started bare: be our own launcher (one rank per GPU)
        return self_spawn(args)
    if args.mode == "mfn":
        return main_mfn(args)
    import numpy as np
    import torch
    import torch.distributed as dist
    _trace("imports done")

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py --gpus %d inside a torch.distributed job of %d rank(s): the two must agree" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback for the product path)")
    if os.environ.get("SLR_BENCH_ONE_DEVICE"):       # dry-run of the N>1 code path on a 1-GPU box (with gloo)
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        backend = os.environ.get("SLR_BENCH_BACKEND", "nccl")     # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    # what the collective leg actually ran on: the backend torch.distributed reports, its rank count and every rank's device
    # (a SCALE run is checked against this: "nccl" = RCCL; distinct UUIDs = distinct physical GPUs)
    collective = None
    if world > 1:
        try:
            uuid = str(torch.cuda.get_device_properties(local).uuid)
        except Exception:
            uuid = None
        mine = {"rank": rank, "local_device": local, "uuid": uuid, "name": torch.cuda.get_device_name(local)}
        allr = [None] * world
        dist.all_gather_object(allr, mine)
        collective = {"backend": dist.get_backend(), "ranks": dist.get_world_size(), "devices": allr,
                      "distinct_devices": len({(r or {}).get("uuid") or ("rank%d" % i) for i, r in enumerate(allr)})}

    slr = importlib.import_module("structure-light-reconstructor_amd")
    synth = importlib.import_module("structure-light-reconstructor_amd.synth")
    W, H = args.width, args.height
    mode = args.mode
    F = args.frames if args.frames > 0 else (4 if mode == "hybrid" else 8)
    PASSES = max(1, args.passes)
    if args.pmc_child:                                   # the rocprofv3 child of live_traffic(): the timed region's batch, two steps
        args.steps, args.warmup, args.profile, args.cpu_baseline, args.host_io, args.traffic = 2, 1, 0, 0, 0, "off"
        args.self_check = 0
        PASSES = 1
        if mode != "mf":
            F, args.steps = 1, 3
    scan_w, scan_h = (W, 0) if mode in ("ge", "hybrid") else ((1280, 1024) if mode == "gray" else (0, 0))
    ncol = synth.gray_num_bits(scan_w) if mode != "mf" else 0
    nrow = synth.gray_num_bits(scan_h) if mode == "gray" else 0
    ppc = 14 if mode == "mf" else 2 + 2 * ncol + 2 * nrow + (12 if mode == "hybrid" else 0)
    if mode == "gray":                                   # the GRAY_ONLY decode with the bucket histogram inside, per camera launch:
        ALG_BYTES["slr_gray_decode"] = float(ppc + 8)    # its planes read, cell + rank written
    rectify = (bool(args.rectify) or mode == "hybrid") and mode != "gray"      # GRAY_ONLY never rectifies (reconstruct.cpp:230-265)

    S = max(1, args.streams)
    streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
    ctxs = [slr.Context(local, stream=st_) for st_ in streams]
    compute, ctx = streams[0], ctxs[0]
    calib, _ = synth.make_calibration(W, H) if mode != "gray" else synth.make_calibration(W, H, baseline=400.0, theta=0.6)
    maps, rig, rig_desc = None, None, None
    if rectify and args.maps.startswith("verged"):
        parts = args.maps.split(":")
        rig_theta = float(parts[1]) if len(parts) > 1 else 0.2
        rig_k1 = float(parts[2]) if len(parts) > 2 else -0.15
        rig = synth.make_verged_rig(W, H, rig_theta, rig_k1)
        calib = rig["calib"]                              # Q is stereoRectify's for this rig
        rig_desc = ("verged stereo head: %.2f rad of toe-in in total, k1 %.2f, rectified by the host mirror's stereoRectify + "
                    "slr_init_rectify_maps (keystone maps); near-identity maps and two more rigs: see realistic_maps" % (rig_theta, rig_k1))
    elif rectify:
        if args.maps != "near-identity":
            raise SystemExit("--maps: verged[:theta:k1] or near-identity")
        maps = [synth.make_rectify_maps(W, H, cam, device=dev) for cam in range(2)]
        torch.cuda.synchronize()

    def install_maps(c_):
        if rig is not None:
            synth.install_verged_maps(c_, rig, W, H)
        else:
            for cam in range(2):
                c_.set_rectify_maps(cam, maps[cam][0], maps[cam][1])
    for c_ in ctxs:
        c_.set_calibration(calib)
        if args.rect_algo:
            c_.set_option(slr.capi.OPT_RECT_DECODE_ALGO, args.rect_algo)
        if args.match_algo:
            c_.set_option(slr.capi.OPT_MF_MATCH_ALGO, args.match_algo)
        if args.match_group:
            c_.set_option(slr.capi.OPT_MF_BATCH_GROUP, args.match_group)
        if args.decode_group:
            c_.set_option(slr.capi.OPT_MF_BATCH_DECODE_GROUP, args.decode_group)
        if args.dma_shape >= 0:
            c_.set_option(slr.capi.OPT_RECT_DMA_SHAPE, args.dma_shape)
        if args.dma_depth >= 0:
            c_.set_option(slr.capi.OPT_RECT_DMA_DEPTH, args.dma_depth)
        if args.batch_streams:
            c_.set_option(slr.capi.OPT_BATCH_STREAMS, args.batch_streams)
        if args.debug_flags:
            c_.set_option(slr.capi.OPT_DEBUG_FLAGS, args.debug_flags)
        if args.rect_resident:
            c_.set_option(slr.capi.OPT_DEBUG_RECT_RESIDENT, args.rect_resident)
        if args.hybrid_one_pass:
            c_.set_option(slr.capi.OPT_HYBRID_ONE_PASS, 1)
        if args.eval_model == "x87":
            c_.set_option(slr.capi.OPT_EVAL_MODEL, 1)
        if rectify:
            install_maps(c_)
    _trace("contexts, calibration and maps installed")
    # F distinct synthetic stereo frames per rank (seeds 1234 + F rank + f), resident in HBM
    pitch = W + max(0, args.pitch_pad)
    stack = torch.zeros((F, 2, ppc, H, pitch), dtype=torch.uint8, device=dev)      # rows padded: see --pitch-pad
    for f in range(F):
        seed = 1234 + F * rank + f                         # config 4: 64 frames over 8 GPUs = seeds 1234 .. 1297
        if mode == "mf":
            rendered = synth.render_mf_stack(W, H, seed=seed, noise=2, device=dev)
        elif mode == "ge":
            rendered = synth.render_gray_stack(W, H, scan_w, seed=seed, noise=2, device=dev)
        elif mode == "hybrid":
            rendered = synth.render_hybrid_stack(W, H, scan_w, seed=seed, noise=2, device=dev)
        else:
            rendered = synth.render_gray_stack(W, H, scan_w, scan_h, seed=seed, noise=2, device=dev, rows=True)
        stack[f, :, :, :, :W] = rendered
        del rendered
    torch.cuda.synchronize()

    npix_ = float(W) * H
    nbuf = 2 if S == 1 else S
    oh, ow = (scan_h, scan_w) if mode == "gray" else (H, W)
    do_gather = world > 1 and args.gather == "step"
    final_gather = world > 1 and args.gather == "final"
    after_gather = world > 1 and args.gather == "after"
    sdist = importlib.import_module("structure-light-reconstructor_amd.dist")
    if do_gather or final_gather or after_gather:
        # The assembled cloud of every rank: [world * F] frames, rank r owning frames [r F, (r + 1) F) ("blocked": its shard is one
        # contiguous piece, which the batch entry point fills in ONE call -- no copy between the kernels and the collective -- and
        # one in-place RCCL all-gather per array assembles the whole).  The library path: dist.gather_point_clouds.
        comm = torch.cuda.Stream(device=dev)
        g_xyz = [torch.empty((world * F, oh, ow, 3), dtype=torch.float32, device=dev) for _ in range(nbuf)]
        g_has = [torch.empty((world * F, oh, ow), dtype=torch.uint8, device=dev) for _ in range(nbuf)]
        xyz = [sdist.local_slots(g, rank, world, "blocked") for g in g_xyz]
        has = [sdist.local_slots(g, rank, world, "blocked") for g in g_has]
        done_compute = [torch.cuda.Event() for _ in range(nbuf)]
        done_gather = [torch.cuda.Event() for _ in range(nbuf)]
    else:
        xyz = [torch.empty((F, oh, ow, 3), dtype=torch.float32, device=dev) for _ in range(nbuf)]
        has = [torch.empty((F, oh, ow), dtype=torch.uint8, device=dev) for _ in range(nbuf)]

    def gather(b):
        sdist.gather_point_clouds(xyz[b], has[b], world * F, out=(g_xyz[b], g_has[b]), assignment="blocked")

    def step(i):
        b = i % nbuf
        if do_gather:
            streams[i % S].wait_event(done_gather[b])   # buffer b is free again once its gather finished
        c_ = ctxs[i % S]
        for _ in range(PASSES):
            if mode == "mf":
                c_.reconstruct_mf_batch(stack, BLACK_THR, rectify, W=W, xyz=xyz[b], has=has[b])
            elif mode == "ge":
                c_.reconstruct_batch(slr.capi.MODE_GE, stack, BLACK_THR, 0, n_col_bits=ncol, scan_w=scan_w, rectify=rectify, W=W,
                                     xyz=xyz[b], has=has[b])
            elif mode == "hybrid":
                c_.reconstruct_hybrid_batch(stack, ncol, BLACK_THR, 0, scan_w, W=W, xyz=xyz[b], has=has[b])
            else:
                c_.reconstruct_batch(slr.capi.MODE_GRAY, stack, BLACK_THR, 0, n_col_bits=ncol, n_row_bits=nrow, scan_w=scan_w,
                                     scan_h=scan_h, rectify=False, W=W, xyz=xyz[b], has=has[b])
        if do_gather:
            done_compute[b].record(streams[i % S])
            comm.wait_event(done_compute[b])
            with torch.cuda.stream(comm):
                gather(b)
                done_gather[b].record(comm)

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    _trace("frames rendered")
    for i in range(args.warmup):
        step(i)
    sync_all()
    _trace("warm-up done")
    if args.profile:
        for c_ in ctxs:
            c_.set_option(slr.capi.OPT_PROFILE_STRIDE, max(1, args.profile_stride))
            c_.profile_enable(True)
            c_.profile_reset()
    sync_all()
    t0 = time.perf_counter()
    ctx.timer_begin()
    for i in range(args.steps):
        step(i)
    ev_ms = ctx.timer_end() if S == 1 else float("nan")
    if final_gather:                                    # assemble the final point cloud on every rank (north_star)
        b = (args.steps - 1) % nbuf
        done_compute[b].record(streams[(args.steps - 1) % S])
        comm.wait_event(done_compute[b])
        with torch.cuda.stream(comm):
            gather(b)
    sync_all()
    elapsed = time.perf_counter() - t0
    _trace("timed region done")
    gather_ms, gather_proof = None, None
    if after_gather or final_gather or do_gather:
        b = (args.steps - 1) % nbuf
        # the proof of the exchange: every rank's checksums of its LOCAL frames (taken before the gather where the gather is still
        # to come) travel by a second, tiny all-gather; every frame of every rank's assembled cloud must reproduce its owner's word
        mine = sdist.frame_checksums(xyz[b], has[b])
        torch.cuda.synchronize()
    if after_gather:                                    # the job's one exchange step, timed on its own (max over ranks)
        sync_all()
        tg = time.perf_counter()
        with torch.cuda.stream(comm):
            gather(b)
        sync_all()
        tt = torch.tensor([time.perf_counter() - tg], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        gather_ms = float(tt.item()) * 1e3
    if after_gather or final_gather or do_gather:
        try:
            nchk = sdist.verify_gathered(g_xyz[b], g_has[b], mine, world * F, assignment="blocked")
            okv = torch.tensor([1], dtype=torch.int64, device=dev)
            err = None
        except RuntimeError as e:
            nchk, okv, err = 0, torch.tensor([0], dtype=torch.int64, device=dev), str(e)
        dist.all_reduce(okv, op=dist.ReduceOp.MIN)      # every rank holds the whole cloud: every rank must agree
        gather_proof = {"frames_verified_on_every_rank": nchk, "all_ranks_ok": bool(int(okv.item()) == 1), "error": err,
                        "how": "per-frame 64-bit position-weighted checksums of each rank's local XYZ + mask, all-gathered and compared "
                               "with the assembled cloud on every rank (dist.verify_gathered)"}
        if not gather_proof["all_ranks_ok"]:
            raise SystemExit("bench.py: the assembled point cloud does not match its owners' checksums: %s" % (err,))
    # What was timed is what is checked: the batch output of the LAST timed step (groups of frames per launch) against the
    # single-frame entry (one pair-decode launch + one match launch per frame) for every distinct frame, by the library's per-pixel
    # position-weighted checksums; and the first call after an idle gap (what a one-scan-at-a-time host sees: the clocks ramp).
    self_check, first_call = None, None
    if args.self_check and not args.pmc_child and mode in ("mf", "hybrid"):
        b = (args.steps - 1) % nbuf
        c_ = ctxs[(args.steps - 1) % S]
        words_batch = c_.cloud_checksums(xyz[b][:F], has[b][:F])
        x1 = torch.empty((1, H, W, 3), dtype=torch.float32, device=dev)
        h1 = torch.empty((1, H, W), dtype=torch.uint8, device=dev)
        words_single = []
        for f in range(F):
            x1.fill_(float("nan")); h1.fill_(0x7B)
            torch.cuda.synchronize()
            if mode == "mf":
                c_.reconstruct_mf(stack[f, 0], stack[f, 1], BLACK_THR, rectify, W=W, xyz=x1[0], has=h1[0])
            else:
                c_.reconstruct_hybrid_batch(stack[f:f + 1], ncol, BLACK_THR, 0, scan_w, W=W, xyz=x1, has=h1)
            words_single.append(int(c_.cloud_checksums(x1, h1)[0]))
        bad = [f for f in range(F) if int(words_batch[f]) != words_single[f]]
        self_check = {"frames": F, "ok": not bad, "mismatching_frames": bad,
                      "matched_fraction_frame0": round(float(has[b][0].float().mean().item()), 4),
                      "how": "slr_cloud_checksums of the last timed step's batch output vs a single-frame call per distinct frame"}
        if bad:
            raise SystemExit("bench.py: the timed batch output differs from the single-frame entry on frames %s" % bad)
        if mode == "mf" and rank == 0:
            try:
                torch.cuda.synchronize()
                time.sleep(1.0)                              # idle: the clocks drop
                c_.timer_begin()
                c_.reconstruct_mf(stack[0, 0], stack[0, 1], BLACK_THR, rectify, W=W, xyz=x1[0], has=h1[0])
                cold = c_.timer_end() * 1e3
                c_.timer_begin()
                for _ in range(40):
                    c_.reconstruct_mf(stack[0, 0], stack[0, 1], BLACK_THR, rectify, W=W, xyz=x1[0], has=h1[0])
                warm_all = c_.timer_end() * 1e3
                c_.timer_begin()
                for _ in range(8):
                    c_.reconstruct_mf(stack[0, 0], stack[0, 1], BLACK_THR, rectify, W=W, xyz=x1[0], has=h1[0])
                warm = c_.timer_end() * 1e3 / 8
                first_call = {"first_call_after_idle_us": round(cold, 1), "single_frame_call_settled_us": round(warm, 1),
                              "mean_of_the_40_calls_in_between_us": round(warm_all / 40, 1),
                              "note": "slr_reconstruct_mf (one frame per call, device buffers) after 1 s of idle, then settled: the GPU's "
                                      "clocks ramp for ~60 ms after an idle gap (profiles/exp/r04/ramp.py) -- what a one-scan-at-a-time host sees"}
            except Exception as e:                              # never break the bench line
                first_call = {"error": repr(e)}
        del x1, h1
    prof = {}
    if args.profile:
        for c_ in ctxs:                                   # merge the per-context HIP-event profiles
            for name, (ms, n) in c_.profile().items():
                a = prof.get(name, (0.0, 0))
                prof[name] = (a[0] + ms, a[1] + n)
            c_.profile_enable(False)
            c_.set_option(slr.capi.OPT_PROFILE_STRIDE, 1)
    kernels_one_stream = None
    if args.profile and args.mode == "gray" and not args.pmc_child:
        # GRAY_ONLY batches pipeline their frames over two streams (SLR_OPT_BATCH_STREAMS = 2): the timed region's per-kernel
        # durations are those of kernels that share the GPU with the other frame's other half.  An untimed pass of two steps on ONE
        # stream gives each kernel's duration on its own.
        prof1 = {}
        for c_ in ctxs:
            c_.set_option(slr.capi.OPT_BATCH_STREAMS, 1)
            c_.set_option(slr.capi.OPT_PROFILE_STRIDE, 1)
            c_.profile_enable(True)
            c_.profile_reset()
        for i in range(2):
            step(i)
        sync_all()
        for c_ in ctxs:
            for name, (ms, n) in c_.profile().items():
                a = prof1.get(name, (0.0, 0))
                prof1[name] = (a[0] + ms, a[1] + n)
            c_.profile_enable(False)
            c_.set_option(slr.capi.OPT_BATCH_STREAMS, 2)
        kernels_one_stream = [{"name": k, "launches": n, "avg_us": round(ms / n * 1e3, 2)} for k, (ms, n) in sorted(prof1.items(), key=lambda kv: -kv[1][0])]
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # BOTH evaluation models in the one line (strict IEEE is the default, x87 is the model more likely to be the
    # reference's shipped Windows binary): the model of the timed region from the timed region itself, the other one from a short
    # leg of the same steps right here (same frames, same buffers, clocks still up), each with the single-frame entry's settled time
    models = None
    if mode == "mf" and world == 1 and rank == 0 and S == 1 and not args.pmc_child and args.both_models:
        models = {}
        timed_flag = 1 if args.eval_model == "x87" else 0
        for name_, flag in (("strict", 0), ("x87", 1)):
            try:
                ctx.set_option(slr.capi.OPT_EVAL_MODEL, flag)
                n_leg = args.steps if flag == timed_flag else max(3, min(args.steps, 8))
                for i in range(2):
                    step(i)
                ctx.set_option(slr.capi.OPT_PROFILE_STRIDE, max(1, args.profile_stride))
                ctx.profile_enable(True); ctx.profile_reset()
                sync_all()
                tl = time.perf_counter()
                for i in range(n_leg):
                    step(i)
                sync_all()
                leg = time.perf_counter() - tl
                pr = ctx.profile()
                ctx.profile_enable(False)
                b = (n_leg - 1) % nbuf
                wb = ctx.cloud_checksums(xyz[b][:F], has[b][:F])
                x1 = torch.empty((H, W, 3), dtype=torch.float32, device=dev)
                h1 = torch.empty((H, W), dtype=torch.uint8, device=dev)
                ws = []
                for f in range(F):
                    ctx.reconstruct_mf(stack[f, 0], stack[f, 1], BLACK_THR, rectify, W=W, xyz=x1, has=h1)
                    ws.append(int(ctx.cloud_checksums(x1[None], h1[None])[0]))
                for _ in range(24):
                    ctx.reconstruct_mf(stack[0, 0], stack[0, 1], BLACK_THR, rectify, W=W, xyz=x1, has=h1)
                ctx.timer_begin()
                for _ in range(8):
                    ctx.reconstruct_mf(stack[0, 0], stack[0, 1], BLACK_THR, rectify, W=W, xyz=x1, has=h1)
                single_us = ctx.timer_end() * 1e3 / 8
                del x1, h1
                ent = {"value": round(npix_ * F * PASSES * n_leg / leg / 1e6, 2), "ms_per_frame": round(leg / n_leg / (F * PASSES) * 1e3, 4),
                       "steps": n_leg, "single_frame_call_settled_us": round(single_us, 1),
                       "batch_equals_single_frame_calls": all(int(wb[f]) == ws[f] for f in range(F)),
                       "cloud_word_frame0": "%016x" % (int(wb[0]) & 0xFFFFFFFFFFFFFFFF)}
                for kn, short in (("slr_mf_rectify_decode_pair", "decode_us_per_frame"), ("slr_mf_match_triangulate", "match_us_per_frame")):
                    if kn in pr and pr[kn][1]:
                        ent[short] = round(pr[kn][0] / pr[kn][1] * 1e3, 2)
                if flag == timed_flag:
                    ent["timed_region"] = {"value": round(npix_ * F * PASSES * args.steps / elapsed / 1e6, 2),
                                           "ms_per_frame": round(elapsed / args.steps / (F * PASSES) * 1e3, 4)}
                models[name_] = ent
            except Exception as e:                          # never break the bench line
                models[name_] = {"error": repr(e)}
        try:
            ctx.set_option(slr.capi.OPT_EVAL_MODEL, timed_flag)
        except Exception:
            pass
        if all("cloud_word_frame0" in m for m in models.values()):
            models["clouds_differ_between_models"] = models["strict"]["cloud_word_frame0"] != models["x87"]["cloud_word_frame0"]
        models["note"] = ("`value` / `ms_per_step` of this line are the timed region under --eval-model %s; the other model ran the same steps "
                          "right after it (fewer of them).  A host that must match the reference's shipped MSVC2010 x87 binary sets "
                          "SLR_OPT_EVAL_MODEL = 1 (slr.h)" % args.eval_model)

    if args.pmc_child:
        for c_ in ctxs:
            c_.close()
        return
    npix = float(W) * H
    value = world * npix * F * PASSES * args.steps / elapsed / 1e6

    kernels, roofline = [], None
    for name, (ms, n) in sorted(prof.items(), key=lambda kv: -kv[1][0]):
        avg_ms = ms / n                                  # (the library's profiler counts a launch over g frames as g: time per FRAME)
        fpl = frames_per_launch(args, mode, rectify, F, name)
        entry = {"name": name, "launches": int(round(n / fpl)), "frames_per_launch": fpl, "avg_launch_us": round(avg_ms * fpl * 1e3, 2),
                 "avg_us": round(avg_ms * 1e3, 2), "total_ms": round(ms, 3)}
        if name in ALG_BYTES:
            gbs = ALG_BYTES[name] * npix / (avg_ms * 1e-3) / 1e9
            entry.update({"alg_bytes_per_px": ALG_BYTES[name], "achieved_GBs": round(gbs, 1),
                          "frac_hbm_peak": round(gbs / HBM_PEAK_GBS, 4)})
        kernels.append(entry)
    if kernels:
        k0 = kernels[0]                                  # dominant kernel by total time in the timed region
        traffic, tsrc, tdetail = None, None, None
        want_live = args.traffic == "live" or (args.traffic == "auto" and world == 1 and shutil.which("rocprofv3"))
        if rank == 0 and want_live:
            traffic, tdetail = live_traffic(args, k0["name"], k0["frames_per_launch"])
            tsrc = "live: rocprofv3 --pmc child runs of this command" if traffic else None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")   # PMC-derived HBM bytes per launch of an earlier profile run
        if traffic is None and args.traffic != "off" and os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                ent = tj.get(k0["name"]) or ({k: 2 * v if isinstance(v, int) else v for k, v in tj.get(k0["name"][:-5], {}).items()}
                                             if k0["name"].endswith("_pair") else None)   # pair launch = 2 x the per-camera launch
                if ent:
                    traffic, tsrc = ent.get("hbm_bytes_per_launch"), "file: profiles/pmc_traffic.json (" + str(ent.get("kernel")) + "), single-frame launches x frames per launch"
                    traffic = int(traffic * k0["frames_per_launch"]) if traffic else traffic
            except Exception:
                traffic = None
        roofline = {"kernel": k0["name"], "bound": "hbm", "achieved": k0.get("achieved_GBs"), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": k0.get("frac_hbm_peak"),
                    "frac_of_achievable_6300": round(k0["achieved_GBs"] / HBM_ACHIEVABLE_GBS, 4) if k0.get("achieved_GBs") else None,
                    "traffic": traffic, "traffic_source": tsrc, "traffic_detail": tdetail if isinstance(tdetail, dict) else None,
                    "traffic_over_algorithmic": round(traffic / (ALG_BYTES.get(k0["name"], 0) * npix * k0["frames_per_launch"]), 3)
                    if traffic and ALG_BYTES.get(k0["name"]) else None,
                    "frames_per_launch": k0["frames_per_launch"], "us_per_frame": k0["avg_us"],
                    "avg_launch_us": k0["avg_launch_us"], "alg_bytes_per_launch": ALG_BYTES.get(k0["name"], 0) * npix * k0["frames_per_launch"]}

    # outside the timed region: the other kernels of the path on the same frame (HIP-event timed, same stream):
    # the unfused phase-decode+unwrap kernel (north_star's named roofline target), the standalone remap and the
    # literal linear-sweep form of the match kernel
    extras = []
    if rank == 0 and args.profile and mode == "mf":
        reps = 10
        ph = torch.empty((H, W), dtype=torch.float32, device=dev)
        vd = torch.empty((H, W), dtype=torch.uint8, device=dev)
        tmp = plane3 = None
        if rectify:
            tmp = torch.empty((H, W), dtype=torch.uint8, device=dev)
            plane3 = stack[0, 0, 3, :, :W].contiguous()
        for timed in (False, True):                      # two untimed launches first (new output buffers, cold code), as
            if timed:                                    # the timed region has its --warmup steps
                ctx.synchronize()
                ctx.profile_enable(True)
                ctx.profile_reset()
            for _ in range(reps if timed else 2):
                ctx.mf_decode(stack[0, 0], BLACK_THR, W=W, phase=ph, valid=vd)
            if rectify:
                for _ in range(reps if timed else 2):
                    ctx.remap_u8(0, plane3, out=tmp)
        for name, (ms, n) in sorted(ctx.profile().items()):
            gbs = ALG_BYTES[name] * npix / (ms / n * 1e-3) / 1e9
            extras.append({"name": name, "launches": n, "avg_us": round(ms / n * 1e3, 2), "alg_bytes_per_px": ALG_BYTES[name],
                           "achieved_GBs": round(gbs, 1), "frac_hbm_peak": round(gbs / HBM_PEAK_GBS, 4)})
        ctx.profile_enable(False)

    # The timed region runs on the maps --maps names (default: the rig verged by 0.2 rad).  What the fused decode does on OTHER maps
    # -- the near-identity maps of rounds 1-3 and verged rigs of 0.1 / 0.2 / 0.3 rad, built by the repo's own stereoRectify +
    # slr_init_rectify_maps -- is measured here, outside the timed region: which form `auto` picks for them, how many tiles do not
    # fit it, the quad / wave read-mode histogram, and the kernel's time
    maps_info, realistic = None, []
    if rank == 0 and rectify and mode == "mf":
        try:
            maps_info = [ctx.rectify_info(cam) for cam in range(2)]
        except Exception as e:
            maps_info = {"error": repr(e)}
    if rank == 0 and world == 1 and args.profile and rectify and mode == "mf" and args.map_sweep and not args.pmc_child:
        sweep = [("near-identity", None, None), ("verged", 0.1, -0.10), ("verged", 0.2, -0.15), ("verged", 0.3, -0.20)]
        for kind, theta, k1 in sweep:
            if kind == "verged" and rig is not None and abs(theta - rig_theta) < 1e-9 and abs(k1 - rig_k1) < 1e-9:
                continue                                     # the maps of the timed region themselves
            if kind == "near-identity" and rig is None:
                continue
            ent = {"rig": "verged, theta %.1f rad, k1 %.2f" % (theta, k1) if kind == "verged" else
                          "near-identity (synth.make_rectify_maps: 0.2 deg roll, k1 -0.08 / -0.06)"}
            try:
                if kind == "verged":
                    synth.install_verged_maps(ctx, synth.make_verged_rig(W, H, theta, k1), W, H)
                else:
                    for cam in range(2):
                        mxy, mfr = synth.make_rectify_maps(W, H, cam, device=dev)
                        ctx.set_rectify_maps(cam, mxy, mfr)
                info = [ctx.rectify_info(cam) for cam in range(2)]
                # stream-event time of F back-to-back calls (the LDS-DMA form's fix-up launches for tiles that do not fit it are
                # part of a call; the per-kernel profiler would only see the main kernel)
                ph_sw = [torch.empty((H, W), dtype=torch.float32, device=dev) for _ in range(2)]
                for _ in range(12):                         # (the clocks have dropped while the maps were built: ~20 ms of work first)
                    ctx.reconstruct_mf_batch(stack, BLACK_THR, True, W=W, xyz=xyz[0], has=has[0])
                for f in range(F):
                    ctx.mf_rectify_decode_pair(stack[f % F, 0], stack[f % F, 1], BLACK_THR, W=W, want_valid=False, phase=ph_sw)
                ctx.synchronize()
                ctx.timer_begin()
                for f in range(F):
                    ctx.mf_rectify_decode_pair(stack[f, 0], stack[f, 1], BLACK_THR, W=W, want_valid=False, phase=ph_sw)
                single_us = ctx.timer_end() / F * 1e3
                # ... and as the timed region runs it: the batch entry (groups of frames per launch), main kernel by the profiler
                ctx.set_option(slr.capi.OPT_PROFILE_STRIDE, 1)
                ctx.profile_enable(True); ctx.profile_reset()
                for _ in range(3):
                    ctx.reconstruct_mf_batch(stack, BLACK_THR, True, W=W, xyz=xyz[0], has=has[0])
                pr = ctx.profile().get("slr_mf_rectify_decode_pair")
                ctx.set_option(slr.capi.OPT_PROFILE_STRIDE, args.profile_stride)
                ctx.profile_enable(False)
                per_frame_us = pr[0] / pr[1] * 1e3 if pr and pr[1] else single_us
                gbs = ALG_BYTES["slr_mf_rectify_decode_pair"] * npix / (per_frame_us * 1e-6) / 1e9
                ent.update({"form_selected": [i["mf_form"] for i in info], "nofit_tiles": [i["dma_nofit_tiles"] for i in info],
                            "dma_tiles": info[0]["dma_tiles"], "quads_by_class": [i["quads_by_class"] for i in info],
                            "waves_by_mode": [i["waves_by_mode"] for i in info], "lds_nofit_tiles": [i["lds_nofit_tiles"] for i in info],
                            "split_tile_extra_entries": [i["dma_extra_entries"] for i in info],
                            "decode_us_per_frame": round(per_frame_us, 1), "decode_us_single_frame_calls": round(single_us, 1),
                            "achieved_GBs": round(gbs, 1), "frac_hbm_peak": round(gbs / HBM_PEAK_GBS, 4)})
            except Exception as e:                          # never break the bench line
                ent["error"] = repr(e)
            realistic.append(ent)
        install_maps(ctx)                                    # back to the maps of the timed region

    _trace("traffic, extras and map sweep done")
    if roofline and isinstance(maps_info, list):
        roofline["form_selected"] = [i["mf_form"] for i in maps_info]
        roofline["nofit_tiles"] = [i["dma_nofit_tiles"] for i in maps_info]
        roofline["quads_by_class"] = [i["quads_by_class"] for i in maps_info]
        roofline["waves_by_mode"] = [i["waves_by_mode"] for i in maps_info]
    ceiling = hostio = None
    if rank == 0:
        try:
            ceiling, mix_rate, memcpy_rate = copy_ceiling(torch, dev, ctx)
        except Exception as e:                              # never break the bench line
            ceiling, mix_rate, memcpy_rate = None, None, repr(e)
        if roofline and roofline.get("achieved") and ceiling:
            roofline["stream_copy_this_box"] = round(ceiling, 1)    # float4 non-temporal copy kernel, one word per thread, 1 GiB (the guide's 6.29 TB/s figure)
            roofline["frac_of_stream_copy_this_box"] = round(roofline["achieved"] / ceiling, 4)
            if mix_rate:                                    # the same kernel with the decode's 20 : 4 read : write mix
                roofline["read_mix_20_4_this_box"] = round(mix_rate, 1)
                roofline["frac_of_read_mix_this_box"] = round(roofline["achieved"] / mix_rate, 4)
            roofline["library_memcpy_this_box"] = round(memcpy_rate, 1) if isinstance(memcpy_rate, float) else memcpy_rate
            roofline["what_bounds_it"] = ("co-limited, not at the copy rate: VALU 73-76 % busy (~127 lane-instructions per camera pixel) beside a "
                                          "fetch side that would take ~0.9 of the kernel's time at the read-mix rate; see DESIGN.md section 4")
        if world == 1 and args.host_io and mode == "mf":
            hostio = host_io_rate(np, torch, ctx, stack, W, H, rectify, slr, calib)

    cpu = None
    if rank == 0 and world == 1 and args.cpu_baseline and mode == "mf":
        rows = args.cpu_rows or H          # whole frame: ~8 s of single-thread CPU work at 4096x3000
        maps_cpu = None if not rectify else [ctx.get_rectify_maps(cam, W, H) for cam in range(2)]
        cpu = cpu_baseline(synth, W, H, np.ascontiguousarray(stack[0, :, :, :, :W].cpu().numpy()), maps_cpu, calib, rows)

    _trace("host-io and cpu baseline done")
    if rank == 0:
        out = {
            "metric": "Mpixels/s decode+unwrap+triangulate, 4096x3000 stereo, 1/2/4/8 GPU",
            "value": round(value, 2), "unit": "Mpix/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8 in, f32 phase/XYZ (f64 undistort + Q reprojection)", "data": "synthetic",
            "ms_per_frame": round(elapsed / args.steps / (F * PASSES) * 1e3, 4),
            "config": {"workload": {"mf": "%dx%d stereo, 3-freq x 4-step (14 planes/camera): rectify+decode+unwrap+match+triangulate",
                                    "ge": "%dx%d stereo, GRAY_EPI (Gray-code columns, 26 planes/camera at a 4096-wide projector): "
                                          "rectify+decode+code match+triangulate",
                                    "gray": "%dx%d stereo, GRAY_ONLY (Gray-code columns+rows, 44 planes/camera, 1280x1024 projector): "
                                            "decode+bucket scatter+ray-ray triangulation",
                                    "hybrid": "%dx%d stereo, Gray-code columns + 3-freq x 4-step fringes in one stack (38 planes/camera, "
                                              "BASELINE config 3): " + ("one-pass rectify+Gray decode+phase decode" if args.hybrid_one_pass else
                                                                       "fused rectify+Gray decode launch, fused rectify+phase decode launch "
                                                                       "(groups of frames)") + ", phase match+triangulate"}[mode] % (W, H)
                                   + "; a step = %d passes over %d distinct HBM-resident frames per GPU (%d frames; every pass re-reads its input from HBM)" % (PASSES, F, PASSES * F),
                       "maps": (rig_desc or "synthetic near-identity rectification maps (synth.make_rectify_maps: 0.2 deg roll, k1 -0.08 / "
                                "-0.06); verged rigs: see realistic_maps") if rectify else None,
                       "mode": mode, "frames_per_gpu_per_step": F * PASSES, "distinct_frames": F, "passes_per_step": PASSES, "rectify": rectify, "streams_per_gpu": S,
                       "debug_flags": args.debug_flags,
                       "stack_row_pitch_bytes": pitch, "hip_event_profile_stride": max(1, args.profile_stride) if args.profile else 0,
                       "parallelism": "frames sharded over %d GPU(s)%s" % (
                           world, ", RCCL all-gather of XYZ+mask after every step (overlapped)" if do_gather else
                           (", one RCCL all-gather of the final XYZ+mask inside the timed region" if final_gather else
                            (", one RCCL all-gather of the final XYZ+mask right after the timed steps (final_allgather_ms)"
                             if after_gather else "")))},
            "device": _device_info(torch, local),
            "collective_backend": None if collective is None else collective["backend"],
            "collective_ranks": None if collective is None else collective["ranks"],
            "collective": collective,
            "final_allgather_ms": None if gather_ms is None else round(gather_ms, 3),
            # the same job with its one exchange step counted: all units / (timed steps + the all-gather that follows them)
            "gather_inclusive_value": (round(world * npix * F * PASSES * args.steps / (elapsed + gather_ms * 1e-3) / 1e6, 2) if gather_ms is not None
                                       else (round(value, 2) if final_gather else None)),
            "gather_proof": gather_proof,
            "final_allgather_bytes_per_rank_out": None if gather_ms is None else int(world * F * oh * ow * 13),
            "stream_event_ms_per_step": round(ev_ms / args.steps, 4) if ev_ms == ev_ms else None,
            "eval_model": args.eval_model, "models": models, "self_check": self_check, "first_call_after_idle": first_call,
            # what ONE GPU holds for the job (every rank holds the same shapes); config 4 at N = 8: 2.75 GB of input, the assembled
            # cloud of all 64 frames (10.2 GB) once per output buffer
            "hbm_footprint_per_gpu": hbm_footprint(torch, dev, {
                "input_stack": stack.numel(),
                "xyz_and_mask_buffers": sum(t.numel() * t.element_size() for t in ((g_xyz + g_has) if (do_gather or final_gather or after_gather) else (xyz + has))),
                "phase_scratch_of_a_group": (8 * W * H * min(F, args.match_group or 8)) if mode in ("mf", "hybrid") else 0,
                "undistortion_tables": 12 * W * H if mode != "gray" else 0,
                "maps_and_tile_tables_both_cameras": 2 * (6 + 4) * W * H if rectify else 0}),
            "roofline": roofline,
            # the kernel north_star's ">= 60 % of the HBM roofline" target names: the UNFUSED phase-decode + unwrap kernel
            # (K2, 19 B/cam-px), measured live right after the timed region on the same frame and stream (10 launches)
            "north_star_kernel": next(({"kernel": e["name"], "bound": "hbm", "achieved": e["achieved_GBs"], "peak": HBM_PEAK_GBS,
                                       "unit": "GB/s", "frac": e["frac_hbm_peak"], "avg_launch_us": e["avg_us"],
                                       "alg_bytes_per_launch": e["alg_bytes_per_px"] * npix, "target_frac": 0.60}
                                      for e in extras if e["name"] == "slr_mf_decode"), None),
            "maps_of_the_timed_region": maps_info, "realistic_maps": realistic,
            "kernels": kernels, "kernels_one_stream_untimed_pass": kernels_one_stream, "other_kernels_untimed_region": extras, "cpu_baseline": cpu,
            "host_buffers_pcie_inclusive": hostio,
        }
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    for c_ in ctxs:
        c_.close()


if __name__ == "__main__":
    main()
