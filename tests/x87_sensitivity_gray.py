#!/usr/bin/env python
"""GRAY_ONLY under the evaluation model of the reference's own binary: Utilities::normalize
(utilities.cpp:19-28), pixelToImageSpace (:47-56) and line_lineIntersection (:399-425, the `fabsf(denom) < 0.1` rejection) restated
under the MSVC2010 x87 / fp:precise model (oracle/slr_oracle_x87.c) against the strict-IEEE oracle, on bench.py --mode gray's
scene (4096x3000 cameras, 1280x1024 projector, baseline 400, theta 0.6; the Gray decode itself is integer work and identical).
Counts, over the projector cells that hold pairs:

  * cells whose pair COUNT changes (a ray pair whose denom = a*c - b*b crosses 0.1 under one model only),
  * cells whose summed XYZ differs by more than north_star's 1e-4 relative (same count), and the largest such difference,
  * cells whose bits differ at all.

CPU only (oracle); test infrastructure, nothing here is product code.  The oracle's K6 is one lane walking every pair: a band of
projector rows keeps it to minutes (--cell-rows, centred; 0 = all 1024).

  python tests/x87_sensitivity_gray.py [--cell-rows 256] > profiles/r05_x87_sensitivity_gray.json
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle as O                                                          # noqa: E402
from util import calib_parts                                                # noqa: E402

BLACK = 40


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=4096)
    ap.add_argument("--height", type=int, default=3000)
    ap.add_argument("--scan-w", type=int, default=1280)
    ap.add_argument("--scan-h", type=int, default=1024)
    ap.add_argument("--cell-rows", type=int, default=256)
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--thetas", default="0.6,0.33,0.32,0.31",
                    help="vergence angles of the scenes: 0.6 rad is bench.py --mode gray's; sin^2 of ~0.32 rad is 0.1, the rejection "
                         "threshold of line_lineIntersection -- rays that close to parallel are where a rounding can flip the test")
    args = ap.parse_args()
    synth = importlib.import_module("structure-light-reconstructor_amd.synth")
    W, H, sw, sh = args.width, args.height, args.scan_w, args.scan_h
    O.build()
    t0 = time.time()
    st = synth.render_gray_stack(W, H, sw, sh, seed=args.seed, noise=2, rows=True).numpy()
    ncol, nrow = synth.gray_num_bits(sw), synth.gray_num_bits(sh)
    dec = [O.gray_decode(st[c], ncol, nrow, BLACK, 0, sw, sh) for c in range(2)]
    nr = args.cell_rows or sh
    j0 = sh // 2 - nr // 2
    # keep only the pixels whose projector row lies in the band: the buckets of the other cells stay empty
    for c in range(2):
        cy = dec[c][1]
        dec[c][2][(cy < j0) | (cy >= j0 + nr)] = 0
    offL, itL = O.gray_bucket(dec[0][0], dec[0][1], dec[0][2], sw, sh)
    offR, itR = O.gray_bucket(dec[1][0], dec[1][1], dec[1][2], sw, sh)
    t_prep = time.time() - t0
    scenes = []
    for theta in [float(x) for x in args.thetas.split(",")]:
        scenes.append(one_scene(synth, args, theta, offL, itL, offR, itR, j0, nr, t_prep))
    print(json.dumps({"what": __doc__.split("\n\n")[0], "scenes": scenes}, indent=1))


def one_scene(synth, args, theta, offL, itL, offR, itR, j0, nr, t_prep):
    W, H, sw, sh = args.width, args.height, args.scan_w, args.scan_h
    calib, _ = synth.make_calibration(W, H, baseline=400.0, theta=theta)    # theta 0.6: bench.py --mode gray's calibration
    camL, camR, _, T = calib_parts(O, calib)
    t1 = time.time()
    sx, sc = O.ray_triangulate(offL, itL, offR, itR, camL, camR, sw, sh, T)
    t2 = time.time()
    xx, xc = O.ray_triangulate_x87(offL, itL, offR, itR, camL, camR, sw, sh, T)
    t3 = time.time()
    lenL = np.diff(offL).reshape(sw, sh).T                                   # bucket ac = i * scan_h + j  ->  cell (j, i)
    lenR = np.diff(offR).reshape(sw, sh).T
    pairs = lenL.astype(np.int64) * lenR
    live = pairs > 0
    same_cnt = live & (sc == xc)
    d = np.abs(sx.astype(np.float64) - xx.astype(np.float64))
    m = np.maximum(np.abs(sx.astype(np.float64)), np.abs(xx.astype(np.float64)))
    rel = np.where(m > 0, d / np.where(m > 0, m, 1), 0.0).max(axis=2)
    bits = (sx.view(np.int32) != xx.view(np.int32)).any(axis=2)
    out = {
        "scene": {"camera": [W, H], "projector": [sw, sh], "projector_rows_in_the_band": [j0, j0 + nr], "seed": args.seed,
                  "calibration": "synth.make_calibration(baseline=400, theta=%g)%s" % (theta, ": bench.py --mode gray" if theta == 0.6 else "")},
        "cells_with_pairs": int(live.sum()), "ray_pairs": int(pairs[live].sum()),
        "pairs_accepted_strict": int(sc[live].astype(np.int64).sum()) if int(pairs.max()) < 256 else None,
        "pairs_accepted_x87": int(xc[live].astype(np.int64).sum()) if int(pairs.max()) < 256 else None,
        "cells_whose_pair_count_changes": int((live & (sc != xc)).sum()),
        "cells_same_count_bits_differ": int((same_cnt & bits).sum()),
        "cells_same_count_xyz_sum_beyond_1e-4_relative": int((same_cnt & (rel > 1e-4)).sum()),
        "largest_relative_difference_same_count": float(rel[same_cnt].max()) if same_cnt.any() else 0.0,
        "cells_accepting_no_pair_in_either_model": int((live & (sc == 0) & (xc == 0)).sum()),
        "seconds": {"render+decode+buckets (shared)": round(t_prep, 1), "strict": round(t2 - t1, 1), "x87": round(t3 - t2, 1)},
    }
    print("done: theta", theta, file=sys.stderr)
    return out


if __name__ == "__main__":
    main()
