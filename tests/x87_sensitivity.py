#!/usr/bin/env python
"""What "parity unpinned" can cost: the multi-frequency path under the evaluation model of the reference's own binary.

The reference is an MSVC2010 32-bit Debug x87 build (Duke/Duke.pro:35-70); oracle/slr_oracle.c -- what every "bit-exact" of this
repo is measured against -- is strict IEEE.  oracle/slr_oracle_x87.c restates the same reference lines
(mfreconstruct.cpp:246-268, :295, :299; utilities.cpp:19-25, 51-53, 399-425) under the x87 / fp:precise model (53-bit stack,
rounding only at assignments, casts and calls).  This script runs BOTH models over the same u8 inputs and counts

  * phases whose bits differ, and how: last-place differences vs flips across the heterodyne cliff (KA3-edge: P12 ~ P23),
  * left pixels whose first-match column (mfreconstruct.cpp:289-295, |dphi| < 0.1, first k wins) changes, or whose match
    appears / disappears,
  * XYZ of pixels matched to the same column in both models that differ by more than north_star's 1e-4 relative,

on BASELINE config 2's scene (4096x3000, seed 1234; near-identity maps) and the three verged rigs of bench.py, plus a +-1-ulp
perturbation of the 511 atanf values under the strict model (the second unpinned ingredient: MSVCR100's atan vs glibc's).
CPU only (oracle + host mirror); ~10 minutes at full size.  Test infrastructure: nothing here is product code.

  python tests/x87_sensitivity.py [--width 4096 --height 3000] [--rows N] > profiles/r04_x87_sensitivity.json
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


def rel_diff(a, b):
    d = np.abs(a.astype(np.float64) - b.astype(np.float64))
    m = np.maximum(np.abs(a.astype(np.float64)), np.abs(b.astype(np.float64)))
    out = np.zeros_like(d)
    nz = m > 0
    out[nz] = d[nz] / m[nz]
    return out


def compare(name, base, other, rows):
    """base / other: dict(phase=[L, R], valid=[L, R], xyz, has, mk)"""
    res = {"variant": name}
    nph = ndiff = ncliff = nvalid_flip = 0
    max_small = 0.0
    for cam in range(2):
        v = (base["valid"][cam] != 0) & (other["valid"][cam] != 0)
        nvalid_flip += int(((base["valid"][cam] != 0) != (other["valid"][cam] != 0)).sum())
        a, b = base["phase"][cam][v], other["phase"][cam][v]
        dif = a.view(np.int32) != b.view(np.int32)
        nph += int(v.sum()); ndiff += int(dif.sum())
        ad = np.abs(a[dif].astype(np.float64) - b[dif].astype(np.float64))
        cliff = ad > 1.0                                                    # a flip across P12 ~ P23 moves the phase by ~255
        ncliff += int(cliff.sum())
        if (~cliff).any():
            max_small = max(max_small, float(ad[~cliff].max()))
    res.update({"valid_phases": nph, "phases_differing_bits": ndiff, "of_which_cliff_flips_gt_1": ncliff,
                "largest_non_cliff_abs_phase_difference": max_small, "validity_flips": nvalid_flip})
    r0, r1 = rows
    hb, ho = base["has"][r0:r1] != 0, other["has"][r0:r1] != 0
    kb, ko = base["mk"][r0:r1], other["mk"][r0:r1]
    both = hb & ho
    res.update({"rows_matched": [r0, r1], "left_pixels_matched_base": int(hb.sum()),
                "match_appears_or_disappears": int((hb != ho).sum()),
                "first_match_column_changes": int((both & (kb != ko)).sum())})
    same = both & (kb == ko)
    rd = rel_diff(base["xyz"][r0:r1][same], other["xyz"][r0:r1][same])
    res.update({"xyz_same_match_pixels": int(same.sum()),
                "xyz_components_differing_bits": int((base["xyz"][r0:r1][same].view(np.int32) != other["xyz"][r0:r1][same].view(np.int32)).sum()),
                "xyz_max_relative_difference_same_match": float(rd.max()) if rd.size else 0.0,
                "xyz_pixels_beyond_1e-4_same_match": int((rd.max(axis=-1) > 1e-4).sum()) if rd.size else 0,
                "pixels_whose_xyz_moves_beyond_1e-4_in_all": int((hb != ho).sum() + (both & (kb != ko)).sum()
                                                                  + ((rd.max(axis=-1) > 1e-4).sum() if rd.size else 0))})
    return res


def run_model(planes, camL, camR, Q, atab, x87_decode, x87_match, rows):
    ph, vd = [], []
    for cam in range(2):
        p, v = O.mf_decode_ev(planes[cam], BLACK, atab, x87_decode)
        ph.append(p); vd.append(v)
    xyz, has, mk = O.mf_triangulate_ev(ph[0], vd[0], ph[1], vd[1], camL, camR, Q, x87_match, rows=rows)
    return {"phase": ph, "valid": vd, "xyz": xyz, "has": has, "mk": mk}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=4096)
    ap.add_argument("--height", type=int, default=3000)
    ap.add_argument("--rows", type=int, default=0, help="rows of the O(W^2) match (centred; 0 = all)")
    ap.add_argument("--seed", type=int, default=1234)
    args = ap.parse_args()
    synth = importlib.import_module("structure-light-reconstructor_amd.synth")
    W, H = args.width, args.height
    nrows = args.rows or H
    rows = (H // 2 - nrows // 2, H // 2 - nrows // 2 + nrows)
    O.build()
    st = synth.render_mf_stack(W, H, seed=args.seed, noise=2).numpy()       # [2][14][H][W] u8: config 2's scene
    tab_glibc, tab_msvc = O.atan_table(0), O.atan_table(1)
    out = {"what": __doc__.split("\n\n")[0], "size": [W, H], "seed": args.seed,
           "atanf_entries_glibc_vs_float_of_double_atan": int((tab_glibc.view(np.int32) != tab_msvc.view(np.int32)).sum()),
           "scenes": []}
    scenes = [("near-identity maps (synth.make_rectify_maps)", None, None), ("verged rig 0.1 rad, k1 -0.10", 0.1, -0.10),
              ("verged rig 0.2 rad, k1 -0.15", 0.2, -0.15), ("verged rig 0.3 rad, k1 -0.20", 0.3, -0.20)]
    for name, theta, k1 in scenes:
        t0 = time.time()
        if theta is None:
            calib, _ = synth.make_calibration(W, H)
            maps = [tuple(m.numpy() for m in synth.make_rectify_maps(W, H, cam)) for cam in range(2)]
        else:
            rig = synth.make_verged_rig(W, H, theta, k1)
            calib = rig["calib"]
            maps = [O.init_undistort_rectify_map(rig["M"][cam], rig["D"][cam], rig["R1" if cam == 0 else "R2"],
                                                 rig["P1" if cam == 0 else "P2"], W, H) for cam in range(2)]
        camL, camR, Q, T = calib_parts(O, calib)
        planes = [np.stack([O.remap_u8(st[cam, p], maps[cam][0], maps[cam][1]) for p in range(14)]) for cam in range(2)]
        base = run_model(planes, camL, camR, Q, tab_glibc, 0, 0, rows)
        # the strict _ev path must BE the oracle (same bits), or nothing below means anything
        for cam in range(2):
            p, v = O.mf_decode(planes[cam], BLACK)
            assert np.array_equal(p.view(np.int32), base["phase"][cam].view(np.int32)) and np.array_equal(v, base["valid"][cam])
        ent = {"scene": name, "variants": []}
        ent["variants"].append(compare("x87 model: decode (P, P123, phase) + match predicate + disparity",
                                       base, run_model(planes, camL, camR, Q, tab_msvc, 1, 1, rows), rows))
        ent["variants"].append(compare("x87 model in the decode only (strict match)",
                                       base, run_model(planes, camL, camR, Q, tab_msvc, 1, 0, rows), rows))
        ent["variants"].append(compare("x87 model in the match predicate + disparity only (strict decode)",
                                       base, run_model(planes, camL, camR, Q, tab_glibc, 0, 1, rows), rows))
        for mode, label in ((2, "strict, every |atanf| one ulp up"), (3, "strict, every |atanf| one ulp down"),
                            (4, "strict, random -1/0/+1 ulp per |q| (seed 4)"), (5, "strict, random -1/0/+1 ulp per |q| (seed 5)")):
            ent["variants"].append(compare(label, base, run_model(planes, camL, camR, Q, O.atan_table(mode), 0, 0, rows), rows))
        ent["seconds"] = round(time.time() - t0, 1)
        out["scenes"].append(ent)
        print("done:", name, ent["seconds"], "s", file=sys.stderr)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
