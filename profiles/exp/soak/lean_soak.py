"""one-off randomized soak of the round-2 kernel forms against the oracle (run on the GPU box from the repo root):
K4 lean / K5 lean at random widths in their ranges, the fused MF and Gray LDS-DMA decodes on random smooth maps of random strength"""
import importlib, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
slr = importlib.import_module("structure-light-reconstructor_amd")
synth = importlib.import_module("structure-light-reconstructor_amd.synth")
import oracle as O
from util import calib_parts
O.build()
ctx = slr.Context(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
T_END = time.time() + (float(sys.argv[2]) if len(sys.argv) > 2 else 120.0)
n = {"k4": 0, "k5": 0, "mf": 0, "gray": 0}
eq = lambda a, b: np.array_equal(np.asarray(a).view(np.uint8), np.asarray(b).view(np.uint8))
while time.time() < T_END:
    # ---- K4 / K5 lean
    W = int(rng.choice([516, 1000, 1024, 2052, 3000, 3584, 4096])); H = int(rng.integers(1, 4))
    calib, _ = synth.make_calibration(max(W, 8), max(H, 8), with_T=bool(rng.integers(0, 2)))
    ctx.set_calibration(calib)
    camL, camR, Q, T = calib_parts(O, calib)
    q = float(rng.choice([0.03, 0.07, 0.25, 1.0]))
    phL = (rng.integers(-60, 500, (H, W)) * q).astype(np.float32); phR = (rng.integers(-60, 500, (H, W)) * q).astype(np.float32)
    if rng.integers(0, 2): phL.sort(axis=1); phR.sort(axis=1)
    vL = (rng.random((H, W)) < 0.9).astype(np.uint8); vR = (rng.random((H, W)) < 0.9).astype(np.uint8)
    e = O.mf_triangulate(phL, vL, phR, vR, camL, camR, Q, T)
    g = ctx.mf_triangulate(phL, vL, phR, vR)
    assert eq(g[2], e[2]) and eq(g[1], e[1]) and eq(g[0], e[0]), ("k4", W, H, q)
    n["k4"] += 1
    if W > 2048:
        nc = int(rng.choice([5, 300, 3000, 9000]))
        cL = rng.integers(0, nc, (H, W)).astype(np.int32); cR = rng.integers(0, nc, (H, W)).astype(np.int32)
        if rng.integers(0, 2): cL.sort(axis=1); cR.sort(axis=1)
        e = O.ge_triangulate(cL, vL, cR, vR, Q, T)
        g = ctx.ge_triangulate(cL, vL, cR, vR)
        assert eq(g[3], e[3]) and eq(g[1], e[1]) and eq(g[0], e[0]), ("k5", W, H, nc)
        n["k5"] += 1
    # ---- fused decodes on random maps
    W = int(rng.choice([256, 400 // 16 * 16, 640, 1024])); H = int(rng.integers(20, 120)); cam = int(rng.integers(0, 2))
    strength = float(rng.choice([0.3, 1.0, 2.0, 3.5]))
    mx, mf = synth.make_rectify_maps(W, H, cam, strength=strength)
    mxn, mfn = mx.numpy(), mf.numpy()
    ctx.set_option(slr.capi.OPT_DEBUG_RECT_RESIDENT, int(rng.choice([0, 8, 16])))
    ctx.set_rectify_maps(cam, mxn, mfn)
    st = synth.render_mf_stack(W, H, seed=int(rng.integers(1, 1 << 30)), noise=3)
    raw = st[cam].numpy()
    rect = np.stack([O.remap_u8(raw[p], mxn, mfn) for p in range(14)])
    eph, ev = O.mf_decode(rect, 40)
    ph, v = ctx.mf_decode(st[cam].cuda(), 40, rectify_cam=cam); ctx.synchronize()
    assert eq(ph.cpu().numpy(), eph) and eq(v.cpu().numpy(), ev), ("mf", W, H, strength)
    n["mf"] += 1
    sw = int(rng.choice([100, 600, 1280, 4096])); rows = bool(rng.integers(0, 2)); sh = int(rng.choice([90, 1024]))
    gs = synth.render_gray_stack(W, H, sw, sh if rows else None, seed=int(rng.integers(1, 1 << 30)), noise=3, rows=rows)
    nc_, nr_ = synth.gray_num_bits(sw), (synth.gray_num_bits(sh) if rows else 0)
    raw = gs[cam].numpy()
    rect = np.stack([O.remap_u8(raw[p], mxn, mfn) for p in range(raw.shape[0])])
    ex, ey, ev = O.gray_decode(rect, nc_, nr_, 40, 3, sw, sh if rows else 0)
    cx, cy, v = ctx.gray_decode(gs[cam].cuda(), nc_, nr_, 40, 3, sw, sh if rows else 0, rectify_cam=cam); ctx.synchronize()
    assert eq(cx.cpu().numpy(), ex) and eq(v.cpu().numpy(), ev) and (not rows or eq(cy.cpu().numpy(), ey)), ("gray", W, H, sw, rows, strength)
    n["gray"] += 1
print("soak ok", n)
