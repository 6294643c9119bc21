"""Second, independent CPU model of the hot path: a NumPy transcription of the compat-mode
pseudo-spec in SURVEY.md Appendix C (C1..C9) and §8c-3(ii), written without looking at oracle/slr_oracle.c.
It exists only so that `tests/test_np_crosscheck.py` can check that two separately written restatements of
Duke/mfreconstruct.cpp, Duke/reconstruct.cpp, Duke/utilities.cpp and cv::remap agree bit for bit.
TEST INFRASTRUCTURE ONLY -- nothing in the product imports this.

dtype discipline: every intermediate is an explicit np.float32 / np.float64 / np.int32 array; NumPy never gets
to pick a promotion.  atanf is taken as the correctly rounded f32 of the f64 arctangent (glibc's atanf agrees
with that for all 511 integer quotients; NumPy's own f32 arctan loop does NOT -- 182 of 511 differ by 1 ulp)."""
import numpy as np

f32, f64, i32 = np.float32, np.float64, np.int32
PI = f32(3.1416)                       # mfreconstruct.cpp:5
TWO_PI = f32(f32(2) * PI)
PI_3_2 = f32(f32(f32(3) * PI) / f32(2))
PI_1_2 = f32(PI / f32(2))


def shadow_mask(white, black, thr):    # C1  mfreconstruct.cpp:190-207
    return ((white.astype(f32) - black.astype(f32)) > f32(thr)).astype(np.uint8)


def wrapped_phase(G1, G2, G3, G4):     # C2  mfreconstruct.cpp:231-263
    G1, G2, G3, G4 = (g.astype(i32) for g in (G1, G2, G3, G4))
    n, d = G4 - G2, G1 - G3
    dd = np.where(d == 0, 1, d)
    q = (np.abs(n) // np.abs(dd)) * np.sign(n) * np.sign(dd)          # C division truncates toward zero
    at = np.arctan(q.astype(f64)).astype(f32)
    out = np.zeros(n.shape, f32)
    ok = np.ones(n.shape, bool)
    done = np.zeros(n.shape, bool)

    def put(cond, val):
        nonlocal done
        m = cond & ~done
        out[m] = val[m] if isinstance(val, np.ndarray) else val
        done |= m

    put((n == 0) & (d > 0), f32(0))
    put((n == 0) & (d < 0), PI)
    put((d == 0) & (n > 0), PI_3_2)
    put((d == 0) & (n < 0), PI_1_2)
    both = (d == 0) & (n == 0) & ~done
    ok[both] = False
    done |= both
    put(d < 0, (at + PI).astype(f32))
    put((d > 0) & (n > 0), (at + TWO_PI).astype(f32))
    put(np.ones(n.shape, bool), at)
    return out, ok


def heterodyne(P0, P1, P2):            # C3  mfreconstruct.cpp:265-268 (P are f64 holding f32 values)
    P0, P1, P2 = (p.astype(f64) for p in (P0, P1, P2))
    two_pi64 = f64(TWO_PI)
    P12 = np.where(P0 > P1, P0 - P1, P0 - P1 + two_pi64).astype(f32)
    P23 = np.where(P1 > P2, P1 - P2, P1 - P2 + two_pi64).astype(f32)
    dlt = (P12 - P23).astype(f32)
    P123 = np.where(P12 > P23, dlt, (dlt + TWO_PI).astype(f32)).astype(f32)
    return ((P123 / TWO_PI).astype(f32) * f32(255)).astype(f32)


def mf_decode(planes, thr):            # a1-a3; phase 0 where shadow-masked; Q5 pixels: P=0, phase kept, valid 0
    mask = shadow_mask(planes[0], planes[1], thr).astype(bool)
    P, ok = [], np.ones(mask.shape, bool)
    for c in range(3):
        p, o = wrapped_phase(*(planes[4 * c + 2 + s] for s in range(4)))
        P.append(p)
        ok &= o
    ph = heterodyne(*P)
    valid = mask & ok
    return np.where(mask, ph, f32(0)).astype(f32), valid.astype(np.uint8)


def gray_decode(planes, ncol, nrow, black_thr, white_thr, scan_w, scan_h):   # C4
    mask = shadow_mask(planes[0], planes[1], black_thr).astype(bool)
    err = np.zeros(mask.shape, bool)

    def word(first, nbits):
        nonlocal err
        g = np.zeros(mask.shape, np.int64)
        for c in range(nbits):
            v1 = planes[first + 2 * c].astype(f64)
            v2 = planes[first + 2 * c + 1].astype(f64)
            err |= np.abs(v1 - v2) < white_thr
            g = (g << 1) | (v1 > v2)
        b = g.copy()
        sh = 1
        while sh < 64:
            b ^= b >> sh
            sh *= 2
        return b

    x = word(2, ncol)
    y = np.zeros_like(x)
    if nrow:
        y = word(2 + 2 * ncol, nrow)
        err |= (y > scan_h) | (x > scan_w)
    else:
        err |= x > scan_w
    valid = mask & ~err
    return x.astype(i32), y.astype(i32), valid.astype(np.uint8)


def remap_u8(src, map_xy, map_frac):   # §8c-3(ii): 15-bit weights, +16384 >> 15, BORDER_CONSTANT 0
    H, W = src.shape
    sx = map_xy[..., 0].astype(np.int64)
    sy = map_xy[..., 1].astype(np.int64)
    f = map_frac.astype(np.int64) & 1023
    fx, fy = f & 31, f >> 5
    w = [(32 - fx) * (32 - fy) * 32, fx * (32 - fy) * 32, (32 - fx) * fy * 32, fx * fy * 32]
    pad = np.zeros((H + 2, W + 2), np.int64)
    pad[1:-1, 1:-1] = src

    def tap(dy, dx):
        yy, xx = sy + dy, sx + dx
        inside = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
        return np.where(inside, pad[np.clip(yy, -1, H) + 1, np.clip(xx, -1, W) + 1], 0)

    acc = tap(0, 0) * w[0] + tap(0, 1) * w[1] + tap(1, 0) * w[2] + tap(1, 1) * w[3]
    return np.clip((acc + 16384) >> 15, 0, 255).astype(np.uint8)


def undistort(px, py, fc, cc, k):      # C7  utilities.cpp:58-94 ; px,py f32 arrays
    fx, fy, cx, cy = f64(f32(fc[0])), f64(f32(fc[1])), f64(f32(cc[0])), f64(f32(cc[1]))
    k1, k2, p1, p2 = (f64(f32(v)) for v in k[:4])
    ifx, ify = f64(1.0) / fx, f64(1.0) / fy
    x0 = (px.astype(f64) - cx) * ifx
    y0 = (py.astype(f64) - cy) * ify
    x, y = x0.copy(), y0.copy()
    for _ in range(5):
        r2 = x * x + y * y
        ic = f64(1.0) / (f64(1.0) + ((f64(0.0) * r2 + k2) * r2 + k1) * r2)
        dx = f64(2.0) * p1 * x * y + p2 * (r2 + f64(2.0) * x * x)
        dy = p1 * (r2 + f64(2.0) * y * y) + f64(2.0) * p2 * x * y
        x = (x0 - dx) * ic
        y = (y0 - dy) * ic
    ox = ((x * fx).astype(f32).astype(f64) + cx).astype(f32)
    oy = ((y * fy).astype(f32).astype(f64) + cy).astype(f32)
    return ox, oy


def reproject(Q, a, b, d):             # p = Q.[a,b,d,1] in f64, X = (f32)(p.xyz / p.w)
    Q = np.asarray(Q, f64).reshape(4, 4)
    v = [a.astype(f64), b.astype(f64), d.astype(f64), np.ones_like(a, f64)]
    p = []
    for r in range(4):
        s = np.zeros_like(v[0])
        for c in range(4):
            s = s + Q[r, c] * v[c]
        p.append(s)
    return np.stack([(p[r] / p[3]).astype(f32) for r in range(3)], -1)


def apply_T(T, X):                     # f32 GEMM with f64 accumulation, narrowed once
    T = np.asarray(T, f32).reshape(3, 4)
    out = []
    for r in range(3):
        s = np.zeros(X.shape[:-1], f64)
        for c in range(3):
            s = s + f64(T[r, c]) * X[..., c].astype(f64)
        s = s + f64(T[r, 3]) * f64(1.0)
        out.append(s.astype(f32))
    return np.stack(out, -1)


def mf_triangulate(phL, vL, phR, vR, camL, camR, Q, T=None):   # C5 (camX = dict(fc,cc,k))
    H, W = phL.shape
    mk = np.full((H, W), -1, i32)
    for i in range(H):
        ks = np.nonzero(vR[i])[0]
        if ks.size == 0:
            continue
        pr = phR[i, ks]
        for j in np.nonzero(vL[i])[0]:
            hit = np.nonzero(np.abs((phL[i, j] - pr).astype(f32)) < f32(0.1))[0]
            if hit.size:
                mk[i, j] = ks[hit[0]]
    has = mk >= 0
    ii, jj = np.nonzero(has)
    xyz = np.zeros((H, W, 3), f32)
    if ii.size:
        kk = mk[ii, jj]
        ulx, uly = undistort(jj.astype(f32), ii.astype(f32), camL["fc"], camL["cc"], camL["k"])
        urx, _ = undistort(kk.astype(f32), ii.astype(f32), camR["fc"], camR["cc"], camR["k"])
        X = reproject(Q, ulx, uly, (ulx - urx).astype(f32))
        if T is not None:
            X = apply_T(T, X)
        xyz[ii, jj] = X
    return xyz, has.astype(np.uint8), mk


def ge_triangulate(cL, vL, cR, vR, Q, T=None, whiteL=None, whiteR=None):   # C6
    H, W = cL.shape
    mk = np.full((H, W), -1, i32)
    for i in range(H):
        ks = 0
        for j in range(W):
            if not vL[i, j]:
                continue
            cand = np.nonzero((vR[i, ks:] != 0) & (cR[i, ks:] == cL[i, j]))[0]
            if cand.size:
                mk[i, j] = ks + cand[0]
                ks = ks + cand[0]
    has = mk >= 0
    ii, jj = np.nonzero(has)
    xyz = np.zeros((H, W, 3), f32)
    color = None if whiteL is None else np.zeros((H, W), np.uint8)
    if ii.size:
        kk = mk[ii, jj]
        X = reproject(Q, jj.astype(f64), ii.astype(f64), (jj - kk).astype(f64))
        if T is not None:
            X = apply_T(T, X)
        xyz[ii, jj] = X
        if whiteL is not None:
            color[ii, jj] = ((whiteL[ii, jj].astype(i32) + whiteR[ii, kk].astype(i32)) // 2).astype(np.uint8)
    return xyz, has.astype(np.uint8), color, mk


def pointcloud_from_grid(xyz, has, scan_w, scan_h):   # C9 / Q11: camera (row i, col j) -> points[j][i]
    H, W = has.shape
    s = np.zeros((scan_h, scan_w, 3), f32)
    c = np.zeros((scan_h, scan_w), np.uint8)
    hh, ww = min(H, scan_w), min(W, scan_h)
    sub = has[:hh, :ww].astype(bool)
    s[:ww, :hh] = np.where(sub[..., None], xyz[:hh, :ww], f32(0)).transpose(1, 0, 2)
    c[:ww, :hh] = sub.T
    return s, c


# ---------------------------------------------------------------------------------------------------------
# cv::stereoRectify (flags = 0, alpha = -1) -- independent fp64 transcription of the published algorithm (OpenCV 2.4
# calib3d: cvStereoRectify / cvRodrigues2 / cvUndistortPoints / cvProjectPoints2), with LAPACK's SVD for the
# re-orthonormalisation step of Rodrigues(matrix).  Checks structure-light-reconstructor_amd/host/calib.cpp (row f4).
# ---------------------------------------------------------------------------------------------------------
def rodrigues_to_matrix(r):
    r = np.asarray(r, np.float64)
    theta = np.sqrt((r * r).sum())
    if theta < np.finfo(np.float64).eps:
        return np.eye(3)
    k = r / theta
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.cos(theta) * np.eye(3) + (1 - np.cos(theta)) * np.outer(k, k) + np.sin(theta) * K


def rodrigues_to_vector(R):
    U, _, Vt = np.linalg.svd(np.asarray(R, np.float64))
    R = U @ Vt                                           # nearest rotation
    v = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    s = np.sqrt((v * v).sum() * 0.25)
    c = np.clip((np.trace(R) - 1) * 0.5, -1.0, 1.0)
    theta = np.arccos(c)
    if s < 1e-5:
        if c > 0:
            return np.zeros(3)
        r = np.sqrt(np.maximum((np.diag(R) + 1) * 0.5, 0))
        r[1] *= -1 if R[0, 1] < 0 else 1
        r[2] *= -1 if R[0, 2] < 0 else 1
        if abs(r[0]) < abs(r[1]) and abs(r[0]) < abs(r[2]) and (R[1, 2] > 0) != (r[1] * r[2] > 0):
            r[2] = -r[2]
        return r * (theta / np.sqrt((r * r).sum()))
    return v * (theta / (2 * s))


def undistort_normalized(u, v, A, k):
    """cvUndistortPoints of one pixel with identity R and P: 5 fixed-point iterations, result narrowed to f32"""
    x0 = x = (u - A[0, 2]) / A[0, 0]
    y0 = y = (v - A[1, 2]) / A[1, 1]
    for _ in range(5):
        r2 = x * x + y * y
        icd = 1.0 / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2)
        dx = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x)
        dy = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y
        x, y = (x0 - dx) * icd, (y0 - dy) * icd
    return np.float64(np.float32(x)), np.float64(np.float32(y))


def stereo_rectify(M1, D1, M2, D2, R, T, nx, ny):
    """-> R1, R2, P1 (3x4), P2 (3x4), Q (4x4)"""
    M1, M2, R, T = (np.asarray(a, np.float64) for a in (M1, M2, R, T))
    D1, D2 = np.asarray(D1, np.float64).reshape(-1), np.asarray(D2, np.float64).reshape(-1)
    T = T.reshape(3)
    r_r = rodrigues_to_matrix(-0.5 * rodrigues_to_vector(R))
    t = r_r @ T
    idx = 0 if abs(t[0]) > abs(t[1]) else 1
    c, nt = t[idx], np.sqrt((t * t).sum())
    uu = np.zeros(3)
    uu[idx] = 1.0 if c > 0 else -1.0
    ww = np.cross(t, uu)
    nw = np.sqrt((ww * ww).sum())
    if nw > 0:
        ww = ww * (np.arccos(abs(c) / nt) / nw)
    wR = rodrigues_to_matrix(ww)
    R1, R2 = wR @ r_r.T, wR @ r_r
    t = R2 @ T
    fc_new = np.inf
    for A, D in ((M1, D1), (M2, D2)):
        fc = A[idx ^ 1, idx ^ 1]
        if D[0] < 0:
            fc = fc * (1 + D[0] * (nx * nx + ny * ny) / (4 * fc * fc))
        fc_new = min(fc_new, fc)
    cc = np.zeros((2, 2))
    for kcam, (A, D, Rk) in enumerate(((M1, D1, R1), (M2, D2, R2))):
        acc = np.zeros(2)
        for i in range(4):
            j = 0 if i < 2 else 1
            px, py = undistort_normalized(np.float64(np.float32((i % 2) * (nx - 1))), np.float64(np.float32(j * (ny - 1))), A, D)
            X = Rk @ np.array([px, py, 1.0])
            acc += np.array([np.float32(fc_new * X[0] / X[2]), np.float32(fc_new * X[1] / X[2])], np.float64)
        cc[kcam] = [(nx - 1) // 2 - acc[0] / 4, (ny - 1) // 2 - acc[1] / 4]
    if idx == 0:
        cc[:, 1] = (cc[0, 1] + cc[1, 1]) * 0.5
    else:
        cc[:, 0] = (cc[0, 0] + cc[1, 0]) * 0.5
    P1 = np.zeros((3, 4)); P2 = np.zeros((3, 4))
    P1[0, 0] = P1[1, 1] = P2[0, 0] = P2[1, 1] = fc_new
    P1[0, 2], P1[1, 2], P2[0, 2], P2[1, 2] = cc[0, 0], cc[0, 1], cc[1, 0], cc[1, 1]
    P1[2, 2] = P2[2, 2] = 1
    P2[idx, 3] = t[idx] * fc_new
    Q = np.zeros((4, 4))
    Q[0, 0] = Q[1, 1] = 1
    Q[0, 3], Q[1, 3], Q[2, 3] = -cc[0, 0], -cc[0, 1], fc_new
    Q[3, 2] = -1.0 / t[idx]
    Q[3, 3] = ((cc[0, 0] - cc[1, 0]) if idx == 0 else (cc[0, 1] - cc[1, 1])) / t[idx]
    return R1, R2, P1, P2, Q
