"""ctypes loader for the CPU ORACLE (test infrastructure only).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.  It is a
checker, never the thing shipped or measured as the product.  PARITY UNPINNED -- see slr_oracle.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# SLR_ORACLE_SANITIZED=1 (set by tests/test_oracle_sanitized.py for its child process, which also preloads the sanitizer
# runtimes): load the AddressSanitizer + UBSan builds of the two libraries (oracle/san/, `make -C oracle sanitized`)
_SAN = os.environ.get("SLR_ORACLE_SANITIZED") == "1"
_LIB_DIR = os.path.join(_HERE, "san") if _SAN else _HERE
_LIB_PATH = os.path.join(_LIB_DIR, "libslr_oracle.so")
_LIT_PATH = os.path.join(_LIB_DIR, "libslr_literal.so")
_lib = None
_lit = None

MF_PLANES = 14
PI_F = np.float32(3.1416)


class Camera(C.Structure):
    """Mirror of slro_camera (virtualcamera.h:27-37 fields the path reads)."""
    _fields_ = [("fc", C.c_float * 2), ("cc", C.c_float * 2), ("k", C.c_float * 5),
                ("R", C.c_float * 9), ("t", C.c_float * 3)]

    @staticmethod
    def make(fc, cc, k, R=None, t=None):
        cam = Camera()
        cam.fc[:] = [float(v) for v in fc]
        cam.cc[:] = [float(v) for v in cc]
        kk = list(k) + [0.0] * (5 - len(k))
        cam.k[:] = [float(v) for v in kk]
        R = np.eye(3) if R is None else np.asarray(R)
        t = np.zeros(3) if t is None else np.asarray(t)
        cam.R[:] = [float(v) for v in R.reshape(-1)]
        cam.t[:] = [float(v) for v in t.reshape(-1)]
        return cam


def build(force=False):
    """Compile the oracle with gcc (oracle/Makefile)."""
    srcs = [os.path.join(_HERE, f) for f in ("slr_oracle.c", "slr_oracle_x87.c", "slr_oracle.h", "slr_literal.cpp")]
    newest = max(os.path.getmtime(f) for f in srcs if os.path.exists(f))
    if (not force and os.path.exists(_LIB_PATH) and os.path.exists(_LIT_PATH)
            and min(os.path.getmtime(_LIB_PATH), os.path.getmtime(_LIT_PATH)) >= newest):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-B"] + (["sanitized"] if _SAN else []), stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.slro_gray_num_bits.restype = C.c_int
        _lib.slro_gen_graycodes.restype = C.c_int
        _lib.slro_gray_to_dec.restype = C.c_int
        _lib.slro_wrapped_phase.restype = C.c_int
        _lib.slro_heterodyne.restype = C.c_float
        _lib.slro_line_line_intersection.restype = C.c_int
        _lib.slro_wrapped_phase_ev.restype = C.c_int
        _lib.slro_heterodyne_ev.restype = C.c_float
        _lib.slro_line_line_intersection_x87.restype = C.c_int
        _lib.slro_x87_quotient_mismatches.restype = C.c_long
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _planes_ptrs(planes):
    """planes: ndarray [N][H][pitch] u8 (C-contiguous) -> array of N row-0 pointers + pitch."""
    assert planes.dtype == np.uint8 and planes.ndim == 3 and planes.flags.c_contiguous
    n = planes.shape[0]
    arr = (C.c_void_p * n)()
    for i in range(n):
        arr[i] = planes[i].ctypes.data
    return arr, planes.shape[2]


# ---- encoders ---------------------------------------------------------------------------------
def gray_num_bits(n):
    return lib().slro_gray_num_bits(C.c_int(n))


def gen_multifreq(projW, projH):
    out = np.empty((MF_PLANES, projH, projW), np.uint8)
    lib().slro_gen_multifreq(C.c_int(projW), C.c_int(projH), _p(out))
    return out


def gen_graycodes(scanW, scanH, use_epi):
    ncol, nrow = gray_num_bits(scanW), gray_num_bits(scanH)
    n = 2 + 2 * ncol + (0 if use_epi else 2 * nrow)
    out = np.empty((n, scanH, scanW), np.uint8)
    r = lib().slro_gen_graycodes(C.c_int(scanW), C.c_int(scanH), C.c_int(1 if use_epi else 0), _p(out))
    assert r == n
    return out


def gray_to_dec(bits):
    b = np.ascontiguousarray(bits, np.uint8)
    return lib().slro_gray_to_dec(_p(b), C.c_int(b.size))


# ---- remap -------------------------------------------------------------------------------------
def remap_u8(src, map_xy, map_frac):
    src = np.ascontiguousarray(src, np.uint8)
    H, W = src.shape
    assert map_xy.shape == (H, W, 2) and map_xy.dtype == np.int16 and map_xy.flags.c_contiguous
    assert map_frac.shape == (H, W) and map_frac.dtype == np.uint16 and map_frac.flags.c_contiguous
    dst = np.empty((H, W), np.uint8)
    lib().slro_remap_u8(_p(src), C.c_int(W), C.c_int(W), C.c_int(H), _p(map_xy), _p(map_frac),
                        _p(dst), C.c_int(W))
    return dst


def init_undistort_rectify_map(M, D, R, P, W, H):
    M = np.ascontiguousarray(M, np.float64).reshape(9)
    D = np.ascontiguousarray(D, np.float64).reshape(5)
    R = np.ascontiguousarray(R, np.float64).reshape(9)
    P = np.ascontiguousarray(P, np.float64).reshape(12)
    xy = np.empty((H, W, 2), np.int16)
    fr = np.empty((H, W), np.uint16)
    lib().slro_init_undistort_rectify_map(_p(M), _p(D), _p(R), _p(P), C.c_int(W), C.c_int(H),
                                          _p(xy), _p(fr))
    return xy, fr


# ---- decode ------------------------------------------------------------------------------------
def mf_decode(planes, black_thr, W=None):
    """planes [14][H][pitch] u8 -> (phase [H][W] f32, valid [H][W] u8)."""
    ptrs, pitch = _planes_ptrs(planes)
    H = planes.shape[1]
    W = pitch if W is None else W
    phase = np.empty((H, W), np.float32)
    valid = np.empty((H, W), np.uint8)
    lib().slro_mf_decode(ptrs, C.c_int(pitch), C.c_int(W), C.c_int(H), C.c_int(black_thr),
                         _p(phase), _p(valid))
    return phase, valid


def mfn_decode_f64(planes, n_freq, n_step, black_thr, W=None):
    """BUILD EXTENSION model (no reference counterpart): planes [2+F*N][H][pitch] float16 -> (phase f64, valid u8)."""
    planes = np.ascontiguousarray(planes).view(np.uint16)
    n, H, pitch = planes.shape
    assert n == 2 + n_freq * n_step
    W = pitch if W is None else W
    ptrs = (C.c_void_p * n)(*[planes[i].ctypes.data for i in range(n)])
    phase = np.empty((H, W), np.float64)
    valid = np.empty((H, W), np.uint8)
    lib().slro_mfn_decode_f64(ptrs, C.c_int(n_freq), C.c_int(n_step), C.c_int(pitch), C.c_int(W), C.c_int(H),
                              C.c_double(black_thr), _p(phase), _p(valid))
    return phase, valid


def mfn_rect_decode_f64(planes, n_freq, n_step, black_thr, map_xy, map_frac, W=None, rows=None):
    """BUILD EXTENSION model of slr_mfn_rectify_decode: planes [2+F*N][H][pitch] float16 (raw camera images) through cv::remap's
    geometry with an exact (f64) bilinear sample -> (phase f64 [H][W], valid u8), rows = (r0, r1) only (default: all)."""
    planes = np.ascontiguousarray(planes).view(np.uint16)
    n, H, pitch = planes.shape
    assert n == 2 + n_freq * n_step
    W = pitch if W is None else W
    r0, r1 = (0, H) if rows is None else rows
    ptrs = (C.c_void_p * n)(*[planes[i].ctypes.data for i in range(n)])
    phase = np.zeros((H, W), np.float64)
    valid = np.zeros((H, W), np.uint8)
    mx, mf = np.ascontiguousarray(map_xy, np.int16), np.ascontiguousarray(map_frac, np.uint16)
    assert mx.shape == (H, W, 2) and mf.shape == (H, W)
    lib().slro_mfn_rect_decode_f64(ptrs, C.c_int(n_freq), C.c_int(n_step), C.c_int(pitch), C.c_int(W), C.c_int(H),
                                   C.c_double(black_thr), _p(mx), _p(mf), C.c_int(r0), C.c_int(r1), _p(phase), _p(valid))
    return phase, valid


def wrapped_phase(G1, G2, G3, G4):
    P = C.c_float(0)
    ok = lib().slro_wrapped_phase(C.c_int(G1), C.c_int(G2), C.c_int(G3), C.c_int(G4), C.byref(P))
    return bool(ok), np.float32(P.value)


def heterodyne(P):
    a = (C.c_double * 3)(*[float(v) for v in P])
    return np.float32(lib().slro_heterodyne(a))


def gray_decode(planes, n_col_bits, n_row_bits, black_thr, white_thr, scan_w, scan_h, W=None):
    ptrs, pitch = _planes_ptrs(planes)
    H = planes.shape[1]
    W = pitch if W is None else W
    cx = np.empty((H, W), np.int32)
    cy = np.empty((H, W), np.int32)
    valid = np.empty((H, W), np.uint8)
    lib().slro_gray_decode(ptrs, C.c_int(n_col_bits), C.c_int(n_row_bits), C.c_int(pitch),
                           C.c_int(W), C.c_int(H), C.c_int(black_thr), C.c_int(white_thr),
                           C.c_int(scan_w), C.c_int(scan_h), _p(cx), _p(cy), _p(valid))
    return cx, cy, valid


# ---- triangulation -----------------------------------------------------------------------------
def undistort_point(px, py, cam):
    ox, oy = C.c_float(0), C.c_float(0)
    lib().slro_undistort_point(C.c_float(px), C.c_float(py), C.byref(cam), C.byref(ox), C.byref(oy))
    return np.float32(ox.value), np.float32(oy.value)


def _q(Q):
    return np.ascontiguousarray(Q, np.float64).reshape(16)


def _t(T):
    return None if T is None else np.ascontiguousarray(T, np.float32).reshape(12)


def mf_triangulate(phaseL, validL, phaseR, validR, camL, camR, Q, T=None, rows=None):
    H, W = phaseL.shape
    xyz = np.zeros((H, W, 3), np.float32)
    has = np.zeros((H, W), np.uint8)
    mk = np.full((H, W), -1, np.int32)
    Qa, Ta = _q(Q), _t(T)
    r0, r1 = (0, H) if rows is None else rows
    lib().slro_mf_triangulate_rows(_p(np.ascontiguousarray(phaseL, np.float32)), _p(np.ascontiguousarray(validL)),
                                   _p(np.ascontiguousarray(phaseR, np.float32)), _p(np.ascontiguousarray(validR)),
                                   C.c_int(W), C.c_int(H), C.c_int(r0), C.c_int(r1),
                                   C.byref(camL), C.byref(camR), _p(Qa), _p(Ta), _p(xyz), _p(has), _p(mk))
    return xyz, has, mk


def literal_lib():
    """the literal-cost model of the reference's MF path (slr_literal.cpp): the oracle's arithmetic on the reference's data
    structures -- a baseline / test helper like everything else in this package"""
    global _lit
    if _lit is None:
        if not os.path.exists(_LIT_PATH):
            build()
        _lit = C.CDLL(_LIT_PATH)
    return _lit


def literal_mf(planesL, planesR, black_thr, camL, camR, Q, T=None, rows=None, row_step=1):
    """planes: [14][H][W] u8, already rectified.  Decodes both cameras with per-pixel heap vectors and by-value matrix headers,
    matches + triangulates image rows rows[0], rows[0] + row_step, ... < rows[1] (default: all).  Returns (xyz, has, t_decode_s,
    t_triangulate_s)."""
    pl, pitch = _planes_ptrs(planesL)
    pr, pitch2 = _planes_ptrs(planesR)
    H, W = planesL.shape[1], planesL.shape[2]
    assert pitch == pitch2 == W and planesL.shape[0] == MF_PLANES and planesR.shape == planesL.shape
    xyz = np.zeros((H, W, 3), np.float32)
    has = np.zeros((H, W), np.uint8)
    t = np.zeros(2, np.float64)
    r0, r1 = (0, H) if rows is None else rows
    Qa, Ta = _q(Q), _t(T)
    literal_lib().slro_literal_mf(pl, pr, C.c_int(pitch), C.c_int(W), C.c_int(H), C.c_int(black_thr), C.byref(camL), C.byref(camR),
                                  _p(Qa), _p(Ta), C.c_int(r0), C.c_int(r1), C.c_int(row_step), _p(xyz), _p(has), _p(t))
    return xyz, has, float(t[0]), float(t[1])


def ge_triangulate(codeL, validL, codeR, validR, Q, T=None, whiteL=None, whiteR=None):
    H, W = codeL.shape
    xyz = np.zeros((H, W, 3), np.float32)
    has = np.zeros((H, W), np.uint8)
    mk = np.full((H, W), -1, np.int32)
    color = np.zeros((H, W), np.uint8) if whiteL is not None else None
    Qa, Ta = _q(Q), _t(T)
    wl = None if whiteL is None else np.ascontiguousarray(whiteL, np.uint8)
    wr = None if whiteR is None else np.ascontiguousarray(whiteR, np.uint8)
    lib().slro_ge_triangulate(_p(np.ascontiguousarray(codeL, np.int32)), _p(np.ascontiguousarray(validL)),
                              _p(np.ascontiguousarray(codeR, np.int32)), _p(np.ascontiguousarray(validR)),
                              C.c_int(W), C.c_int(H), _p(Qa), _p(Ta), _p(wl), _p(wr),
                              _p(xyz), _p(has), _p(color), _p(mk))
    return xyz, has, color, mk


def gray_bucket(code_x, code_y, valid, scan_w, scan_h):
    H, W = code_x.shape
    nb = scan_w * scan_h
    offsets = np.zeros(nb + 1, np.int32)
    items = np.zeros(max(1, H * W), np.uint32)
    lib().slro_gray_bucket(_p(np.ascontiguousarray(code_x, np.int32)), _p(np.ascontiguousarray(code_y, np.int32)),
                           _p(np.ascontiguousarray(valid)), C.c_int(W), C.c_int(H),
                           C.c_int(scan_w), C.c_int(scan_h), _p(offsets), _p(items))
    return offsets, items[:offsets[-1]].copy()


def ray_triangulate(offL, itemsL, offR, itemsR, camL, camR, scan_w, scan_h, T=None):
    xyz = np.zeros((scan_h, scan_w, 3), np.float32)
    cnt = np.zeros((scan_h, scan_w), np.uint8)
    Ta = _t(T)
    iL = np.ascontiguousarray(itemsL, np.uint32) if itemsL.size else np.zeros(1, np.uint32)
    iR = np.ascontiguousarray(itemsR, np.uint32) if itemsR.size else np.zeros(1, np.uint32)
    lib().slro_ray_triangulate(_p(offL), _p(iL), _p(offR), _p(iR), C.byref(camL), C.byref(camR), _p(Ta),
                               C.c_int(scan_w), C.c_int(scan_h), _p(xyz), _p(cnt))
    return xyz, cnt


def line_line_intersection(p1, v1, p2, v2):
    a = [np.ascontiguousarray(v, np.float32) for v in (p1, v1, p2, v2)]
    out = np.zeros(3, np.float32)
    ok = lib().slro_line_line_intersection(_p(a[0]), _p(a[1]), _p(a[2]), _p(a[3]), _p(out))
    return bool(ok), out


def cam2world(cam, p):
    a = np.ascontiguousarray(p, np.float32).copy()
    lib().slro_cam2world(C.byref(cam), _p(a))
    return a


def pointcloud_from_grid(xyz, has, scan_w, scan_h, color=None):
    H, W = has.shape
    s = np.zeros((scan_h, scan_w, 3), np.float32)
    c = np.zeros((scan_h, scan_w), np.uint8)
    col = np.zeros((scan_h, scan_w), np.uint8) if color is not None else None
    lib().slro_pointcloud_from_grid(_p(np.ascontiguousarray(xyz, np.float32)), _p(np.ascontiguousarray(has)),
                                    _p(None if color is None else np.ascontiguousarray(color)),
                                    C.c_int(W), C.c_int(H), C.c_int(scan_w), C.c_int(scan_h),
                                    _p(s), _p(c), _p(col))
    return s, c, col


def pointcloud_get(pc_sum, pc_count):
    n = pc_count.size
    out = np.zeros((n, 3), np.float32)
    lib().slro_pointcloud_get(_p(np.ascontiguousarray(pc_sum, np.float32)), _p(np.ascontiguousarray(pc_count)),
                              C.c_int(n), _p(out))
    return out.reshape(pc_count.shape + (3,))


# ---- second evaluation model: the reference's MSVC2010 x87 build (slr_oracle_x87.c; sensitivity analysis) -----------------------
def atan_table(mode):
    """511 floats: atan of the integer quotients -255 .. 255.  mode 0 glibc atanf, 1 MSVC x86's (float)atan((double)q), 2 / 3
    every |entry| one ulp up / down, >= 4 seeded random -1 / 0 / +1 ulp."""
    tab = np.empty(511, np.float32)
    lib().slro_atan_table(C.c_int(mode), _p(tab))
    return tab


def wrapped_phase_ev(G1, G2, G3, G4, atab, x87):
    P = C.c_double(0)
    ok = lib().slro_wrapped_phase_ev(C.c_int(G1), C.c_int(G2), C.c_int(G3), C.c_int(G4), _p(atab), C.c_int(x87), C.byref(P))
    return bool(ok), float(P.value)


def heterodyne_ev(P, x87):
    a = (C.c_double * 3)(*[float(v) for v in P])
    return np.float32(lib().slro_heterodyne_ev(a, C.c_int(x87)))


def mf_decode_ev(planes, black_thr, atab, x87, W=None):
    ptrs, pitch = _planes_ptrs(planes)
    H = planes.shape[1]
    W = pitch if W is None else W
    phase = np.empty((H, W), np.float32)
    valid = np.empty((H, W), np.uint8)
    lib().slro_mf_decode_ev(ptrs, C.c_int(pitch), C.c_int(W), C.c_int(H), C.c_int(black_thr), _p(atab), C.c_int(x87),
                            _p(phase), _p(valid))
    return phase, valid


def mf_triangulate_ev(phaseL, validL, phaseR, validR, camL, camR, Q, x87, T=None, rows=None):
    H, W = phaseL.shape
    xyz = np.zeros((H, W, 3), np.float32)
    has = np.zeros((H, W), np.uint8)
    mk = np.full((H, W), -1, np.int32)
    Qa, Ta = _q(Q), _t(T)
    r0, r1 = (0, H) if rows is None else rows
    lib().slro_mf_triangulate_rows_ev(_p(np.ascontiguousarray(phaseL, np.float32)), _p(np.ascontiguousarray(validL)),
                                      _p(np.ascontiguousarray(phaseR, np.float32)), _p(np.ascontiguousarray(validR)),
                                      C.c_int(W), C.c_int(H), C.c_int(r0), C.c_int(r1), C.byref(camL), C.byref(camR),
                                      _p(Qa), _p(Ta), C.c_int(x87), _p(xyz), _p(has), _p(mk))
    return xyz, has, mk


def line_line_intersection_x87(p1, v1, p2, v2):
    a = [np.ascontiguousarray(v, np.float32) for v in (p1, v1, p2, v2)]
    out = np.zeros(3, np.float32)
    ok = lib().slro_line_line_intersection_x87(_p(a[0]), _p(a[1]), _p(a[2]), _p(a[3]), _p(out))
    return bool(ok), out


def ray_triangulate_x87(offL, itemsL, offR, itemsR, camL, camR, scan_w, scan_h, T=None):
    xyz = np.zeros((scan_h, scan_w, 3), np.float32)
    cnt = np.zeros((scan_h, scan_w), np.uint8)
    Ta = _t(T)
    iL = np.ascontiguousarray(itemsL, np.uint32) if itemsL.size else np.zeros(1, np.uint32)
    iR = np.ascontiguousarray(itemsR, np.uint32) if itemsR.size else np.zeros(1, np.uint32)
    lib().slro_ray_triangulate_x87(_p(offL), _p(iL), _p(offR), _p(iR), C.byref(camL), C.byref(camR), _p(Ta),
                                   C.c_int(scan_w), C.c_int(scan_h), _p(xyz), _p(cnt))
    return xyz, cnt


def x87_quotient_mismatches(lo, hi):
    """integers d in [lo, hi) for which the device's multiply + 2 fma form of P123 / (2*PI) * 255 (x87 model) differs from the division"""
    return int(lib().slro_x87_quotient_mismatches(C.c_long(lo), C.c_long(hi)))
