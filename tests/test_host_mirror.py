"""The C++ host mirror (structure-light-reconstructor_amd/host: VirtualCamera, stereoRect, PointCloudImage, GrayCodes,
MultiFrequency, Reconstruct, MFReconstruct, MeshCreator) driven through libslr_host.so's extern "C" hooks.
CPU half: encoders, matrix text I/O (6-digit precision loss, Q14), PNG/PGM I/O, map builder, PointCloudImage semantics,
stereoRectify sanity.  GPU half: a synthetic project directory in the reference's layout through all three modes."""
import ctypes as C
import os
import struct
import zlib

import numpy as np
import pytest
import torch  # noqa: F401  (HIP runtime first)

from util import bits_equal

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "structure-light-reconstructor_amd", "libslr_host.so")


@pytest.fixture(scope="module")
def host(slr):
    slr.capi.load_library()
    assert os.path.exists(LIB), "build it: make -C structure-light-reconstructor_amd/host"
    return C.CDLL(LIB)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def test_encoders_match_oracle(host, oracle):
    for (w, h, epi) in ((64, 24, True), (100, 37, False), (1280, 4, True)):
        exp = oracle.gen_graycodes(w, h, epi)
        assert host.duke_gray_num_imgs(w, h, int(epi)) == exp.shape[0]
        out = np.zeros_like(exp)
        assert host.duke_gen_graycodes(w, h, int(epi), _p(out)) == exp.shape[0]
        assert np.array_equal(out, exp)
    exp = oracle.gen_multifreq(1280, 3)
    out = np.zeros_like(exp)
    host.duke_gen_multifreq(1280, 3, _p(out))
    assert np.array_equal(out, exp)
    for bits in ([1, 0, 0], [1, 1, 0], [0, 1, 1, 1, 0, 1]):
        b = np.array(bits, np.uint8)
        assert host.duke_gray_to_dec(_p(b), len(bits)) == oracle.gray_to_dec(bits)


def test_matrix_text_io_keeps_the_float_round_trip(host, tmp_path):
    m = np.array([[1234.56789012, 0.000123456789, 321.987654321], [0, 987.654321987, 255.5], [0, 0, 1]], np.float64)
    path = str(tmp_path / "cam_matrix.txt").encode()
    assert host.duke_export_mat(path, _p(m), 3, 3) == 1
    text = open(path).read()
    assert "1234.57\t" in text and "0.000123457\t" in text            # default ostream precision: 6 significant digits
    out = np.zeros((3, 3), np.float32)
    assert host.duke_load_matrix(path, 3, 3, _p(out)) == 1
    assert out[0, 0] == np.float32(1234.57) and out[1, 1] == np.float32(987.654)
    assert host.duke_load_matrix(str(tmp_path / "missing.txt").encode(), 3, 3, _p(out)) == -1


def _png_bytes(img, filt, idat_split=0):
    """minimal PNG writer with a chosen filter type per row (exercises the decoder's unfilter paths); idat_split > 0 cuts the
    zlib stream into IDAT chunks of that many bytes"""
    h, w = img.shape
    raw = bytearray()
    prev = np.zeros(w, np.int32)
    for y in range(h):
        cur = img[y].astype(np.int32)
        f = filt[y % len(filt)]
        left = np.concatenate([[0], cur[:-1]])
        upleft = np.concatenate([[0], prev[:-1]])
        if f == 0: enc = cur
        elif f == 1: enc = cur - left
        elif f == 2: enc = cur - prev
        elif f == 3: enc = cur - ((left + prev) >> 1)
        else:
            p = left + prev - upleft
            pa, pb, pc = abs(p - left), abs(p - prev), abs(p - upleft)
            pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, prev, upleft))
            enc = cur - pred
        raw.append(f)
        raw.extend((enc & 255).astype(np.uint8).tobytes())
        prev = cur

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)
    z = zlib.compress(bytes(raw))
    idat = chunk(b"IDAT", z) if not idat_split else b"".join(chunk(b"IDAT", z[i:i + idat_split]) for i in range(0, len(z), idat_split))
    return b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 0, 0, 0, 0)) + idat + chunk(b"IEND", b"")


def test_png_banded_decode(host, tmp_path):
    """the decoder inflates a band of scanlines at a time: an image taller than one band (256 KiB of filtered bytes), every
    filter type crossing the band edges, the zlib stream cut into many IDAT chunks (one of them a single byte)"""
    rng = np.random.default_rng(6)
    h, w = 301, 3001                                         # 87 rows per band -> 4 bands, the last one partial
    img = (rng.integers(0, 40, size=(h, w)) + np.linspace(0, 200, w)[None, :]).astype(np.uint8)
    for filt, split in (([4, 3, 2, 1, 0, 2, 4], 0), ([3, 4], 4099), ([2], 1 << 14), ([1, 4, 4], 1)):
        if split == 1:
            img_s, name = img[:9, :40].copy(), "one.png"     # every IDAT chunk one byte long
        else:
            img_s, name = img, "band%d.png" % split
        open(tmp_path / name, "wb").write(_png_bytes(img_s, filt, split))
        out = np.zeros_like(img_s)
        ww, hh = C.c_int(0), C.c_int(0)
        assert host.duke_imread(str(tmp_path / name).encode(), _p(out), out.size, C.byref(ww), C.byref(hh)) == 1
        assert (ww.value, hh.value) == (img_s.shape[1], img_s.shape[0]) and np.array_equal(out, img_s)
    good = _png_bytes(img, [4, 1], 5000)
    cut = good[:len(good) * 2 // 3]                          # whole chunks missing from the middle on: no IEND
    open(tmp_path / "cut.png", "wb").write(cut)
    assert host.duke_imread(str(tmp_path / "cut.png").encode(), _p(out), out.size, C.byref(ww), C.byref(hh)) == 0


def test_png_and_pgm_io(host, tmp_path):
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, size=(37, 53), dtype=np.uint8)
    for png in (0, 1):
        path = str(tmp_path / ("a.png" if png else "a.pgm")).encode()
        assert host.duke_imwrite(path, _p(img), 53, 37, png) == 1
        out = np.zeros_like(img)
        w, h = C.c_int(0), C.c_int(0)
        assert host.duke_imread(path, _p(out), out.size, C.byref(w), C.byref(h)) == 1
        assert (w.value, h.value) == (53, 37) and np.array_equal(out, img)
    path = str(tmp_path / "filters.png")
    open(path, "wb").write(_png_bytes(img, [0, 1, 2, 3, 4]))
    out = np.zeros_like(img)
    w, h = C.c_int(0), C.c_int(0)
    assert host.duke_imread(path.encode(), _p(out), out.size, C.byref(w), C.byref(h)) == 1
    assert np.array_equal(out, img)
    assert host.duke_imread(str(tmp_path / "nope.png").encode(), _p(out), out.size, C.byref(w), C.byref(h)) == 0


def test_png_decoder_rejects_what_it_cannot_trust(host, tmp_path):
    """chunk CRCs, announced sizes and the inflated length are verified; interlaced (Adam7), 16-bit and colour PNGs decode;
    nothing here may crash or throw through the C boundary"""
    rng = np.random.default_rng(9)
    w, h = C.c_int(0), C.c_int(0)

    def read(path, shape):
        out = np.zeros(shape, np.uint8)
        return host.duke_imread(str(path).encode(), _p(out), out.size, C.byref(w), C.byref(h)), out
    for (ww, hh) in ((53, 37), (1, 1), (3, 2), (8, 8), (9, 17), (5, 1), (1, 6)):
        img = rng.integers(0, 256, size=(hh, ww), dtype=np.uint8)
        assert host.duke_imwrite(str(tmp_path / "i.png").encode(), _p(img), ww, hh, 2) == 1          # Adam7
        rc, out = read(tmp_path / "i.png", (hh, ww))
        assert rc == 1 and (w.value, h.value) == (ww, hh) and np.array_equal(out, img), (ww, hh)
    img = rng.integers(0, 256, size=(21, 34), dtype=np.uint8)
    good = _png_bytes(img, [4, 1, 3])
    open(tmp_path / "ok.png", "wb").write(good)
    assert read(tmp_path / "ok.png", img.shape)[0] == 1
    bad = bytearray(good); bad[60] ^= 0x40                                                           # a bit flip inside IDAT
    open(tmp_path / "crc.png", "wb").write(bytes(bad))
    assert read(tmp_path / "crc.png", img.shape)[0] == 0
    open(tmp_path / "cut.png", "wb").write(good[:len(good) // 2])
    assert read(tmp_path / "cut.png", img.shape)[0] == 0

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)
    sig = b"\x89PNG\r\n\x1a\n"
    huge = sig + chunk(b"IHDR", struct.pack(">IIBBBBB", 0x7FFFFFFF, 0x7FFFFFFF, 8, 0, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(b"\0")) + chunk(b"IEND", b"")
    open(tmp_path / "huge.png", "wb").write(huge)
    assert read(tmp_path / "huge.png", (4, 4))[0] == 0
    raw = b"".join(b"\0" + bytes(img[y]) for y in range(img.shape[0]))
    short = sig + chunk(b"IHDR", struct.pack(">IIBBBBB", 34, 22, 8, 0, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b"")
    open(tmp_path / "short.png", "wb").write(short)                                                  # IHDR announces one row more
    assert read(tmp_path / "short.png", (22, 34))[0] == 0
    trailing = sig + chunk(b"IHDR", struct.pack(">IIBBBBB", 34, 21, 8, 0, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw) + b"junk") + chunk(b"IEND", b"")
    open(tmp_path / "trail.png", "wb").write(trailing)                                               # bytes behind the zlib stream
    rc, out = read(tmp_path / "trail.png", img.shape)
    assert rc == 1 and np.array_equal(out, img)
    rawf = bytearray(raw); rawf[5 * 35] = 5                                                         # a scanline with filter type 5
    badf = sig + chunk(b"IHDR", struct.pack(">IIBBBBB", 34, 21, 8, 0, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(bytes(rawf))) + chunk(b"IEND", b"")
    open(tmp_path / "filter5.png", "wb").write(badf)                                                 # valid CRCs and sizes, invalid content
    assert read(tmp_path / "filter5.png", img.shape)[0] == 0
    # 16-bit grey keeps the high byte; RGB goes through OpenCV's fixed-point grey conversion
    g16 = rng.integers(0, 65536, size=(7, 11)).astype(">u2")
    raw16 = b"".join(b"\0" + g16[y].tobytes() for y in range(7))
    open(tmp_path / "g16.png", "wb").write(sig + chunk(b"IHDR", struct.pack(">IIBBBBB", 11, 7, 16, 0, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw16)) + chunk(b"IEND", b""))
    rc, out = read(tmp_path / "g16.png", (7, 11))
    assert rc == 1 and np.array_equal(out, (g16.astype(np.uint16) >> 8).astype(np.uint8))
    rgb = rng.integers(0, 256, size=(6, 9, 3), dtype=np.uint8)
    rawc = b"".join(b"\0" + rgb[y].tobytes() for y in range(6))
    open(tmp_path / "rgb.png", "wb").write(sig + chunk(b"IHDR", struct.pack(">IIBBBBB", 9, 6, 8, 2, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(rawc)) + chunk(b"IEND", b""))
    rc, out = read(tmp_path / "rgb.png", (6, 9))
    exp = ((rgb[..., 0].astype(np.int64) * 4899 + rgb[..., 1].astype(np.int64) * 9617 + rgb[..., 2].astype(np.int64) * 1868 + 8192) >> 14).astype(np.uint8)
    assert rc == 1 and np.array_equal(out, exp)
    # PGM with an absurd header
    open(tmp_path / "bad.pgm", "wb").write(b"P5\n99999999999 3\n255\n" + b"x" * 10)
    assert read(tmp_path / "bad.pgm", (4, 4))[0] == 0
    open(tmp_path / "cut.pgm", "wb").write(b"P5\n10 10\n255\n" + b"x" * 50)
    assert read(tmp_path / "cut.pgm", (10, 10))[0] == 0


def test_png_and_pgm_codec_against_pillow(host, tmp_path):
    """row f1's codec against an INDEPENDENT implementation (Pillow / libpng-free zlib PNG reader and writer of its own): files
    Pillow writes (grey 8-bit at several compression settings, 16-bit grey, grey + alpha, RGB, palette-free RGBA, binary PGM)
    decode here to what cv::imread(path, 0) gives -- grey as is, 16-bit by its high byte, colour by OpenCV's fixed-point grey
    conversion -- and files this codec writes (PNG, Adam7-interlaced PNG, PGM) read back identically through Pillow."""
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(12)
    w, h = C.c_int(0), C.c_int(0)

    def read(path, shape):
        out = np.zeros(shape, np.uint8)
        rc = host.duke_imread(str(path).encode(), _p(out), out.size, C.byref(w), C.byref(h))
        return rc, out

    for (hh, ww) in ((37, 53), (1, 1), (64, 200), (301, 1000)):
        smooth = (np.linspace(0, 255, ww)[None, :] * np.linspace(0.2, 1.0, hh)[:, None]).astype(np.uint8)
        for img in (rng.integers(0, 256, size=(hh, ww), dtype=np.uint8), smooth):
            for k, kw in enumerate(({"compress_level": 0}, {"compress_level": 1}, {"compress_level": 9}, {"optimize": True})):
                path = tmp_path / ("pil_%d_%d_%d.png" % (hh, ww, k))
                Image.fromarray(img, "L").save(path, **kw)
                rc, out = read(path, img.shape)
                assert rc == 1 and (w.value, h.value) == (ww, hh) and np.array_equal(out, img), (hh, ww, kw)
            path = tmp_path / "pil.pgm"
            Image.fromarray(img, "L").save(path)                                     # binary P5
            rc, out = read(path, img.shape)
            assert rc == 1 and np.array_equal(out, img)
            # ... and the other way: what this codec writes, Pillow reads
            for mode, name in ((1, "our.png"), (2, "our_adam7.png"), (0, "our.pgm")):
                assert host.duke_imwrite(str(tmp_path / name).encode(), _p(np.ascontiguousarray(img)), ww, hh, mode) == 1
                with Image.open(tmp_path / name) as im:
                    assert im.mode == "L" and im.size == (ww, hh)
                    if mode == 2:
                        assert im.info.get("interlace") == 1
                    assert np.array_equal(np.asarray(im), img), (hh, ww, name)
    # 16-bit grey: the high byte (cv::imread's depth conversion)
    g16 = rng.integers(0, 65536, size=(23, 31)).astype(np.uint16)
    Image.fromarray(g16).save(tmp_path / "pil16.png")
    rc, out = read(tmp_path / "pil16.png", g16.shape)
    assert rc == 1 and np.array_equal(out, (g16 >> 8).astype(np.uint8))
    # colour and alpha: OpenCV's BGR -> grey fixed point on the colour samples, alpha dropped
    rgb = rng.integers(0, 256, size=(19, 27, 3), dtype=np.uint8)
    exp = ((rgb[..., 0].astype(np.int64) * 4899 + rgb[..., 1].astype(np.int64) * 9617 + rgb[..., 2].astype(np.int64) * 1868 + 8192) >> 14).astype(np.uint8)
    Image.fromarray(rgb, "RGB").save(tmp_path / "pilrgb.png")
    rc, out = read(tmp_path / "pilrgb.png", exp.shape)
    assert rc == 1 and np.array_equal(out, exp)
    rgba = np.dstack([rgb, rng.integers(0, 256, size=(19, 27, 1), dtype=np.uint8)])
    Image.fromarray(rgba, "RGBA").save(tmp_path / "pilrgba.png")
    rc, out = read(tmp_path / "pilrgba.png", exp.shape)
    assert rc == 1 and np.array_equal(out, exp)
    la = np.dstack([rgb[..., 0], rgba[..., 3]])
    Image.fromarray(la, "LA").save(tmp_path / "pilla.png")
    rc, out = read(tmp_path / "pilla.png", exp.shape)
    assert rc == 1 and np.array_equal(out, rgb[..., 0])


def test_map_builder_matches_oracle(host, oracle):
    W, H = 96, 64
    M = np.array([[110.0, 0, 47.3], [0, 112.0, 31.8], [0, 0, 1]])
    D = np.array([-0.12, 0.03, 1e-3, -7e-4, 0.01])
    th = 0.02
    R = np.array([[np.cos(th), 0, np.sin(th)], [0, 1, 0], [-np.sin(th), 0, np.cos(th)]])
    P = np.array([[105.0, 0, 48.0, 0], [0, 105.0, 32.0, 0], [0, 0, 1, 0]])
    exy, efr = oracle.init_undistort_rectify_map(M, D, R, P, W, H)
    xy = np.zeros((H, W, 2), np.int16)
    fr = np.zeros((H, W), np.uint16)
    host.duke_init_undistort_rectify_map(_p(M.reshape(-1)), _p(D), _p(R.reshape(-1)), _p(P.reshape(-1)), W, H, _p(xy), _p(fr))
    assert np.array_equal(xy, exy) and np.array_equal(fr, efr)
    assert (fr < 1024).all() and abs(int(xy[H // 2, W // 2, 0]) - W // 2) < 4


def test_pointcloudimage_semantics(host, oracle):
    """first hit sets, later hits add and count+1 (u8 wrap), out-of-range dropped, getPoint = sum/count"""
    w, h = 5, 4
    pts = []
    rng = np.random.default_rng(1)
    for _ in range(40):
        pts.append([rng.integers(0, 7), rng.integers(0, 6)] + list(rng.standard_normal(3)))
    pts += [[2, 1, 1.5, -2.0, 0.25]] * 300                                 # wraps the u8 counter
    pts = np.array(pts, np.float32)
    s = np.zeros((h, w, 3), np.float32); c = np.zeros((h, w), np.uint8); m = np.zeros((h, w, 3), np.float32)
    host.duke_pointcloud_accumulate(w, h, _p(pts), len(pts), _p(s), _p(c), _p(m))
    es = np.zeros((h, w, 3), np.float32); ec = np.zeros((h, w), np.uint8)
    for i_w, j_h, x, y, z in pts:
        i_w, j_h = int(i_w), int(j_h)
        if i_w >= w or j_h >= h:
            continue
        p = np.array([x, y, z], np.float32)
        if ec[j_h, i_w] == 0:
            es[j_h, i_w] = p; ec[j_h, i_w] = 1
        else:
            es[j_h, i_w] = p + es[j_h, i_w]; ec[j_h, i_w] = np.uint8((int(ec[j_h, i_w]) + 1) & 255)
    assert np.array_equal(c, ec) and bits_equal(s, es)
    assert bits_equal(m, oracle.pointcloud_get(es, ec))


def _write_project(host, tmp_path, synth, W, H, calib, st_by_mode, sn=0):
    proj = str(tmp_path)
    for d in ("calib/left", "calib/right", "scan/left/%d" % sn, "scan/right/%d" % sn, "reconstruction"):
        os.makedirs(os.path.join(proj, d), exist_ok=True)

    def put(rel, arr):
        a = np.ascontiguousarray(np.asarray(arr, np.float64))
        a2 = a.reshape(a.shape[0], -1)
        assert host.duke_export_mat(os.path.join(proj, rel).encode(), _p(a2), a2.shape[0], a2.shape[1]) == 1
    for side, cam in (("left", calib.cam[0]), ("right", calib.cam[1])):
        K = np.array([[cam.fc[0], 0, cam.cc[0]], [0, cam.fc[1], cam.cc[1]], [0, 0, 1]])
        put("calib/%s/cam_matrix.txt" % side, K)
        put("calib/%s/cam_distortion.txt" % side, np.array(list(cam.k)).reshape(5, 1))
        put("calib/%s/cam_rotation_matrix.txt" % side, np.array(list(cam.R)).reshape(3, 3))
        put("calib/%s/cam_trans_vectror.txt" % side, np.array(list(cam.t)).reshape(3, 1))
        put("calib/%s/cam_stereo.txt" % side, K)
        put("calib/%s/distortion_stereo.txt" % side, np.array(list(cam.k)).reshape(5, 1))
    th = 0.01
    put("calib/R_stereo.txt", np.array([[np.cos(th), 0, np.sin(th)], [0, 1, 0], [-np.sin(th), 0, np.cos(th)]]))
    put("calib/T_stereo.txt", np.array([[-120.0], [0.8], [-1.5]]))
    put("calib/fundamental_stereo.txt", np.eye(3))
    put("calib/H1_mat.txt", np.eye(3)); put("calib/H2_mat.txt", np.eye(3))
    for cam, side, pre in ((0, "left", "L"), (1, "right", "R")):
        st = st_by_mode[cam]
        for i in range(st.shape[0]):
            path = os.path.join(proj, "scan/%s/%d/%s%d.png" % (side, sn, pre, i)).encode()
            assert host.duke_imwrite(path, _p(np.ascontiguousarray(st[i])), W, H, 1) == 1
    return proj


def _float_text(path, shape):
    return np.array(open(path).read().split(), np.float32).reshape(shape)


def _oracle_calib_from_project(O, proj):
    cams = []
    for side in ("left", "right"):
        K = _float_text(os.path.join(proj, "calib/%s/cam_matrix.txt" % side), (3, 3))
        cams.append(O.Camera.make((K[0, 0], K[1, 1]), (K[0, 2], K[1, 2]),
                                  _float_text(os.path.join(proj, "calib/%s/cam_distortion.txt" % side), (5,)),
                                  _float_text(os.path.join(proj, "calib/%s/cam_rotation_matrix.txt" % side), (3, 3)),
                                  _float_text(os.path.join(proj, "calib/%s/cam_trans_vectror.txt" % side), (3,))))
    return cams


@pytest.mark.parametrize("th,T", [(0.01, (-120.0, 0.8, -1.5)), (0.2, (-80.0, -3.0, 6.0)), (-0.05, (2.0, 95.0, -4.0)), (0.0, (60.0, 0.0, 0.0))])
def test_stereo_rectify_against_the_numpy_model(host, synth, tmp_path, th, T):
    """row f4: the C++ restatement of cv::stereoRectify (one-sided Jacobi SVD in Rodrigues(matrix)) against the independent fp64
    NumPy transcription with LAPACK's SVD, to 1e-12; horizontal and vertical baselines, zero rotation"""
    import np_model
    W, H = 128, 96
    calib, _ = synth.make_calibration(W, H)
    proj = _write_project(host, tmp_path, synth, W, H, calib, [np.zeros((0, H, W), np.uint8)] * 2)
    ax = np.array([0.3, 1.0, -0.2]); ax /= np.linalg.norm(ax)
    Rm = np_model.rodrigues_to_matrix(ax * th) @ (np.eye(3) + 3e-7 * np.arange(9).reshape(3, 3))     # not exactly orthonormal
    for name, arr in (("calib/R_stereo.txt", Rm), ("calib/T_stereo.txt", np.array(T).reshape(3, 1))):
        a = np.ascontiguousarray(arr, np.float64)
        assert host.duke_export_mat(os.path.join(proj, name).encode(), _p(a), a.shape[0], a.shape[1]) == 1
    R1, R2, P1, P2, Q = np.zeros(9), np.zeros(9), np.zeros(12), np.zeros(12), np.zeros(16)
    assert host.duke_stereo_rect(proj.encode(), W, H, _p(R1), _p(R2), _p(P1), _p(P2), _p(Q), None, None, None, None) == 1
    rd = lambda rel, shape: _float_text(os.path.join(proj, rel), shape).astype(np.float64)
    e = np_model.stereo_rectify(rd("calib/left/cam_stereo.txt", (3, 3)), rd("calib/left/distortion_stereo.txt", (5,)),
                                rd("calib/right/cam_stereo.txt", (3, 3)), rd("calib/right/distortion_stereo.txt", (5,)),
                                rd("calib/R_stereo.txt", (3, 3)), rd("calib/T_stereo.txt", (3,)), W, H)
    for got, exp in zip((R1.reshape(3, 3), R2.reshape(3, 3), P1.reshape(3, 4), P2.reshape(3, 4), Q.reshape(4, 4)), e):
        assert np.allclose(got, exp, rtol=1e-12, atol=1e-12), (got, exp)


def test_stereo_rectify_sanity(host, synth, tmp_path):
    W, H = 128, 96
    calib, _ = synth.make_calibration(W, H)
    proj = _write_project(host, tmp_path, synth, W, H, calib, [np.zeros((0, H, W), np.uint8)] * 2)
    R1, R2, P1, P2, Q = np.zeros(9), np.zeros(9), np.zeros(12), np.zeros(12), np.zeros(16)
    assert host.duke_stereo_rect(proj.encode(), W, H, _p(R1), _p(R2), _p(P1), _p(P2), _p(Q), None, None, None, None) == 1
    R1, R2, P1, P2, Q = R1.reshape(3, 3), R2.reshape(3, 3), P1.reshape(3, 4), P2.reshape(3, 4), Q.reshape(4, 4)
    for R in (R1, R2):
        assert np.allclose(R @ R.T, np.eye(3), atol=1e-9) and abs(np.linalg.det(R) - 1) < 1e-9
    T = _float_text(os.path.join(proj, "calib/T_stereo.txt"), (3,)).astype(np.float64)
    t = R2 @ T
    assert abs(t[1]) < 1e-6 * abs(t[0]) and abs(t[2]) < 1e-6 * abs(t[0])          # baseline on the new x axis
    f = P1[0, 0]
    assert P1[1, 1] == f and P2[0, 0] == f and np.isclose(P2[0, 3], t[0] * f) and P1[1, 2] == P2[1, 2]
    assert np.allclose(Q, [[1, 0, 0, -P1[0, 2]], [0, 1, 0, -P1[1, 2]], [0, 0, 0, f], [0, 0, -1 / t[0], (P1[0, 2] - P2[0, 2]) / t[0]]])


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [2, 1, 0])
def test_project_directory_through_the_host_mirror(host, oracle, synth, tmp_path, mode):
    """mode 2 = MFReconstruct::runReconstruction, 1 = Reconstruct::runReconstruction_GE, 0 = runReconstruction"""
    W, H, scan_w, scan_h, BLACK, WHITE = 160, 96, 120, 150, 40, 3
    calib, _ = synth.make_calibration(W, H, baseline=400.0, theta=0.6)
    if mode == 2:
        st = synth.render_mf_stack(W, H, seed=5).numpy()
    else:
        st = synth.render_gray_stack(W, H, scan_w, scan_h, seed=5, noise=2, rows=(mode == 0)).numpy()
    proj = _write_project(host, tmp_path, synth, W, H, calib, st, sn=0)
    pc_sum = np.zeros((scan_h, scan_w, 3), np.float32)
    pc_cnt = np.zeros((scan_h, scan_w), np.uint8)
    pc_col = np.zeros((scan_h, scan_w, 3), np.uint8)
    err = C.create_string_buffer(512)
    ply = os.path.join(proj, "reconstruction", "0.ply")
    ok = host.duke_run_project(proj.encode(), mode, 0, scan_w, scan_h, W, H, BLACK, WHITE, 1 if mode == 1 else 0, b".png",
                               ply.encode(), _p(pc_sum), _p(pc_cnt), _p(pc_col), err, 512)
    assert ok == 1, err.value
    camL, camR = _oracle_calib_from_project(oracle, proj)
    ncol, nrow = synth.gray_num_bits(scan_w), synth.gray_num_bits(scan_h)
    if mode == 0:
        dec = [oracle.gray_decode(st[c], ncol, nrow, BLACK, WHITE, scan_w, scan_h) for c in range(2)]
        offL, itL = oracle.gray_bucket(dec[0][0], dec[0][1], dec[0][2], scan_w, scan_h)
        offR, itR = oracle.gray_bucket(dec[1][0], dec[1][1], dec[1][2], scan_w, scan_h)
        es, ec = oracle.ray_triangulate(offL, itL, offR, itR, camL, camR, scan_w, scan_h, None)
    else:
        R1, R2, P1, P2, Q = np.zeros(9), np.zeros(9), np.zeros(12), np.zeros(12), np.zeros(16)
        m11 = np.zeros((H, W, 2), np.int16); m12 = np.zeros((H, W), np.uint16)
        m21 = np.zeros((H, W, 2), np.int16); m22 = np.zeros((H, W), np.uint16)
        assert host.duke_stereo_rect(proj.encode(), W, H, _p(R1), _p(R2), _p(P1), _p(P2), _p(Q), _p(m11), _p(m12), _p(m21), _p(m22)) == 1
        maps = [(m11, m12), (m21, m22)]
        rect = [np.stack([oracle.remap_u8(st[c][p], maps[c][0], maps[c][1]) for p in range(st[c].shape[0])]) for c in range(2)]
        if mode == 2:
            dec = [oracle.mf_decode(rect[c], BLACK) for c in range(2)]
            xyz, has, _ = oracle.mf_triangulate(dec[0][0], dec[0][1], dec[1][0], dec[1][1], camL, camR, Q.reshape(4, 4), None)
            col = None
        else:
            dec = [oracle.gray_decode(rect[c], ncol, 0, BLACK, WHITE, scan_w, 0) for c in range(2)]
            xyz, has, col, _ = oracle.ge_triangulate(dec[0][0], dec[0][2], dec[1][0], dec[1][2], Q.reshape(4, 4), None,
                                                     rect[0][0], rect[1][0])
        es, ec, ecol = oracle.pointcloud_from_grid(xyz, has, scan_w, scan_h, col)
        if mode == 1:
            assert np.array_equal(pc_col[..., 0], ecol) and np.array_equal(pc_col[..., 2], ecol)
    assert np.array_equal(pc_cnt, ec) and bits_equal(pc_sum, es)
    assert ec.sum() > 50
    # PLY written by MeshCreator::exportPlyMesh: the whole file against the format spec restated in Python
    assert open(ply).read() == _expected_ply(es, ec, ecol if mode == 1 else None)


def _fmt(v):
    return "%g" % float(v)


def _cloud_points(pc_sum, pc_cnt):
    """PointCloudImage::getPoint: f32(f64(sum) / f64(f32(count))) per component"""
    return (pc_sum.astype(np.float64) / pc_cnt.astype(np.float32).astype(np.float64)[..., None]).astype(np.float32)


def _mesh_numbers(pc_cnt, first):
    h, w = pc_cnt.shape
    ids = np.full((h, w), -1, np.int64)
    k = first
    for i in range(w):
        for j in range(h):
            if pc_cnt[j, i]:
                ids[j, i] = k
                k += 1
    return ids, k - first


def _mesh_faces(ids, first):
    h, w = ids.shape
    ok = lambda i, j: 0 <= i < w and 0 <= j < h and ids[j, i] >= 0 and not (first == 0 and ids[j, i] == 0)
    out = []
    for i in range(w):
        for j in range(h):
            if ok(i, j) and ok(i + 1, j):
                if ok(i, j + 1):
                    out.append((ids[j, i], ids[j, i + 1], ids[j + 1, i]))
                if ok(i + 1, j - 1):
                    out.append((ids[j, i], ids[j - 1, i + 1], ids[j, i + 1]))
    return out


def _expected_ply(pc_sum, pc_cnt, col=None):
    """meshcreator.cpp:67-166 as a format: vertices column by column numbered from 0 (number 0 can never be a face corner),
    colour 100 100 100 without a colour image, two triangles per pixel over the right / lower / upper-right neighbours"""
    ids, n = _mesh_numbers(pc_cnt, 0)
    faces = _mesh_faces(ids, 0)
    pts = _cloud_points(pc_sum, np.maximum(pc_cnt, 1))
    h, w = pc_cnt.shape
    lines = ["ply", "format ascii 1.0", "element vertex %d" % n, "property float x", "property float y", "property float z",
             "property uchar red", "property uchar green", "property uchar blue", "element face %d" % len(faces),
             "property list uchar int vertex_indices", "end_header"]
    for i in range(w):
        for j in range(h):
            if pc_cnt[j, i]:
                c = 100 if col is None else int(round(float(col[j, i]) / float(np.float32(pc_cnt[j, i]))))
                lines.append("%s %s %s %d %d %d" % (_fmt(pts[j, i, 0]), _fmt(pts[j, i, 1]), _fmt(pts[j, i, 2]), c, c, c))
    lines += ["3 %d %d %d" % f for f in faces]
    return "\n".join(lines) + "\n"


@pytest.mark.gpu
def test_mesh_files_against_the_format_spec(host, tmp_path):
    """row f2: PLY and OBJ of random clouds, byte for byte -- vertex order, numbering from 0 / 1, the vertex-0 sentinel of the PLY
    numbering, face winding, borders, an empty cloud"""
    rng = np.random.default_rng(11)
    for (w, h, p) in ((7, 5, 0.7), (40, 33, 0.5), (64, 64, 0.97), (3, 3, 0.0), (1, 9, 1.0), (130, 2, 0.8)):
        cnt = (rng.random((h, w)) < p).astype(np.uint8) * rng.integers(1, 4, (h, w)).astype(np.uint8)
        if p >= 0.5:
            cnt[0, 0] = 2                                   # vertex 0 exists and has usable neighbours: the sentinel matters
            cnt[0, 1 % w] = 1; cnt[1 % h, 0] = 1
        s = (rng.standard_normal((h, w, 3)) * 100).astype(np.float32)
        ply, obj = str(tmp_path / "m.ply"), str(tmp_path / "m.obj")
        assert host.duke_export_mesh(ply.encode(), 0, w, h, _p(s), _p(cnt)) == 1
        assert host.duke_export_mesh(obj.encode(), 1, w, h, _p(s), _p(cnt)) == 1
        assert open(ply).read() == _expected_ply(s, cnt), (w, h)
        ids, n = _mesh_numbers(cnt, 1)
        pts = _cloud_points(s, np.maximum(cnt, 1))
        lines = ["v %s %s %s" % tuple(_fmt(x) for x in pts[j, i]) for i in range(w) for j in range(h) if cnt[j, i]]
        lines += ["f %d/%d %d/%d %d/%d" % (a, a, b, b, c, c) for a, b, c in _mesh_faces(ids, 1)]
        assert open(obj).read() == ("\n".join(lines) + "\n" if lines else ""), (w, h)
        if p >= 0.5 and w > 1 and h > 1:
            assert len(_mesh_faces(_mesh_numbers(cnt, 0)[0], 0)) < len(_mesh_faces(ids, 1))          # the PLY lost vertex 0's faces


@pytest.mark.gpu
def test_scan_series_is_pipelined_and_identical_to_single_scans(host, synth, tmp_path):
    """row f1: MFReconstruct::runReconstructionSeries (PNG decode into page-locked memory, two contexts with SLR_OPT_ASYNC_HOST,
    decode of scan i+1 overlapping the GPU work of scan i) == one duke_run_project per scan"""
    W, H, scan_w, scan_h = 320, 96, 200, 340
    calib, _ = synth.make_calibration(W, H, baseline=400.0, theta=0.6)
    n = 4
    proj = None
    for sn in range(n):
        st = synth.render_mf_stack(W, H, seed=70 + sn).numpy()
        proj = _write_project(host, tmp_path, synth, W, H, calib, st, sn=sn)
    for sn in (1, 2, 3):                                     # every scan after the first brings its transfer matrix (mfreconstruct.cpp:278-282)
        Tm = np.array([[1, 0, 0, 5.0 * sn], [0, 1, 0, -2.0], [0, 0, 1, 0.5]], np.float64)
        assert host.duke_export_mat(os.path.join(proj, "scan/transfer_mat%d.txt" % sn).encode(), _p(Tm), 3, 4) == 1
    err = C.create_string_buffer(512)
    ss = np.zeros((n, scan_h, scan_w, 3), np.float32)
    sc = np.zeros((n, scan_h, scan_w), np.uint8)
    pre = os.path.join(proj, "reconstruction", "s")
    assert host.duke_run_series(proj.encode(), 0, n, scan_w, scan_h, W, H, 40, 0, b".png", pre.encode(), _p(ss), _p(sc), err, 512) == n, err.value
    for sn in range(n):
        es = np.zeros((scan_h, scan_w, 3), np.float32)
        ec = np.zeros((scan_h, scan_w), np.uint8)
        ply = os.path.join(proj, "reconstruction", "one%d.ply" % sn)
        assert host.duke_run_project(proj.encode(), 2, sn, scan_w, scan_h, W, H, 40, 0, 0, b".png", ply.encode(), _p(es), _p(ec), None, err, 512) == 1
        assert np.array_equal(sc[sn], ec) and bits_equal(ss[sn], es), sn
        assert open(pre + "%d.ply" % sn).read() == open(ply).read()
        assert ec.sum() > 50
    assert not bits_equal(ss[0], ss[1])
    # a missing scan stops the series where the reference's loop would
    assert host.duke_run_series(proj.encode(), 2, 4, scan_w, scan_h, W, H, 40, 0, b".png", None, None, None, err, 512) == 2
    assert b"not found" in err.value
    # round 6: the series over several pipelines ("devices" 0, 0, 0 on this one-GPU box: three reconstructors on three host threads,
    # scans dealt round-robin): the same clouds and the same files
    ms = np.zeros((n, scan_h, scan_w, 3), np.float32)
    mc = np.zeros((n, scan_h, scan_w), np.uint8)
    mpre = os.path.join(proj, "reconstruction", "m")
    devs = (C.c_int * 3)(0, 0, 0)
    assert host.duke_run_series_multi(proj.encode(), 0, n, scan_w, scan_h, W, H, 40, 0, b".png", mpre.encode(), devs, 3, _p(ms), _p(mc), err, 512) == n, err.value
    assert np.array_equal(mc, sc) and bits_equal(ms, ss)
    for sn in range(n):
        assert open(mpre + "%d.ply" % sn).read() == open(pre + "%d.ply" % sn).read()
    # a missing scan: the pipelines that meet it stop, the others finish theirs; the error is reported
    done = host.duke_run_series_multi(proj.encode(), 2, 4, scan_w, scan_h, W, H, 40, 0, b".png", None, devs, 2, None, None, err, 512)
    assert done == 2 and b"not found" in err.value
    assert host.duke_run_series_multi(proj.encode(), 0, n, scan_w, scan_h, W, H, 40, 0, b".png", None, None, 0, None, None, err, 512) == 0


@pytest.mark.gpu
def test_missing_images_fail_like_the_reference(host, synth, tmp_path):
    W, H = 64, 48
    calib, _ = synth.make_calibration(W, H)
    proj = _write_project(host, tmp_path, synth, W, H, calib, [np.zeros((3, H, W), np.uint8)] * 2)   # only 3 of 14 images
    err = C.create_string_buffer(512)
    ok = host.duke_run_project(proj.encode(), 2, 0, 64, 64, W, H, 40, 0, 0, b".png", None, None, None, None, err, 512)
    assert ok == 0 and b"not found" in err.value
    ok = host.duke_run_project(str(tmp_path / "nowhere").encode(), 2, 0, 64, 64, W, H, 40, 0, 0, b".png", None, None, None, None, err, 512)
    assert ok == 0 and b"Calibration" in err.value


def test_bench_frames_per_launch_bookkeeping():
    """bench.py's kernel entries divide the profiler's per-frame times back into launches: the grouping rules of
    slr_reconstruct_mf_batch (SLR_OPT_MF_BATCH_GROUP / _DECODE_GROUP) restated there must match the library's defaults"""
    import importlib.util
    import types
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    a = types.SimpleNamespace(match_group=0, decode_group=0, match_algo=0, rect_algo=0)
    assert bench.frames_per_launch(a, "mf", True, 8, "slr_mf_match_triangulate") == 8.0
    assert bench.frames_per_launch(a, "mf", True, 8, "slr_mf_rectify_decode_pair") == 8.0
    assert bench.frames_per_launch(a, "mf", False, 8, "slr_mf_rectify_decode_pair") == 1.0      # unrectified: the unfused decode, per frame
    assert bench.frames_per_launch(a, "ge", True, 8, "slr_mf_match_triangulate") == 1.0
    a.match_group, a.decode_group = 8, 3                                                          # 8 frames: decode launches of 3 + 3 + 2
    assert bench.frames_per_launch(a, "mf", True, 8, "slr_mf_rectify_decode_pair") == 8.0 / 3
    a.match_group = 1
    assert bench.frames_per_launch(a, "mf", True, 8, "slr_mf_match_triangulate") == 1.0
    assert bench.frames_per_launch(a, "mf", True, 8, "slr_mf_rectify_decode_pair") == 1.0
    a.match_group, a.decode_group = 5, 8                                                          # groups of 5 + 3
    assert bench.frames_per_launch(a, "mf", True, 8, "slr_mf_match_triangulate") == 4.0
