"""shared helpers for the parity tests (inputs, oracle-side pipelines, comparison)"""
import numpy as np


def to_oracle_cam(O, cam):
    return O.Camera.make(list(cam.fc), list(cam.cc), list(cam.k), np.array(list(cam.R)).reshape(3, 3), list(cam.t))


def calib_parts(O, calib):
    camL, camR = to_oracle_cam(O, calib.cam[0]), to_oracle_cam(O, calib.cam[1])
    Q = np.array(list(calib.Q), np.float64).reshape(4, 4)
    T = np.array(list(calib.T), np.float32).reshape(3, 4) if calib.has_T else None
    return camL, camR, Q, T


def bits_equal(a, b):
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    assert a.shape == b.shape and a.dtype == b.dtype, (a.shape, b.shape, a.dtype, b.dtype)
    if a.dtype.kind == "f":
        return np.array_equal(a.view(np.uint32), b.view(np.uint32))
    return np.array_equal(a, b)


def assert_float_parity(got, exp, rel=1e-4, what=""):
    """north_star tolerance: 1e-4 relative.  Also reports how far from bit-exact we are (expected: 0)."""
    got = np.asarray(got, np.float32)
    exp = np.asarray(exp, np.float32)
    assert got.shape == exp.shape
    nbad_bits = int((got.view(np.uint32) != exp.view(np.uint32)).sum())
    denom = np.maximum(np.abs(exp), 1e-30)
    err = np.abs(got.astype(np.float64) - exp.astype(np.float64)) / denom
    err = np.where(exp == got, 0.0, err)
    assert np.all(np.isfinite(got) == np.isfinite(exp)), what
    worst = float(np.nanmax(err)) if err.size else 0.0
    assert worst <= rel, "%s: max rel err %.3e > %.1e (%d elements not bit-equal)" % (what, worst, rel, nbad_bits)
    return nbad_bits


def np_of(t):
    """torch tensor (cpu/cuda) or ndarray -> ndarray"""
    if hasattr(t, "detach"):
        return t.detach().cpu().numpy()
    return np.asarray(t)


def forms(ctx, slr, opt, values, default=0, required=()):
    """the values of a form option (SLR_OPT_RECT_DECODE_ALGO / _RECT_DMA_SHAPE / _MF_MATCH_ALGO) this build of the library has:
    the measured-dominated forms are compiled with `make FORMS=all` only and answer SLR_ERR_UNSUPPORTED otherwise; `required`
    values must be there.  Leaves the option at `default`."""
    ok = []
    for v in values:
        try:
            ctx.set_option(opt, v)
            ok.append(v)
        except slr.capi.SlrError as e:
            assert e.status == slr.capi.ERR_UNSUPPORTED and v not in required, (opt, v, str(e))
    ctx.set_option(opt, default)
    for v in required:
        assert v in ok, (opt, v)
    return ok
