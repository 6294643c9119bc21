"""The oracle's second evaluation model (oracle/slr_oracle_x87.c: the reference's MSVC2010 x87 / fp:precise build, DESIGN.md
section 2) against an independent NumPy transcription of the same rules, and its strict switch against the oracle proper.  CPU."""
import numpy as np
import pytest

import oracle as O

PI = np.float32(3.1416)


def np_wrapped_x87(G1, G2, G3, G4, atab):
    """mfreconstruct.cpp:246-261 with float + float evaluated at 53 bits and stored to a double (returns None when undefined)"""
    if G4 == G2 and G1 > G3:
        return 0.0
    if G4 == G2 and G1 < G3:
        return float(PI)
    if G1 == G3 and G4 > G2:
        return 3.0 * float(PI) / 2.0
    if G1 == G3 and G4 < G2:
        return float(PI) / 2.0
    if G1 == G3 and G4 == G2:
        return None
    q = int((G4 - G2) / (G1 - G3))                      # C division truncates toward zero
    a = float(atab[q + 255])
    if G1 < G3:
        return a + float(PI)
    if G1 > G3 and G4 > G2:
        return a + 2.0 * float(PI)
    return a


def np_heterodyne_x87(P):
    two_pi = 2.0 * float(PI)
    P12 = np.float32((P[0] - P[1]) if P[0] > P[1] else (P[0] - P[1] + two_pi))
    P23 = np.float32((P[1] - P[2]) if P[1] > P[2] else (P[1] - P[2] + two_pi))
    d = float(P12) - float(P23)
    P123 = np.float32(d if P12 > P23 else d + two_pi)
    return np.float32(float(P123) / two_pi * 255.0)


def test_strict_switch_is_the_oracle():
    rng = np.random.default_rng(3)
    tab = O.atan_table(0)
    for levels in (4, 32, 256):                          # few grey levels: the equality branches and Q5 occur
        planes = (rng.integers(0, levels, size=(14, 40, 64)) * (255 // (levels - 1))).astype(np.uint8)
        planes[0] = 220
        planes[1] = rng.integers(0, 255, size=(40, 64))
        p0, v0 = O.mf_decode(planes, 40)
        p1, v1 = O.mf_decode_ev(planes, 40, tab, 0)
        assert np.array_equal(p0.view(np.int32), p1.view(np.int32)) and np.array_equal(v0, v1)
    for _ in range(2000):
        g = [int(x) for x in rng.integers(0, 256, size=4)]
        ok0, P0 = O.wrapped_phase(*g)
        ok1, P1 = O.wrapped_phase_ev(*g, tab, 0)
        assert ok0 == ok1 and (not ok0 or float(P0) == P1)
        P = rng.uniform(-1, 8, size=3).astype(np.float32).astype(np.float64)
        assert O.heterodyne(P).view(np.int32) == O.heterodyne_ev(P, 0).view(np.int32)


def test_x87_model_against_numpy():
    rng = np.random.default_rng(4)
    tab = O.atan_table(1)
    assert np.array_equal(tab.view(np.int32), O.atan_table(0).view(np.int32))     # glibc atanf == (float)atan((double)q) here
    cases = [(200, 135, 70, 135), (70, 135, 200, 135), (135, 70, 135, 200), (135, 200, 135, 70), (135, 135, 135, 135),
             (100, 120, 180, 160), (180, 100, 100, 190), (180, 190, 100, 100)]
    cases += [tuple(int(x) for x in rng.integers(0, 256, size=4)) for _ in range(3000)]
    cases += [tuple(int(x) for x in rng.integers(100, 104, size=4)) for _ in range(300)]
    for g in cases:
        ok, P = O.wrapped_phase_ev(*g, tab, 1)
        want = np_wrapped_x87(*g, tab)
        assert ok == (want is not None) and (want is None or P == want), g
    # the sums keep bits an f32 cannot hold: atanf(1) + 2*PI
    ok, P = O.wrapped_phase_ev(180, 100, 100, 190, tab, 1)
    assert P == float(np.float32(np.arctan(np.float32(1)))) + 2.0 * float(PI) and P != float(np.float32(P))
    for _ in range(3000):
        g3 = [np_wrapped_x87(*[int(x) for x in rng.integers(0, 256, size=4)], tab) for _ in range(3)]
        if any(p is None for p in g3):
            continue
        assert O.heterodyne_ev(g3, 1).view(np.int32) == np_heterodyne_x87(g3).view(np.int32), g3
    # the cliff of KA3-edge under both models: same operands, the strict and the x87 evaluation may land on different sides
    P = [7.068598, 3.1416, -0.7853982]
    assert abs(float(O.heterodyne_ev(P, 0)) - 254.99998) < 1e-4


def test_x87_mf_decode_against_numpy():
    rng = np.random.default_rng(5)
    tab = O.atan_table(1)
    planes = rng.integers(60, 200, size=(14, 12, 20)).astype(np.uint8)
    planes[0], planes[1] = 230, rng.integers(150, 220, size=(12, 20))
    ph, vd = O.mf_decode_ev(planes, 40, tab, 1)
    for r in range(12):
        for c in range(20):
            mask = float(planes[0, r, c]) - float(planes[1, r, c]) > 40
            want_ph, want_v = np.float32(0), int(mask)
            if mask:
                P = []
                for f in range(3):
                    g = [int(planes[4 * f + 2 + s, r, c]) for s in range(4)]
                    p = np_wrapped_x87(*g, tab)
                    if p is None:
                        want_v, p = 0, 0.0
                    P.append(p)
                want_ph = np_heterodyne_x87(P)
            assert vd[r, c] == want_v and ph[r, c].view(np.int32) == want_ph.view(np.int32)


def np_dot_x87(a, b):
    s = np.float32(0)
    for i in range(3):
        s = np.float32(float(s) + float(a[i]) * float(b[i]))
    return s


def test_x87_line_line_intersection_against_numpy():
    rng = np.random.default_rng(6)
    ndiff = 0
    for _ in range(2000):
        p1 = rng.uniform(-50, 50, 3).astype(np.float32)
        p2 = rng.uniform(-50, 50, 3).astype(np.float32)
        v1 = rng.normal(size=3); v1 = (v1 / np.linalg.norm(v1)).astype(np.float32)
        v2 = rng.normal(size=3); v2 = (v2 / np.linalg.norm(v2)).astype(np.float32)
        ok, got = O.line_line_intersection_x87(p1, v1, p2, v2)
        v12 = (p1 - p2).astype(np.float32)
        a, c, b = np_dot_x87(v1, v1), np_dot_x87(v2, v2), np_dot_x87(v1, v2)
        d1, d2 = np_dot_x87(v12, v1), np_dot_x87(v12, v2)
        denom = np.float32(float(a) * float(c) - float(b) * float(b))
        if abs(float(denom)) < 0.1:
            assert not ok
            continue
        s = np.float32((float(b) / float(denom)) * float(d2) - (float(c) / float(denom)) * float(d1))
        t = np.float32(-(float(b) / float(denom)) * float(d1) + (float(a) / float(denom)) * float(d2))
        want = np.array([np.float32(0.5 * float(np.float32(np.float32(p1[k] + np.float32(s * v1[k])) + np.float32(p2[k] + np.float32(t * v2[k])))))
                         for k in range(3)], np.float32)
        assert ok and np.array_equal(got.view(np.int32), want.view(np.int32))
        ok0, strict = O.line_line_intersection(p1, v1, p2, v2)
        ndiff += int(ok0 and not np.array_equal(strict.view(np.int32), got.view(np.int32)))
    assert ndiff > 0                                    # the two models do differ in the last places (that is the point)


def test_x87_quotient_by_constant():
    """decode_common.hpp's het_finish_x87 forms P123 / (2*PI) -- an f64 division under the x87 model -- as one multiply and two
    fused multiply-adds with constants.  Every value P123 can take on the device (all integers d = P123 * 2^24 before the f32
    rounding of the store, 1 .. 2 * (2*PI * 2^24) + 1) gives the division's bits; the range beyond, too."""
    two_pi_q24 = int(np.float32(2) * PI * np.float32(16777216.0))
    assert two_pi_q24 == 105414600
    assert O.x87_quotient_mismatches(0, 2 * two_pi_q24 + 2) == 0
    assert O.x87_quotient_mismatches((1 << 30) - 3000000, 1 << 30) == 0
