"""Host logic, no GPU: the two LDS tables the decode kernels use instead of the reference's branch chain
(structure-light-reconstructor_amd/csrc/kernels_decode.hip `wrapped_phase_q24` / `heterodyne_q24`, filled in
slr_capi.hip `slr_create`) restated in NumPy and checked, for EVERY pair of differences n, d in [-255, 255], against the
independent NumPy model of mfreconstruct.cpp:231-268 (tests/np_model.py).  The device side of the same statement is the
exhaustive 511 x 511 image of tests/test_gpu_parity.py."""
import numpy as np

import np_model as M

f32, f64 = np.float32, np.float64
SLOT = {-1: 2, 0: 9, 1: 6}
LUT_WORDS = 512 + 10 * 256 + 1


def build_tables():
    lut = np.zeros(LUT_WORDS, np.int64)
    used = np.zeros(LUT_WORDS, bool)
    off = {(-1, -1): M.PI, (-1, 0): M.PI, (-1, 1): M.PI,
           (0, -1): M.PI_1_2, (0, 0): f32(0), (0, 1): M.PI_3_2,
           (1, -1): f32(0), (1, 0): f32(0), (1, 1): M.TWO_PI}
    for d in range(-255, 256):
        sd = (d > 0) - (d < 0)
        R = 0 if d == 0 else 65536 // abs(d) + 1
        lut[d + 255] = R | SLOT[sd] << 24
    for sd in (-1, 0, 1):
        for sn in (-1, 0, 1):
            for qa in range(256):
                if (sd == 0 or sn == 0) and qa:
                    continue
                q = sd * sn * qa
                sidx = ~qa if (sn < 0 and sd != 0) else qa
                P = f32(f32(np.arctan(f64(q))) + off[(sd, sn)])
                scaled = f64(P) * 16777216.0
                assert scaled == int(scaled) and abs(scaled) < 2 ** 28
                w = 512 + ((SLOT[sd] + sn) << 8) + sidx
                assert 512 <= w < LUT_WORDS and not used[w]          # the slots of the nine sign cases never collide
                used[w] = True
                lut[w] = int(scaled)
    return lut


def device_wrapped_phase_q24(lut, n, d):
    e = lut[255 + d]
    R = e & 0xFFFFFF                                                 # v_mul_i32_i24 ignores the top byte
    s = (n * R) >> 16                                                # arithmetic shift of the signed product
    return lut[512 + (((e >> 24) + np.sign(n)) << 8) + s]


def test_signed_reciprocal_quotient_is_exact():
    n = np.arange(-255, 256, dtype=np.int64)
    for d in range(1, 256):
        s = (n * (65536 // d + 1)) >> 16
        assert np.array_equal(np.where(n < 0, ~s, s), np.abs(n) // d)


def test_wrapped_phase_table_matches_the_model_for_every_difference_pair():
    lut = build_tables()
    n, d = np.meshgrid(np.arange(-255, 256), np.arange(-255, 256), indexing="ij")
    # realise (n, d) with grey levels: G4 - G2 = n, G1 - G3 = d
    G4, G2 = np.where(n >= 0, n, 0), np.where(n >= 0, 0, -n)
    G1, G3 = np.where(d >= 0, d, 0), np.where(d >= 0, 0, -d)
    want, ok = M.wrapped_phase(G1, G2, G3, G4)
    got = device_wrapped_phase_q24(lut, n.astype(np.int64), d.astype(np.int64))
    assert np.array_equal(got.astype(f64) / 16777216.0, want.astype(f64))
    assert np.array_equal(ok, (n | d) != 0)


def test_integer_heterodyne_equals_the_f64_narrowing_one():
    lut = build_tables()
    vals = np.unique(lut[512:])
    rng = np.random.default_rng(7)
    P = rng.choice(vals, size=(3, 400000))
    # neighbours and equal phases: the `>` ties and the KA3-edge cliff
    P[1, :50000] = P[0, :50000]
    P[2, 50000:100000] = P[1, 50000:100000]
    two_pi_q = int(f64(M.TWO_PI) * 16777216.0)
    assert two_pi_q == f64(M.TWO_PI) * 16777216.0
    d12 = P[0] - P[1] + np.where(P[0] > P[1], 0, two_pi_q)
    d23 = P[1] - P[2] + np.where(P[1] > P[2], 0, two_pi_q)
    F12, F23 = d12.astype(f32), d23.astype(f32)                      # int -> f32: round to nearest even
    F123 = np.where(F12 > F23, F12 - F23, (F12 - F23).astype(f32) + f32(f64(M.TWO_PI) * 16777216.0)).astype(f32)
    P123 = (F123 * f32(1.0 / 16777216.0)).astype(f32)
    got = (P123 / M.TWO_PI).astype(f32) * f32(255)
    Pf = (P.astype(f64) / 16777216.0).astype(f32)
    want = M.heterodyne(Pf[0], Pf[1], Pf[2])
    assert np.array_equal(got.astype(f32), want)
