"""Oracle pinning: RNG core against published known-answer vectors, math helpers against numpy.

The Threefry-2x32-20 vectors are the Random123 known-answer tests (also used by JAX's
`testThreefry2x32`); everything the engine's RNG does beyond the block function is the build's own
choice (parity unpinned against the real reference)."""
import ctypes as C

import numpy as np


def test_threefry_known_answers(oracle):
    L = oracle.lib()
    out = (C.c_uint32 * 2)()
    kats = [((0, 0), (0, 0), (0x6b200159, 0x99ba4efe)),
            ((0xffffffff, 0xffffffff), (0xffffffff, 0xffffffff), (0x1cb996fc, 0xbb002be7)),
            ((0x13198a2e, 0x03707344), (0x243f6a88, 0x85a308d3), (0xc4923a9c, 0x483df7a0))]
    for key, ctr, exp in kats:
        L.hsref_threefry(key[0], key[1], ctr[0], ctr[1], out)
        assert (out[0], out[1]) == exp


def test_rng_draws_are_counter_based(oracle):
    L = oracle.lib()
    a = (C.c_uint32 * 16)()
    b = (C.c_uint32 * 8)()
    L.hsref_rng_draws(11, 22, 16, a)
    L.hsref_rng_draws(11, 22, 8, b)
    assert list(a)[:8] == list(b)                 # a prefix of the same stream
    assert len(set(a)) == 16
    out = (C.c_uint32 * 2)()
    L.hsref_threefry(11, 22, 3, 0, out)           # draw i == threefry(key, (i, 0)), a ^ b
    assert a[3] == out[0] ^ out[1]


def test_sample_i32_half_open_and_empty_range(oracle):
    L = oracle.lib()
    vals = [L.hsref_sample_i32(k, 99, 3, 10) for k in range(2000)]
    assert min(vals) == 3 and max(vals) == 9      # [a, b)
    assert all(L.hsref_sample_i32(k, 1, 3, 3) == 3 for k in range(50))   # empty range -> a (level_gen.cpp:87-88)
    assert set(L.hsref_sample_i32(k, 5, 0, 2) for k in range(200)) == {0, 1}


def test_sincos_atan2_asin_close_to_libm(oracle):
    L = oracle.lib()
    s, c = C.c_float(), C.c_float()
    for x in np.linspace(-7.0, 7.0, 1001, dtype=np.float32):
        L.hsref_sincos(C.c_float(x), C.byref(s), C.byref(c))
        assert abs(s.value - np.sin(np.float64(x))) < 5e-7
        assert abs(c.value - np.cos(np.float64(x))) < 5e-7
    rng = np.random.default_rng(0)
    for y, x in rng.uniform(-3, 3, size=(500, 2)).astype(np.float32):
        assert abs(L.hsref_atan2(C.c_float(y), C.c_float(x)) - np.arctan2(np.float64(y), np.float64(x))) < 1e-6
    for x in np.linspace(-1, 1, 401, dtype=np.float32):
        assert abs(L.hsref_asin(C.c_float(x)) - np.arcsin(np.float64(x))) < 1e-6
    assert L.hsref_atan2(C.c_float(1.0), C.c_float(0.0)) == np.float32(np.pi / 2)
    assert L.hsref_atan2(C.c_float(0.0), C.c_float(0.0)) == 0.0


def test_quat_to_euler_identities(oracle):
    """quatToEuler src/sim.cpp:372-399 incl. the gimbal clamp (:382-387)."""
    L = oracle.lib()
    out = (C.c_float * 3)()

    def euler(q):
        qa = (C.c_float * 4)(*q)
        L.hsref_quat_to_euler(qa, out)
        return np.array(out[:])
    assert np.allclose(euler([1, 0, 0, 0]), 0)
    h = np.sqrt(0.5)
    assert np.allclose(euler([h, 0, 0, h]), [0, 0, np.pi / 2], atol=1e-6)     # yaw 90
    assert np.allclose(euler([h, h, 0, 0]), [np.pi / 2, 0, 0], atol=1e-6)     # roll 90
    assert np.allclose(euler([h, 0, h, 0])[1], np.pi / 2, atol=1e-3)          # pitch clamp
    assert np.allclose(euler([h, 0, -h, 0])[1], -np.pi / 2, atol=1e-3)
