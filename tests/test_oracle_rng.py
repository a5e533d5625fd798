"""Pins oracle/orc_rng.h: rand-0.8.5 Xoshiro256++ semantics
and Philox4x32-10 (Random123 known-answer vectors)."""
import ctypes as C

import numpy as np

import orc


def test_xoshiro256pp_kat():
    # reference vector of xoshiro256++ for state [1,2,3,4] (rand_xoshiro test suite)
    r = orc.rng_ref_state([1, 2, 3, 4])
    got = [orc.lib().orc_rng_next_u64(C.byref(r)) for _ in range(10)]
    assert got == [41943041, 58720359, 3588806011781223, 3591011842654386, 9228616714210784205,
                   9973669472204895162, 14011001112246962877, 12406186145184390807, 15849039046786891736,
                   10450023813501588000]


def test_zero_seed_is_splitmix64_of_zero():
    # SmallRng::from_seed([0;32]) -> seed_from_u64(0): the seed every reference unit test uses
    r = orc.rng_ref(bytes(32))
    assert list(r.s) == [0xe220a8397b1dcdaf, 0x6e789e6aa1b965f4, 0x06c45d188009454f, 0xf88bb8a8724c81ec]
    got = [orc.lib().orc_rng_next_u64(C.byref(r)) for _ in range(4)]
    assert got == [5987356902031041503, 7051070477665621255, 6633766593972829180, 211316841551650330]


def test_from_seed_little_endian_and_child():
    seed = bytes(range(1, 33))
    r = orc.rng_ref(seed)
    assert r.s[0] == int.from_bytes(seed[:8], "little")
    assert r.s[3] == int.from_bytes(seed[24:], "little")
    # rng_get(): child seeded with 4 consecutive parent outputs (utils/random.rs:19-22)
    parent = orc.rng_ref_state([1, 2, 3, 4])
    child = orc.rng_ref_child(parent)
    assert list(child.s) == [41943041, 58720359, 3588806011781223, 3591011842654386]
    assert parent.raw_draws == 4


def _py_gen_range(next_u64, rng_range):
    lz = 64 - rng_range.bit_length()
    zone = ((rng_range << lz) & (2**64 - 1)) - 1
    zone &= 2**64 - 1
    while True:
        v = next_u64()
        m = v * rng_range
        if (m & (2**64 - 1)) <= zone:
            return m >> 64


def test_gen_range_u64_matches_python_restatement_and_rejects():
    a = orc.rng_ref(bytes(32))
    b = orc.rng_ref(bytes(32))
    nxt = lambda: orc.lib().orc_rng_next_u64(C.byref(b))
    for rng_range in [1, 2, 3, 5, 7, 16, 33, 34, 156, 1000, 2**31 + 5, 2**40 + 12345]:
        for _ in range(50):
            x = orc.lib().orc_rng_gen_range_u64(C.byref(a), rng_range)
            assert x == _py_gen_range(nxt, rng_range)
            assert 0 <= x < rng_range
    assert a.raw_draws == b.raw_draws
    # range=1 rejects ~50% of raw words (zone = 2^63-1): data-dependent stream position
    c = orc.rng_ref(bytes(32))
    for _ in range(2000):
        orc.lib().orc_rng_gen_range_u64(C.byref(c), 1)
    assert 3600 < c.raw_draws < 4400


def test_gen_range_floats():
    a = orc.rng_ref(bytes(32))
    b = orc.rng_ref(bytes(32))
    for _ in range(100):
        f = orc.lib().orc_rng_gen_range_f32(C.byref(a), 1.0)
        u = (orc.lib().orc_rng_next_u64(C.byref(b)) >> 32) >> 9      # next_u32 = upper half
        exp = np.array([0x3F800000 | u], dtype=np.uint32).view(np.float32)[0] - np.float32(1.0)
        assert np.float32(f) == exp and 0.0 <= f < 1.0
    for _ in range(100):
        f = orc.lib().orc_rng_gen_range_f64(C.byref(a), 0.2, 5.0)
        u = orc.lib().orc_rng_next_u64(C.byref(b)) >> 12
        v = np.array([0x3FF0000000000000 | u], dtype=np.uint64).view(np.float64)[0]
        assert f == (v - 1.0) * (5.0 - 0.2) + 0.2 and 0.2 <= f < 5.0


def test_philox4x32_10_random123_kat():
    assert orc.philox_raw([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert orc.philox_raw([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert orc.philox_raw([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_named_draw_is_two_level_philox():
    seed, call, tag, id_, d0, d1 = 0x1234567890abcdef, 77, orc.TAG_NS_HOMO, (5 << 32) | 9, 3, 1
    ck = orc.philox_raw([call & 0xffffffff, call >> 32, tag, 0x74636867], [seed & 0xffffffff, seed >> 32])
    exp = orc.philox_raw([d0, d1, id_ & 0xffffffff, id_ >> 32], ck[:2])
    assert orc.philox_named_draw(seed, call, tag, id_, d0, d1) == exp
