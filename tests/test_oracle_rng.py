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


def test_bounded_word_is_lemires_exact_method():
    """philox-mode's slot draw (round 4): one 32-bit word, multiply-shift, exact rejection.  The accepted words of a range
    map onto every value equally often: counted in closed form (accepted words with value v = words w with
    (w * range) >> 32 == v and low half >= 2^32 mod range), and spot-checked against the C function."""
    for rng_range in [1, 2, 3, 5, 6, 7, 10, 1000, 65537, 370000, 2**20 + 3, 2**31 - 1, 2**31 + 5, 2**32 - 1]:
        t = (2**32) % rng_range
        per_value = (2**32 - t) // rng_range            # what Lemire's proof promises: every value equally often
        assert (2**32 - t) % rng_range == 0
        # count the accepted words of three values exactly: words w with floor(w * range / 2^32) == v are
        # ceil(v * 2^32 / range) .. ceil((v + 1) * 2^32 / range) - 1; the rejected ones are the first ones with low half < t
        for v in {0, rng_range // 2, rng_range - 1}:
            w_lo = -((-v * 2**32) // rng_range)
            w_hi = -((-(v + 1) * 2**32) // rng_range)
            rejected = 0
            w = w_lo
            while w < w_hi and ((w * rng_range) & 0xFFFFFFFF) < t:   # low halves rise by `range` per word: only a prefix can be < t
                rejected += 1
                w += 1
            assert (w_hi - w_lo) - rejected == per_value
            # the C function agrees on the boundary words
            for ww in {w_lo, min(w_lo + rejected, w_hi - 1), w_hi - 1}:
                val, ok = orc.bounded_word(ww, rng_range)
                assert val == (ww * rng_range) >> 32 == v
                assert ok == (((ww * rng_range) & 0xFFFFFFFF) >= t)
    assert orc.bounded_word(0, 3) == (0, False)          # 2^32 mod 3 = 1: the word 0 is the one rejected word of value 0
    assert orc.bounded_word(1, 3) == (0, True)


def test_slot_draw_words_blocks_and_fallback():
    """Slot s takes word s & 3 of block s >> 2; a rejected word is replaced by the 64-bit draw of block (s, d1 | 'F')."""
    seed, call, tag, d1 = 7, 3, orc.TAG_NS_HOMO, 0
    fell = 0
    for id_ in range(40):
        for s in range(12):
            for rng_range in (9, 370001, 2**31 + 12345, 2**32 - 5):
                r = orc.rng_philox(seed, call)
                got, fb = orc.slot_draw(r, tag, id_, d1, s, rng_range)
                w = orc.philox_named_draw(seed, call, tag, id_, s >> 2, d1)[s & 3]
                m = w * rng_range
                accept = (m & 0xFFFFFFFF) >= (2**32) % rng_range
                assert fb == (not accept)
                if accept:
                    assert got == m >> 32
                else:
                    f = orc.philox_named_draw(seed, call, tag, id_, s, d1 | 0x46)
                    assert got == ((f[0] | (f[1] << 32)) * rng_range) >> 64
                    fell += 1
                assert 0 <= got < rng_range
    assert fell > 100        # the two large ranges reject about half / nearly all of their words
    # ranges of 2^32 and more never look at the word
    r = orc.rng_philox(seed, call)
    got, fb = orc.slot_draw(r, tag, 5, d1, 2, 2**40 + 7)
    f = orc.philox_named_draw(seed, call, tag, 5, 2, d1 | 0x46)
    assert fb and got == ((f[0] | (f[1] << 32)) * (2**40 + 7)) >> 64
