"""CPU half of the law check (runs without a GPU): the oracle's philox-mode operators (what the device equals bit for
bit) against its ref-mode (the reference's stream and literal loops) -- chi-square homogeneity on whole-operator
outputs.  tests/test_gpu_distribution_vs_ref.py repeats it with the device producing the philox side."""
import numpy as np

import orc
from dist_helpers import assert_different_law, assert_same_law
from helpers import load_karate

N = 40000


def _slots(ptrs, idx, k, sampler, rng, weights=None):
    o = orc.ns_homo(ptrs, idx, np.zeros(N, dtype=np.int64), [k], rng, sampler=sampler, weights=weights)
    return o[3].reshape(N, -1) - int(ptrs[0])


def test_neighbor_sampling_laws_agree():
    ei, n = load_karate()
    ptrs, idx, _ = orc.to_csc(ei, n)
    w = np.ones(idx.size)
    w[:16] = np.arange(1, 17)
    for sampler, weights in ((0, None), (1, None), (2, w)):
        a = _slots(ptrs, idx, 5, sampler, orc.rng_philox(11, 3), weights)
        b = _slots(ptrs, idx, 5, sampler, orc.rng_ref_child(orc.rng_ref()), weights)
        for s in range(5):
            assert_same_law(a[:, s], b[:, s], "sampler %d slot %d" % (sampler, s), n_bins=16)
        assert_same_law(a[:, 0] * 16 + a[:, 4], b[:, 0] * 16 + b[:, 4], "sampler %d slots (0,4)" % sampler, n_bins=256)
    a = _slots(ptrs, idx, 5, 0, orc.rng_philox(11, 3))
    b = _slots(ptrs, idx, 5, 1, orc.rng_ref_child(orc.rng_ref()))
    assert_different_law(a[:, 0], b[:, 0], "no-replace vs replace", n_bins=16)


def test_random_walk_laws_agree():
    ei, n = load_karate()
    rptrs, ridx, _ = orc.to_csr(ei, n)
    start = np.zeros(N, dtype=np.int64)
    a = orc.random_walk(rptrs, ridx, start, 3, 0.5, 2.0, orc.rng_philox(5, 0))
    b = orc.random_walk(rptrs, ridx, start, 3, 0.5, 2.0, orc.rng_ref_child(orc.rng_ref()))
    for step in (1, 2, 3):
        assert_same_law(a[:, step], b[:, step], "walk step %d" % step, n_bins=n)
    assert_same_law(a[:, 1] * n + a[:, 2], b[:, 1] * n + b[:, 2], "walk steps (1,2)", n_bins=n * n)
