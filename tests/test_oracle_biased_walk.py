"""Oracle checks for biased_tempo_random_walk (random_walk.rs:160-288): the reference's own invariant test
(:398-450) in both RNG modes, the weight definitions of BiasType::apply as selection frequencies, the panic and the
restart quirks.  Sampled values are "parity unpinned" (the reference holds no vectors; see oracle/orc_rng.h)."""
import numpy as np
import pytest

import orc
from helpers import load_karate


def _karate_csr():
    ei, n = load_karate()
    ptrs, idx, _ = orc.to_csr(ei, n)
    return ptrs, idx, n


@pytest.mark.parametrize("mode", ["ref", "philox"])
@pytest.mark.parametrize("bias", ["uniform", "linear", "exponential"])
def test_reference_invariants_on_karate(mode, bias):
    ptrs, idx, n = _karate_csr()
    rs = np.random.default_rng(0)
    nts, ets = rs.integers(-1, 5, n), rs.integers(-1, 5, len(idx))
    start, sts = np.array([0, 1, 2, 3]), np.array([0, -1, 2, 3])
    rng = orc.rng_ref() if mode == "ref" else orc.rng_philox(3, 0)
    walks, wts = orc.biased_tempo_random_walk(ptrs, idx, nts, ets, start, sts, 10, bias, True, 10, rng)
    assert walks.shape == (4, 10) and np.array_equal(walks[:, 0], start) and np.array_equal(wts[:, 0], sts)
    for i in range(4):                                      # random_walk.rs:441-447
        if sts[i] == -1:
            continue
        t = wts[i][wts[i] != -1]
        assert np.all(t >= sts[i])
        assert np.all(np.diff(t) >= 0)                      # time never runs backwards along a walk
    valid = walks[walks >= 0]
    assert valid.size and valid.max() < n


def _star(times, node_times=None):
    """vertex 0 -> vertices 1..k with the given edge timestamps"""
    k = len(times)
    ptrs = np.array([0, k] + [k] * k)
    idx = np.arange(1, k + 1)
    return ptrs, idx, (np.zeros(k + 1, dtype=np.int64) if node_times is None else np.asarray(node_times)), np.asarray(times)


def _frequencies(times, bias, forward, n=60000, t0=0, seed=5):
    ptrs, idx, nts, ets = _star(times)
    start, sts = np.zeros(n, dtype=np.int64), np.full(n, t0)
    walks, _ = orc.biased_tempo_random_walk(ptrs, idx, nts, ets, start, sts, 2, bias, forward, 1, orc.rng_philox(seed, 0))
    return np.bincount(walks[:, 1], minlength=len(times) + 1)[1:] / n


def test_bias_weights_are_selection_frequencies():
    times = [3, 9, 5, 9, 4]
    f = _frequencies(times, "uniform", True)
    assert np.allclose(f, 0.2, atol=0.01)
    # linear: weight[c] = argsort(descending)[c] / sum -- the permutation itself, random_walk.rs:171-173
    perm = np.array(sorted(range(5), key=lambda c: (-times[c], c)))
    f = _frequencies(times, "linear", True)
    assert np.allclose(f, perm / perm.sum(), atol=0.01)
    # exponential: softmax(t - times) forward, softmax(times - t) backward (:175-178)
    for forward in (True, False):
        d = np.array([-t for t in times] if forward else times, dtype=np.float64)
        want = np.exp(d - d.max()) / np.exp(d - d.max()).sum()
        f = _frequencies(times, "exponential", forward)
        assert np.allclose(f, want, atol=0.01)
    # walkers without a timestamp see uniform weights whatever the bias (:258-259)
    f = _frequencies(times, "exponential", True, t0=-1)
    assert np.allclose(f, 0.2, atol=0.01)


def test_candidates_respect_time_and_unknown_timestamps():
    ptrs, idx, nts, ets = _star([1, 5, -1, 7], node_times=[0, 0, 0, -1, 0])
    n = 4000
    walks, wts = orc.biased_tempo_random_walk(ptrs, idx, nts, ets, np.zeros(n, dtype=np.int64), np.full(n, 5), 2, "uniform",
                                              True, 1, orc.rng_philox(1, 0))
    assert set(np.unique(walks[:, 1])) == {2, 3, 4}         # edge at t=1 lies in the past; unknown (-1) passes
    assert np.all(wts[walks[:, 1] == 3, 1] == -1)           # an unknown timestamp is reported as -1 (:283-285)


def test_underflowed_weights_panic():
    ptrs, idx, nts, ets = _star([0, 0, 1000])
    with pytest.raises(RuntimeError):                       # first two softmax weights are exp(-1000) = 0
        orc.biased_tempo_random_walk(ptrs, idx, nts, ets, np.array([0]), np.array([0]), 2, "exponential", False, 1,
                                     orc.rng_philox(1, 0))


def test_restart_keeps_stale_timestamps():
    # 0 -> 1 -> 4 (dead end), 0 -> 2 (dead end): an attempt through 1 writes a timestamp at position 2 that a later,
    # shorter attempt through 2 does not erase (random_walk.rs:223-225 resets the vertices only)
    ptrs = np.array([0, 2, 3, 3, 3, 3])
    idx = np.array([1, 2, 4])
    ets = np.array([5, 7, 6])
    nts = np.zeros(5, dtype=np.int64)
    seen_stale = False
    for call in range(40):
        walks, wts = orc.biased_tempo_random_walk(ptrs, idx, nts, ets, np.array([0]), np.array([0]), 4, "uniform", True, 2,
                                                  orc.rng_philox(2, call))
        assert walks[0, 0] == 0 and walks[0, 3] == -1
        if walks[0, 1] == 2:
            assert walks[0, 2] == -1 and wts[0, 1] == 7
            seen_stale |= wts[0, 2] == 6
    assert seen_stale
    w0, t0 = orc.biased_tempo_random_walk(ptrs, idx, nts, ets, np.array([0]), np.array([0]), 4, "uniform", True, 0,
                                          orc.rng_philox(2, 0))
    assert np.all(w0 == -1) and np.all(t0 == -1)            # retry_count == 0: nothing is written (:217)
