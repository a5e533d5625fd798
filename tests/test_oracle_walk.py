"""Oracle random walks (src/algo/random_walk.rs)."""
import ctypes as C

import numpy as np
import pytest

import orc
from helpers import has_edge, load_karate


@pytest.fixture(scope="module")
def karate_csr():
    ei, n = load_karate()
    ptrs, idx, _ = orc.to_csr(ei, n)
    return ptrs, idx, n


def test_b4_random_walk_vectors(karate_csr):
    """SURVEY App. B4 (derived vectors; reference test config random_walk.rs:301-331)."""
    ptrs, idx, _ = karate_csr
    rng = orc.rng_ref()
    w = orc.random_walk(ptrs, idx, [0, 1, 2, 3], 10, 1.0, 1.5, rng)
    assert rng.raw_draws == 122
    assert w.tolist() == [[0, 6, 4, 6, 4, 0, 2, 13, 2, 1, 3], [1, 7, 0, 7, 3, 1, 2, 7, 0, 4, 0],
                          [2, 1, 13, 3, 12, 3, 12, 3, 2, 1, 13], [3, 13, 1, 7, 2, 8, 2, 0, 4, 10, 5]]
    rng = orc.rng_ref()
    w = orc.random_walk(ptrs, idx, [0, 1, 2, 3], 10, 1.0, 1.0, rng)
    assert rng.raw_draws == 104
    assert w.tolist() == [[0, 6, 4, 6, 16, 5, 0, 2, 13, 2, 1], [1, 3, 2, 13, 0, 7, 3, 1, 2, 7, 0],
                          [2, 3, 1, 17, 0, 17, 1, 0, 17, 1, 21], [3, 7, 3, 12, 3, 13, 33, 32, 8, 32, 14]]


@pytest.mark.parametrize("mode", ["ref", "philox"])
@pytest.mark.parametrize("pq", [(1.0, 1.5), (1.0, 1.0), (0.25, 4.0), (4.0, 0.25)])
def test_random_walk_invariants(karate_csr, mode, pq):
    ptrs, idx, _ = karate_csr
    rng = orc.rng_ref() if mode == "ref" else orc.rng_philox(9)
    start = [0, 1, 2, 3, 33]
    w = orc.random_walk(ptrs, idx, start, 10, pq[0], pq[1], rng)
    assert w.shape == (5, 11) and w[:, 0].tolist() == start          # random_walk.rs:322-330
    for row in w:
        for a, b in zip(row[:-1], row[1:]):
            assert has_edge(ptrs, idx, a, b)


def test_random_walk_dead_end_pads_minus_one():
    ptrs = np.array([0, 1, 1, 2], dtype=np.int64)   # 0 -> 1, 1 has no out-edges, 2 -> 0
    idx = np.array([1, 0], dtype=np.int64)
    w = orc.random_walk(ptrs, idx, [2, 1], 4, 1.0, 1.0, orc.rng_philox(1))
    assert w.tolist() == [[2, 0, 1, -1, -1], [1, -1, -1, -1, -1]]


@pytest.mark.parametrize("mode", ["ref", "philox"])
@pytest.mark.parametrize("algo", [orc.RES_TICKETS, orc.RES_LITERAL])
def test_tempo_random_walk_window(karate_csr, mode, algo):
    """random_walk.rs:333-383: ts U{-1..4}, start ts [0,-1,2,3], L=10, window (0,2)."""
    ptrs, idx, n = karate_csr
    rng = orc.rng_ref() if mode == "ref" else orc.rng_philox(4)
    g = np.random.default_rng(7)
    node_ts = g.integers(-1, 4, n)
    edge_ts = g.integers(-1, 4, len(idx))
    start, start_ts = [0, 1, 2, 3], [0, -1, 2, 3]
    w, wt = orc.tempo_random_walk(ptrs, idx, node_ts, edge_ts, start, start_ts, 10, (0, 2), rng, reservoir_algo=algo)
    assert w.shape == (4, 10) and wt.shape == (4, 10)
    assert w[:, 0].tolist() == start and wt[:, 0].tolist() == start_ts
    for i in range(4):
        t0 = start_ts[i]
        if t0 == -1:
            continue
        for t in wt[i, 1:]:
            if t != -1:
                assert t0 <= t < t0 + 2
    assert np.all(w >= 0)   # a tempo walk restarts instead of stopping (random_walk.rs:144-148)


def test_tempo_walk_k1_reservoir_quirk():
    """With one slot the reservoir index for the 2nd candidate is drawn from 0..1, so candidate 0 can only
    survive when it is the only one (sampling.rs:17-22)."""
    ptrs = np.array([0, 3, 3, 3, 3], dtype=np.int64)
    idx = np.array([1, 2, 3], dtype=np.int64)
    z = np.full(4, -1, dtype=np.int64)
    for mode in ("ref", "philox"):
        seen = set()
        ref = orc.rng_ref()
        for c in range(64):
            rng = ref if mode == "ref" else orc.rng_philox(3, c)   # ref: one stream, 64 consecutive calls
            w, _ = orc.tempo_random_walk(ptrs, idx, z, np.full(3, -1), [0], [-1], 2, (0, 1), rng)
            seen.add(int(w[0, 1]))
        assert seen == {2, 3}


def test_tempo_walk_chunked_reservoir_law():
    """philox-mode's one-slot reservoir draws once per chunk of 64 row positions (orc_reservoir_one_chunked); its law is the
    literal loop's: with n admissible neighbours never candidate 0, every other one with probability 1/(n-1).  A row of 150
    neighbours (three chunks), every second one outside the window, 40 000 walkers; ref-mode (the literal loop on the
    reference's stream) measured the same way."""
    n_nb = 150
    ptrs = np.zeros(n_nb + 2, dtype=np.int64)
    ptrs[1:] = n_nb                                     # vertex 0 -> 1..150, everyone else has no out-edges
    idx = np.arange(1, n_nb + 1, dtype=np.int64)
    edge_ts = np.where(np.arange(n_nb) % 2 == 0, 5, 50).astype(np.int64)     # even positions admissible (window [0, 10))
    node_ts = np.full(n_nb + 1, -1, dtype=np.int64)
    walkers = 40000
    start, start_ts = np.zeros(walkers, dtype=np.int64), np.zeros(walkers, dtype=np.int64)
    n_cand = n_nb // 2
    for rng in (orc.rng_philox(123), orc.rng_ref()):
        w, _ = orc.tempo_random_walk(ptrs, idx, node_ts, edge_ts, start, start_ts, 2, (0, 10), rng)
        picks = w[:, 1]
        assert np.all((picks - 1) % 2 == 0)                                  # admissible neighbours only
        counts = np.bincount((picks - 1) // 2, minlength=n_cand)
        assert counts[0] == 0                                                # candidate 0 never survives (sampling.rs:19)
        expect = walkers / (n_cand - 1)
        chi2 = float(((counts[1:] - expect) ** 2 / expect).sum())
        assert chi2 < 130, chi2                                              # 73 degrees of freedom: mean 73, sd 12
