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


# ---------------------------------------------------------------- round 4: the operators whose philox-mode is not the
# literal loop (blocked running sums, reservoir by tickets inside other operators, addressed candidate draws)
def hgt_law_graph(duplicates=True):
    """Two node types; the input a0 has the in-neighbours b0 x3, b1 x2, b2..b5 x1 (multi-edges: budget scores 3/9, 2/9,
    1/9 ..., hgt_sampling.rs:72-98), so `sample_from` (:104-135) draws 2 of 6 budget entries with weights score^2."""
    rows = [0, 0, 0, 1, 1, 2, 3, 4, 5] if duplicates else [0, 1, 2, 3, 4, 5]
    ei = np.array([rows, [0] * len(rows)], dtype=np.int64)
    P, I, _ = orc.to_csc(ei, (6, 1))
    return ["a", "b"], [("b", "r", "a")], {"b__r__a": P}, {"b__r__a": I}


def hgt_outcomes(rng_of_call, n_calls, duplicates=True):
    nt, et, P, I = hgt_law_graph(duplicates)
    out = np.empty(n_calls, dtype=np.int64)
    for c in range(n_calls):
        s, _, _, _, _ = orc.hgt(nt, et, P, I, None, {"a": [0]}, None, {"a": [2], "b": [2]}, 1, rng_of_call(c))
        assert len(s["b"]) == 2
        out[c] = s["b"][0] * 6 + s["b"][1]
    return out


def test_hgt_sample_from_laws_agree():
    n_calls = 12000
    parent = orc.rng_ref()
    a = hgt_outcomes(lambda c: orc.rng_philox(21, c), n_calls)
    b = hgt_outcomes(lambda c: orc.rng_ref_child(parent), n_calls)
    assert_same_law(a, b, "hgt sample_from ordered pair", n_bins=36)
    assert_same_law(a // 6, b // 6, "hgt sample_from slot 0", n_bins=6)
    assert_same_law(a % 6, b % 6, "hgt sample_from slot 1", n_bins=6)
    flat = hgt_outcomes(lambda c: orc.rng_philox(21, c), n_calls, duplicates=False)     # all scores equal: another law
    assert_different_law(a, flat, "hgt scores 9:4:1:1:1:1 vs equal scores", n_bins=36)


def biased_law_graph():
    """vertex 0 has 12 out-edges to vertices 1..12 with edge times 5, 5, 6, 7, 7, 7, 9, 12, 12, 20, 30, 31 (ties on
    purpose: the linear bias ranks by argsort, random_walk.rs:171-173); the walker starts at time 5, so every edge is a
    candidate (:228-251)."""
    ts = np.array([5, 5, 6, 7, 7, 7, 9, 12, 12, 20, 30, 31], dtype=np.int64)
    n = 13
    ptrs = np.zeros(n + 1, dtype=np.int64)
    ptrs[1:] = 12
    idx = np.arange(1, 13, dtype=np.int64)
    return ptrs, idx, np.full(n, -1, dtype=np.int64), ts


def test_biased_walk_first_step_laws_agree():
    ptrs, idx, node_ts, edge_ts = biased_law_graph()
    start, start_ts = np.zeros(N, dtype=np.int64), np.full(N, 5, dtype=np.int64)
    first = {}
    for bias in ("uniform", "linear", "exponential"):
        a, _ = orc.biased_tempo_random_walk(ptrs, idx, node_ts, edge_ts, start, start_ts, 2, bias, True, 1,
                                            orc.rng_philox(31, 1))
        b, _ = orc.biased_tempo_random_walk(ptrs, idx, node_ts, edge_ts, start, start_ts, 2, bias, True, 1,
                                            orc.rng_ref_child(orc.rng_ref()))
        assert_same_law(a[:, 1], b[:, 1], "biased walk (%s) step 1" % bias, n_bins=13)
        first[bias] = a[:, 1]
    assert_different_law(first["uniform"], first["linear"], "uniform vs linear bias", n_bins=13)
    assert_different_law(first["uniform"], first["exponential"], "uniform vs exponential bias", n_bins=13)


def test_negative_sampling_laws_agree():
    """negative_sampling.rs:31-45: a candidate is uniform over the node range and retried while it is a neighbour, so the
    accepted nodes of an input are uniform over its non-neighbours."""
    ei, n = load_karate()
    rptrs, ridx, _ = orc.to_csr(ei, n)
    inputs = np.zeros(N // 4, dtype=np.int64)                     # vertex 0: 16 neighbours of 34 nodes

    def accepted(rng, try_count):
        s, rows, cols, _ = orc.neg_homo(rptrs, ridx, (n, n), inputs, 4, try_count, rng)
        return s[cols]

    a = accepted(orc.rng_philox(41, 2), 8)
    b = accepted(orc.rng_ref_child(orc.rng_ref()), 8)
    assert len(a) > 0.95 * N and len(b) > 0.95 * N
    assert_same_law(a, b, "negatives of vertex 0", n_bins=n)
    one_try = accepted(orc.rng_philox(41, 2), 1)                  # fewer retries: fewer accepted, same conditional law ...
    assert len(one_try) < 0.7 * len(a)
    hub_last = accepted(orc.rng_philox(41, 2), 8)
    inputs[:] = 33                                                # ... but another input vertex has another law
    other = accepted(orc.rng_philox(41, 2), 8)
    assert_different_law(hub_last, other, "negatives of vertex 0 vs vertex 33", n_bins=n)


def test_budget_sampling_laws_agree():
    """budget_sampling.rs:137-151: `Budget::sample` is reservoir_sampling over the node's candidate list (the first <= 50
    column entries); philox-mode draws it by tickets."""
    ei, n = load_karate()
    P, I, _ = orc.to_csc(ei, n)
    nt, et = ["a"], [("a", "r", "a")]
    m = N // 2
    inputs = {"a": np.zeros(m, dtype=np.int64)}                   # vertex 0: 16 candidates, 5 kept per input

    def slots(rng, k):
        s, ts, rows, cols, eidx = orc.budget(nt, et, {"a__r__a": P}, {"a__r__a": I}, None, inputs, None, {"a": [k]}, 1, rng)
        e = eidx["a__r__a"]
        assert len(e) == m * k and np.array_equal(cols["a__r__a"], np.repeat(np.arange(m), k))
        return e.reshape(m, k)

    a = slots(orc.rng_philox(51, 4), 5)
    b = slots(orc.rng_ref_child(orc.rng_ref()), 5)
    for s in range(5):
        assert_same_law(a[:, s], b[:, s], "budget slot %d" % s, n_bins=16)
    assert_same_law(a[:, 0] * 16 + a[:, 3], b[:, 0] * 16 + b[:, 3], "budget slots (0,3)", n_bins=256)
    assert_different_law(a[:, 0], slots(orc.rng_philox(51, 4), 3)[:, 0], "k = 5 vs k = 3, slot 0", n_bins=16)
