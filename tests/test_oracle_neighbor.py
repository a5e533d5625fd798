"""Oracle neighbor sampling (src/algo/neighbor_sampling.rs).

ref-mode (sequential Xoshiro256++) is cross-checked against the derived
known-answer vectors of SURVEY.md App. B (an independent restatement -- not
output of the Rust binary: parity unpinned) and against the reference's own
invariant tests; philox-mode runs the same invariants."""
import ctypes as C

import numpy as np
import pytest

import orc
from helpers import load_fake_hetero, load_karate, rel_key, roots_of, validate_neighbor_samples

INPUTS = [0, 1, 4, 5]


@pytest.fixture(scope="module")
def karate_csc():
    ei, n = load_karate()
    ptrs, idx, _ = orc.to_csc(ei, n)
    return ptrs, idx


def test_b1_baseline_cfg1(karate_csc):
    """BASELINE cfg1: karate, inputs [0,1,4,5], fanout [5,5], default sampler, SmallRng::from_seed([0;32])."""
    ptrs, idx = karate_csc
    rng = orc.rng_ref()
    samples, rows, cols, eidx, lo = orc.ns_homo(ptrs, idx, INPUTS, [5, 5], rng)
    assert rng.raw_draws == 56
    assert lo == [(4, 0, 4), (21, 17, 21)]
    assert samples.tolist() == [0, 1, 4, 5, 12, 31, 8, 4, 19, 17, 2, 19, 30, 21, 0, 6, 10, 0, 6, 10, 16, 0, 3, 0, 33,
                                25, 28, 32, 0, 2, 30, 32, 33, 0, 6, 10, 0, 1, 33, 0, 1, 9, 32, 3, 28, 8, 0, 1, 33, 1,
                                8, 32, 33, 0, 1, 8, 2, 31, 12, 5, 0, 4, 5, 16, 0, 4, 5, 21, 7, 12, 6, 5, 0, 4, 5, 16,
                                0, 4, 5, 5, 6]
    assert rows.tolist() == list(range(4, 81))
    assert cols.tolist() == [0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 2, 2, 2, 3, 3, 3, 3, 4, 4, 5, 5, 5, 5, 5, 6, 6, 6, 6, 6, 7,
                             7, 7, 8, 8, 8, 9, 9, 10, 10, 10, 10, 10, 11, 11, 11, 12, 12, 12, 12, 13, 13, 14, 14, 14,
                             14, 14, 15, 15, 15, 15, 16, 16, 16, 17, 17, 17, 17, 17, 18, 18, 18, 18, 19, 19, 19, 20, 20]
    assert eidx.tolist() == [10, 15, 7, 3, 13, 21, 17, 22, 24, 23, 41, 42, 43, 44, 45, 46, 47, 67, 68, 121, 126, 123,
                             124, 125, 56, 57, 58, 59, 60, 41, 42, 43, 84, 85, 86, 80, 81, 30, 34, 27, 33, 29, 84, 85,
                             86, 117, 118, 119, 120, 89, 90, 7, 1, 15, 10, 4, 48, 49, 50, 51, 63, 64, 65, 14, 6, 10, 5,
                             4, 48, 49, 50, 51, 63, 64, 65, 78, 79]
    validate_neighbor_samples(ptrs, idx, rows, cols, samples, samples, lo, [5, 5])


def test_b2_reference_unit_test_config_with_replacement(karate_csc):
    """neighbor_sampling.rs:437-464: fanout [4,3], UnweightedSampler<true>."""
    ptrs, idx = karate_csc
    rng = orc.rng_ref()
    samples, rows, cols, eidx, lo = orc.ns_homo(ptrs, idx, INPUTS, [4, 3], rng, sampler=orc.SAMPLER_UNIFORM_REPL)
    assert rng.raw_draws == 123
    assert lo == [(4, 0, 4), (20, 16, 20)]
    assert cols.tolist() == [i for i in range(4) for _ in range(4)] + [i for i in range(4, 20) for _ in range(3)]
    assert samples.tolist() == [0, 1, 4, 5, 6, 7, 1, 1, 2, 3, 2, 0, 6, 6, 0, 10, 0, 6, 6, 16, 16, 5, 5, 1, 1, 0, 0, 2,
                                3, 17, 21, 2, 0, 27, 7, 7, 12, 0, 8, 32, 8, 13, 21, 19, 5, 16, 16, 5, 16, 16, 11, 7,
                                12, 5, 4, 0, 13, 13, 4, 5, 16, 5, 0, 0, 0, 6, 6, 6]
    assert eidx.tolist() == [5, 6, 0, 0, 17, 18, 17, 16, 42, 42, 41, 43, 44, 45, 45, 47, 51, 50, 50, 53, 53, 52, 16,
                             17, 18, 21, 23, 17, 25, 32, 28, 38, 39, 35, 29, 34, 29, 11, 14, 13, 50, 51, 51, 50, 51,
                             51, 9, 6, 10, 65, 64, 63, 11, 11, 3, 50, 51, 50, 48, 48, 48, 79, 79, 79]
    validate_neighbor_samples(ptrs, idx, rows, cols, samples, samples, lo, [4, 3])


def test_b3_without_replacement(karate_csc):
    ptrs, idx = karate_csc
    rng = orc.rng_ref()
    samples, rows, cols, eidx, lo = orc.ns_homo(ptrs, idx, INPUTS, [4, 3], rng)
    assert rng.raw_draws == 80
    assert (len(samples), len(rows)) == (59, 55) and lo == [(4, 0, 4), (19, 15, 19)]
    assert samples[4:19].tolist() == [21, 19, 10, 17, 13, 17, 21, 7, 0, 6, 10, 0, 6, 10, 16]
    assert eidx[:15].tolist() == [14, 13, 8, 12, 20, 21, 23, 19, 41, 42, 43, 44, 45, 46, 47]


def _rng(mode, call=0):
    return orc.rng_ref() if mode == "ref" else orc.rng_philox(0xC0FFEE, call)


@pytest.mark.parametrize("mode", ["ref", "philox"])
@pytest.mark.parametrize("algo", [orc.RES_TICKETS, orc.RES_LITERAL])
def test_uniform_invariants(karate_csc, mode, algo):
    ptrs, idx = karate_csc
    for sampler in (orc.SAMPLER_UNIFORM, orc.SAMPLER_UNIFORM_REPL):
        samples, rows, cols, eidx, lo = orc.ns_homo(ptrs, idx, INPUTS, [4, 3], _rng(mode), sampler=sampler,
                                                    reservoir_algo=algo)
        validate_neighbor_samples(ptrs, idx, rows, cols, samples, samples, lo, [4, 3])
        assert rows.tolist() == list(range(4, 4 + len(rows)))          # forest: rows[e] = n_inputs + e
        assert np.array_equal(samples[rows], idx[eidx])                # sample = indices[edge_ptr]
        deg = np.diff(ptrs)
        per_dst = np.bincount(cols, minlength=len(samples))
        front = np.arange(lo[1][2])                                    # all vertices that were expanded
        k_of = np.where(front < 4, 4, 3)
        if sampler == orc.SAMPLER_UNIFORM:                             # exactly min(deg, k), distinct edges
            assert np.array_equal(per_dst[front], np.minimum(deg[samples[front]], k_of))
            for i in front:
                e = eidx[cols == i]
                assert len(set(e.tolist())) == len(e)
        else:                                                          # exactly k if deg > 0
            assert np.array_equal(per_dst[front], np.where(deg[samples[front]] > 0, k_of, 0))


@pytest.mark.parametrize("mode", ["ref", "philox"])
def test_weighted_invariants(karate_csc, mode):
    """neighbor_sampling.rs:466-495 (weights U(0.2, 5.0) f64 drawn from the same rng first)."""
    ptrs, idx = karate_csc
    rng = _rng(mode)
    if mode == "ref":
        w = np.array([orc.lib().orc_rng_gen_range_f64(C.byref(rng), 0.2, 5.0) for _ in range(len(idx))])
    else:
        w = np.random.default_rng(1).uniform(0.2, 5.0, len(idx))
    samples, rows, cols, eidx, lo = orc.ns_homo(ptrs, idx, INPUTS, [4, 3], rng, sampler=orc.SAMPLER_WEIGHTED,
                                                weights=w)
    validate_neighbor_samples(ptrs, idx, rows, cols, samples, samples, lo, [4, 3])
    assert np.array_equal(samples[rows], idx[eidx])


def test_weighted_zero_weight_sum_is_the_reference_panic(karate_csc):
    ptrs, idx = karate_csc
    with pytest.raises(RuntimeError):
        orc.ns_homo(ptrs, idx, INPUTS, [2, 2], orc.rng_ref(), sampler=orc.SAMPLER_WEIGHTED,
                    weights=np.zeros(len(idx)))


@pytest.mark.parametrize("mode", ["ref", "philox"])
def test_temporal_static_and_relative(karate_csc, mode):
    """neighbor_sampling.rs:497-570."""
    ptrs, idx = karate_csc
    rng = _rng(mode)
    if mode == "ref":
        ts = np.array([orc.lib().orc_rng_gen_range_u64(C.byref(rng), 4) for _ in range(len(idx))], dtype=np.int64)
    else:
        ts = np.random.default_rng(2).integers(0, 4, len(idx))
    in_ts = [0, 1, 2, 3]
    samples, rows, cols, eidx, lo = orc.ns_homo(ptrs, idx, INPUTS, [4, 3], rng, filter_mode=orc.FILTER_STATIC,
                                                window=(0, 2), timestamps=ts, inputs_state=in_ts)
    validate_neighbor_samples(ptrs, idx, rows, cols, samples, samples, lo, [4, 3])
    assert np.all((ts[eidx] >= 0) & (ts[eidx] <= 2))
    samples, rows, cols, eidx, lo = orc.ns_homo(ptrs, idx, INPUTS, [4, 3], rng, filter_mode=orc.FILTER_RELATIVE,
                                                forward=False, window=(0, 2), timestamps=ts, inputs_state=in_ts)
    validate_neighbor_samples(ptrs, idx, rows, cols, samples, samples, lo, [4, 3])
    root = roots_of(cols, 4, len(samples))
    t0 = np.array(in_ts)[root[rows]]
    assert np.all((ts[eidx] >= t0 - 2) & (ts[eidx] <= t0))


@pytest.mark.parametrize("forward", [True, False])
def test_temporal_dynamic_state_follows_the_edge(karate_csc, forward):
    ptrs, idx = karate_csc
    ts = np.random.default_rng(3).integers(0, 6, len(idx))
    in_ts = [3, 3, 3, 3]
    samples, rows, cols, eidx, lo = orc.ns_homo(ptrs, idx, INPUTS, [4, 3], orc.rng_philox(5), forward=forward,
                                                filter_mode=orc.FILTER_DYNAMIC, window=(0, 1), timestamps=ts,
                                                inputs_state=in_ts)
    state = np.zeros(len(samples), dtype=np.int64)
    state[:4] = in_ts
    for e, (j, i) in enumerate(zip(rows, cols)):
        d = ts[eidx[e]] - state[i]
        d = d if forward else -d
        assert 0 <= d <= 1
        state[j] = ts[eidx[e]]


def _hetero_graph():
    counts, edges = load_fake_hetero()
    node_types = sorted(counts)
    edge_types = sorted(edges)
    P, I = {}, {}
    for et in edge_types:
        s, _, d = et
        p, i, _ = orc.to_csc(edges[et], (counts[s], counts[d]))
        P[rel_key(et)], I[rel_key(et)] = p, i
    return node_types, edge_types, P, I


@pytest.mark.parametrize("mode", ["ref", "philox"])
def test_hetero_reference_test_config(mode):
    """neighbor_sampling.rs:572-648: inputs [0,1,4,5] per type, [4,3] per relation, 2 hops."""
    node_types, edge_types, P, I = _hetero_graph()
    inputs = {t: INPUTS for t in node_types}
    nn = {rel_key(e): [4, 3] for e in edge_types}
    samples, rows, cols, eidx, los = orc.ns_hetero(node_types, edge_types, P, I, inputs, nn, 2, _rng(mode))
    for et in edge_types:
        r = rel_key(et)
        validate_neighbor_samples(P[r], I[r], rows[r], cols[r], samples[et[0]], samples[et[2]], los[r], [4, 3])
        assert np.array_equal(samples[et[0]][rows[r]], I[r][eidx[r]])
        assert len(los[r]) == 2 and los[r][0][1] == 0


def test_hetero_with_one_relation_equals_homogeneous():
    ei, n = load_karate()
    ptrs, idx, _ = orc.to_csc(ei, n)
    hs, hr, hc, he, hlo = orc.ns_homo(ptrs, idx, INPUTS, [4, 3], orc.rng_ref())
    et = ("a", "to", "a")
    s, r, c, e, lo = orc.ns_hetero(["a"], [et], {rel_key(et): ptrs}, {rel_key(et): idx}, {"a": INPUTS},
                                   {rel_key(et): [4, 3]}, 2, orc.rng_ref())
    assert np.array_equal(s["a"], hs) and np.array_equal(r[rel_key(et)], hr)
    assert np.array_equal(c[rel_key(et)], hc) and np.array_equal(e[rel_key(et)], he) and lo[rel_key(et)] == hlo


def test_empty_inputs_and_isolated_vertices():
    ptrs = np.array([0, 0, 2, 2], dtype=np.int64)      # vertex 1 has in-neighbours {0, 2}
    idx = np.array([0, 2], dtype=np.int64)
    s, r, c, e, lo = orc.ns_homo(ptrs, idx, [], [3, 3], orc.rng_ref())
    assert len(s) == 0 and len(r) == 0 and lo == [(0, 0, 0), (0, 0, 0)]
    s, r, c, e, lo = orc.ns_homo(ptrs, idx, [0, 1, 2, 1], [3, 3], orc.rng_philox(1))
    assert s.tolist() == [0, 1, 2, 1, 0, 2, 0, 2] and c.tolist() == [1, 1, 3, 3]
    assert lo == [(4, 0, 4), (8, 4, 8)]


def test_fakedataset_fixture_invariants_in_both_modes():
    """the reference's third data file (tests/fakedataset.npz): validate_neighbor_samples in ref- and philox-mode"""
    from helpers import load_fake_dataset
    ei, n = load_fake_dataset()
    assert ei.shape == (2, 22648) and n == 1144
    ptrs, idx, _ = orc.to_csc(ei, n)
    seeds = np.tile(np.arange(8), 4)
    for rng in (orc.rng_ref(), orc.rng_philox(1, 0)):
        s, r, c, e, lo = orc.ns_homo(ptrs, idx, seeds, [4, 3], rng)
        validate_neighbor_samples(ptrs, idx, r, c, s, s, lo, [4, 3])
        assert np.array_equal(s[:len(seeds)], seeds) and np.array_equal(idx[e], s[r])
