"""Oracle budget_sampling (src/algo/budget_sampling.rs, SURVEY 8(f) "next" row): the reference test config and the
structural properties of the algorithm."""
import numpy as np
import pytest

import orc
from helpers import load_fake_hetero, rel_key


def graph():
    counts, edges = load_fake_hetero()
    node_types, edge_types = sorted(counts), sorted(edges)
    P, I = {}, {}
    for et in edge_types:
        P[rel_key(et)], I[rel_key(et)], _ = orc.to_csc(edges[et], (counts[et[0]], counts[et[2]]))
    return node_types, edge_types, P, I


def validate(node_types, edge_types, P, I, samples, ts, rows, cols, eidx, n_in, quotas, hops):
    for et in edge_types:
        k = rel_key(et)
        src, dst = samples[et[0]], samples[et[2]]
        for i, j, col_idx in zip(rows[k], cols[k], eidx[k]):
            w = dst[j]
            assert 0 <= col_idx < min(50, P[k][w + 1] - P[k][w])                 # :100 a column prefix, :116 its index
            assert I[k][P[k][w] + col_idx] == src[i]                             # the edge is real (:268-275 validator)
        assert len(rows[k]) == len(set(rows[k].tolist()))                       # each new node has exactly one parent
    for t in node_types:
        per = np.zeros(len(samples[t]), dtype=int)
        for et in edge_types:
            if et[2] == t:
                np.add.at(per, cols[rel_key(et)], 1)
        assert per.max(initial=0) <= max(quotas[t][:hops] or [0])                # <= num_neighbors per node
        assert len(ts[t]) == len(samples[t])
    total_new = sum(len(samples[t]) - n_in.get(t, 0) for t in node_types)
    assert total_new == sum(len(v) for v in rows.values())                        # a forest: one edge per new node


@pytest.mark.parametrize("mode", ["ref", "philox"])
def test_budget_reference_config(mode):
    """budget_sampling.rs:401-499: edge ts U{0..7}, inputs [0,1,4,5] per type with ts U{0..7}, [3,4], 2 hops,
    window 0..2 backward, not relative."""
    node_types, edge_types, P, I = graph()
    g = np.random.default_rng(0)
    RTS = {k: g.integers(0, 7, len(I[k])) for k in I}
    inputs = {t: [0, 1, 4, 5] for t in node_types}
    in_ts = {t: g.integers(0, 7, 4) for t in node_types}
    nn = {t: [3, 4] for t in node_types}
    rng = orc.rng_ref() if mode == "ref" else orc.rng_philox(2)
    s, ts, r, c, e = orc.budget(node_types, edge_types, P, I, RTS, inputs, in_ts, nn, 2, rng, window=(0, 2),
                                forward=False, relative=False)
    validate(node_types, edge_types, P, I, s, ts, r, c, e, {t: 4 for t in node_types}, nn, 2)
    for et in edge_types:                                                        # the window really filtered
        k = rel_key(et)
        for i, j, col_idx in zip(r[k], c[k], e[k]):
            w_t, v_t = ts[et[2]][j], RTS[k][P[k][s[et[2]][j]] + col_idx]
            assert 0 <= -(v_t - w_t) < 2
            assert ts[et[0]][i] == v_t                                           # not relative: the edge's time moves on


@pytest.mark.parametrize("relative", [False, True])
@pytest.mark.parametrize("forward", [False, True])
def test_budget_filter_variants(forward, relative):
    node_types, edge_types, P, I = graph()
    g = np.random.default_rng(1)
    RTS = {k: g.integers(-1, 12, len(I[k])) for k in list(I)[:4]}               # two relations without timestamps
    inputs = {"v0": [0, 1, 4, 5, 9], "v1": [2]}
    in_ts = {"v0": [3, -1, 5, 6, 7]}                                             # v1 gets NAN timestamps (:195)
    nn = {t: [4, 2, 2] for t in node_types}
    s, ts, r, c, e = orc.budget(node_types, edge_types, P, I, RTS, inputs, in_ts, nn, 3, orc.rng_philox(5),
                                window=(0, 4), forward=forward, relative=relative)
    validate(node_types, edge_types, P, I, s, ts, r, c, e, {"v0": 5, "v1": 1}, nn, 3)


def test_budget_without_filter_and_panics():
    node_types, edge_types, P, I = graph()
    nn = {t: [60, 2] for t in node_types}                                        # quota above any budget of a node
    s, ts, r, c, e = orc.budget(node_types, edge_types, P, I, None, {"v2": [1, 2, 3]}, None, nn, 2, orc.rng_philox(7))
    validate(node_types, edge_types, P, I, s, ts, r, c, e, {"v2": 3}, nn, 2)
    assert all(np.all(t == -1) for t in ts.values())
    with pytest.raises(RuntimeError):                                            # :226 missing num_neighbors key
        orc.budget(node_types, edge_types, P, I, None, {"v2": [1]}, None, {"v2": [1]}, 1, orc.rng_philox(7))
