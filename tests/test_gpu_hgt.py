"""GPU parity: hgt_sampling through the operator surface == oracle philox-mode (canonical orders)."""
import numpy as np
import pytest
import torch

import orc
from helpers import has_edge, load_fake_hetero, rel_key

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tg():
    import tch_geometric
    return tch_geometric


@pytest.fixture(scope="module")
def graph():
    counts, edges = load_fake_hetero()
    node_types, edge_types = sorted(counts), sorted(edges)
    P, I = {}, {}
    for et in edge_types:
        P[rel_key(et)], I[rel_key(et)], _ = orc.to_csc(edges[et], (counts[et[0]], counts[et[2]]))
    return node_types, edge_types, P, I


def _cuda(d):
    return {k: torch.from_numpy(np.asarray(v, dtype=np.int64)).cuda() for k, v in d.items()} if d is not None else None


def _compare(tg, graph, inputs, in_ts, ns, hops, seed, rts=None, timerange=None):
    node_types, edge_types, P, I = graph
    tg.seed(seed)
    s, t, r, c, e = tg.hgt_sampling(node_types, edge_types, _cuda(P), _cuda(I), _cuda(rts), _cuda(inputs),
                                    _cuda(in_ts), ns, hops, timerange)
    o = orc.hgt(node_types, edge_types, P, I, rts, inputs, in_ts, ns, hops, orc.rng_philox(seed, 0),
                timerange=timerange)
    for nt in node_types:
        assert np.array_equal(s[nt].cpu().numpy(), o[0][nt]), nt
        assert np.array_equal(t[nt].cpu().numpy(), o[1][nt]), nt
    for et in edge_types:
        k = rel_key(et)
        assert np.array_equal(r[k].cpu().numpy(), o[2][k]), k
        assert np.array_equal(c[k].cpu().numpy(), o[3][k]), k
        assert np.array_equal(e[k].cpu().numpy(), o[4][k]), k
    return o


def test_hgt_reference_config(tg, graph):
    """hgt_sampling.rs:356-429: inputs [0,1,4,5] per type, [20,15] per type, 2 hops."""
    node_types, edge_types, P, I = graph
    o = _compare(tg, graph, {t: [0, 1, 4, 5] for t in node_types}, None, {t: [20, 15] for t in node_types}, 2, 11)
    for et in edge_types:
        k = rel_key(et)
        for j, i in zip(o[2][k], o[3][k]):                                  # :300-305
            assert has_edge(P[k], I[k], o[0][et[2]][i], o[0][et[0]][j])


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_hgt_larger_quotas_three_hops(tg, graph, seed):
    node_types = graph[0]
    _compare(tg, graph, {"v0": list(range(0, 120, 3)), "v2": [5, 5, 9]},      # a duplicate input
             None, {t: [64, 48, 32] for t in node_types}, 3, seed)


def test_hgt_timestamps_and_timerange(tg, graph):
    node_types, edge_types, P, I = graph
    g = np.random.default_rng(5)
    rts = {k: g.integers(-1, 30, len(I[k])) for k in I}
    _compare(tg, graph, {"v0": [0, 1, 4, 5], "v2": [7, 8]}, {"v0": [3, 10, -1, 20], "v2": [5, 25]},
             {t: [10, 6] for t in node_types}, 2, 9, rts=rts, timerange=(5, 20))
    some = {k: v for k, v in list(rts.items())[:3]}                            # timestamps on some relations only
    _compare(tg, graph, {"v1": [2, 3]}, {"v1": [7, 8]}, {t: [30, 30] for t in node_types}, 2, 10, rts=some)


def test_hgt_quota_above_budget_and_edge_cases(tg, graph):
    node_types = graph[0]
    _compare(tg, graph, {"v1": [3]}, None, {t: [1000, 1000] for t in node_types}, 2, 4)     # take whole budgets
    _compare(tg, graph, {t: list(range(60)) for t in node_types}, None, {t: [] for t in node_types}, 0, 5)
    _compare(tg, graph, {"v0": []}, None, {t: [5] for t in node_types}, 1, 6)               # empty input list
    with pytest.raises(RuntimeError, match="reference panics"):                             # :202
        tg.hgt_sampling(graph[0], graph[1], _cuda(graph[2]), _cuda(graph[3]), None, _cuda({"v1": [3]}), None,
                        {"v1": [5, 5]}, 2)


def test_hgt_more_than_8192_samples_per_layer(tg):
    """quotas above what the LDS slot tables hold (8192) use global slot tables; results still equal the oracle"""
    rs = np.random.default_rng(21)
    nA, nB = 40000, 30000
    e1 = np.stack([rs.integers(0, nA, 400000), rs.integers(0, nB, 400000)])     # A -> B
    e2 = np.stack([rs.integers(0, nB, 300000), rs.integers(0, nA, 300000)])     # B -> A
    node_types, edge_types = ["A", "B"], [("A", "x", "B"), ("B", "y", "A")]
    P, I = {}, {}
    P["A__x__B"], I["A__x__B"], _ = orc.to_csc(e1, (nA, nB))
    P["B__y__A"], I["B__y__A"], _ = orc.to_csc(e2, (nB, nA))
    g = (node_types, edge_types, P, I)
    o = _compare(tg, g, {"B": rs.integers(0, nB, 3000)}, None, {"A": [9000, 12000], "B": [10000, 100]}, 2, 3)
    assert len(o[0]["A"]) > 8192


def test_hgt_hub_sources_long_contribution_runs(tg):
    """Budget entries that hundreds of samples contribute to (hubs, parallel edges): their score is still the left-to-right
    f64 sum in contribution order -- the runs here span one lane (<= 8), one wavefront (<= 64) and the wavefront-wide
    radix sort (> 64, up to ~1500 contributions to one entry)."""
    rs = np.random.default_rng(77)
    nA, nB = 5000, 3000
    src, dst = [], []
    for b in range(nB):                         # A -> B: the first neighbours of every column are hubs 0..5
        hubs = [0, 0, 1] + [2] * (b % 3) + ([3] if b % 2 else []) + ([4] if b % 7 == 0 else []) + ([5] if b % 40 == 0 else [])
        rest = rs.integers(6, nA, int(rs.integers(0, 12))).tolist()
        for a in hubs + rest:
            src.append(a), dst.append(b)
    e1 = np.stack([np.asarray(src), np.asarray(dst)])
    e2 = np.stack([rs.integers(0, nB, 40000), rs.integers(0, nA, 40000)])     # B -> A
    node_types, edge_types = ["A", "B"], [("A", "x", "B"), ("B", "y", "A")]
    P, I = {}, {}
    P["A__x__B"], I["A__x__B"], _ = orc.to_csc(e1, (nA, nB))
    P["B__y__A"], I["B__y__A"], _ = orc.to_csc(e2, (nB, nA))
    g = (node_types, edge_types, P, I)
    for seed in (1, 2):
        _compare(tg, g, {"B": rs.permutation(nB)[:700]}, None, {"A": [300, 200], "B": [600, 100]}, 2, seed)
    _compare(tg, g, {"B": np.arange(60)}, None, {"A": [50, 50, 50], "B": [70, 70, 70]}, 3, 3)


@pytest.mark.parametrize("seed", [5, 6])
def test_hgt_many_node_types_and_relations(tg, seed):
    """11 node types and 26 relations: more than one launch's 8 types (init, sample_from), more than 4 independent
    update_budget steps per round, relations that feed the same budget (kept in canonical order), more than 8 relations in
    the edge rebuild; timestamps on some relations."""
    rs = np.random.default_rng(900 + seed)
    T = 11
    node_types = ["n%02d" % i for i in range(T)]
    counts = {t: int(rs.integers(40, 300)) for t in node_types}
    edge_types, P, I, rts = [], {}, {}, {}
    pairs = [(i, (i + 1) % T) for i in range(T)] + [(i, (i + 3) % T) for i in range(T)] + [(0, 5), (5, 0), (2, 2), (7, 2)]
    for r, (a, b) in enumerate(pairs):
        et = (node_types[a], "r%02d" % r, node_types[b])
        e = int(rs.integers(200, 3000))
        ei = np.stack([rs.integers(0, counts[et[0]], e), rs.integers(0, counts[et[2]], e)])
        ei[0, rs.integers(0, e, e // 5)] = rs.integers(0, 3)                    # a few busy sources
        edge_types.append(et)
        k = rel_key(et)
        P[k], I[k], _ = orc.to_csc(ei, (counts[et[0]], counts[et[2]]))
        if r % 3 == 0:
            rts[k] = rs.integers(-1, 25, len(I[k]))
    g = (node_types, edge_types, P, I)
    inputs = {t: rs.integers(0, counts[t], int(rs.integers(3, 25))) for t in node_types[::2]}
    _compare(tg, g, inputs, None, {t: [int(rs.integers(5, 60)), int(rs.integers(5, 60)), 20] for t in node_types}, 3, seed,
             rts=rts)
