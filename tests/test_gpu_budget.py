"""GPU parity: budget_sampling through the operator surface == oracle philox-mode."""
import numpy as np
import pytest
import torch

import orc
from helpers import load_fake_hetero, rel_key

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


def _compare(tg, graph, inputs, in_ts, nn, hops, seed, rts=None, window=None, forward=False, relative=False):
    node_types, edge_types, P, I = graph
    tg.seed(seed)
    s, t, r, c, e = tg.budget_sampling(node_types, edge_types, _cuda(P), _cuda(I), _cuda(rts), _cuda(inputs),
                                       _cuda(in_ts), nn, hops, window, forward, relative)
    o = orc.budget(node_types, edge_types, P, I, rts, inputs, in_ts, nn, hops, orc.rng_philox(seed, 0), window=window,
                   forward=forward, relative=relative)
    for nt in node_types:
        assert np.array_equal(s[nt].cpu().numpy(), o[0][nt]), nt
        assert np.array_equal(t[nt].cpu().numpy(), o[1][nt]), nt
    for et in edge_types:
        k = rel_key(et)
        assert np.array_equal(r[k].cpu().numpy(), o[2][k]), k
        assert np.array_equal(c[k].cpu().numpy(), o[3][k]), k
        assert np.array_equal(e[k].cpu().numpy(), o[4][k]), k
    return o


def test_budget_reference_config(tg, graph):
    """budget_sampling.rs:401-499."""
    node_types, edge_types, P, I = graph
    g = np.random.default_rng(0)
    rts = {k: g.integers(0, 7, len(I[k])) for k in I}
    o = _compare(tg, graph, {t: [0, 1, 4, 5] for t in node_types}, {t: g.integers(0, 7, 4) for t in node_types},
                 {t: [3, 4] for t in node_types}, 2, 21, rts=rts, window=(0, 2), forward=False, relative=False)
    assert sum(len(v) for v in o[2].values()) > 20


@pytest.mark.parametrize("relative", [False, True])
@pytest.mark.parametrize("forward", [False, True])
def test_budget_filter_variants_three_hops(tg, graph, forward, relative):
    node_types, edge_types, P, I = graph
    g = np.random.default_rng(1)
    rts = {k: g.integers(-1, 12, len(I[k])) for k in list(I)[:4]}
    _compare(tg, graph, {"v0": [0, 1, 4, 5, 9], "v1": [2]}, {"v0": [3, -1, 5, 6, 7]}, {t: [4, 2, 2] for t in node_types},
             3, 5, rts=rts, window=(0, 4), forward=forward, relative=relative)


def test_budget_no_filter_large_quota_and_edge_cases(tg, graph):
    node_types = graph[0]
    _compare(tg, graph, {"v2": [1, 2, 3]}, None, {t: [60, 2] for t in node_types}, 2, 7)       # quota above any budget
    _compare(tg, graph, {t: list(range(0, 400, 7)) for t in node_types}, None, {t: [5, 5] for t in node_types}, 2, 8)
    _compare(tg, graph, {"v0": []}, None, {t: [5] for t in node_types}, 1, 9)
    _compare(tg, graph, {"v0": [1, 2]}, None, {t: [] for t in node_types}, 0, 10)
    with pytest.raises(RuntimeError, match="reference panics"):
        tg.budget_sampling(graph[0], graph[1], _cuda(graph[2]), _cuda(graph[3]), None, _cuda({"v2": [1]}), None,
                           {"v2": [1]}, 1, None, False, False)
