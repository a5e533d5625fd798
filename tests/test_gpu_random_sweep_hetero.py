"""Randomised parity sweep of the heterogeneous operators: random typed multigraphs (self relations, several relations
between the same pair of types, empty relations, node types nobody points to) and random configurations; every case
through the operator surface must equal the oracle's philox-mode bit for bit."""
import numpy as np
import pytest
import torch

import orc
from helpers import rel_key

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tg():
    import tch_geometric
    return tch_geometric


def random_hetero(rs):
    T = int(rs.integers(2, 5))
    node_types = ["t%d" % i for i in range(T)]
    counts = {t: int(rs.integers(3, 400)) for t in node_types}
    R = int(rs.integers(2, 7))
    edge_types, edges = [], {}
    for r in range(R):
        s, d = node_types[int(rs.integers(0, T))], node_types[int(rs.integers(0, T))]
        et = (s, "r%d" % r, d)
        e = int(rs.integers(0, 12 * max(counts[s], counts[d]))) if rs.random() > 0.1 else 0
        ei = np.stack([rs.integers(0, counts[s], e), rs.integers(0, counts[d], e)]).astype(np.int64).reshape(2, e)
        if e > 20:                                             # one heavy column
            ei[1, rs.integers(0, e, e // 4)] = rs.integers(0, counts[d])
        edge_types.append(et)
        edges[et] = ei
    return node_types, edge_types, counts, edges


def _cuda(d):
    return {k: torch.from_numpy(np.ascontiguousarray(np.asarray(v))).cuda() for k, v in d.items()} if d is not None else None


def _eq_dicts(got, want, keys, what):
    for k in keys:
        assert np.array_equal(got[k].cpu().numpy(), want[k]), (what, k)


@pytest.mark.parametrize("case", range(10))
def test_hetero_neighbor_sampling_random(tg, case):
    rs = np.random.default_rng(3000 + case)
    node_types, edge_types, counts, edges = random_hetero(rs)
    P, I = {}, {}
    for et in edge_types:
        P[rel_key(et)], I[rel_key(et)], _ = orc.to_csc(edges[et], (counts[et[0]], counts[et[2]]))
    hops = int(rs.integers(1, 4))
    nn = {rel_key(et): [int(rs.integers(1, 9)) for _ in range(hops)] for et in edge_types}
    inputs = {t: rs.integers(0, counts[t], int(rs.integers(1, 30))) for t in node_types if rs.random() > 0.3}
    if not inputs:
        inputs = {node_types[0]: rs.integers(0, counts[node_types[0]], 5)}
    variant = case % 4
    sampler, flt, kw = None, None, {}
    if variant == 1:
        sampler, kw = tg.UniformEdgeSampler(True), dict(sampler=orc.SAMPLER_UNIFORM_REPL)
    elif variant == 2:
        W = {r: rs.uniform(0.1, 4.0, len(I[r])) for r in I}
        sampler, kw = tg.WeightedEdgeSampler(_cuda(W)), dict(sampler=orc.SAMPLER_WEIGHTED, weights=W)
    elif variant == 3:
        TS = {r: rs.integers(0, 12, len(I[r])) for r in I}
        ST = {t: rs.integers(0, 12, len(v)) for t, v in inputs.items()}
        mode, fwd = int(rs.integers(0, 3)), bool(rs.integers(0, 2))
        flt = (tg.TemporalEdgeFilter((0, 5), _cuda(TS), fwd, mode), _cuda(ST))
        kw = dict(filter_mode=mode, forward=fwd, window=(0, 5), timestamps=TS, inputs_state=ST)
    tg.seed(case)
    s, r, c, e, lo = tg.neighbor_sampling_heterogenous(node_types, edge_types, _cuda(P), _cuda(I), _cuda(inputs), nn, hops,
                                                       sampler, flt)
    o = orc.ns_hetero(node_types, edge_types, P, I, inputs, nn, hops, orc.rng_philox(case, 0), **kw)
    _eq_dicts(s, o[0], node_types, "samples")
    rels = [rel_key(et) for et in edge_types]
    for name, got, want in (("rows", r, o[1]), ("cols", c, o[2]), ("edge_index", e, o[3])):
        _eq_dicts(got, want, rels, name)
    for k in rels:
        assert [tuple(x) for x in lo[k]] == o[4][k], k


@pytest.mark.parametrize("variant", ["temporal", "weighted", "weighted+temporal"])
def test_hetero_many_relations_take_several_rounds(tg, variant):
    """23 relations over 3 node types under a filter / with weights: a hop's relations go through the segmented flat hop
    in rounds of <= 16 entries and <= 8 frontier segments (tchgeo.h), and the rounds must add up to the reference's
    relation-by-relation order (oracle, bit for bit) -- list positions, layer offsets, frontier slices."""
    rs = np.random.default_rng(9100)
    node_types = ["a", "b", "c"]
    counts = {"a": 150, "b": 90, "c": 40}
    edge_types, edges = [], {}
    for r in range(23):
        s, d = node_types[int(rs.integers(0, 3))], node_types[int(rs.integers(0, 3))]
        if r in (4, 17):
            d = "c"                                            # "c" has no inputs: empty frontier in hop 0
        et = (s, "r%d" % r, d)
        e = 0 if r == 9 else int(rs.integers(50, 900))        # one relation without edges
        ei = np.stack([rs.integers(0, counts[s], e), rs.integers(0, counts[d], e)]).astype(np.int64).reshape(2, e)
        edge_types.append(et)
        edges[et] = ei
    P, I = {}, {}
    for et in edge_types:
        P[rel_key(et)], I[rel_key(et)], _ = orc.to_csc(edges[et], (counts[et[0]], counts[et[2]]))
    rels = [rel_key(et) for et in edge_types]
    hops = 3
    nn = {k: [int(rs.integers(1, 6)) for _ in range(hops)] for k in rels}
    inputs = {"a": rs.integers(0, counts["a"], 12), "b": rs.integers(0, counts["b"], 7)}
    sampler, flt, kw = None, None, {}
    if "weighted" in variant:
        W = {r: rs.uniform(0.1, 4.0, len(I[r])) for r in I}
        sampler, kw = tg.WeightedEdgeSampler(_cuda(W)), dict(sampler=orc.SAMPLER_WEIGHTED, weights=W)
    if "temporal" in variant:
        TS = {r: rs.integers(0, 12, len(I[r])) for r in I}
        ST = {t: rs.integers(0, 12, len(v)) for t, v in inputs.items()}
        flt = (tg.TemporalEdgeFilter((0, 6), _cuda(TS), True, 2), _cuda(ST))   # dynamic: the states travel too
        kw.update(filter_mode=2, forward=True, window=(0, 6), timestamps=TS, inputs_state=ST)
    tg.seed(77)
    s, r, c, e, lo = tg.neighbor_sampling_heterogenous(node_types, edge_types, _cuda(P), _cuda(I), _cuda(inputs), nn, hops,
                                                       sampler, flt)
    o = orc.ns_hetero(node_types, edge_types, P, I, inputs, nn, hops, orc.rng_philox(77, 0), **kw)
    assert sum(len(o[1][k]) for k in rels) > 500 and sum(1 for k in rels if len(o[1][k])) > 16
    _eq_dicts(s, o[0], node_types, "samples")
    for name, got, want in (("rows", r, o[1]), ("cols", c, o[2]), ("edge_index", e, o[3])):
        _eq_dicts(got, want, rels, name)
    for k in rels:
        assert [tuple(x) for x in lo[k]] == o[4][k], k


@pytest.mark.parametrize("case", range(8))
def test_hgt_and_budget_random(tg, case):
    rs = np.random.default_rng(4000 + case)
    node_types, edge_types, counts, edges = random_hetero(rs)
    P, I = {}, {}
    for et in edge_types:
        P[rel_key(et)], I[rel_key(et)], _ = orc.to_csc(edges[et], (counts[et[0]], counts[et[2]]))
    rels = [rel_key(et) for et in edge_types]
    hops = int(rs.integers(1, 4))
    inputs = {t: rs.integers(0, counts[t], int(rs.integers(1, 20))) for t in node_types if rs.random() > 0.4}
    if not inputs:
        inputs = {node_types[-1]: rs.integers(0, counts[node_types[-1]], 4)}
    temporal = case % 2 == 1
    rts = {k: rs.integers(-1, 20, len(I[k])) for k in rels} if temporal else None
    in_ts = {t: rs.integers(-1, 20, len(v)) for t, v in inputs.items()} if temporal else None
    ns = {t: [int(rs.integers(1, 40)) for _ in range(hops)] for t in node_types}
    timerange = (2, 15) if temporal and case % 4 == 1 else None
    tg.seed(100 + case)
    got = tg.hgt_sampling(node_types, edge_types, _cuda(P), _cuda(I), _cuda(rts), _cuda(inputs), _cuda(in_ts), ns, hops,
                          timerange)
    want = orc.hgt(node_types, edge_types, P, I, rts, inputs, in_ts, ns, hops, orc.rng_philox(100 + case, 0),
                   timerange=timerange)
    for j, keys in enumerate((node_types, node_types, rels, rels, rels)):
        _eq_dicts(got[j], want[j], keys, "hgt[%d]" % j)
    window = (0, 8) if temporal else None
    fwd, rel = bool(rs.integers(0, 2)), bool(rs.integers(0, 2))
    got = tg.budget_sampling(node_types, edge_types, _cuda(P), _cuda(I), _cuda(rts), _cuda(inputs), _cuda(in_ts), ns, hops,
                             window, fwd, rel)
    want = orc.budget(node_types, edge_types, P, I, rts, inputs, in_ts, ns, hops, orc.rng_philox(100 + case, 1),
                      window=window, forward=fwd, relative=rel)
    for j, keys in enumerate((node_types, node_types, rels, rels, rels)):
        _eq_dicts(got[j], want[j], keys, "budget[%d]" % j)


@pytest.mark.parametrize("case", range(6))
def test_hetero_negative_sampling_random(tg, case):
    rs = np.random.default_rng(5000 + case)
    node_types, edge_types, counts, edges = random_hetero(rs)
    P, I, sizes = {}, {}, {}
    for et in edge_types:
        k = rel_key(et)
        sizes[k] = (counts[et[0]], counts[et[2]])
        P[k], I[k], _ = orc.to_csr(edges[et], sizes[k])
    has_out = {et[0] for et in edge_types}                    # a type without outgoing relation panics in the reference
    inputs = {t: rs.integers(0, counts[t], int(rs.integers(1, 40))) for t in node_types if t in has_out and rs.random() > 0.3}
    if not inputs:
        t0 = edge_types[0][0]
        inputs = {t0: rs.integers(0, counts[t0], 7)}
    num_neg, tries = int(rs.integers(0, 5)), int(rs.integers(1, 5))
    tg.seed(200 + case)
    s, r, c, cnt = tg.negative_sample_neighbors_heterogenous(node_types, edge_types, _cuda(P), _cuda(I), sizes, _cuda(inputs),
                                                             num_neg, tries, False)
    o = orc.neg_hetero(node_types, edge_types, P, I, sizes, inputs, num_neg, tries, False, orc.rng_philox(200 + case, 0))
    _eq_dicts(s, o[0], node_types, "samples")
    rels = [rel_key(et) for et in edge_types]
    _eq_dicts(r, o[1], rels, "rows")
    _eq_dicts(c, o[2], rels, "cols")
    assert {t: int(cnt[t]) for t in node_types} == {t: int(o[3][t]) for t in node_types}


def test_large_hetero_call_leaves_the_fused_kernel(tg):
    """more than 4096 inputs in one call: the host drives whole-device hops per relation; results unchanged"""
    rs = np.random.default_rng(77)
    node_types, edge_types, counts, edges = random_hetero(rs)
    P, I = {}, {}
    for et in edge_types:
        P[rel_key(et)], I[rel_key(et)], _ = orc.to_csc(edges[et], (counts[et[0]], counts[et[2]]))
    inputs = {t: rs.integers(0, counts[t], 3000) for t in node_types}
    nn = {rel_key(et): [3, 2] for et in edge_types}
    tg.seed(5)
    s, r, c, e, lo = tg.neighbor_sampling_heterogenous(node_types, edge_types, _cuda(P), _cuda(I), _cuda(inputs), nn, 2)
    o = orc.ns_hetero(node_types, edge_types, P, I, inputs, nn, 2, orc.rng_philox(5, 0))
    _eq_dicts(s, o[0], node_types, "samples")
    rels = [rel_key(et) for et in edge_types]
    _eq_dicts(r, o[1], rels, "rows")
    _eq_dicts(c, o[2], rels, "cols")
    _eq_dicts(e, o[3], rels, "edge_index")
    for k in rels:
        assert [tuple(x) for x in lo[k]] == o[4][k]
