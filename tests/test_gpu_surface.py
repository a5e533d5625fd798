"""The operator surface (C++ host module mirroring src/python.rs) on a GPU: same call shapes as the
reference, results equal to the oracle's philox-mode for the seeded (seed, call counter) state."""
import numpy as np
import pytest
import torch

import orc
from helpers import load_fake_hetero, load_karate, rel_key, validate_neighbor_samples

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tg():
    import tch_geometric
    return tch_geometric


def _np(t):
    return t.cpu().numpy()


@pytest.mark.parametrize("where", ["cpu", "cuda"])
def test_to_csc_to_csr(tg, where):
    ei, n = load_karate()
    t = torch.from_numpy(ei).to(where)
    for fn, ofn in ((tg.to_csc, orc.to_csc), (tg.to_csr, orc.to_csr)):
        p, i, perm = fn(t, n)
        op, oi, operm = ofn(ei, n)
        assert p.device.type == where
        assert np.array_equal(_np(p), op) and np.array_equal(_np(i), oi) and np.array_equal(_np(perm), operm)
    # storage.rs:165-184 + GraphSize as a tuple
    e2 = torch.tensor([[1, 2, 3, 4, 9, 5, 6, 7], [0, 0, 0, 1, 4, 1, 2, 2]]).to(where)
    p, i, _ = tg.to_csc(e2, (10, 10))
    assert _np(i[p[0]:p[1]]).tolist() == [1, 2, 3] and _np(i[p[1]:p[2]]).tolist() == [4, 5]


@pytest.mark.parametrize("where", ["cpu", "cuda"])
def test_neighbor_sampling_homogenous_call_shapes(tg, where):
    ei, n = load_karate()
    ptrs, idx, _ = orc.to_csc(ei, n)
    P, I = torch.from_numpy(ptrs).to(where), torch.from_numpy(idx).to(where)
    inputs = torch.tensor([0, 1, 4, 5]).to(where)
    tg.seed(1234)
    # trailing Optionals omitted, as the reference's examples do (examples/neighbor_sampling.py:18-20)
    s, r, c, e, lo = tg.neighbor_sampling_homogenous(P, I, inputs, [5, 5])
    o = orc.ns_homo(ptrs, idx, [0, 1, 4, 5], [5, 5], orc.rng_philox(1234, 0))
    assert s.device.type == where and isinstance(lo, list) and isinstance(lo[0], tuple)
    assert lo == o[4]
    for a, b in zip((s, r, c, e), o[:4]):
        assert a.dtype == torch.int64 and np.array_equal(_np(a), b)
    # second call consumes the next call id; sampler objects are duck-typed
    s, r, c, e, lo = tg.neighbor_sampling_homogenous(P, I, inputs, [4, 3], tg.UniformEdgeSampler(True), None)
    o = orc.ns_homo(ptrs, idx, [0, 1, 4, 5], [4, 3], orc.rng_philox(1234, 1), sampler=orc.SAMPLER_UNIFORM_REPL)
    assert np.array_equal(_np(s), o[0]) and np.array_equal(_np(e), o[3])
    validate_neighbor_samples(ptrs, idx, _np(r), _np(c), _np(s), _np(s), lo, [4, 3])
    assert tg.rng_state() == (1234, 2)


def test_weighted_and_temporal_arguments(tg):
    ei, n = load_karate()
    ptrs, idx, _ = orc.to_csc(ei, n)
    P, I = torch.from_numpy(ptrs).cuda(), torch.from_numpy(idx).cuda()
    inputs = torch.tensor([0, 1, 4, 5]).cuda()
    g = np.random.default_rng(0)
    w, ts = g.uniform(0.2, 5.0, len(idx)), g.integers(0, 4, len(idx))
    st = np.array([0, 1, 2, 3])
    tg.seed(9)
    out = tg.neighbor_sampling_homogenous(P, I, inputs, [4, 3], tg.WeightedEdgeSampler(torch.from_numpy(w).cuda()))
    o = orc.ns_homo(ptrs, idx, [0, 1, 4, 5], [4, 3], orc.rng_philox(9, 0), sampler=orc.SAMPLER_WEIGHTED, weights=w)
    assert np.array_equal(_np(out[0]), o[0]) and np.array_equal(_np(out[3]), o[3])
    for mode, fwd in ((tg.TEMPORAL_SAMPLE_STATIC, False), (tg.TEMPORAL_SAMPLE_RELATIVE, False),
                      (tg.TEMPORAL_SAMPLE_RELATIVE, True), (tg.TEMPORAL_SAMPLE_DYNAMIC, True)):
        flt = (tg.TemporalEdgeFilter((0, 2), torch.from_numpy(ts).cuda(), fwd, mode), torch.from_numpy(st).cuda())
        call = tg.rng_state()[1]
        out = tg.neighbor_sampling_homogenous(P, I, inputs, [4, 3], None, flt)
        o = orc.ns_homo(ptrs, idx, [0, 1, 4, 5], [4, 3], orc.rng_philox(9, call), filter_mode=mode,
                        forward=True if mode == 0 else fwd, window=(0, 2), timestamps=ts, inputs_state=st)
        assert out[4] == o[4]
        for a, b in zip(out[:4], o[:4]):
            assert np.array_equal(_np(a), b)
    # an unknown mode falls back to the identity filter (python.rs:249)
    flt = (tg.TemporalEdgeFilter((0, 0), torch.from_numpy(ts).cuda(), False, 7), torch.from_numpy(st).cuda())
    call = tg.rng_state()[1]
    out = tg.neighbor_sampling_homogenous(P, I, inputs, [4, 3], None, flt)
    o = orc.ns_homo(ptrs, idx, [0, 1, 4, 5], [4, 3], orc.rng_philox(9, call))
    assert np.array_equal(_np(out[0]), o[0])


def test_error_behaviour(tg):
    P, I = torch.tensor([0, 1, 2]).cuda(), torch.tensor([1, 0]).cuda()
    with pytest.raises(ValueError, match="Expected Int64 but got Float"):       # utils/tensor.rs:14-15
        tg.neighbor_sampling_homogenous(P, I, torch.tensor([0.5]).cuda(), [2])
    with pytest.raises(ValueError, match="Expected Int64 but got Int"):
        tg.neighbor_sampling_homogenous(P.int(), I, torch.tensor([0]).cuda(), [2])
    with pytest.raises(ValueError, match="Expected Double but got Float"):      # weights are f64 (python.rs:214)
        tg.neighbor_sampling_homogenous(P, I, torch.tensor([0]).cuda(), [2],
                                        tg.WeightedEdgeSampler(torch.ones(2).cuda()))
    with pytest.raises(ValueError):
        tg.neighbor_sampling_homogenous(P, I, torch.tensor([0]).cuda(), [0])
    with pytest.raises(ValueError, match="Unknown bias type: cubic"):           # python.rs:670
        tg.biased_tempo_random_walk(P, I, torch.zeros(2, dtype=torch.int64).cuda(), torch.zeros(2, dtype=torch.int64).cuda(),
                                    torch.tensor([0]).cuda(), torch.tensor([0]).cuda(), 3, "cubic", True, 2)
    assert issubclass(tg.PanicException, RuntimeError)
    with pytest.raises(tg.PanicException):                                      # sampling.rs:49 panic
        tg.neighbor_sampling_homogenous(torch.tensor([0, 2, 2]).cuda(), torch.tensor([0, 1]).cuda(),
                                        torch.tensor([0]).cuda(), [1],      # 2 candidates, 1 slot: one float draw
                                        tg.WeightedEdgeSampler(torch.zeros(2, dtype=torch.float64).cuda()))


@pytest.mark.parametrize("variant", ["uniform", "replace", "weighted", "temporal"])
def test_neighbor_sampling_heterogenous(tg, variant):
    """neighbor_sampling.rs:572-648 config (inputs [0,1,4,5] per type, [4,3] per relation, 2 hops)."""
    counts, edges = load_fake_hetero()
    node_types, edge_types = sorted(counts), sorted(edges)
    P, I, Pd, Id = {}, {}, {}, {}
    for et in edge_types:
        p, i, _ = orc.to_csc(edges[et], (counts[et[0]], counts[et[2]]))
        P[rel_key(et)], I[rel_key(et)] = p, i
        Pd[rel_key(et)], Id[rel_key(et)] = torch.from_numpy(p).cuda(), torch.from_numpy(i).cuda()
    inputs = {t: [0, 1, 4, 5] for t in node_types}
    inputs_d = {t: torch.tensor(v).cuda() for t, v in inputs.items()}
    nn = {rel_key(e): [4, 3] for e in edge_types}
    g = np.random.default_rng(3)
    kw, sampler, flt = {}, None, None
    if variant == "replace":
        sampler, kw = tg.UniformEdgeSampler(True), dict(sampler=orc.SAMPLER_UNIFORM_REPL)
    elif variant == "weighted":
        W = {r: g.uniform(0.2, 5.0, len(I[r])) for r in I}
        sampler = tg.WeightedEdgeSampler({r: torch.from_numpy(w).cuda() for r, w in W.items()})
        kw = dict(sampler=orc.SAMPLER_WEIGHTED, weights=W)
    elif variant == "temporal":
        TS = {r: g.integers(0, 10, len(I[r])) for r in I}
        ST = {t: np.array([3, 4, 5, 6]) for t in node_types}
        flt = (tg.TemporalEdgeFilter((0, 4), {r: torch.from_numpy(t).cuda() for r, t in TS.items()}, True,
                                     tg.TEMPORAL_SAMPLE_DYNAMIC),
               {t: torch.from_numpy(v).cuda() for t, v in ST.items()})
        kw = dict(filter_mode=orc.FILTER_DYNAMIC, forward=True, window=(0, 4), timestamps=TS, inputs_state=ST)
    tg.seed(77)
    s, r, c, e, lo = tg.neighbor_sampling_heterogenous(node_types, edge_types, Pd, Id, inputs_d, nn, 2, sampler, flt)
    os_, or_, oc, oe, olo = orc.ns_hetero(node_types, edge_types, P, I, inputs, nn, 2, orc.rng_philox(77, 0), **kw)
    for t in node_types:
        assert np.array_equal(_np(s[t]), os_[t]), t
    for et in edge_types:
        k = rel_key(et)
        assert [tuple(x) for x in lo[k]] == olo[k], k
        assert np.array_equal(_np(r[k]), or_[k]) and np.array_equal(_np(c[k]), oc[k]) and np.array_equal(_np(e[k]), oe[k])
        validate_neighbor_samples(P[k], I[k], _np(r[k]), _np(c[k]), _np(s[et[0]]), _np(s[et[2]]), lo[k], [4, 3])


def test_walks_through_the_surface(tg):
    ei, n = load_karate()
    ptrs, idx, _ = orc.to_csr(ei, n)
    P, I = torch.from_numpy(ptrs).cuda(), torch.from_numpy(idx).cuda()
    tg.seed(5)
    w = tg.random_walk(P, I, torch.tensor([0, 1, 2, 3]).cuda(), 10, 1.0, 1.5)     # random_walk.rs:301-331
    assert w.shape == (4, 11)
    assert np.array_equal(_np(w), orc.random_walk(ptrs, idx, [0, 1, 2, 3], 10, 1.0, 1.5, orc.rng_philox(5, 0)))
    w = tg.random_walk(P.cpu(), I.cpu(), torch.tensor([0, 1, 2, 3]), 10, 1.0, 1.0)
    assert w.device.type == "cpu"
    assert np.array_equal(_np(w), orc.random_walk(ptrs, idx, [0, 1, 2, 3], 10, 1.0, 1.0, orc.rng_philox(5, 1)))
    g = np.random.default_rng(7)
    nts, ets = g.integers(-1, 4, n), g.integers(-1, 4, len(idx))
    a, b = tg.tempo_random_walk(P, I, torch.from_numpy(nts).cuda(), torch.from_numpy(ets).cuda(),
                                torch.tensor([0, 1, 2, 3]).cuda(), torch.tensor([0, -1, 2, 3]).cuda(), 10, (0, 2))
    oa, ob = orc.tempo_random_walk(ptrs, idx, nts, ets, [0, 1, 2, 3], [0, -1, 2, 3], 10, (0, 2), orc.rng_philox(5, 2))
    assert a.shape == (4, 10) and np.array_equal(_np(a), oa) and np.array_equal(_np(b), ob)


def test_large_node2vec_calls_build_and_reuse_the_edge_set(tg):
    """p != q asks has_edge per proposal: a call of >= 2^20 walker steps builds the graph's edge set, later calls on the same
    tensors (small ones too) answer has_edge from it; walks equal the oracle's either way; a graph changed in place gets a
    new set"""
    rs = np.random.default_rng(21)
    n = 1 << 12
    ei = np.stack([rs.integers(0, n, n * 12), rs.integers(0, n, n * 12)])
    ptrs, idx, _ = orc.to_csr(ei, n)
    P, I = torch.from_numpy(ptrs).cuda(), torch.from_numpy(idx).cuda()
    tg.graph_cache_clear()
    small = rs.integers(0, n, 100)
    tg.seed(8)
    w0 = tg.random_walk(P, I, torch.from_numpy(small).cuda(), 20, 0.5, 2.0)
    assert tg.graph_cache_info()["edge_sets"] == 0                                  # a small call does not pay for the set
    assert np.array_equal(_np(w0), orc.random_walk(ptrs, idx, small, 20, 0.5, 2.0, orc.rng_philox(8, 0)))
    big = rs.integers(0, n, 1 << 15)
    w1 = tg.random_walk(P, I, torch.from_numpy(big).cuda(), 40, 0.5, 2.0)           # 1.3 M steps: builds it
    info = tg.graph_cache_info()
    assert info["edge_sets"] == 1 and info["edge_set_builds"] == 1 and info["edge_set_bytes"] >= 16 * len(idx)
    assert np.array_equal(_np(w1), orc.random_walk(ptrs, idx, big, 40, 0.5, 2.0, orc.rng_philox(8, 1)))
    w2 = tg.random_walk(P, I, torch.from_numpy(small).cuda(), 20, 0.5, 2.0)          # reuses it
    assert tg.graph_cache_info()["edge_set_hits"] == 1
    assert np.array_equal(_np(w2), orc.random_walk(ptrs, idx, small, 20, 0.5, 2.0, orc.rng_philox(8, 2)))
    tg.random_walk(P, I, torch.from_numpy(big).cuda(), 40, 1.0, 1.0)                 # p = q = 1 never asks has_edge
    assert tg.graph_cache_info()["edge_set_hits"] == 1
    I[ptrs[5]:ptrs[6]] = torch.sort(torch.from_numpy(rs.integers(0, n, int(ptrs[6] - ptrs[5]))).cuda()).values  # row 5 rewritten in place
    idx2 = _np(I)
    w3 = tg.random_walk(P, I, torch.from_numpy(big).cuda(), 40, 0.5, 2.0)
    assert tg.graph_cache_info()["edge_set_builds"] == 2
    assert np.array_equal(_np(w3), orc.random_walk(ptrs, idx2, big, 40, 0.5, 2.0, orc.rng_philox(8, 4)))
    tg.graph_cache_clear()
    assert tg.graph_cache_info()["edge_sets"] == 0


def test_unsorted_rows_keep_the_binary_search_whatever_the_call_size(tg):
    """has_edge is the reference's binary search of the row (graph.rs:80-83): on rows that do not ascend it can miss an
    edge the hash set would find, so such a graph never gets a set -- small and large calls, first or later, walk alike
    (and like the oracle, which searches the same way); a call on another stream waits for a set built elsewhere"""
    rs = np.random.default_rng(22)
    n = 1 << 12
    ei = np.stack([rs.integers(0, n, n * 12), rs.integers(0, n, n * 12)])
    ptrs, idx, _ = orc.to_csr(ei, n)
    mixed = idx.copy()
    for v in range(n):                                                      # every row in a random order
        mixed[ptrs[v]:ptrs[v + 1]] = rs.permutation(mixed[ptrs[v]:ptrs[v + 1]])
    assert not np.array_equal(mixed, idx)
    P, I = torch.from_numpy(ptrs).cuda(), torch.from_numpy(mixed).cuda()
    tg.graph_cache_clear()
    builds0 = tg.graph_cache_info()["edge_set_builds"]                      # the counters run on across clears
    small, big = rs.integers(0, n, 100), rs.integers(0, n, 1 << 15)
    tg.seed(9)
    w0 = tg.random_walk(P, I, torch.from_numpy(small).cuda(), 20, 0.5, 2.0)
    w1 = tg.random_walk(P, I, torch.from_numpy(big).cuda(), 40, 0.5, 2.0)           # 1.3 M steps: would build a set
    info = tg.graph_cache_info()
    assert info["edge_set_builds"] == builds0 and info["edge_set_bytes"] == 0
    w2 = tg.random_walk(P, I, torch.from_numpy(small).cuda(), 20, 0.5, 2.0)
    assert tg.graph_cache_info()["edge_set_builds"] == builds0
    for w, s, L, c in ((w0, small, 20, 0), (w1, big, 40, 1), (w2, small, 20, 2)):
        assert np.array_equal(_np(w), orc.random_walk(ptrs, mixed, s, L, 0.5, 2.0, orc.rng_philox(9, c)))
    # sorted rows: the set built on the default stream serves a call on another stream (event behind the build)
    Is = torch.from_numpy(idx).cuda()
    tg.seed(9)
    tg.random_walk(P, Is, torch.from_numpy(big).cuda(), 40, 0.5, 2.0)
    assert tg.graph_cache_info()["edge_set_builds"] == builds0 + 1
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        w = tg.random_walk(P, Is, torch.from_numpy(small).cuda(), 20, 0.5, 2.0)
    side.synchronize()
    assert np.array_equal(_np(w), orc.random_walk(ptrs, idx, small, 20, 0.5, 2.0, orc.rng_philox(9, 1)))
    tg.graph_cache_clear()


def test_out_of_range_node_ids_raise_instead_of_faulting(tg):
    """the reference panics (index out of bounds) on a node id outside the graph; here: IndexError, no device fault"""
    P, I = torch.tensor([0, 1, 2]).cuda(), torch.tensor([1, 0]).cuda()       # 2 nodes
    bad = torch.tensor([0, 2]).cuda()
    with pytest.raises(IndexError, match="outside the graph"):
        tg.neighbor_sampling_homogenous(P, I, bad, [2])
    with pytest.raises(IndexError):
        tg.neighbor_sampling_homogenous(P, I, torch.tensor([-1]).cuda(), [2])
    with pytest.raises(IndexError):
        tg.random_walk(P, I, bad, 3, 1.0, 1.0)
    with pytest.raises(IndexError):
        tg.tempo_random_walk(P, I, torch.zeros(2, dtype=torch.int64).cuda(), torch.zeros(2, dtype=torch.int64).cuda(),
                             bad, torch.zeros(2, dtype=torch.int64).cuda(), 3, (0, 1))
    with pytest.raises(IndexError):
        tg.negative_sample_neighbors_homogenous(P, I, (2, 2), bad, 2, 2)
    et = ("a", "to", "a")
    k = "a__to__a"
    with pytest.raises(IndexError):
        tg.neighbor_sampling_heterogenous(["a"], [et], {k: P}, {k: I}, {"a": bad}, {k: [2]}, 1)
    # the filtered / weighted forms validate on the device and raise with the call's final read-back
    ts, w = torch.tensor([5, 5]).cuda(), torch.tensor([1.0, 2.0], dtype=torch.float64).cuda()
    flt = lambda st: (tg.TemporalEdgeFilter((0, 9), ts, False, tg.TEMPORAL_SAMPLE_STATIC), st)
    with pytest.raises(IndexError, match="outside the graph"):
        tg.neighbor_sampling_homogenous(P, I, bad, [2, 2], None, flt(torch.zeros(2, dtype=torch.int64).cuda()))
    with pytest.raises(IndexError):
        tg.neighbor_sampling_homogenous(P, I, bad, [2], tg.WeightedEdgeSampler(w))
    with pytest.raises(IndexError, match="outside the graph"):
        tg.neighbor_sampling_heterogenous(["a"], [et], {k: P}, {k: I}, {"a": bad}, {k: [2]}, 1, None,
                                          (tg.TemporalEdgeFilter((0, 9), {k: ts}, False, tg.TEMPORAL_SAMPLE_STATIC),
                                           {"a": torch.zeros(2, dtype=torch.int64).cuda()}))
    with pytest.raises(IndexError):
        tg.neighbor_sampling_heterogenous(["a"], [et], {k: P}, {k: I}, {"a": bad}, {k: [2]}, 1, tg.WeightedEdgeSampler({k: w}))
    with pytest.raises(IndexError):
        tg.hgt_sampling(["a"], [et], {k: P}, {k: I}, None, {"a": bad}, None, {"a": [2]}, 1)
    with pytest.raises(IndexError):
        tg.budget_sampling(["a"], [et], {k: P}, {k: I}, None, {"a": bad}, None, {"a": [2]}, 1, None, False, False)
    # and valid calls still work afterwards
    assert tg.neighbor_sampling_homogenous(P, I, torch.tensor([0, 1]).cuda(), [2])[0].numel() == 4


def test_large_single_call_takes_the_whole_device_path(tg):
    """more than 2048 seeds in ONE call: hop by hop over the whole device (tg_ns_hop), same results as the oracle"""
    ptrs, idx, n = None, None, 1 << 13
    rs = np.random.default_rng(12)
    ei = np.stack([rs.integers(0, n, n * 8), rs.integers(0, n, n * 8)])
    ptrs, idx, _ = orc.to_csc(ei, n)
    seeds = rs.integers(0, n, 20000)
    P, I, S = (torch.from_numpy(a).cuda() for a in (ptrs, idx, seeds))
    for sampler, kw in ((None, {}), (tg.UniformEdgeSampler(True), dict(sampler=orc.SAMPLER_UNIFORM_REPL))):
        tg.seed(31)
        s, r, c, e, lo = tg.neighbor_sampling_homogenous(P, I, S, [6, 4], sampler)
        o = orc.ns_homo(ptrs, idx, seeds, [6, 4], orc.rng_philox(31, 0), **kw)
        assert lo == o[4]
        for a, b in zip((s, r, c, e), o[:4]):
            assert np.array_equal(_np(a), b)


def test_more_hops_than_the_one_launch_kernel_takes(tg):
    """the reference takes any number of hops (neighbor_sampling.rs:188); beyond TG_MAX_HOPS = 8 the operators go hop by hop"""
    rs = np.random.default_rng(15)
    n = 4000
    ei = np.stack([rs.integers(0, n, n * 6), rs.integers(0, n, n * 6)])
    ptrs, idx, _ = orc.to_csc(ei, n)
    seeds = rs.integers(0, n, 7)
    fan = [2, 1, 2, 1, 1, 2, 1, 1, 2, 1, 1]                                  # 11 hops
    P, I, S = (torch.from_numpy(a).cuda() for a in (ptrs, idx, seeds))
    tg.seed(43)
    s, r, c, e_, lo = tg.neighbor_sampling_homogenous(P, I, S, fan)
    o = orc.ns_homo(ptrs, idx, seeds, fan, orc.rng_philox(43, 0))
    assert lo == o[4] and len(lo) == 11
    for a, b in zip((s, r, c, e_), o[:4]):
        assert np.array_equal(_np(a), b)
    et, k = ("a", "to", "a"), "a__to__a"
    tg.seed(44)
    hs, hr, hc, he, hlo = tg.neighbor_sampling_heterogenous(["a"], [et], {k: P}, {k: I}, {"a": S}, {k: fan[:10]}, 10)
    ho = orc.ns_hetero(["a"], [et], {k: ptrs}, {k: idx}, {"a": seeds}, {k: fan[:10]}, 10, orc.rng_philox(44, 0))
    assert np.array_equal(_np(hs["a"]), ho[0]["a"]) and np.array_equal(_np(he[k]), ho[3][k])


@pytest.mark.parametrize("fan", [[200], [300, 2], [1000], [4096]])
def test_fanouts_above_255_take_the_wavefront_per_vertex_path(tg, fan):
    rs = np.random.default_rng(14)
    n, e = 3000, 400000                                   # mean degree 133, hubs above 5000
    ei = np.stack([rs.integers(0, n, e), rs.integers(0, n, e)])
    ei[1, rs.integers(0, e, e // 10)] = 7
    ei[1, rs.integers(0, e, e // 20)] = 8
    ptrs, idx, _ = orc.to_csc(ei, n)
    seeds = np.concatenate([[7, 8, 7], rs.integers(0, n, 20)])
    P, I, S = (torch.from_numpy(a).cuda() for a in (ptrs, idx, seeds))
    for sampler, kw in ((None, {}), (tg.UniformEdgeSampler(True), dict(sampler=orc.SAMPLER_UNIFORM_REPL))):
        if fan[0] > 1000 and sampler is not None:
            continue
        tg.seed(41)
        s, r, c, e_, lo = tg.neighbor_sampling_homogenous(P, I, S, fan, sampler)
        o = orc.ns_homo(ptrs, idx, seeds, fan, orc.rng_philox(41, 0), **kw)
        assert lo == o[4]
        for a, b in zip((s, r, c, e_), o[:4]):
            assert np.array_equal(_np(a), b), fan
    with pytest.raises(ValueError):
        tg.neighbor_sampling_homogenous(P, I, S, [5000])


@pytest.mark.parametrize("k", [65, 200, 1024])
def test_filtered_and_weighted_fanouts_above_64(tg, k):
    rs = np.random.default_rng(15)
    n, e = 2000, 300000
    ei = np.stack([rs.integers(0, n, e), rs.integers(0, n, e)])
    ei[1, rs.integers(0, e, e // 8)] = 3                      # a hub column of ~37 K edges
    ptrs, idx, _ = orc.to_csc(ei, n)
    seeds = np.concatenate([[3, 3], rs.integers(0, n, 12)])
    ts, w = rs.integers(0, 10, len(idx)), rs.uniform(0.1, 3.0, len(idx))
    st = rs.integers(0, 10, len(seeds))
    P, I, S = (torch.from_numpy(a).cuda() for a in (ptrs, idx, seeds))
    flt = (tg.TemporalEdgeFilter((0, 6), torch.from_numpy(ts).cuda(), True, tg.TEMPORAL_SAMPLE_DYNAMIC), torch.from_numpy(st).cuda())
    fkw = dict(filter_mode=orc.FILTER_DYNAMIC, forward=True, window=(0, 6), timestamps=ts, inputs_state=st)
    cases = [(None, flt, fkw), (tg.UniformEdgeSampler(True), flt, dict(sampler=orc.SAMPLER_UNIFORM_REPL, **fkw))]
    if k <= 200:
        cases.append((tg.WeightedEdgeSampler(torch.from_numpy(w).cuda()), None, dict(sampler=orc.SAMPLER_WEIGHTED, weights=w)))
    for sampler, f, kw in cases:
        tg.seed(51)
        s, r, c, e_, lo = tg.neighbor_sampling_homogenous(P, I, S, [k], sampler, f)
        o = orc.ns_homo(ptrs, idx, seeds, [k], orc.rng_philox(51, 0), **kw)
        assert lo == o[4]
        for a, b in zip((s, r, c, e_), o[:4]):
            assert np.array_equal(_np(a), b), (k, kw.get("sampler"))
    with pytest.raises(ValueError):
        tg.neighbor_sampling_homogenous(P, I, S, [1025], None, flt)


def test_concurrent_callers_on_their_own_streams(tg):
    """worker threads (each with its own HIP stream) call the surface at once: every result is a valid sample of some
    call id of the shared counter, and every call id is used exactly once"""
    import threading
    ei, n = load_karate()
    ptrs, idx, _ = orc.to_csc(ei, n)
    P, I = torch.from_numpy(ptrs).cuda(), torch.from_numpy(idx).cuda()
    seeds = np.array([0, 1, 4, 5])
    S = torch.from_numpy(seeds).cuda()
    tg.seed(123)
    n_threads, per = 4, 15
    results, errors = [[] for _ in range(n_threads)], []

    def work(t):
        try:
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                for _ in range(per):
                    o = tg.neighbor_sampling_homogenous(P, I, S, [4, 3])
                    st.synchronize()
                    results[t].append(tuple(x.cpu().numpy() for x in o[:4]) + (o[4],))
        except Exception as e:      # noqa: BLE001
            errors.append(e)

    ths = [threading.Thread(target=work, args=(t,)) for t in range(n_threads)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    assert not errors, errors
    expected = {}
    for c in range(n_threads * per):
        o = orc.ns_homo(ptrs, idx, seeds, [4, 3], orc.rng_philox(123, c))
        expected[c] = o
    used = set()
    for t in range(n_threads):
        for s, r, c_, e, lo in results[t]:
            match = [c for c, o in expected.items() if c not in used and np.array_equal(o[0], s) and np.array_equal(o[3], e)
                     and np.array_equal(o[1], r) and np.array_equal(o[2], c_) and o[4] == lo]
            assert match, "a concurrent call returned something no call id of the counter produces"
            used.add(match[0])
    assert len(used) == n_threads * per and tg.rng_state()[1] == n_threads * per


def test_cpu_resident_graph_is_uploaded_once(tg):
    """The reference borrows CPU tensors (utils/tensor.rs:50-59); here a CPU-resident adjacency is uploaded on first use
    and kept on the device, keyed on the tensor's identity and content version: later calls reuse the copy, an in-place
    write or another tensor uploads again.  Results equal the call with device tensors."""
    DEV = "cuda:0"
    ei = torch.from_numpy(load_karate()[0])
    ptrs_d, idx_d, _ = tg.to_csc(ei.to(DEV), 34)
    ptrs, idx = ptrs_d.cpu(), idx_d.cpu()
    inputs = torch.tensor([0, 1, 4, 5])
    tg.graph_cache_clear()
    base = tg.graph_cache_info()
    outs = []
    for _ in range(3):
        tg.seed(11)
        outs.append(tg.neighbor_sampling_homogenous(ptrs, idx, inputs, [5, 5]))
    info = tg.graph_cache_info()
    assert info["uploads"] - base["uploads"] == 2 and info["hits"] - base["hits"] == 4 and info["entries"] == 2
    tg.seed(11)
    ref = tg.neighbor_sampling_homogenous(ptrs_d, idx_d, inputs.to(DEV), [5, 5])
    for o in outs:
        assert all(torch.equal(a.cpu(), b.cpu()) for a, b in zip(o[:4], ref[:4])) and o[4] == ref[4]
        assert not o[0].is_cuda                                        # results come back where the inputs live
    idx[0] = idx[0]                                                    # an in-place write: the content version moves on
    tg.seed(11)
    tg.neighbor_sampling_homogenous(ptrs, idx, inputs, [5, 5])
    assert tg.graph_cache_info()["uploads"] - base["uploads"] == 3
    idx2 = idx.clone()                                                 # another tensor with the same content
    tg.neighbor_sampling_homogenous(ptrs, idx2, inputs, [5, 5])
    assert tg.graph_cache_info()["uploads"] - base["uploads"] == 4
    del idx2
    tg.graph_cache_clear()
    assert tg.graph_cache_info()["entries"] == 0


def test_inference_mode_strided_views_and_aliased_writes(tg):
    """(a) tensors made under torch.inference_mode() have no version counter -- the identity memos must not ask
    for one (an eval loop calls the samplers there, with CPU adjacency as the reference does); (b) two strided views of one
    storage share address, length and version -- only contiguous tensors are memoised; (c) a write that does not bump the
    version (numpy alias) is caught by the content fingerprint of the resident copy."""
    DEV = "cuda:0"
    ei = torch.from_numpy(load_karate()[0])
    ptrs_d, idx_d, _ = tg.to_csc(ei.to(DEV), 34)
    rptrs_d, ridx_d, _ = tg.to_csr(ei.to(DEV), 34)
    inputs = torch.tensor([0, 1, 4, 5])
    tg.seed(3)
    ref = tg.neighbor_sampling_homogenous(ptrs_d, idx_d, inputs.to(DEV), [5, 5])
    tg.seed(3)
    ref_w = tg.random_walk(rptrs_d, ridx_d, inputs.to(DEV), 6, 0.5, 2.0)
    tg.graph_cache_clear()
    with torch.inference_mode():
        ptrs, idx = ptrs_d.cpu().clone(), idx_d.cpu().clone()           # inference tensors, on the CPU
        rptrs, ridx = rptrs_d.cpu().clone(), ridx_d.cpu().clone()
        assert ptrs.is_inference()
        for _ in range(2):
            tg.seed(3)
            o = tg.neighbor_sampling_homogenous(ptrs, idx, inputs.clone(), [5, 5])
            assert all(torch.equal(a, b.cpu()) for a, b in zip(o[:4], ref[:4])) and o[4] == ref[4]
            tg.seed(3)
            w = tg.random_walk(rptrs, ridx, inputs.clone(), 6, 0.5, 2.0)
            assert torch.equal(w, ref_w.cpu())
        tg.seed(3)                                                        # ... and device-resident inference tensors
        o = tg.neighbor_sampling_homogenous(ptrs_d.clone(), idx_d.clone(), inputs.to(DEV), [5, 5])
        assert all(torch.equal(a, b) for a, b in zip(o[:4], ref[:4]))
    # (b) idx_a and idx_b: same storage, address, length, version -- different content
    both = torch.stack([idx_d.cpu(), torch.zeros_like(idx_d.cpu())], dim=1).reshape(-1)   # idx interleaved with zeros
    idx_a, idx_b = both[0::2], both[1::2][: idx_d.numel()]
    base_p = ptrs_d.cpu()
    tg.seed(3)
    oa = tg.neighbor_sampling_homogenous(base_p, idx_a, inputs, [5, 5])
    assert all(torch.equal(a, b.cpu()) for a, b in zip(oa[:4], ref[:4]))
    view0 = both[0::2]
    shifted = torch.stack([torch.zeros_like(idx_d.cpu()), idx_d.cpu()], dim=1).reshape(-1)
    tg.seed(3)
    ob = tg.neighbor_sampling_homogenous(base_p, shifted[1::2], inputs, [5, 5])          # another strided view: not a stale hit
    assert all(torch.equal(a, b.cpu()) for a, b in zip(ob[:4], ref[:4]))
    del idx_b, view0
    # (c) numpy alias: the write below does not move the version counter
    arr = idx_d.cpu().numpy().copy()
    t = torch.from_numpy(arr)
    tg.seed(3)
    o1 = tg.neighbor_sampling_homogenous(base_p, t, inputs, [5, 5])
    assert torch.equal(o1[0], ref[0].cpu())
    v = t._version
    arr[:] = np.roll(arr, 1)                                              # behind torch's back
    assert t._version == v
    tg.seed(3)
    o2 = tg.neighbor_sampling_homogenous(base_p, t, inputs, [5, 5])
    tg.seed(3)
    want = tg.neighbor_sampling_homogenous(ptrs_d, torch.from_numpy(arr.copy()).to(DEV), inputs.to(DEV), [5, 5])
    assert torch.equal(o2[0], want[0].cpu()) and torch.equal(o2[3], want[3].cpu())
    tg.graph_cache_clear()


@pytest.mark.parametrize("deg", [20_000, 100_000, 700_000])
@pytest.mark.parametrize("filtered", [False, True])
def test_weighted_sampling_of_long_columns(tg, deg, filtered):
    """the weighted sampler's three forms by column length (csrc/ns_hop_scan.hip): one wavefront (<= 16 K edges), a
    workgroup (chunk totals, carries, atomic-max slots), and -- beyond the workgroup's scratch of 8 192 chunks = 512 K
    edges -- one wavefront walking the column inside that kernel; all equal the oracle's blocked running sum bit for bit"""
    rs = np.random.default_rng(deg)
    n = 2000
    # vertex 7 owns a column of `deg` in-edges (multi-edges allowed, storage.rs keeps them); the others a few each
    rows = np.concatenate([rs.integers(0, n, deg), rs.integers(0, n, 6 * n)])
    cols = np.concatenate([np.full(deg, 7), rs.integers(0, n, 6 * n)])
    ptrs, idx, _ = orc.to_csc(np.stack([rows, cols]).astype(np.int64), n)
    w = rs.uniform(0.05, 3.0, len(idx))
    ts = rs.integers(0, 100, len(idx))
    P, I = torch.from_numpy(ptrs).cuda(), torch.from_numpy(idx).cuda()
    inputs = np.array([7, 3, 7, 11, 500], dtype=np.int64)
    st = np.array([50, 10, 90, 0, 20])
    k = [12, 3]
    tg.seed(21)
    call = tg.rng_state()[1]
    sampler = tg.WeightedEdgeSampler(torch.from_numpy(w).cuda())
    flt = None
    kw = {}
    if filtered:
        flt = (tg.TemporalEdgeFilter((-30, 30), torch.from_numpy(ts).cuda(), True, tg.TEMPORAL_SAMPLE_RELATIVE),
               torch.from_numpy(st).cuda())
        kw = dict(filter_mode=tg.TEMPORAL_SAMPLE_RELATIVE, forward=True, window=(-30, 30), timestamps=ts, inputs_state=st)
    out = tg.neighbor_sampling_homogenous(P, I, torch.from_numpy(inputs).cuda(), k, sampler, flt)
    o = orc.ns_homo(ptrs, idx, inputs, k, orc.rng_philox(21, call), sampler=orc.SAMPLER_WEIGHTED, weights=w, **kw)
    assert out[4] == o[4] and int(out[1].numel()) > 20
    for a, b in zip(out[:4], o[:4]):
        assert np.array_equal(_np(a), b)
