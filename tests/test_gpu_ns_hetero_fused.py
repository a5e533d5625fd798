"""tg_ns_hetero_batched (one launch for all hops and relations of many seed batches) == the oracle's
neighbor_sampling_heterogenous, batch by batch, bit for bit."""
import numpy as np
import pytest
import torch

import orc
from helpers import load_fake_hetero, rel_key, validate_neighbor_samples
from test_gpu_random_sweep_hetero import random_hetero

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def cabi():
    from tch_geometric import _cabi
    return _cabi


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.int64))).to(DEV)


def run_fused(cabi, node_types, edge_types, counts, edges, inputs, nn, hops, seed, call, sampler=0):
    """inputs: {type: [n_batches, n] array}; returns per-batch dict results and the CSC used"""
    tix = {t: i for i, t in enumerate(node_types)}
    P, I, rels = {}, {}, []
    for et in edge_types:
        k = rel_key(et)
        P[k], I[k], _ = orc.to_csc(edges[et], (counts[et[0]], counts[et[2]]))
        rels.append((tix[et[0]], tix[et[2]], _t(P[k]), _t(I[k]), nn[k]))
    nb = next(iter(inputs.values())).shape[0]
    ins = [(_t(inputs[t]) if t in inputs and inputs[t].shape[1] else None) for t in node_types]
    h = cabi.NsHeteroBatched(len(node_types), rels, ins, hops, nb, torch.device(DEV), sampler=sampler)
    h.run(seed, call)
    torch.cuda.synchronize()
    return h, P, I


def check_against_oracle(h, node_types, edge_types, P, I, inputs, nn, hops, seed, call, **kw):
    T, counts = len(node_types), h.counts.cpu().numpy()
    lo = h.layer_offsets.cpu().numpy()
    for b in range(h.nb):
        ins = {t: inputs[t][b] for t in inputs if inputs[t].shape[1]}
        o = orc.ns_hetero(node_types, edge_types, P, I, ins, nn, hops, orc.rng_philox(seed, call + b), **kw)
        for t, nt in enumerate(node_types):
            assert np.array_equal(h.samples[t][b, :counts[b, t]].cpu().numpy(), o[0][nt]), (b, nt)
        for r, et in enumerate(edge_types):
            k, m = rel_key(et), counts[b, T + r]
            assert np.array_equal(h.rows[r][b, :m].cpu().numpy(), o[1][k]), (b, k)
            assert np.array_equal(h.cols[r][b, :m].cpu().numpy(), o[2][k]), (b, k)
            assert np.array_equal(h.edge_index[r][b, :m].cpu().numpy(), o[3][k]), (b, k)
            assert [tuple(x) for x in lo[b, r, :hops]] == o[4][k], (b, k)


@pytest.mark.parametrize("sampler", [0, 1])
def test_fake_hetero_reference_config(cabi, sampler):
    counts, edges = load_fake_hetero()
    node_types, edge_types = sorted(counts), sorted(edges)
    rs = np.random.default_rng(2)
    inputs = {t: np.stack([[0, 1, 4, 5]] + [rs.integers(0, counts[t], 4) for _ in range(4)]) for t in node_types}
    nn = {rel_key(e): [4, 3] for e in edge_types}
    h, P, I = run_fused(cabi, node_types, edge_types, counts, edges, inputs, nn, 2, 77, 10, sampler=sampler)
    check_against_oracle(h, node_types, edge_types, P, I, inputs, nn, 2, 77, 10, sampler=sampler)
    c = h.counts.cpu().numpy()
    for r, et in enumerate(edge_types):            # neighbor_sampling.rs:370-401 on batch 0
        k, m = rel_key(et), c[0, len(node_types) + r]
        s_src = h.samples[node_types.index(et[0])][0].cpu().numpy()
        s_dst = h.samples[node_types.index(et[2])][0].cpu().numpy()
        lo = [tuple(x) for x in h.layer_offsets[0, r].cpu().numpy()]
        if sampler == 0:
            validate_neighbor_samples(P[k], I[k], h.rows[r][0, :m].cpu().numpy(), h.cols[r][0, :m].cpu().numpy(), s_src, s_dst,
                                      lo, [4, 3])


@pytest.mark.parametrize("case", range(10))
def test_random_typed_graphs(cabi, case):
    rs = np.random.default_rng(6000 + case)
    node_types, edge_types, counts, edges = random_hetero(rs)
    hops = int(rs.integers(0, 4))
    big = case % 3 == 2
    nn = {rel_key(et): [int(rs.integers(1, 33 if big else 12)) for _ in range(hops)] for et in edge_types}
    if case % 4 == 1 and edge_types:               # a relation that is not sampled at all
        nn[rel_key(edge_types[0])] = [0] * hops
    nb = int(rs.integers(1, 6))
    inputs = {t: rs.integers(0, counts[t], (nb, int(rs.integers(0, 40)))) for t in node_types}
    sampler = int(rs.integers(0, 2))
    h, P, I = run_fused(cabi, node_types, edge_types, counts, edges, inputs, nn, hops, case, 3, sampler=sampler)
    active = [et for et in edge_types if any(nn[rel_key(et)])] if hops else list(edge_types)
    # the oracle takes only sampled relations' fan-outs; a fan-out of 0 everywhere = relation absent from num_neighbors
    T = len(node_types)
    cnt = h.counts.cpu().numpy()
    for b in range(nb):
        ins = {t: inputs[t][b] for t in inputs if inputs[t].shape[1]}
        nn_o = {rel_key(et): (nn[rel_key(et)] if et in active else [1] * hops) for et in edge_types}
        # relations with fan-out 0 keep their index (tag) but contribute nothing: emulate by an empty graph
        P_o = {k: (P[k] if any(nn[k]) or not hops else np.zeros_like(P[k])) for k in P}
        o = orc.ns_hetero(node_types, edge_types, P_o, I, ins, nn_o, hops, orc.rng_philox(case, 3 + b), sampler=sampler)
        for t, nt in enumerate(node_types):
            assert np.array_equal(h.samples[t][b, :cnt[b, t]].cpu().numpy(), o[0][nt]), (case, b, nt)
        for r, et in enumerate(edge_types):
            k, m = rel_key(et), cnt[b, T + r]
            assert np.array_equal(h.rows[r][b, :m].cpu().numpy(), o[1][k]), (case, b, k)
            assert np.array_equal(h.cols[r][b, :m].cpu().numpy(), o[2][k])
            assert np.array_equal(h.edge_index[r][b, :m].cpu().numpy(), o[3][k])


def test_argument_errors(cabi):
    import ctypes as C
    pb = cabi.TgHetProblem(9, 0, 0, 0, None, None, None, None, None, (C.c_int64 * 9)())
    cap = (C.c_int64 * 16)()
    assert cabi.lib.tg_ns_hetero_capacity(C.byref(pb), cap, cap) == 1          # TG_ERR_INVALID: too many node types
    pb = cabi.TgHetProblem(1, 0, 0, 2, None, None, None, None, None, (C.c_int64 * 1)())
    assert cabi.lib.tg_ns_hetero_capacity(C.byref(pb), cap, cap) == 1          # weighted sampler is not fused
