"""GPU parity: tg_random_walk / tg_tempo_random_walk (HIP, C ABI) == CPU oracle philox-mode, bit for bit."""
import numpy as np
import pytest
import torch

import orc
from helpers import has_edge, load_karate

pytestmark = pytest.mark.gpu
SEED = 0xA11CE


@pytest.fixture(scope="module")
def cabi():
    from tch_geometric import _cabi
    return _cabi


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


def _csr_rmat(scale, seed):
    n = 1 << scale
    row, col = orc.rmat_edges(scale, n * 16, seed)
    ptrs, idx, _ = orc.to_csr(np.stack([row, col]), n)
    return ptrs, idx, n


def _to(dev, *arrs):
    return [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in arrs]


@pytest.mark.parametrize("pq", [(1.0, 1.0), (1.0, 1.5), (0.25, 4.0), (4.0, 0.5)])
@pytest.mark.parametrize("walk_length", [1, 10, 16, 33, 80])
def test_random_walk_karate(cabi, dev, pq, walk_length):
    ei, n = load_karate()
    ptrs, idx, _ = orc.to_csr(ei, n)
    start = np.arange(n, dtype=np.int64).repeat(3)
    p_d, i_d, s_d = _to(dev, ptrs, idx, start)
    w = cabi.random_walk(cabi.graph_view(p_d, i_d), s_d, walk_length, pq[0], pq[1], SEED, 7).cpu().numpy()
    ref = orc.random_walk(ptrs, idx, start, walk_length, pq[0], pq[1], orc.rng_philox(SEED, 7))
    assert np.array_equal(w, ref)
    for row in w[:10]:                                   # random_walk.rs:322-330
        for a, b in zip(row[:-1], row[1:]):
            assert has_edge(ptrs, idx, a, b)
    # has_edge answered from the edge set (one hash probe instead of a binary search of the row): the same walks
    g = cabi.graph_view(p_d, i_d)
    es = cabi.edge_set(g, dev)
    assert np.array_equal(cabi.random_walk(g, s_d, walk_length, pq[0], pq[1], SEED, 7, edge_set=es).cpu().numpy(), ref)


@pytest.mark.parametrize("pq", [(1.0, 1.0), (2.0, 0.5)])
def test_random_walk_rmat_with_dead_ends(cabi, dev, pq):
    ptrs, idx, n = _csr_rmat(13, 99)                     # RMAT has many zero-out-degree vertices
    start = orc.seed_batches(0x57A27, 0, 1, 5000, n)[0]
    p_d, i_d, s_d = _to(dev, ptrs, idx, start)
    w = cabi.random_walk(cabi.graph_view(p_d, i_d), s_d, 40, pq[0], pq[1], SEED, 1).cpu().numpy()
    ref = orc.random_walk(ptrs, idx, start, 40, pq[0], pq[1], orc.rng_philox(SEED, 1))
    assert np.array_equal(w, ref)
    assert (w == -1).any() and (w[:, -1] >= 0).any()
    # the optional u32 shadows of the CSR change the bytes read, not the walk
    g32 = cabi.graph_view(p_d, i_d, indices32=i_d.to(torch.int32), ptrs32=p_d.to(torch.int32))
    assert np.array_equal(cabi.random_walk(g32, s_d, 40, pq[0], pq[1], SEED, 1).cpu().numpy(), ref)
    es = cabi.edge_set(g32, dev)                         # and so does the edge set (RMAT: multi-edges, self loops, hubs)
    assert es.numel() >= 2 * idx.size and (es.numel() & (es.numel() - 1)) == 0
    assert np.array_equal(cabi.random_walk(g32, s_d, 40, pq[0], pq[1], SEED, 1, edge_set=es).cpu().numpy(), ref)
    keys = es[es != -1].cpu().numpy().astype(np.uint64)  # the set holds exactly the distinct edges
    rows = np.repeat(np.arange(n), np.diff(ptrs)).astype(np.uint64)
    assert np.array_equal(np.sort(keys), np.unique((rows << np.uint64(32)) | idx.astype(np.uint64)))


def test_edge_set_of_an_empty_graph_and_a_wrong_size(cabi, dev):
    ptrs = torch.zeros(6, dtype=torch.int64, device=dev)
    idx = torch.zeros(0, dtype=torch.int64, device=dev)
    g = cabi.graph_view(ptrs, idx)
    es = cabi.edge_set(g, dev)
    assert (es == -1).all()
    start = torch.arange(5, device=dev)
    w = cabi.random_walk(g, start, 3, 2.0, 0.5, SEED, 0, edge_set=es)
    assert (w[:, 0] == start).all() and (w[:, 1:] == -1).all()
    with pytest.raises(RuntimeError, match="was not built for this graph"):
        cabi.random_walk(g, start, 3, 2.0, 0.5, SEED, 0, edge_set=torch.empty(8, dtype=torch.int64, device=dev))


def test_random_walk_edge_cases(cabi, dev):
    ptrs = np.array([0, 1, 1, 2], dtype=np.int64)       # 0 -> 1, 1 dead end, 2 -> 0
    idx = np.array([1, 0], dtype=np.int64)
    p_d, i_d, s_d = _to(dev, ptrs, idx, np.array([2, 1, 0], dtype=np.int64))
    g = cabi.graph_view(p_d, i_d)
    assert cabi.random_walk(g, s_d, 4, 1.0, 1.0, 1, 1).cpu().tolist() == \
        [[2, 0, 1, -1, -1], [1, -1, -1, -1, -1], [0, 1, -1, -1, -1]]
    assert cabi.random_walk(g, s_d, 0, 1.0, 1.0, 1, 1).cpu().tolist() == [[2], [1], [0]]
    assert cabi.random_walk(g, s_d[:0], 4, 1.0, 1.0, 1, 1).shape == (0, 5)
    with pytest.raises(cabi.TchGeoError):
        cabi.random_walk(g, s_d, 4, 0.0, 1.0, 1, 1)


@pytest.mark.parametrize("walk_length", [2, 10, 25])
def test_tempo_random_walk_karate(cabi, dev, walk_length):
    """random_walk.rs:333-383 config, then compared word for word with the oracle."""
    ei, n = load_karate()
    ptrs, idx, _ = orc.to_csr(ei, n)
    g = np.random.default_rng(7)
    node_ts, edge_ts = g.integers(-1, 4, n), g.integers(-1, 4, len(idx))
    start = np.array([0, 1, 2, 3, 33, 5, 6], dtype=np.int64)
    start_ts = np.array([0, -1, 2, 3, 1, 0, 9], dtype=np.int64)       # 9: nothing admissible -> restarts
    p_d, i_d, nt_d, et_d, s_d, st_d = _to(dev, ptrs, idx, node_ts, edge_ts, start, start_ts)
    w, wt = cabi.tempo_random_walk(cabi.graph_view(p_d, i_d), nt_d, et_d, s_d, st_d, walk_length, (0, 2), SEED, 3)
    rw, rwt = orc.tempo_random_walk(ptrs, idx, node_ts, edge_ts, start, start_ts, walk_length, (0, 2),
                                    orc.rng_philox(SEED, 3))
    assert np.array_equal(w.cpu().numpy(), rw) and np.array_equal(wt.cpu().numpy(), rwt)
    for i in range(len(start)):                          # :375-381 window check
        if start_ts[i] != -1:
            t = rwt[i, 1:]
            assert np.all((t == -1) | ((t >= start_ts[i]) & (t < start_ts[i] + 2)))


def test_tempo_random_walk_rmat(cabi, dev):
    ptrs, idx, n = _csr_rmat(12, 5)
    g = np.random.default_rng(1)
    node_ts, edge_ts = g.integers(-1, 50, n), g.integers(-1, 50, len(idx))
    start = orc.seed_batches(1, 0, 1, 700, n)[0]
    start_ts = g.integers(-1, 40, 700)
    p_d, i_d, nt_d, et_d, s_d, st_d = _to(dev, ptrs, idx, node_ts, edge_ts, start, start_ts)
    w, wt = cabi.tempo_random_walk(cabi.graph_view(p_d, i_d), nt_d, et_d, s_d, st_d, 12, (0, 15), SEED, 4)
    rw, rwt = orc.tempo_random_walk(ptrs, idx, node_ts, edge_ts, start, start_ts, 12, (0, 15),
                                    orc.rng_philox(SEED, 4))
    assert np.array_equal(w.cpu().numpy(), rw) and np.array_equal(wt.cpu().numpy(), rwt)
