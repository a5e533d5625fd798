"""NeighborLoader: every mini-batch equals the oracle's neighbor_sampling_homogenous for its (seed, call id), whatever
the prefetch depth, and carries the right attribute rows."""
import numpy as np
import pytest
import torch

import orc
from helpers import load_karate

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _graph(n=500, e=6000, seed=1):
    from tch_geometric.transforms import Graph
    rs = np.random.default_rng(seed)
    ei = np.stack([rs.integers(0, n, e), rs.integers(0, n, e)]).astype(np.int64)
    x = rs.standard_normal((n, 12)).astype(np.float32)
    y = rs.integers(0, 7, n)
    ea = rs.standard_normal((e, 2)).astype(np.float32)
    g = Graph(edge_index=torch.from_numpy(ei).to(DEV), num_nodes=n, x=torch.from_numpy(x).to(DEV),
              y=torch.from_numpy(y).to(DEV), edge_attr=torch.from_numpy(ea).to(DEV))
    return g, ei, x, y, ea


@pytest.mark.parametrize("prefetch", [1, 3, 64])
@pytest.mark.parametrize("replace", [False, True])
def test_batches_equal_oracle_and_carry_attributes(prefetch, replace):
    from tch_geometric.loader import NeighborLoader
    g, ei, x, y, ea = _graph()
    ptrs, idx, perm = orc.to_csc(ei, 500)
    loader = NeighborLoader(g, [5, 4], batch_size=64, prefetch=prefetch, replace=replace, seed=9, call_id0=100)
    assert len(loader) == 8                                   # 500 seeds: 7 full batches + a ragged one
    seen = 0
    for j, b in enumerate(loader):
        seeds = np.arange(j * 64, min((j + 1) * 64, 500))
        o = orc.ns_homo(ptrs, idx, seeds, [5, 4], orc.rng_philox(9, 100 + j), sampler=1 if replace else 0)
        s = b.n_id.cpu().numpy()
        assert b.call_id == 100 + j and b.batch_size == len(seeds)
        assert np.array_equal(s, o[0]) and np.array_equal(b.edge_index.cpu().numpy(), np.stack([o[1], o[2]]))
        assert np.array_equal(b.e_id.cpu().numpy(), perm[o[3]]) and b.layer_offsets == o[4]
        assert np.array_equal(b.x.cpu().numpy(), x[s]) and np.array_equal(b.y.cpu().numpy(), y[s])
        assert np.array_equal(b.edge_attr.cpu().numpy(), ea[perm[o[3]]])
        seen += 1
    assert seen == 8
    assert sum(1 for _ in NeighborLoader(g, [5, 4], batch_size=64, prefetch=prefetch, drop_last=True)) == 7


def test_input_nodes_and_shuffle():
    from tch_geometric.loader import NeighborLoader
    g, ei, x, y, ea = _graph(seed=2)
    nodes = torch.tensor([5, 7, 7, 400, 3, 9, 11, 2, 450, 1])
    plain = [b.n_id[:b.batch_size].cpu() for b in NeighborLoader(g, [3], input_nodes=nodes, batch_size=4, prefetch=2)]
    assert torch.equal(torch.cat(plain), nodes)
    sh = NeighborLoader(g, [3], input_nodes=nodes, batch_size=4, prefetch=2, shuffle=True, seed=3)
    e1 = torch.cat([b.n_id[:b.batch_size].cpu() for b in sh])
    e2 = torch.cat([b.n_id[:b.batch_size].cpu() for b in sh])
    assert sorted(e1.tolist()) == sorted(nodes.tolist()) == sorted(e2.tolist())
    assert not torch.equal(e1, e2) or not torch.equal(e1, nodes)   # epochs reshuffle


def test_hetero_loader_equals_oracle_and_carries_attributes():
    from tch_geometric.loader import HeteroNeighborLoader
    from tch_geometric.transforms import HeteroGraph
    from helpers import load_fake_hetero, rel_key
    counts, edges = load_fake_hetero()
    node_types, edge_types = sorted(counts), sorted(edges)
    rs = np.random.default_rng(4)
    data, feats, ets = HeteroGraph(), {}, {}
    for nt in node_types:
        feats[nt] = rs.standard_normal((counts[nt], 6)).astype(np.float32)
        data[nt].x, data[nt].num_nodes = torch.from_numpy(feats[nt]).to(DEV), counts[nt]
    for et in edge_types:
        data[et].edge_index = torch.from_numpy(edges[et]).to(DEV)
        ets[et] = rs.integers(0, 100, edges[et].shape[1])
        data[et].timestamps = torch.from_numpy(ets[et]).to(DEV)
    P, I, PERM = {}, {}, {}
    for et in edge_types:
        P[rel_key(et)], I[rel_key(et)], PERM[rel_key(et)] = orc.to_csc(edges[et], (counts[et[0]], counts[et[2]]))
    nt0 = node_types[0]
    nodes = torch.from_numpy(rs.integers(0, counts[nt0], 150))
    loader = HeteroNeighborLoader(data, [4, 3], nt0, input_nodes=nodes, batch_size=32, prefetch=3, seed=5, call_id0=40)
    assert len(loader) == 5
    nn = {rel_key(et): [4, 3] for et in edge_types}
    n_seen = 0
    for j, b in enumerate(loader):
        seeds = nodes[j * 32:(j + 1) * 32].numpy()
        o = orc.ns_hetero(node_types, edge_types, P, I, {nt0: seeds}, nn, 2, orc.rng_philox(5, 40 + j))
        for nt in node_types:
            s = b[nt].n_id.cpu().numpy()
            assert np.array_equal(s, o[0][nt]) and np.array_equal(b[nt].x.cpu().numpy(), feats[nt][s])
        for et in edge_types:
            k = rel_key(et)
            assert np.array_equal(b[et].edge_index.cpu().numpy(), np.stack([o[1][k], o[2][k]]))
            assert np.array_equal(b[et].e_id.cpu().numpy(), PERM[k][o[3][k]])
            assert np.array_equal(b[et].timestamps.cpu().numpy(), ets[et][PERM[k][o[3][k]]])
            assert b[et].layer_offsets == o[4][k]
        assert b[nt0].batch_size == len(seeds) and b.call_id == 40 + j
        n_seen += 1
    assert n_seen == 5


def test_epochs_draw_afresh_and_reproducibly():
    """mini-batch j of epoch e draws with call id call_id0 + e*len(loader) + j: a second epoch over the same seeds
    samples other neighbours (the reference's global stream advances on every call, utils/random.rs:19-22), and a
    fresh loader replays epoch by epoch"""
    from tch_geometric.loader import NeighborLoader
    g, ei, x, y, ea = _graph(seed=3)
    ptrs, idx, perm = orc.to_csc(ei, 500)
    mk = lambda: NeighborLoader(g, [5, 4], batch_size=64, prefetch=3, seed=9, call_id0=100)
    a = mk()
    e0 = [(b.call_id, b.n_id.cpu()) for b in a]
    e1 = [(b.call_id, b.n_id.cpu()) for b in a]
    assert [c for c, _ in e0] == list(range(100, 108)) and [c for c, _ in e1] == list(range(108, 116))
    assert any(not torch.equal(u, v) for (_, u), (_, v) in zip(e0, e1))
    for j, (c, s) in enumerate(e1):                                          # epoch 1 equals the oracle at its call ids
        seeds = np.arange(j * 64, min((j + 1) * 64, 500))
        o = orc.ns_homo(ptrs, idx, seeds, [5, 4], orc.rng_philox(9, c))
        assert np.array_equal(s.numpy(), o[0])
    b2 = mk()
    r0 = [b.n_id.cpu() for b in b2]
    r1 = [b.n_id.cpu() for b in b2]
    assert all(torch.equal(u, v) for (_, u), v in zip(e0, r0)) and all(torch.equal(u, v) for (_, u), v in zip(e1, r1))


def test_out_of_range_input_nodes_raise():
    from tch_geometric.loader import NeighborLoader
    g, *_ = _graph()
    for bad in ([0, 500], [-1, 3]):
        with pytest.raises(IndexError):
            NeighborLoader(g, [3], input_nodes=torch.tensor(bad), batch_size=2)


def test_two_live_iterators_and_an_abandoned_epoch():
    """two iterators over ONE loader (zip(loader, loader); a restart after `break`) own their slab sets: neither's
    prefetched launch lands in the slabs the other's pending mini-batches are cut from; the ragged last mini-batch has
    its own small slab; a mini-batch takes user attributes"""
    from tch_geometric.loader import NeighborLoader
    g, ei, x, y, ea = _graph(seed=4)
    ptrs, idx, perm = orc.to_csc(ei, 500)
    loader = NeighborLoader(g, [5, 4], batch_size=64, prefetch=2, seed=9, call_id0=0)

    def check(b, epoch, j):
        seeds = np.arange(j * 64, min((j + 1) * 64, 500))
        o = orc.ns_homo(ptrs, idx, seeds, [5, 4], orc.rng_philox(9, epoch * 8 + j))
        assert b.call_id == epoch * 8 + j
        assert np.array_equal(b.n_id.cpu().numpy(), o[0]) and np.array_equal(b.e_id.cpu().numpy(), perm[o[3]])
        assert np.array_equal(b.x.cpu().numpy(), x[o[0]])

    n = 0
    for j, (u, v) in enumerate(zip(loader, loader)):           # epochs 0 and 1, interleaved
        check(u, 0, j)
        check(v, 1, j)
        u.note = "mine"                                        # a consumer's own attribute
        assert u.note == "mine" and not hasattr(v, "note")
        n += 1
    assert n == 8 and len(loader._pool) == 2
    it = iter(loader)                                          # epoch 2, abandoned with its next launch in flight
    first = next(it)
    check(first, 2, 0)
    it.close()
    for j, b in enumerate(loader):                             # epoch 3 reuses pooled slabs, ragged one included
        check(b, 3, j)
    big = {id(s["out"]) for slabs in loader._pool for k, s in slabs.items() if k != "ragged"}
    rag = [s["out"] for slabs in loader._pool for k, s in slabs.items() if k == "ragged"]
    assert len(loader._pool) == 2 and len(big) <= 4 and all(r.n_batches == 1 and r.n_seeds == 500 - 7 * 64 for r in rag)
