"""Size-independent properties at BASELINE.json's full sizes (RMAT-24, fanout [15,10], batch 1024; cfg3 walks),
where the oracle is too slow to replay everything: forest structure, exact per-vertex counts, distinct edges per
vertex, gather consistency, determinism, launch-geometry independence -- plus an oracle spot check of a few batches."""
import numpy as np
import pytest
import torch

import orc

pytestmark = pytest.mark.gpu
SCALE = 24


@pytest.fixture(scope="module")
def cabi():
    from tch_geometric import _cabi
    return _cabi


@pytest.fixture(scope="module")
def rmat24(cabi):
    dev = torch.device("cuda:0")
    n = 1 << SCALE
    row, col = cabi.rmat_edges(SCALE, n * 16, 0x5EED0000 + SCALE, dev)
    ptrs, idx, _ = cabi.coo_to_csx(row, col, n, n, True)
    cptrs, cidx, _ = cabi.coo_to_csx(row, col, n, n, False)
    del row, col
    return dev, n, ptrs, idx, cptrs, cidx


def test_rmat24_ingest_properties(rmat24):
    dev, n, ptrs, idx, cptrs, cidx = rmat24
    assert int(ptrs[0]) == 0 and int(ptrs[-1]) == n * 16 and bool((ptrs[1:] >= ptrs[:-1]).all())
    # rows ascending inside every column (sort key col*N+row, storage.rs:119): idx non-decreasing except at column starts
    starts = torch.zeros(n * 16, dtype=torch.bool, device=dev)
    starts[ptrs[:-1][ptrs[:-1] < n * 16]] = True
    assert bool(((idx[1:] >= idx[:-1]) | starts[1:]).all())
    assert int(idx.sum()) == int(cptrs.new_tensor(0)) + int(torch.repeat_interleave(
        torch.arange(n, device=dev), cptrs[1:] - cptrs[:-1]).sum())      # same multiset of rows in CSC and CSR


def test_neighbor_sampling_fullsize_properties(cabi, rmat24):
    dev, n, ptrs, idx, _, _ = rmat24
    g = cabi.graph_view(ptrs, idx)
    nb, B, fan = 256, 1024, [15, 10]
    seeds = cabi.seed_batches(0xBA7C4, 0, nb, B, n, dev)
    out = cabi.NsBatchedOut(nb, B, fan, dev)
    cabi.ns_homo_batched(g, seeds, fan, 0, 0, out)
    torch.cuda.synchronize()
    counts, lo = out.counts, out.layer_offsets
    ns, ne = counts[:, 0], counts[:, 1]
    assert bool((ns == B + ne).all())                                    # one new sample per edge: a forest
    assert bool((lo[:, 0, 0] == B).all() and (lo[:, 0, 1] == 0).all() and (lo[:, 1, 0] == B + lo[:, 1, 1]).all())
    deg = ptrs[1:] - ptrs[:-1]
    total = 0
    for b in range(0, nb, 37):
        s, r, c, e, _ = out.batch(b)
        m = int(ne[b])
        total += m
        assert torch.equal(r, torch.arange(B, B + m, device=dev))        # rows[e] = n_seeds + e (:212-217)
        assert torch.equal(s[B:], idx[e])                                # sample = indices[edge_ptr]
        parent = s[c]
        assert bool(((e >= ptrs[parent]) & (e < ptrs[parent + 1])).all())  # the edge belongs to its parent's column
        assert bool((c[1:] >= c[:-1]).all())                             # parents visited in frontier order
        h1 = int(lo[b, 1, 1])
        front = B + h1                                                   # vertices that were expanded
        per = torch.bincount(c, minlength=front)[:front]
        k_of = torch.where(torch.arange(front, device=dev) < B, 15, 10)
        assert torch.equal(per, torch.minimum(deg[s[:front]], k_of))     # exactly min(deg, k) per vertex
        key = c * (16 * n) + e
        assert int(torch.unique(key).numel()) == m                       # without replacement: distinct edges
    assert total > 0
    # determinism + independence of launch geometry: the same batches alone, in a different launch shape
    out2 = cabi.NsBatchedOut(3, B, fan, dev)
    cabi.ns_homo_batched(g, seeds[100:103].contiguous(), fan, 0, 100, out2)
    for j in range(3):
        a, b2 = out.batch(100 + j), out2.batch(j)
        assert all(torch.equal(x, y) for x, y in zip(a[:4], b2[:4])) and a[4] == b2[4]


def test_neighbor_sampling_fullsize_oracle_spot_check(cabi, rmat24):
    """two full-size batches replayed by the CPU oracle (philox-mode is O(k) per vertex, so this is quick)"""
    dev, n, ptrs, idx, _, _ = rmat24
    hp, hi = ptrs.cpu().numpy(), idx.cpu().numpy()
    seeds = cabi.seed_batches(0xBA7C4, 7, 2, 1024, n, dev)
    out = cabi.NsBatchedOut(2, 1024, [15, 10], dev)
    cabi.ns_homo_batched(cabi.graph_view(ptrs, idx), seeds, [15, 10], 0, 7, out)
    hs = seeds.cpu().numpy()
    for b in range(2):
        gs, gr, gc, ge, glo = out.batch(b)
        o = orc.ns_homo(hp, hi, hs[b], [15, 10], orc.rng_philox(0, 7 + b))
        assert glo == o[4]
        for x, y in zip((gs, gr, gc, ge), o[:4]):
            assert np.array_equal(x.cpu().numpy(), y)


def test_random_walk_fullsize_properties(cabi, rmat24):
    """cfg3: 1M walkers, walk_length 80, p = q = 1."""
    dev, n, _, _, cptrs, cidx = rmat24
    g = cabi.graph_view(cptrs, cidx)
    nw, L = 1 << 20, 80
    start = cabi.seed_batches(0x57A27, 0, 1, nw, n, dev)[0].contiguous()
    w = cabi.random_walk(g, start, L, 1.0, 1.0, 0, 0)
    assert w.shape == (nw, L + 1) and torch.equal(w[:, 0], start)
    alive = w >= 0
    assert bool((alive[:, 1:] <= alive[:, :-1]).all())                   # -1 only as a suffix (dead end, :45-47)
    a, b2 = w[:, :-1], w[:, 1:]
    step = alive[:, 1:]
    src, dst = a[step], b2[step]
    # every executed step is an edge: dst lies in src's row (binary search on the sorted row)
    lo_, hi_ = cptrs[src], cptrs[src + 1]
    pos = lo_.clone()
    span = hi_ - lo_
    while int(span.max()) > 0:                                            # vectorised lower_bound
        half = span // 2
        mid = pos + half
        go = (cidx[torch.clamp(mid, max=cidx.numel() - 1)] < dst) & (span > 0)
        pos = torch.where(go, mid + 1, pos)
        span = torch.where(go, span - half - 1, half)
    assert bool(((pos < hi_) & (cidx[torch.clamp(pos, max=cidx.numel() - 1)] == dst)).all())
    dead = ~alive[:, 1:] & alive[:, :-1]                                  # the step where a walk stopped
    assert bool(((cptrs[a[dead] + 1] - cptrs[a[dead]]) == 0).all())      # ... was taken at a vertex without out-edges
    assert torch.equal(w, cabi.random_walk(g, start, L, 1.0, 1.0, 0, 0))  # deterministic under a fixed (seed, call id)
    # oracle spot check of the first 2000 walkers
    ref = orc.random_walk(cptrs.cpu().numpy(), cidx.cpu().numpy(), start[:2000].cpu().numpy(), L, 1.0, 1.0,
                          orc.rng_philox(0, 0))
    assert np.array_equal(w[:2000].cpu().numpy(), ref)
