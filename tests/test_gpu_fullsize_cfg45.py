"""BASELINE.json cfg4 and cfg5 at their full sizes.

cfg4: 3 node types (A 2^23, B 2^22, C 2^22 nodes) / 5 relations x 20 M edges, built exactly as bench.py builds it;
`neighbor_sampling_heterogenous` (reference neighbor_sampling.rs:233-356) over 64 seed batches in one
`tg_ns_hetero_batched` launch and `hgt_sampling` (hgt_sampling.rs:138-278) through the operator surface:
size-independent properties on every checked batch + a replay of two batches / of the whole HGT call by the oracle.

cfg5: RMAT-27 (2.1 G edges, the last offset = 2^31) range-partitioned over an emulated world of 8 on the one GPU:
`tg_part_begin / requests / sample / emit` with bucket p of the requests answered from shard p; must equal the
replicated launch `tg_ns_homo_batched` bit for bit (SURVEY.md 8(e) mode 2), which itself is replayed by the oracle
for one batch."""
import numpy as np
import pytest
import torch

import orc
from helpers import rel_key
from helpers_part import emulated_world_sample

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

NODE_TYPES = ["A", "B", "C"]
SCALES = {"A": 23, "B": 22, "C": 22}
EDGE_TYPES = [("A", "e0", "A"), ("A", "e1", "B"), ("B", "e2", "A"), ("B", "e3", "C"), ("C", "e4", "A")]


@pytest.fixture(scope="module")
def cabi():
    from tch_geometric import _cabi
    return _cabi


@pytest.fixture(scope="module")
def cfg4(cabi):
    dev = torch.device(DEV)
    P, I = {}, {}
    for r, (s, _, d) in enumerate(EDGE_TYPES):
        rw, cl = cabi.rmat_edges_rect(SCALES[s], SCALES[d], 20_000_000, 0xC0F4 + r, dev)
        P[rel_key(EDGE_TYPES[r])], I[rel_key(EDGE_TYPES[r])], _ = cabi.coo_to_csx(rw, cl, 1 << SCALES[s], 1 << SCALES[d], True)
    return dev, P, I


def test_cfg4_hetero_neighbor_sampling(cabi, cfg4):
    dev, P, I = cfg4
    T, nb, B, fan, hops = 3, 64, 1024, [15, 10], 2
    tix = {t: i for i, t in enumerate(NODE_TYPES)}
    rels = [(tix[s], tix[d], P[rel_key((s, r, d))], I[rel_key((s, r, d))], fan) for s, r, d in EDGE_TYPES]
    seeds = cabi.seed_batches(0xBA7C4, 5000, nb, B, 1 << SCALES["A"], dev)
    h = cabi.NsHeteroBatched(T, rels, [seeds, None, None], hops, nb, dev)
    h.run(0, 5000)
    torch.cuda.synchronize()
    counts, lo = h.counts.cpu().numpy(), h.layer_offsets.cpu().numpy()
    assert counts[:, T:].sum() > nb * 10_000
    for b in range(0, nb, 9):
        n_of = [int(counts[b, t]) for t in range(T)]
        smp = [h.samples[t][b, :n_of[t]] for t in range(T)]
        assert torch.equal(smp[0][:B], seeds[b]) and n_of[0] >= B
        appended = [0] * T                                   # every sample beyond the inputs comes from exactly one edge
        n_in = [B, 0, 0]
        hop0 = [n_in[t] + sum(int(lo[b, r, 1, 1]) for r, et in enumerate(EDGE_TYPES) if tix[et[0]] == t) for t in range(T)]
        slices = [[(0, n_in[t]) for t in range(T)], [(n_in[t], hop0[t]) for t in range(T)]]
        for r, (s, _, d) in enumerate(EDGE_TYPES):
            k, m = rel_key(EDGE_TYPES[r]), int(counts[b, T + r])
            rows, cols, eidx = h.rows[r][b, :m], h.cols[r][b, :m], h.edge_index[r][b, :m]
            appended[tix[s]] += m
            src, dst = smp[tix[s]], smp[tix[d]]
            assert torch.equal(src[rows], I[k][eidx])                        # sample = indices[edge pointer] (:323)
            parent = dst[cols]
            assert bool(((eidx >= P[k][parent]) & (eidx < P[k][parent + 1])).all())   # edge lies in its parent's column
            assert bool((cols[1:] >= cols[:-1]).all())                      # frontier visited in order
            assert int(torch.unique(rows).numel()) == m                     # a forest: one new sample per edge
            assert int(torch.unique(cols * (1 << 40) + eidx).numel()) == m  # without replacement
            # exactly min(deg, k) per frontier vertex and hop; the frontier of hop h is the slice of dst's list that
            # existed when the hop STARTED (`slices`, neighbor_sampling.rs:285-288, :349-352)
            deg = P[k][1:] - P[k][:-1]
            e_cut = [int(lo[b, r, hh, 1]) for hh in range(hops)] + [m]
            for hh in range(hops):
                begin, end = slices[hh][tix[d]]
                ce = cols[e_cut[hh]:e_cut[hh + 1]]
                per = torch.bincount(ce - begin, minlength=end - begin) if end > begin else ce[:0]
                assert torch.equal(per, torch.clamp(deg[dst[begin:end]], max=fan[hh])), (b, k, hh)
        assert [n_of[t] - (B if t == 0 else 0) for t in range(T)] == appended
    # oracle replay of two batches
    hP = {k: v.cpu().numpy() for k, v in P.items()}
    hI = {k: v.cpu().numpy() for k, v in I.items()}
    nn = {rel_key(et): fan for et in EDGE_TYPES}
    hs = seeds.cpu().numpy()
    for b in (0, nb - 1):
        o = orc.ns_hetero(NODE_TYPES, EDGE_TYPES, hP, hI, {"A": hs[b]}, nn, hops, orc.rng_philox(0, 5000 + b))
        for t, nt in enumerate(NODE_TYPES):
            assert np.array_equal(h.samples[t][b, :counts[b, t]].cpu().numpy(), o[0][nt]), (b, nt)
        for r, et in enumerate(EDGE_TYPES):
            k, m = rel_key(et), counts[b, T + r]
            assert np.array_equal(h.rows[r][b, :m].cpu().numpy(), o[1][k]), (b, k)
            assert np.array_equal(h.cols[r][b, :m].cpu().numpy(), o[2][k]), (b, k)
            assert np.array_equal(h.edge_index[r][b, :m].cpu().numpy(), o[3][k]), (b, k)
            assert [tuple(x) for x in lo[b, r, :hops]] == o[4][k], (b, k)
    # determinism and independence of launch geometry: three of the batches alone
    h2 = cabi.NsHeteroBatched(T, rels, [seeds[20:23].contiguous(), None, None], hops, 3, dev)
    h2.run(0, 5020)
    torch.cuda.synchronize()
    c2 = h2.counts.cpu().numpy()
    for j in range(3):
        assert np.array_equal(c2[j], counts[20 + j])
        for t in range(T):
            assert torch.equal(h2.samples[t][j, :c2[j, t]], h.samples[t][20 + j, :c2[j, t]])
        for r in range(len(EDGE_TYPES)):
            m = c2[j, T + r]
            assert torch.equal(h2.edge_index[r][j, :m], h.edge_index[r][20 + j, :m])


def test_cfg4_hgt_sampling(cabi, cfg4):
    """hgt_sampling at cfg4: 1 024 inputs of type A, {A,B,C: [512, 512]}, 2 hops, through the operator surface."""
    import tch_geometric as tg
    dev, P, I = cfg4
    seeds = cabi.seed_batches(0xBA7C4, 5000, 1, 1024, 1 << SCALES["A"], dev)[0].contiguous()
    ns = {t: [512, 512] for t in NODE_TYPES}
    tg.seed(4)
    s, ts, r, c, e = tg.hgt_sampling(NODE_TYPES, EDGE_TYPES, P, I, None, {"A": seeds}, None, ns, 2)
    torch.cuda.synchronize()
    assert torch.equal(s["A"][:1024], seeds)                                # inputs first, duplicates kept (:171-180)
    for t in NODE_TYPES:
        new = s[t][1024:] if t == "A" else s[t]
        assert 0 < int(new.numel()) <= 1024                                  # <= 512 per layer and type (:207)
        assert int(torch.unique(new).numel()) == int(new.numel())            # budget samples are distinct nodes (:222-231)
        if t == "A":
            assert not bool(torch.isin(new, seeds).any())                    # ... and never an already sampled node
    n_edges = 0
    for (sn, rn, dn) in EDGE_TYPES:
        k = rel_key((sn, rn, dn))
        rows, cols, eidx = r[k], c[k], e[k]
        n_edges += int(rows.numel())
        if rows.numel() == 0:
            continue
        src, dst = s[sn][rows], s[dn][cols]
        # edge rebuild (:244-268): edge pointer lies in dst's column and holds src
        assert bool(((eidx >= P[k][dst]) & (eidx < P[k][dst + 1])).all())
        assert torch.equal(I[k][eidx], src)
    assert n_edges > 0
    # the whole call replayed by the oracle
    hP = {k: v.cpu().numpy() for k, v in P.items()}
    hI = {k: v.cpu().numpy() for k, v in I.items()}
    o = orc.hgt(NODE_TYPES, EDGE_TYPES, hP, hI, None, {"A": seeds.cpu().numpy()}, None, ns, 2, orc.rng_philox(4, 0))
    for t in NODE_TYPES:
        assert np.array_equal(s[t].cpu().numpy(), o[0][t]), t
        assert np.array_equal(ts[t].cpu().numpy(), o[1][t]), t
    for et in EDGE_TYPES:
        k = rel_key(et)
        assert np.array_equal(r[k].cpu().numpy(), o[2][k]), k
        assert np.array_equal(c[k].cpu().numpy(), o[3][k]), k
        assert np.array_equal(e[k].cpu().numpy(), o[4][k]), k


@pytest.mark.parametrize("variant", ["temporal-dynamic", "weighted"])
def test_cfg4_hetero_under_filter_and_weights(cabi, cfg4, variant):
    """neighbor_sampling_heterogenous at cfg4 under a temporal filter (neighbor_sampling.rs:36-77, dynamic mode: the
    sample's timestamp becomes its state) and with the weighted sampler (sampling.rs:28-55), through the operator
    surface: all relations of a hop go through one segmented flat hop (hub columns of 10^4..10^5 edges included).
    Window / membership properties on every relation, and the whole call replayed by the oracle."""
    import tch_geometric as tg
    dev, P, I = cfg4
    g = torch.Generator(device=dev)
    g.manual_seed(11)
    seeds = cabi.seed_batches(0xBA7C4, 7000, 1, 1024, 1 << SCALES["A"], dev)[0].contiguous()
    nn = {rel_key(et): [15, 10] for et in EDGE_TYPES}
    sampler, flt, kw = None, None, {}
    if variant == "weighted":
        W = {k: torch.rand(I[k].numel(), device=dev, generator=g, dtype=torch.float64) + 0.1 for k in P}
        sampler = tg.WeightedEdgeSampler(W)
        kw = dict(sampler=orc.SAMPLER_WEIGHTED, weights={k: v.cpu().numpy() for k, v in W.items()})
    else:
        TS = {k: torch.randint(0, 100, (I[k].numel(),), device=dev, generator=g) for k in P}
        ST = torch.full_like(seeds, 50)
        flt = (tg.TemporalEdgeFilter((0, 30), TS, True, tg.TEMPORAL_SAMPLE_DYNAMIC), {"A": ST})
        kw = dict(filter_mode=2, forward=True, window=(0, 30), timestamps={k: v.cpu().numpy() for k, v in TS.items()},
                  inputs_state={"A": ST.cpu().numpy()})
    tg.seed(6)
    s, r, c, e, lo = tg.neighbor_sampling_heterogenous(NODE_TYPES, EDGE_TYPES, P, I, {"A": seeds}, nn, 2, sampler, flt)
    torch.cuda.synchronize()
    assert torch.equal(s["A"][:1024], seeds)
    total = 0
    for (sn, rn, dn) in EDGE_TYPES:
        k = rel_key((sn, rn, dn))
        rows, cols, eidx = r[k], c[k], e[k]
        total += int(rows.numel())
        if rows.numel() == 0:
            continue
        parent = s[dn][cols]
        assert torch.equal(s[sn][rows], I[k][eidx])                                     # sample = indices[edge pointer]
        assert bool(((eidx >= P[k][parent]) & (eidx < P[k][parent + 1])).all())         # in its parent's column
        assert int(torch.unique(cols * (1 << 40) + eidx).numel()) == int(rows.numel())  # without replacement
        assert int(torch.bincount(cols).max()) <= 15
    assert total > 5_000
    hP = {k: v.cpu().numpy() for k, v in P.items()}
    hI = {k: v.cpu().numpy() for k, v in I.items()}
    o = orc.ns_hetero(NODE_TYPES, EDGE_TYPES, hP, hI, {"A": seeds.cpu().numpy()}, nn, 2, orc.rng_philox(6, 0), **kw)
    for t in NODE_TYPES:
        assert np.array_equal(s[t].cpu().numpy(), o[0][t]), t
    for et in EDGE_TYPES:
        k = rel_key(et)
        assert np.array_equal(r[k].cpu().numpy(), o[1][k]) and np.array_equal(c[k].cpu().numpy(), o[2][k]), k
        assert np.array_equal(e[k].cpu().numpy(), o[3][k]), k
        assert [tuple(x) for x in lo[k]] == o[4][k], k


def test_cfg5_rmat27_partitioned_world8(cabi):
    from tch_geometric import partitioned
    dev = torch.device(DEV)
    scale, world = 27, 8
    n = 1 << scale
    row, col = cabi.rmat_edges(scale, n * 16, 0x5EED0000 + scale, dev)
    ptrs, idx, perm = cabi.coo_to_csx(row, col, n, n, True)
    del row, col, perm
    torch.cuda.empty_cache()
    assert int(ptrs[-1]) == n * 16 == 2 ** 31           # the last offset no longer fits int32
    nb, B, fan, seed, first = 4, 1024, [15, 10], 0, 8 * 1024 * 3
    seeds = cabi.seed_batches(0xBA7C4, first, nb, B, n, dev)
    ref = cabi.NsBatchedOut(nb, B, fan, dev)
    cabi.ns_homo_batched(cabi.graph_view(ptrs, idx), seeds, fan, seed, first, ref)
    shards = [partitioned.CscShard.from_full(ptrs, idx, r, world) for r in range(world)]
    assert sum(int(s.indices.numel()) for s in shards) == n * 16
    out, crossed = emulated_world_sample(cabi, shards, seeds, fan, seed, first)
    torch.cuda.synchronize()
    assert crossed > 0
    beyond = 0
    for b in range(nb):
        got, want = out.batch(b), ref.batch(b)
        assert got[4] == want[4]
        for x, y in zip(got[:4], want[:4]):
            assert torch.equal(x, y), b
        beyond += int((got[3] >= 2 ** 30).sum())
    assert beyond > 0                                        # edge pointers in the upper half of the 2^31 range occurred
    # ... and with the slot replies (27 vertex bits + the longest column's position bits: two-chunk slots for both hops;
    # the edge pointer = the owner shard's 2^28-edge offset + 32-bit local parts)
    out2, _ = emulated_world_sample(cabi, shards, seeds, fan, seed, first, slots=True)
    torch.cuda.synchronize()
    for b in range(nb):
        got, want = out2.batch(b), ref.batch(b)
        assert got[4] == want[4]
        for x, y in zip(got[:4], want[:4]):
            assert torch.equal(x, y), b
    del shards, out2
    # one batch of the replicated launch replayed by the oracle on the host copy of the 18 GB CSC
    hp, hi = ptrs.cpu().numpy(), idx.cpu().numpy()
    o = orc.ns_homo(hp, hi, seeds[1].cpu().numpy(), fan, orc.rng_philox(seed, first + 1))
    g = ref.batch(1)
    assert g[4] == o[4]
    for x, y in zip(g[:4], o[:4]):
        assert np.array_equal(x.cpu().numpy(), y)
