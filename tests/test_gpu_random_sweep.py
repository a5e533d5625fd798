"""Randomised parity sweep: random multigraphs (self loops, duplicate edges, isolated vertices, a few hubs) and random
operator configurations; every case must equal the oracle's philox-mode bit for bit."""
import numpy as np
import pytest
import torch

import orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cabi():
    from tch_geometric import _cabi
    return _cabi


def random_graph(rs, n, e, hubs=2):
    row = rs.integers(0, n, e)
    col = rs.integers(0, n, e)
    for h in range(hubs):                                   # a few heavy columns and rows
        m = rs.integers(e // 20, max(e // 8, e // 20 + 1))  # (graphs of fewer than 8 edges: an empty range otherwise)
        col[rs.integers(0, e, m)] = rs.integers(0, n)
        row[rs.integers(0, e, m)] = rs.integers(0, n)
    return np.stack([row, col]).astype(np.int64)


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev) if a is not None else None


@pytest.mark.parametrize("case", range(24))
def test_neighbor_sampling_random_cases(cabi, case):
    dev = torch.device("cuda:0")
    rs = np.random.default_rng(1000 + case)
    n = int(rs.integers(5, 3000))
    e = int(rs.integers(1, 40 * n))
    ei = random_graph(rs, n, e)
    ptrs, idx, _ = orc.to_csc(ei, n)
    hops = int(rs.integers(1, 4))
    fan = [int(rs.integers(1, 21)) for _ in range(hops)]
    if case % 6 == 5:
        fan[0] = int(rs.integers(33, 60))                   # LDS ticket path (uniform) / large k (scan path)
    nb, B = int(rs.integers(1, 5)), int(rs.integers(1, 40))
    seeds = rs.integers(0, n, (nb, B))
    sampler = int(rs.integers(0, 3))
    filt = int(rs.integers(-1, 3))
    kw, w, ts, st = {}, None, None, None
    if sampler == 2:
        w = rs.uniform(0.01, 3.0, e)
        kw["weights"] = w
    if filt >= 0:
        ts = rs.integers(0, 50, e)
        st = rs.integers(0, 50, (nb, B))
        kw.update(filter_mode=filt, forward=bool(rs.integers(0, 2)), window=(int(rs.integers(0, 10)), int(rs.integers(10, 40))),
                  timestamps=ts)
    g = cabi.graph_view(_t(ptrs, dev), _t(idx, dev), _t(w, dev), _t(ts, dev))
    out = cabi.NsBatchedOut(nb, B, fan, dev, with_states=filt >= 0)
    cabi.ns_homo_batched(g, _t(seeds, dev), fan, 77, 5, out, sampler=sampler, filter_mode=filt,
                         forward=kw.get("forward", False), window=kw.get("window", (0, 0)), seeds_state=_t(st, dev))
    torch.cuda.synchronize()
    counts = out.counts.cpu()
    for b in range(nb):
        o = orc.ns_homo(ptrs, idx, seeds[b], fan, orc.rng_philox(77, 5 + b), sampler=sampler,
                        inputs_state=st[b] if st is not None else None, **kw)
        gs, gr, gc, ge, glo = out.batch(b, counts)
        assert glo == o[4], (case, b)
        for x, y in zip((gs, gr, gc, ge), o[:4]):
            assert np.array_equal(x.cpu().numpy(), y), (case, b, sampler, filt, fan)


@pytest.mark.parametrize("case", range(8))
def test_walks_and_negatives_random_cases(cabi, case):
    import tch_geometric as tg
    dev = torch.device("cuda:0")
    rs = np.random.default_rng(2000 + case)
    n = int(rs.integers(4, 1500))
    e = int(rs.integers(1, 30 * n))
    ei = random_graph(rs, n, e)
    ptrs, idx, _ = orc.to_csr(ei, n)
    P, I = _t(ptrs, dev), _t(idx, dev)
    start = rs.integers(0, n, int(rs.integers(1, 300)))
    L = int(rs.integers(1, 40))
    p, q = float(rs.choice([0.25, 0.5, 1.0, 2.0])), float(rs.choice([0.25, 1.0, 1.5, 4.0]))
    tg.seed(500 + case)
    w = tg.random_walk(P, I, _t(start, dev), L, p, q)
    assert np.array_equal(w.cpu().numpy(), orc.random_walk(ptrs, idx, start, L, p, q, orc.rng_philox(500 + case, 0)))
    nts, ets = rs.integers(-1, 30, n), rs.integers(-1, 30, len(idx))
    sts = rs.integers(-1, 25, len(start))
    win = (int(rs.integers(-3, 3)), int(rs.integers(3, 15)))
    a, b = tg.tempo_random_walk(P, I, _t(nts, dev), _t(ets, dev), _t(start, dev), _t(sts, dev), L, win)
    oa, ob = orc.tempo_random_walk(ptrs, idx, nts, ets, start, sts, L, win, orc.rng_philox(500 + case, 1))
    assert np.array_equal(a.cpu().numpy(), oa) and np.array_equal(b.cpu().numpy(), ob)
    num_neg, tries = int(rs.integers(0, 6)), int(rs.integers(1, 6))
    size1 = int(rs.integers(1, n + 1))
    s, r, c, sc = tg.negative_sample_neighbors_homogenous(P, I, (n, size1), _t(start, dev), num_neg, tries)
    o = orc.neg_homo(ptrs, idx, (n, size1), start, num_neg, tries, orc.rng_philox(500 + case, 2))
    assert sc == o[3] and np.array_equal(s.cpu().numpy(), o[0])
    assert np.array_equal(r.cpu().numpy(), o[1]) and np.array_equal(c.cpu().numpy(), o[2])
