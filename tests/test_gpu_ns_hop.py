"""tg_ns_hop (whole-device flat hop) == the oracle: per frontier vertex, the samples of a one-seed oracle call
addressed with that vertex's (call id, draw id)."""
import numpy as np
import pytest
import torch

import orc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("sampler", [0, 1])
@pytest.mark.parametrize("k", [1, 10, 15, 20, 40])
def test_flat_hop_matches_oracle(sampler, k):
    from tch_geometric import _cabi
    dev = torch.device("cuda:0")
    n = 1 << 12
    row, col = orc.rmat_edges(12, n * 16, 4)
    ptrs, idx, _ = orc.to_csc(np.stack([row, col]), n)
    g = _cabi.graph_view(torch.from_numpy(ptrs).to(dev), torch.from_numpy(idx).to(dev))
    rs = np.random.default_rng(k)
    m = 700
    verts = rs.integers(0, n, m)
    verts[::17] = -1                                         # empty slots
    ids = rs.integers(0, 1 << 40, m)
    calls = rs.integers(0, 1000, m)
    cnt, off, nbr, ep, par = _cabi.ns_hop(g, torch.from_numpy(verts).to(dev), k, 77, sampler=sampler,
                                          ids=torch.from_numpy(ids).to(dev), call_ids=torch.from_numpy(calls).to(dev))
    cnt, off, nbr, ep, par = (x.cpu().numpy() for x in (cnt, off, nbr, ep, par))
    assert off[0] == 0 and np.array_equal(np.diff(off), cnt)
    for i in range(m):
        lo, hi = off[i], off[i + 1]
        if verts[i] < 0:
            assert hi == lo
            continue
        o = orc.ns_homo(ptrs, idx, [verts[i]], [k], orc.rng_philox(77, int(calls[i])), sampler=sampler,
                        id_base=int(ids[i]))
        assert np.array_equal(nbr[lo:hi], o[0][1:]) and np.array_equal(ep[lo:hi], o[3]), i
        assert np.all(par[lo:hi] == i)
    # id_base / single call id form == the batched kernel's first hop
    seeds = torch.from_numpy(rs.integers(0, n, 300)).to(dev)
    cnt, off, nbr, ep, par = _cabi.ns_hop(g, seeds, k, 5, call_id=9, sampler=sampler)
    out = _cabi.NsBatchedOut(1, 300, [k], dev)
    _cabi.ns_homo_batched(g, seeds.reshape(1, -1), [k], 5, 9, out, sampler=sampler)
    s, r, c, e, lo = out.batch(0)
    total = int(off[300])
    assert total == e.numel() and torch.equal(nbr[:total], s[300:]) and torch.equal(ep[:total], e)
    assert torch.equal(par[:total], c)


def test_flat_hop_empty():
    from tch_geometric import _cabi
    dev = torch.device("cuda:0")
    g = _cabi.graph_view(torch.tensor([0, 1], device=dev), torch.tensor([0], device=dev))
    cnt, off, nbr, ep, par = _cabi.ns_hop(g, torch.zeros(0, dtype=torch.int64, device=dev), 3, 1)
    assert cnt.numel() == 0 and off.tolist() == [0]
