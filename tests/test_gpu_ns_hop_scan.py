"""tg_ns_hop_scan (whole-device filtered hop) == the per-batch scan kernel and the oracle."""
import numpy as np
import pytest
import torch

import orc

pytestmark = pytest.mark.gpu


def _setup(dev):
    from tch_geometric import _cabi
    n = 1 << 13
    row, col = orc.rmat_edges(13, n * 16, 0xABC)
    ptrs, idx, _ = orc.to_csc(np.stack([row, col]), n)
    ts = np.random.default_rng(11).integers(0, 100, len(idx))
    g = _cabi.graph_view(torch.from_numpy(ptrs).to(dev), torch.from_numpy(idx).to(dev), None, torch.from_numpy(ts).to(dev))
    return _cabi, ptrs, idx, ts, g, n


@pytest.mark.parametrize("mode,forward,window", [(0, False, (10, 60)), (1, False, (0, 25)), (1, True, (0, 25)),
                                                 (2, True, (0, 30)), (2, False, (0, 30))])
@pytest.mark.parametrize("sampler", [0, 1])
@pytest.mark.parametrize("k", [1, 7, 15, 40])
def test_filtered_flat_hop_matches_batched_kernel_and_oracle(mode, forward, window, sampler, k):
    dev = torch.device("cuda:0")
    cabi, ptrs, idx, ts, g, n = _setup(dev)
    rs = np.random.default_rng(k)
    seeds = orc.seed_batches(7, 0, 1, 600, n)
    st = rs.integers(20, 80, seeds.shape)
    seeds_d, st_d = torch.from_numpy(seeds).to(dev), torch.from_numpy(st).to(dev)
    out = cabi.NsBatchedOut(1, 600, [k], dev, with_states=True)
    cabi.ns_homo_batched(g, seeds_d, [k], 5, 9, out, sampler=sampler, filter_mode=mode, forward=forward, window=window,
                         seeds_state=st_d)
    s, r, c, e, lo = out.batch(0)
    cnt, off, nbr, ep, par, st_out, status = cabi.ns_hop_scan(g, seeds_d[0].contiguous(), st_d[0].contiguous(), k, 5,
                                                              mode, window, forward=forward, call_id=9, sampler=sampler)
    total = int(off[600])
    assert int(status) == 0 and total == e.numel()
    assert torch.equal(nbr[:total], s[600:]) and torch.equal(ep[:total], e) and torch.equal(par[:total], c)
    assert torch.equal(st_out[:total], out.states[0, 600:600 + total])
    o = orc.ns_homo(ptrs, idx, seeds[0], [k], orc.rng_philox(5, 9), sampler=sampler, filter_mode=mode, forward=forward,
                    window=window, timestamps=ts, inputs_state=st[0])
    assert np.array_equal(ep[:total].cpu().numpy(), o[3])


def test_per_vertex_ids_empty_slots_and_overflow():
    dev = torch.device("cuda:0")
    cabi, ptrs, idx, ts, g, n = _setup(dev)
    rs = np.random.default_rng(3)
    m = 500
    verts = rs.integers(0, n, m)
    verts[::13] = -1
    ids, calls, st = rs.integers(0, 1 << 40, m), rs.integers(0, 50, m), rs.integers(0, 100, m)
    t = lambda a: torch.from_numpy(a).to(dev)
    cnt, off, nbr, ep, par, st_out, status = cabi.ns_hop_scan(g, t(verts), t(st), 6, 3, 1, (0, 40), forward=False,
                                                              ids=t(ids), call_ids=t(calls))
    assert int(status) == 0
    off_h, ep_h = off.cpu().numpy(), ep.cpu().numpy()
    for i in range(m):
        if verts[i] < 0:
            assert off_h[i + 1] == off_h[i]
            continue
        o = orc.ns_homo(ptrs, idx, [verts[i]], [6], orc.rng_philox(3, int(calls[i])), filter_mode=1, forward=False,
                        window=(0, 40), timestamps=ts, inputs_state=[st[i]], id_base=int(ids[i]))
        assert np.array_equal(ep_h[off_h[i]:off_h[i + 1]], o[3]), i
    # a workspace that cannot hold the frontier's groups reports it instead of sampling
    cnt, off, nbr, ep, par, st_out, status = cabi.ns_hop_scan(g, t(verts), t(st), 6, 3, 1, (0, 40), group_cap=8)
    assert int(status) == 1 and int(off[m]) == 0
