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


@pytest.mark.parametrize("packed", [False, True])
@pytest.mark.parametrize("weighted", [False, True])
def test_segmented_hop_equals_one_hop_per_segment(weighted, packed):
    """tg_ns_hop_segments over a frontier of three segments (two graphs, different fan-outs and draw tags) == the
    single-graph hop run once per segment, concatenated; with the host-side (padded) layout and with a device-side
    layout that packs the segments' real frontiers back to back."""
    dev = torch.device("cuda:0")
    cabi, ptrs, idx, ts, g, n = _setup(dev)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    rs = np.random.default_rng(5)
    w_a = rs.uniform(0.1, 3.0, len(idx))
    ga = cabi.graph_view(t(ptrs), t(idx), t(w_a), t(ts))
    n2 = 1 << 11                                              # a second, smaller graph: ids of the frontier stay below n2
    row, col = orc.rmat_edges(11, n2 * 8, 0x77)
    p2, i2, _ = orc.to_csc(np.stack([row, col]), n2)
    ts2, w2 = rs.integers(0, 100, len(i2)), rs.uniform(0.1, 3.0, len(i2))
    gb = cabi.graph_view(t(p2), t(i2), t(w2), t(ts2))
    caps, real, fans, tags = [300, 200, 260], [300, 120, 0], [7, 12, 4], [0x11, 0x2200 | 0x11, 0x3300 | 0x11]
    graphs = [ga, gb, ga]
    sampler = 2 if weighted else 0
    mode, window = (1, (0, 40))
    # per segment: vertices, states, draw ids; segment 2's real frontier is empty
    verts = [rs.integers(0, n2, c) for c in caps]
    states = [rs.integers(20, 80, c) for c in caps]
    idsv = [rs.integers(0, 1 << 30, c) for c in caps]
    if packed:      # real frontiers back to back; what follows them is never read
        begins = np.concatenate([[0], np.cumsum(real)])
        tail = np.zeros(sum(caps) - sum(real), np.int64)
        V, S, IDS = (np.concatenate([a[:r] for a, r in zip(arrs, real)] + [tail]) for arrs in (verts, states, idsv))
        layout = t(begins.astype(np.int64))
    else:           # every segment padded to its capacity with -1
        begins = np.concatenate([[0], np.cumsum(caps)])
        padded = []
        for v, r in zip(verts, real):
            v = v.copy()
            v[r:] = -1
            padded.append(v)
        V, S, IDS = np.concatenate(padded), np.concatenate(states), np.concatenate(idsv)
        layout = None
    host_begins = np.concatenate([[0], np.cumsum(caps)])
    segs = [(graphs[j], int(host_begins[j]), fans[j], tags[j]) for j in range(3)]
    cnt, off, nbr, ep, par, st_out, status = cabi.ns_hop_segments(segs, t(V), t(S), 5, filter_mode=mode, window=window,
                                                                  call_id=9, sampler=sampler, ids=t(IDS), layout=layout)
    assert int(status) == 0
    m_eff = int(begins[-1])
    off_h = off.cpu().numpy()
    for j in range(3):
        r = real[j]
        if r == 0:
            continue
        if weighted:
            c1, o1, n1, e1, p1, s1, st1 = _one(cabi, graphs[j], t(verts[j][:r]), t(states[j][:r]), fans[j], mode, window,
                                               t(idsv[j][:r]), tags[j], weighted=True)
        else:
            c1, o1, n1, e1, p1, s1, st1 = cabi.ns_hop_scan(graphs[j], t(verts[j][:r]), t(states[j][:r]), fans[j], 5, mode,
                                                           window, call_id=9, ids=t(idsv[j][:r]), rng_tag=tags[j])
        assert int(st1) == 0
        tot = int(o1[r])
        b = int(begins[j])
        lo, hi = int(off_h[b]), int(off_h[b + r])
        assert hi - lo == tot and tot > 0, j
        assert torch.equal(cnt[b:b + r], c1[:r])
        assert torch.equal(nbr[lo:hi], n1[:tot]) and torch.equal(ep[lo:hi], e1[:tot]), j
        assert torch.equal(par[lo:hi] - b, p1[:tot]) and torch.equal(st_out[lo:hi], s1[:tot]), j
    assert int(off_h[m_eff]) == int(off_h[int(begins[2])])      # the empty third segment adds nothing


def _one(cabi, graph, verts, states, k, mode, window, ids, tag, weighted):
    """tg_ns_hop_weighted for one segment (the wrapper of the unweighted hop is cabi.ns_hop_scan)"""
    import ctypes as C
    m, dev = verts.numel(), verts.device
    o = dict(dtype=torch.int64, device=dev)
    cnt, offsets = torch.empty(m, **o), torch.empty(m + 1, **o)
    nbr, ep, par, st_out = (torch.empty(m * k, **o) for _ in range(4))
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    hin, hout, flt = cabi.TgHopIn(), cabi.TgHopOut(), cabi.TgHopFilter()
    hin.vertices, hin.ids, hin.m, hin.fanout, hin.sampler, hin.rng_tag = verts.data_ptr(), ids.data_ptr(), m, k, 2, tag
    hout.cnt, hout.offsets = cnt.data_ptr(), offsets.data_ptr()
    hout.neighbors, hout.edge_ptrs, hout.parents = nbr.data_ptr(), ep.data_ptr(), par.data_ptr()
    flt.filter_mode, flt.forward, flt.win_lo, flt.win_hi, flt.states = mode, 0, window[0], window[1], states.data_ptr()
    nbytes = C.c_int64(0)
    cabi.check(cabi.lib.tg_ns_hop_scan_workspace_bytes(C.c_int64(m), C.c_int32(k), C.c_int64(1), C.byref(nbytes)))
    ws = torch.empty(nbytes.value // 8 + 1, **o)
    rng = cabi.TgRng(5, 9)
    cabi.check(cabi.lib.tg_ns_hop_weighted(C.byref(graph), C.byref(hin), C.byref(flt), C.byref(rng), C.byref(hout),
                                           cabi.ptr(st_out), cabi.ptr(status), cabi.ptr(ws), C.c_int64(nbytes.value),
                                           cabi.stream_ptr(dev)))
    return cnt, offsets, nbr, ep, par, st_out, status


@pytest.mark.parametrize("mode,window", [(-1, (0, 0)), (0, (10, 60)), (2, (0, 30))])
@pytest.mark.parametrize("k", [1, 9, 40])
def test_weighted_group_form_equals_the_column_form(mode, window, k):
    """tg_ns_hop_weighted_groups (totals and draws flat over the 512-edge groups of all columns) == tg_ns_hop_weighted (a
    wavefront / a workgroup per column): same counts, neighbours, edge pointers, parents, states; a hub graph so that
    columns span many groups; too small a group bound raises status bit 1 and samples nothing."""
    import ctypes as C
    dev = torch.device("cuda:0")
    cabi, ptrs, idx, ts, g, n = _setup(dev)
    rs = np.random.default_rng(31 + k)
    ei = np.stack([rs.integers(0, n, 300000), rs.integers(0, n, 300000)])
    ei[1, rs.integers(0, 300000, 120000)] = rs.integers(0, 6, 120000)       # six hub columns of ~20 K edges
    ptrs, idx, _ = orc.to_csc(ei, n)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    w = rs.uniform(0.05, 4.0, len(idx))
    w[rs.integers(0, len(idx), len(idx) // 20)] = 1e-300                     # some vanishing weights
    tsv = rs.integers(0, 100, len(idx))
    graph = cabi.graph_view(t(ptrs), t(idx), t(w), t(tsv))
    verts = np.concatenate([np.arange(6), rs.integers(0, n, 1500), [-1, -1, 3]])
    states, ids = rs.integers(20, 80, len(verts)), rs.integers(0, 1 << 40, len(verts))
    V, S, IDS = t(verts), t(states), t(ids)
    ref = _one(cabi, graph, V, S, k, mode, window, IDS, 0x31, True)

    def groups(group_cap):
        m = V.numel()
        o = dict(dtype=torch.int64, device=dev)
        cnt, offsets = torch.empty(m, **o), torch.empty(m + 1, **o)
        nbr, ep, par, st_out = (torch.full((m * k,), -7, **o) for _ in range(4))
        status = torch.zeros(1, dtype=torch.int32, device=dev)
        hin, hout, flt = cabi.TgHopIn(), cabi.TgHopOut(), cabi.TgHopFilter()
        hin.vertices, hin.ids, hin.m, hin.fanout, hin.sampler, hin.rng_tag = V.data_ptr(), IDS.data_ptr(), m, k, 2, 0x31
        hout.cnt, hout.offsets = cnt.data_ptr(), offsets.data_ptr()
        hout.neighbors, hout.edge_ptrs, hout.parents = nbr.data_ptr(), ep.data_ptr(), par.data_ptr()
        flt.filter_mode, flt.forward, flt.win_lo, flt.win_hi, flt.states = mode, 0, window[0], window[1], S.data_ptr()
        nbytes = C.c_int64(0)
        cabi.check(cabi.lib.tg_ns_hop_weighted_workspace_bytes(C.c_int64(m), C.c_int32(k), C.c_int64(group_cap), C.byref(nbytes)))
        ws = torch.empty(nbytes.value // 8 + 1, **o)
        rng = cabi.TgRng(5, 9)
        cabi.check(cabi.lib.tg_ns_hop_weighted_groups(C.byref(graph), C.byref(hin), C.byref(flt), C.byref(rng), C.byref(hout),
                                                      cabi.ptr(st_out), cabi.ptr(status), cabi.ptr(ws), C.c_int64(nbytes.value),
                                                      C.c_int64(group_cap), cabi.stream_ptr(dev)))
        return cnt, offsets, nbr, ep, par, st_out, status

    got = groups(8192)
    assert int(got[6].item()) == int(ref[6].item()) == 0
    assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1])
    total = int(ref[1][-1].item())
    assert total > 6 * min(k, 1) and int(ref[0][:6].min().item()) == k     # the hubs fill every slot
    for a, b in zip(got[2:5], ref[2:5]):
        assert torch.equal(a[:total], b[:total])
    if mode != -1:
        assert torch.equal(got[5][:total], ref[5][:total])
    low = groups(1024)                                                       # ~1 750 groups needed: the bound is reached
    assert int(low[6].item()) & 1 and int(low[0].sum().item()) == 0 and int(low[1][-1].item()) == 0
