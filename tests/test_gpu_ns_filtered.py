"""GPU parity for the scan path of tg_ns_homo_batched: temporal filters (STATIC / RELATIVE / DYNAMIC,
forward and backward), the weighted sampler, and with-replacement under a filter == oracle philox-mode."""
import numpy as np
import pytest
import torch

import orc
from helpers import load_karate, roots_of, validate_neighbor_samples

pytestmark = pytest.mark.gpu
SEED = 0xF117E2


@pytest.fixture(scope="module")
def cabi():
    from tch_geometric import _cabi
    return _cabi


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


def _t(dev, a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev) if a is not None else None


def _run(cabi, dev, ptrs, idx, seeds, fanout, sampler=0, weights=None, filter_mode=-1, forward=False, window=(0, 0),
         ts=None, seeds_state=None, call_id=40):
    g = cabi.graph_view(_t(dev, ptrs), _t(dev, idx), _t(dev, weights), _t(dev, ts))
    out = cabi.NsBatchedOut(seeds.shape[0], seeds.shape[1], fanout, dev, with_states=filter_mode != -1)
    cabi.ns_homo_batched(g, _t(dev, seeds), fanout, SEED, call_id, out, sampler=sampler, filter_mode=filter_mode,
                         forward=forward, window=window, seeds_state=_t(dev, seeds_state))
    torch.cuda.synchronize()
    counts = out.counts.cpu()
    res = []
    for b in range(seeds.shape[0]):
        gs, gr, gc, ge, glo = out.batch(b, counts)
        o = orc.ns_homo(ptrs, idx, seeds[b], fanout, orc.rng_philox(SEED, call_id + b), sampler=sampler,
                        weights=weights, filter_mode=filter_mode, forward=forward, window=window, timestamps=ts,
                        inputs_state=seeds_state[b] if seeds_state is not None else None)
        assert glo == o[4], (b, glo, o[4])
        for g_, o_ in zip((gs, gr, gc, ge), o[:4]):
            assert np.array_equal(g_.cpu().numpy(), o_), b
        res.append(o)
    return res


@pytest.fixture(scope="module")
def karate():
    ei, n = load_karate()
    ptrs, idx, _ = orc.to_csc(ei, n)
    return ptrs, idx


@pytest.fixture(scope="module")
def rmat13():
    n = 1 << 13
    row, col = orc.rmat_edges(13, n * 16, 0xABC)
    ptrs, idx, _ = orc.to_csc(np.stack([row, col]), n)
    return ptrs, idx, n


SEEDS_K = np.array([[0, 1, 4, 5], [33, 32, 2, 3]], dtype=np.int64)


def test_temporal_static_and_relative_reference_config(cabi, dev, karate):
    """neighbor_sampling.rs:497-570."""
    ptrs, idx = karate
    ts = np.random.default_rng(2).integers(0, 4, len(idx))
    st = np.array([[0, 1, 2, 3], [3, 2, 1, 0]], dtype=np.int64)
    res = _run(cabi, dev, ptrs, idx, SEEDS_K, [4, 3], filter_mode=0, window=(0, 2), ts=ts, seeds_state=st)
    for s, r, c, e, lo in res:
        validate_neighbor_samples(ptrs, idx, r, c, s, s, lo, [4, 3])
        assert np.all((ts[e] >= 0) & (ts[e] <= 2))
    res = _run(cabi, dev, ptrs, idx, SEEDS_K, [4, 3], filter_mode=1, forward=False, window=(0, 2), ts=ts,
               seeds_state=st)
    for b, (s, r, c, e, lo) in enumerate(res):
        root = roots_of(c, 4, len(s))
        t0 = st[b][root[r]]
        assert np.all((ts[e] >= t0 - 2) & (ts[e] <= t0))


@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("forward", [False, True])
@pytest.mark.parametrize("sampler", [0, 1])
def test_temporal_modes_rmat(cabi, dev, rmat13, mode, forward, sampler):
    ptrs, idx, n = rmat13
    g = np.random.default_rng(11)
    ts = g.integers(0, 100, len(idx))
    seeds = orc.seed_batches(5, 0, 6, 96, n)
    st = g.integers(20, 80, seeds.shape)
    window = (10, 60) if mode == 0 else (0, 25)
    _run(cabi, dev, ptrs, idx, seeds, [7, 5], sampler=sampler, filter_mode=mode, forward=forward, window=window,
         ts=ts, seeds_state=st)


@pytest.mark.parametrize("filtered", [False, True])
def test_weighted_sampler(cabi, dev, karate, rmat13, filtered):
    """neighbor_sampling.rs:466-495 (weights U(0.2, 5.0) f64), karate then RMAT hubs."""
    for ptrs, idx, seeds, fan in ((karate[0], karate[1], SEEDS_K, [4, 3]),
                                  (rmat13[0], rmat13[1], orc.seed_batches(6, 0, 5, 128, rmat13[2]), [10, 5])):
        g = np.random.default_rng(4)
        w = g.uniform(0.2, 5.0, len(idx))
        kw = {}
        if filtered:
            kw = dict(filter_mode=1, forward=True, window=(0, 40), ts=g.integers(0, 100, len(idx)),
                      seeds_state=g.integers(0, 60, seeds.shape))
        res = _run(cabi, dev, ptrs, idx, seeds, fan, sampler=2, weights=w, **kw)
        for s, r, c, e, lo in res:
            validate_neighbor_samples(ptrs, idx, r, c, s, s, lo, fan)


def test_weighted_zero_sum_reports_the_reference_panic(cabi, dev, karate):
    ptrs, idx = karate
    g = cabi.graph_view(_t(dev, ptrs), _t(dev, idx), _t(dev, np.zeros(len(idx))), None)
    out = cabi.NsBatchedOut(1, 4, [2, 2], dev)
    cabi.ns_homo_batched(g, _t(dev, SEEDS_K[:1]), [2, 2], 1, 1, out, sampler=2)
    assert int(out.counts.cpu()[0, 0]) == -1          # sampling.rs:49 panics on an empty float range


def test_filter_that_admits_nothing_and_missing_buffers(cabi, dev, karate):
    ptrs, idx = karate
    ts = np.zeros(len(idx), dtype=np.int64)
    res = _run(cabi, dev, ptrs, idx, SEEDS_K, [3, 3], filter_mode=0, window=(5, 6), ts=ts,
               seeds_state=np.zeros_like(SEEDS_K))
    assert all(len(r[1]) == 0 for r in res)
    g = cabi.graph_view(_t(dev, ptrs), _t(dev, idx))
    out = cabi.NsBatchedOut(1, 4, [2], dev)
    with pytest.raises(cabi.TchGeoError):
        cabi.ns_homo_batched(g, _t(dev, SEEDS_K[:1]), [2], 1, 1, out, sampler=2)          # no weights
    with pytest.raises(cabi.TchGeoError):
        cabi.ns_homo_batched(g, _t(dev, SEEDS_K[:1]), [2], 1, 1, out, filter_mode=0)      # no timestamps


# ---------------------------------------------------------------- round 4: few batches take the flat hops by themselves
def _flat_vs_batched(cabi, dev, ptrs, idx, seeds, fanout, sampler=0, weights=None, filter_mode=-1, forward=False,
                     window=(0, 0), ts=None, seeds_state=None, call_id=70, oracle_batches=()):
    """tg_ns_homo_batched_ws with the workspace of tg_ns_homo_batched_workspace_bytes (the flat path) == tg_ns_homo_batched
    (one workgroup per batch) on the device, word for word, and == the oracle for `oracle_batches`."""
    g = cabi.graph_view(_t(dev, ptrs), _t(dev, idx), _t(dev, weights), _t(dev, ts))
    nb, B = seeds.shape
    ws = cabi.ns_homo_batched_workspace(g, nb, B, fanout, dev, sampler=sampler, filter_mode=filter_mode)
    assert ws is not None and ws.numel() > 0
    kw = dict(sampler=sampler, filter_mode=filter_mode, forward=forward, window=window, seeds_state=_t(dev, seeds_state))
    a = cabi.NsBatchedOut(nb, B, fanout, dev, with_states=filter_mode != -1)
    b = cabi.NsBatchedOut(nb, B, fanout, dev, with_states=filter_mode != -1)
    for t in (a.samples, a.rows, a.cols, a.edge_index, b.samples, b.rows, b.cols, b.edge_index):
        t.fill_(-7)
    cabi.ns_homo_batched(g, _t(dev, seeds), fanout, SEED, call_id, a, ws=ws, **kw)
    cabi.ns_homo_batched(g, _t(dev, seeds), fanout, SEED, call_id, b, **kw)
    torch.cuda.synchronize()
    assert torch.equal(a.counts, b.counts) and torch.equal(a.layer_offsets, b.layer_offsets)
    ca = a.counts.cpu()
    for j in range(nb):
        if int(ca[j, 0]) < 0:
            continue                                                  # the reference panics here: nothing else is defined
        for x, y in zip(a.batch(j, ca)[:4], b.batch(j, ca)[:4]):
            assert torch.equal(x, y), j
    for j in oracle_batches:
        o = orc.ns_homo(ptrs, idx, seeds[j], fanout, orc.rng_philox(SEED, call_id + j), sampler=sampler, weights=weights,
                        filter_mode=filter_mode, forward=forward, window=window, timestamps=ts,
                        inputs_state=seeds_state[j] if seeds_state is not None else None)
        got = a.batch(j, ca)
        assert got[4] == o[4]
        for x, y in zip(got[:4], o[:4]):
            assert np.array_equal(x.cpu().numpy(), y), j
    return a


@pytest.mark.parametrize("nb", [8, 64, 256])
@pytest.mark.parametrize("case", ["dynamic_fwd", "static", "relative_bwd_repl", "weighted", "weighted_filtered"])
def test_few_batches_take_the_flat_hops(cabi, dev, rmat13, nb, case):
    ptrs, idx, n = rmat13
    g = np.random.default_rng(21)
    ts = g.integers(0, 100, len(idx))
    w = g.uniform(0.2, 5.0, len(idx))
    seeds = orc.seed_batches(9, 3, nb, 48, n)
    st = g.integers(20, 80, seeds.shape)
    kw = {"dynamic_fwd": dict(filter_mode=2, forward=True, window=(0, 30), ts=ts, seeds_state=st),
          "static": dict(filter_mode=0, window=(10, 60), ts=ts, seeds_state=st),
          "relative_bwd_repl": dict(sampler=1, filter_mode=1, forward=False, window=(0, 25), ts=ts, seeds_state=st),
          "weighted": dict(sampler=2, weights=w),
          "weighted_filtered": dict(sampler=2, weights=w, filter_mode=1, forward=True, window=(0, 40), ts=ts, seeds_state=st)}[case]
    _flat_vs_batched(cabi, dev, ptrs, idx, seeds, [7, 5], oracle_batches=(0, nb - 1), **kw)


def test_flat_path_falls_back_where_it_cannot_finish(cabi, dev):
    """(a) a frontier that repeats a hub needs more 512-edge column groups than the workspace holds; (b) a weighted column
    whose sum is not positive is the reference's panic (the batch's counts[0] = -1): both raise the flat path's status word
    and the per-batch kernel behind it produces the result."""
    n, hub = 64, 100_000
    rs = np.random.default_rng(5)
    deg = np.full(n, 3, dtype=np.int64)
    deg[0] = hub
    ptrs = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(deg, out=ptrs[1:])
    idx = rs.integers(0, n, int(ptrs[-1]))
    idx[: hub: 3] = 0
    ts = rs.integers(0, 50, len(idx))
    seeds = np.zeros((64, 16), dtype=np.int64)                        # 1 024 frontier slots, all the hub: ~200 K groups
    _flat_vs_batched(cabi, dev, ptrs, idx, seeds, [6, 4], filter_mode=0, window=(5, 30), ts=ts,
                     seeds_state=np.zeros_like(seeds), oracle_batches=(0, 63))
    ei, nk = load_karate()
    kp, ki, _ = orc.to_csc(ei, nk)
    w = np.ones(len(ki))
    w[int(kp[5]):int(kp[6])] = 0.0                                    # vertex 5's column sums to zero
    seeds = np.array([[0, 1, 2, 3], [4, 5, 6, 7], [8, 9, 10, 11]], dtype=np.int64)
    a = _flat_vs_batched(cabi, dev, kp, ki, seeds, [2, 2], sampler=2, weights=w, oracle_batches=(0, 2))
    assert a.counts.cpu()[:, 0].tolist()[1] == -1 and int(a.counts.cpu()[0, 0]) > 0


def test_workspace_query_by_configuration(cabi, dev, rmat13):
    ptrs, idx, n = rmat13
    g = cabi.graph_view(_t(dev, ptrs), _t(dev, idx), _t(dev, np.ones(len(idx))), _t(dev, np.zeros(len(idx), dtype=np.int64)))
    assert cabi.ns_homo_batched_workspace(g, 64, 32, [5, 5], dev, sampler=2) is not None           # few weighted batches: flat
    assert cabi.ns_homo_batched_workspace(g, 1024, 32, [5, 5], dev, sampler=2) is None             # many: a workgroup per batch
    assert cabi.ns_homo_batched_workspace(g, 64, 32, [5, 5], dev, filter_mode=2) is not None
    assert cabi.ns_homo_batched_workspace(g, 256, 32, [5, 5], dev, filter_mode=2) is not None
    assert cabi.ns_homo_batched_workspace(g, 257, 32, [5, 5], dev, filter_mode=2) is None
    plain = cabi.ns_homo_batched_workspace(g, 64, 32, [5, 5], dev)
    assert plain is not None and plain.numel() == cabi.ns_homo_workspace(64, 32, [5, 5], dev, graph=g).numel()
