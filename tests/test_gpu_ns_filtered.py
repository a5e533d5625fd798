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
