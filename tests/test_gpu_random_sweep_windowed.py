"""Randomised parity sweep of the launch forms that round 2 added, directly against the oracle's philox-mode: the
window-ordered many-batch launch (narrow and wide items, with and without the u32 shadows) and the range-partitioned
sampler (emulated worlds; uniform, with replacement, under filters and with weights) over random multigraphs with self
loops, duplicate edges, isolated vertices and a few hubs."""
import numpy as np
import pytest
import torch

import orc
from helpers_part import emulated_world_sample
from test_gpu_random_sweep import _t, random_graph

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def cabi():
    from tch_geometric import _cabi
    return _cabi


def _check(out, ptrs, idx, seeds, fan, seed, call, **okw):
    c = out.counts.cpu()
    for b in range(seeds.shape[0]):
        o = orc.ns_homo(ptrs, idx, seeds[b], fan, orc.rng_philox(seed, call + b), **okw)
        x = out.batch(b, c)
        assert x[4] == o[4], b
        for u, v in zip(x[:4], o[:4]):
            assert np.array_equal(u.cpu().numpy(), v), b
    return c


@pytest.mark.parametrize("case", range(16))
def test_windowed_launch_random_cases(cabi, case):
    dev = torch.device(DEV)
    rs = np.random.default_rng(7000 + case)
    n = int(rs.integers(5, 4000))
    e = int(rs.integers(1, 30 * n))
    ei = random_graph(rs, n, e)
    ptrs, idx, _ = orc.to_csc(ei, n)
    hops = int(rs.integers(1, 4))
    fan = [int(rs.integers(1, 33)) for _ in range(hops)]
    nb, B = int(rs.integers(1, 12)), int(rs.integers(1, 90))
    seeds = rs.integers(0, n, (nb, B))
    sampler = int(rs.integers(0, 2))
    shadows = bool(rs.integers(0, 2))
    tp, ti = _t(ptrs, dev), _t(idx, dev)
    g = cabi.graph_view(tp, ti, indices32=ti.to(torch.int32) if shadows else None,
                        ptrs32=tp.to(torch.int32) if shadows else None)
    form = 1 if case % 2 == 0 else 3                        # narrow / wide work items
    out = cabi.NsBatchedOut(nb, B, fan, dev)
    for t in (out.samples, out.rows, out.cols, out.edge_index):
        t.fill_(-9)
    ws = cabi.ns_homo_workspace(nb, B, fan, dev)
    cabi.ns_homo_batched(g, _t(seeds, dev), fan, 31, 400, out, sampler=sampler, ws=ws, form=form)
    torch.cuda.synchronize()
    _check(out, ptrs, idx, seeds, fan, 31, 400, sampler=sampler)


@pytest.mark.parametrize("case", range(12))
def test_partitioned_random_cases(cabi, case):
    from tch_geometric import partitioned
    dev = torch.device(DEV)
    rs = np.random.default_rng(9000 + case)
    n = int(rs.integers(20, 3000))
    e = int(rs.integers(n, 30 * n))
    ei = random_graph(rs, n, e)
    ptrs, idx, _ = orc.to_csc(ei, n)
    world = int(rs.integers(1, 7))
    hops = int(rs.integers(1, 4))
    fan = [int(rs.integers(1, 17)) for _ in range(hops)]
    nb, B = int(rs.integers(1, 6)), int(rs.integers(1, 60))
    seeds = rs.integers(0, n, (nb, B))
    sampler, filt = int(rs.integers(0, 3)), int(rs.integers(-1, 3))
    if case % 3 == 0:
        sampler, filt = int(rs.integers(0, 2)), -1          # the dedicated unweighted / unfiltered owner kernels
    okw, w, ts, st = dict(sampler=sampler), None, None, None
    if sampler == 2:
        w = rs.uniform(0.05, 3.0, e)
        okw["weights"] = w
    fkw = {}
    if filt >= 0:
        ts = rs.integers(0, 50, e)
        st = rs.integers(0, 50, (nb, B))
        fkw = dict(filter_mode=filt, forward=bool(rs.integers(0, 2)), window=(int(rs.integers(0, 10)), int(rs.integers(10, 40))))
        okw.update(fkw, timestamps=ts)
    tp, ti = _t(ptrs, dev), _t(idx, dev)
    tw = _t(w, dev).to(torch.float64) if w is not None else None
    tts = _t(ts, dev)
    shards = [partitioned.CscShard.from_full(tp, ti, r, world, weights=tw, timestamps=tts) for r in range(world)]
    out, _ = emulated_world_sample(cabi, shards, _t(seeds, dev), fan, 13, 50, sampler=sampler,
                                   seeds_state=_t(st, dev), **(fkw or {}))
    for b in range(nb):                                     # against the oracle, states included
        o = orc.ns_homo(ptrs, idx, seeds[b], fan, orc.rng_philox(13, 50 + b), inputs_state=st[b] if st is not None else None,
                        **okw)
        x = out.batch(b)
        assert x[4] == o[4], (case, b)
        for u, v in zip(x[:4], o[:4]):
            assert np.array_equal(u.cpu().numpy(), v), (case, b)
