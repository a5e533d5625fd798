"""Every homogeneous operator on the reference's mid-size fixture graph (tests/fakedataset.npz, 1144 nodes / 22648
edges): surface == oracle philox-mode, plus the reference's structural invariants."""
import numpy as np
import pytest
import torch

import orc
from helpers import has_edge, load_fake_dataset, validate_neighbor_samples

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tg():
    import tch_geometric
    return tch_geometric


def _c(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_neighbor_sampling_variants(tg):
    ei, n = load_fake_dataset()
    ptrs, idx, perm = orc.to_csc(ei, n)
    p, i, pm = tg.to_csc(_c(ei), n)
    assert np.array_equal(p.cpu().numpy(), ptrs) and np.array_equal(i.cpu().numpy(), idx) and np.array_equal(pm.cpu().numpy(), perm)
    rs = np.random.default_rng(1)
    seeds = np.tile(np.arange(8), 4)                        # examples/neighbor_sampling.py:17-19
    w, ts = rs.uniform(0.1, 2.0, len(idx)), rs.integers(0, 5, len(idx))
    st = rs.integers(0, 5, len(seeds))
    cases = [(None, None, {}),
             (tg.UniformEdgeSampler(True), None, dict(sampler=orc.SAMPLER_UNIFORM_REPL)),
             (tg.WeightedEdgeSampler(_c(w)), None, dict(sampler=orc.SAMPLER_WEIGHTED, weights=w)),
             (None, (tg.TemporalEdgeFilter((0, 3), _c(ts), False, tg.TEMPORAL_SAMPLE_RELATIVE), _c(st)),
              dict(filter_mode=orc.FILTER_RELATIVE, forward=False, window=(0, 3), timestamps=ts, inputs_state=st))]
    for sampler, flt, kw in cases:
        tg.seed(3)
        s, r, c, e, lo = tg.neighbor_sampling_homogenous(p, i, _c(seeds), [4, 3], sampler, flt)
        o = orc.ns_homo(ptrs, idx, seeds, [4, 3], orc.rng_philox(3, 0), **kw)
        assert lo == o[4]
        for a, b in zip((s, r, c, e), o[:4]):
            assert np.array_equal(a.cpu().numpy(), b)
        if sampler is None or not getattr(sampler, "with_replacement", False):
            validate_neighbor_samples(ptrs, idx, o[1], o[2], o[0], o[0], o[4], [4, 3])


def test_walks_and_negatives(tg):
    ei, n = load_fake_dataset()
    ptrs, idx, _ = orc.to_csr(ei, n)
    P, I = _c(ptrs), _c(idx)
    start = np.tile(np.arange(8), 4)
    tg.seed(8)
    w = tg.random_walk(P, I, _c(start), 9, 1.0, 1.5).cpu().numpy()       # examples/random_walk.py
    assert np.array_equal(w, orc.random_walk(ptrs, idx, start, 9, 1.0, 1.5, orc.rng_philox(8, 0)))
    for row in w:
        for a, b in zip(row[:-1], row[1:]):
            assert b < 0 or has_edge(ptrs, idx, a, b)
    s, r, c, cnt = tg.negative_sample_neighbors_homogenous(P, I, (n, n), _c(np.arange(n)), 5, 5)   # examples/negative_sampling.py
    o = orc.neg_homo(ptrs, idx, (n, n), np.arange(n), 5, 5, orc.rng_philox(8, 1))
    assert cnt == o[3] and np.array_equal(s.cpu().numpy(), o[0]) and np.array_equal(r.cpu().numpy(), o[1])
    assert np.array_equal(c.cpu().numpy(), o[2])
    sv = s.cpu().numpy()
    for a, b in zip(o[1][:200], o[2][:200]):                              # negative_sampling.rs:167-170
        assert not has_edge(ptrs, idx, sv[a], sv[b])
