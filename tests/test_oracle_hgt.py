"""Oracle hgt_sampling (src/algo/hgt_sampling.rs) -- the reference's test config and the structural
properties its algorithm implies."""
import numpy as np
import pytest

import orc
from helpers import has_edge, load_fake_hetero, rel_key


def _graph():
    counts, edges = load_fake_hetero()
    node_types, edge_types = sorted(counts), sorted(edges)
    P, I = {}, {}
    for et in edge_types:
        P[rel_key(et)], I[rel_key(et)], _ = orc.to_csc(edges[et], (counts[et[0]], counts[et[2]]))
    return counts, node_types, edge_types, P, I


def _validate(node_types, edge_types, P, I, samples, rows, cols, eidx):
    for et in edge_types:
        k = rel_key(et)
        src, dst = samples[et[0]], samples[et[2]]
        for j, i, ep in zip(rows[k], cols[k], eidx[k]):                 # hgt_sampling.rs:300-305
            assert has_edge(P[k], I[k], dst[i], src[j])
            assert I[k][ep] == src[j] and P[k][dst[i]] <= ep < P[k][dst[i] + 1]
        per_dst = np.bincount(cols[k], minlength=len(dst)) if len(cols[k]) else np.zeros(len(dst), dtype=int)
        assert per_dst.max(initial=0) <= 50                             # MAX_NEIGHBORS
    for t in node_types:
        assert len(set(samples[t].tolist())) == len(samples[t])         # budget sampling never repeats a node


@pytest.mark.parametrize("mode", ["ref", "philox"])
def test_hgt_reference_config(mode):
    """hgt_sampling.rs:356-429: inputs [0,1,4,5] per type, [20,15] per type, 2 hops, no timestamps."""
    counts, node_types, edge_types, P, I = _graph()
    inputs = {t: [0, 1, 4, 5] for t in node_types}
    ns = {t: [20, 15] for t in node_types}
    rng = orc.rng_ref() if mode == "ref" else orc.rng_philox(3)
    samples, ts, rows, cols, eidx = orc.hgt(node_types, edge_types, P, I, None, inputs, None, ns, 2, rng)
    _validate(node_types, edge_types, P, I, samples, rows, cols, eidx)
    for t in node_types:
        assert samples[t][:4].tolist() == [0, 1, 4, 5]
        assert len(samples[t]) == 4 + 20 + 15                           # budgets are larger than the quotas here
        assert np.all(ts[t] == -1)


def test_hgt_timestamps_and_timerange():
    counts, node_types, edge_types, P, I = _graph()
    g = np.random.default_rng(5)
    RTS = {k: g.integers(-1, 30, len(I[k])) for k in I}
    inputs = {"v0": [0, 1, 4, 5], "v2": [7, 8]}
    in_ts = {"v0": [3, 10, -1, 20], "v2": [5, 25]}
    ns = {t: [10, 6] for t in node_types}
    samples, ts, rows, cols, eidx = orc.hgt(node_types, edge_types, P, I, RTS, inputs, in_ts, ns, 2,
                                            orc.rng_philox(9), timerange=(5, 20))
    _validate(node_types, edge_types, P, I, samples, rows, cols, eidx)
    assert ts["v0"][:4].tolist() == [3, 10, -1, 20] and ts["v2"][:2].tolist() == [5, 25]
    for t in node_types:
        n_in = len(inputs.get(t, []))
        new = ts[t][n_in:]
        assert np.all((new == -1) | ((new >= 5) & (new < 20)))           # hgt_sampling.rs:88-92 half-open range


def test_hgt_types_without_inputs_and_small_budgets():
    counts, node_types, edge_types, P, I = _graph()
    inputs = {"v1": [3]}
    ns = {t: [1000, 1000] for t in node_types}                          # quota far above the budget: take all
    samples, ts, rows, cols, eidx = orc.hgt(node_types, edge_types, P, I, None, inputs, None, ns, 2, orc.rng_philox(1))
    _validate(node_types, edge_types, P, I, samples, rows, cols, eidx)
    assert samples["v1"][0] == 3 and sum(len(v) for v in samples.values()) > 10
    with pytest.raises(RuntimeError):                                    # :202 num_samples[type] missing -> panic
        orc.hgt(node_types, edge_types, P, I, None, inputs, None, {"v1": [5, 5]}, 2, orc.rng_philox(1))


def test_hgt_zero_hops_only_rebuilds_edges_among_inputs():
    counts, node_types, edge_types, P, I = _graph()
    inputs = {t: list(range(60)) for t in node_types}
    samples, ts, rows, cols, eidx = orc.hgt(node_types, edge_types, P, I, None, inputs, None,
                                            {t: [] for t in node_types}, 0, orc.rng_ref())
    _validate(node_types, edge_types, P, I, samples, rows, cols, eidx)
    assert all(len(samples[t]) == 60 for t in node_types) and sum(len(v) for v in rows.values()) > 0
