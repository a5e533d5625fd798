"""Oracle negative sampling (src/algo/negative_sampling.rs)."""
import numpy as np
import pytest

import orc
from helpers import has_edge, load_fake_hetero, load_karate, rel_key


@pytest.mark.parametrize("mode", ["ref", "philox"])
def test_negative_homogeneous(mode):
    """negative_sampling.rs:146-171: karate, all nodes, 10 negatives, 5 tries."""
    ei, n = load_karate()
    ptrs, idx, _ = orc.to_csr(ei, n)
    rng = orc.rng_ref() if mode == "ref" else orc.rng_philox(11)
    inputs = np.arange(n)
    samples, rows, cols, sc = orc.neg_homo(ptrs, idx, (n, n), inputs, 10, 5, rng)
    assert sc == n and samples[:n].tolist() == inputs.tolist()
    assert len(rows) <= n * 10 and len(rows) > n * 5
    for i, j in zip(rows, cols):
        v, w = samples[i], samples[j]
        assert v != w and not has_edge(ptrs, idx, v, w)
    assert len(set(samples.tolist())) == len(samples)      # dedup: every node has one local id


def test_negative_duplicate_inputs_map_to_last_occurrence():
    ptrs = np.array([0, 0, 0, 0], dtype=np.int64)
    idx = np.zeros(0, dtype=np.int64)
    samples, rows, cols, sc = orc.neg_homo(ptrs, idx, (3, 3), [1, 1, 1], 4, 8, orc.rng_philox(2))
    assert sc == 3
    # a negative that hits an input value is mapped to that value's LAST slot (HashMap::extend overwrites)
    assert all(c >= 2 for c in cols)
    assert samples[:3].tolist() == [1, 1, 1]


@pytest.mark.parametrize("mode", ["ref", "philox"])
@pytest.mark.parametrize("inbound", [False, True])
def test_negative_heterogeneous(mode, inbound):
    """negative_sampling.rs:173-233: 3 negatives, 10 tries."""
    counts, edges = load_fake_hetero()
    node_types, edge_types = sorted(counts), sorted(edges)
    if inbound:   # has_edge(w, v) indexes the src->dst CSR by a dst id: only defined when |dst| <= |src|
        edge_types = [e for e in edge_types if counts[e[2]] <= counts[e[0]]]
        node_types = sorted({e[0] for e in edge_types} | {e[2] for e in edge_types})
    P, I, S = {}, {}, {}
    for et in edge_types:
        p, i, _ = orc.to_csr(edges[et], (counts[et[0]], counts[et[2]]))
        P[rel_key(et)], I[rel_key(et)], S[rel_key(et)] = p, i, (counts[et[0]], counts[et[2]])
    inputs = {t: np.arange(20) for t in node_types}
    rng = orc.rng_ref() if mode == "ref" else orc.rng_philox(21)
    samples, rows, cols, sc = orc.neg_hetero(node_types, edge_types, P, I, S, inputs, 3, 10, inbound, rng)
    assert sc == {t: 20 for t in node_types}
    total = 0
    for et in edge_types:
        r = rel_key(et)
        total += len(rows[r])
        for i, j in zip(rows[r], cols[r]):
            v, w = samples[et[0]][i], samples[et[2]][j]
            if inbound:
                # reference quirk: has_edge(w, v) on the src->dst CSR (negative_sampling.rs:113); only
                # meaningful when w is a valid row of that CSR
                if w < counts[et[0]]:
                    assert not has_edge(P[r], I[r], w, v)
            else:
                assert not has_edge(P[r], I[r], v, w)
            assert v != w
    assert total > len(node_types) * 20 * 3 * 0.9


def test_negative_heterogeneous_inbound_out_of_range_is_the_reference_panic():
    counts, edges = load_fake_hetero()
    et = ("v0", "e0", "v2")      # 897 src rows, 982 dst nodes: w can exceed the CSR's row count
    p, i, _ = orc.to_csr(edges[et], (counts["v0"], counts["v2"]))
    with pytest.raises(RuntimeError):
        orc.neg_hetero(["v0", "v2"], [et], {rel_key(et): p}, {rel_key(et): i},
                       {rel_key(et): (counts["v0"], counts["v2"])}, {"v0": np.arange(200)}, 5, 5, True,
                       orc.rng_philox(1))
