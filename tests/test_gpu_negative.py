"""GPU parity for negative sampling through the operator surface == oracle philox-mode."""
import numpy as np
import pytest
import torch

import orc
from helpers import has_edge, load_fake_hetero, load_karate, rel_key

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tg():
    import tch_geometric
    return tch_geometric


def _np(t):
    return t.cpu().numpy()


@pytest.mark.parametrize("where", ["cpu", "cuda"])
def test_negative_homogeneous_reference_config(tg, where):
    """negative_sampling.rs:146-171: karate, all nodes, 10 negatives, 5 tries."""
    ei, n = load_karate()
    ptrs, idx, _ = orc.to_csr(ei, n)
    P, I = torch.from_numpy(ptrs).to(where), torch.from_numpy(idx).to(where)
    inputs = torch.arange(n).to(where)
    tg.seed(31)
    s, r, c, sc = tg.negative_sample_neighbors_homogenous(P, I, (n, n), inputs, 10, 5)
    os_, or_, oc, osc = orc.neg_homo(ptrs, idx, (n, n), np.arange(n), 10, 5, orc.rng_philox(31, 0))
    assert sc == osc == n and s.device.type == where
    assert np.array_equal(_np(s), os_) and np.array_equal(_np(r), or_) and np.array_equal(_np(c), oc)
    for i, j in zip(_np(r), _np(c)):                       # :167-170
        assert not has_edge(ptrs, idx, _np(s)[i], _np(s)[j])


def test_negative_homogeneous_rmat_dedup_order_and_duplicate_inputs(tg):
    n = 1 << 12
    row, col = orc.rmat_edges(12, n * 16, 21)
    ptrs, idx, _ = orc.to_csr(np.stack([row, col]), n)
    P, I = torch.from_numpy(ptrs).cuda(), torch.from_numpy(idx).cuda()
    inputs = orc.seed_batches(2, 0, 1, 3000, 64)[0]        # only 64 distinct values: heavy duplication
    tg.seed(8)
    for num_neg, tries, size in ((7, 3, n), (1, 1, n), (20, 4, 50)):   # size 50: few distinct negatives, many repeats
        call = tg.rng_state()[1]
        s, r, c, sc = tg.negative_sample_neighbors_homogenous(P, I, (n, size), torch.from_numpy(inputs).cuda(),
                                                              num_neg, tries)
        o = orc.neg_homo(ptrs, idx, (n, size), inputs, num_neg, tries, orc.rng_philox(8, call))
        assert sc == o[3]
        assert np.array_equal(_np(s), o[0]) and np.array_equal(_np(r), o[1]) and np.array_equal(_np(c), o[2])


def test_negative_homogeneous_edge_cases(tg):
    P, I = torch.tensor([0, 1, 2]).cuda(), torch.tensor([1, 0]).cuda()     # 2 nodes, both directions linked
    s, r, c, sc = tg.negative_sample_neighbors_homogenous(P, I, (2, 2), torch.tensor([0, 1]).cuda(), 3, 4)
    assert sc == 2 and s.tolist() == [0, 1] and r.numel() == 0 and c.numel() == 0      # nothing admissible
    s, r, c, sc = tg.negative_sample_neighbors_homogenous(P, I, (2, 2), torch.zeros(0, dtype=torch.int64).cuda(), 3, 4)
    assert sc == 0 and s.numel() == 0 and r.numel() == 0
    s, r, c, sc = tg.negative_sample_neighbors_homogenous(P, I, (2, 2), torch.tensor([0]).cuda(), 0, 4)
    assert sc == 1 and s.tolist() == [0] and r.numel() == 0


@pytest.mark.parametrize("inbound", [False, True])
def test_negative_heterogeneous(tg, inbound):
    """negative_sampling.rs:173-233 config (3 negatives, 10 tries), all three node types as inputs."""
    counts, edges = load_fake_hetero()
    node_types, edge_types = sorted(counts), sorted(edges)
    if inbound:
        edge_types = [e for e in edge_types if counts[e[2]] <= counts[e[0]]]
        node_types = sorted({e[0] for e in edge_types} | {e[2] for e in edge_types})
    P, I, S, Pd, Id = {}, {}, {}, {}, {}
    for et in edge_types:
        k = rel_key(et)
        P[k], I[k], _ = orc.to_csr(edges[et], (counts[et[0]], counts[et[2]]))
        S[k] = (counts[et[0]], counts[et[2]])
        Pd[k], Id[k] = torch.from_numpy(P[k]).cuda(), torch.from_numpy(I[k]).cuda()
    inputs = {t: np.arange(0, 200, 2) % 60 for t in node_types}       # duplicates inside the inputs
    if not inbound:
        del inputs[node_types[-1]]                                    # a type without an `inputs` entry
    tg.seed(404)
    s, r, c, sc = tg.negative_sample_neighbors_heterogenous(
        node_types, edge_types, Pd, Id, S, {t: torch.from_numpy(v).cuda() for t, v in inputs.items()}, 3, 10, inbound)
    os_, or_, oc, osc = orc.neg_hetero(node_types, edge_types, P, I, S, inputs, 3, 10, inbound, orc.rng_philox(404, 0))
    assert sc == osc
    for t in node_types:
        assert np.array_equal(_np(s[t]), os_[t]), t
    for et in edge_types:
        k = rel_key(et)
        assert np.array_equal(_np(r[k]), or_[k]) and np.array_equal(_np(c[k]), oc[k]), k
        if not inbound:
            for i, j in zip(_np(r[k]), _np(c[k])):         # :228-231
                assert not has_edge(P[k], I[k], _np(s[et[0]])[i], _np(s[et[2]])[j])


def test_negative_heterogeneous_inbound_panic_is_reported(tg):
    counts, edges = load_fake_hetero()
    et = ("v0", "e0", "v2")
    p, i, _ = orc.to_csr(edges[et], (counts["v0"], counts["v2"]))
    k = rel_key(et)
    with pytest.raises(RuntimeError, match="reference panics"):
        tg.negative_sample_neighbors_heterogenous(["v0", "v2"], [et], {k: torch.from_numpy(p).cuda()},
                                                  {k: torch.from_numpy(i).cuda()}, {k: (counts["v0"], counts["v2"])},
                                                  {"v0": torch.arange(200).cuda()}, 5, 5, True)
