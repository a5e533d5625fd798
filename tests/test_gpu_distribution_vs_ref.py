"""Whole operators on the device (counter-addressed Philox, reservoir by tickets) against the reference's own
algorithm and stream (oracle ref-mode: one sequential rand-0.8.5 Xoshiro256++ stream, `gen_range` with rejection, the
literal loops of src/utils/sampling.rs:6-69 and src/algo/random_walk.rs:52-66): the two cannot agree draw by draw
(hard part 1), so what is pinned here is that they produce the same OUTPUT LAW -- chi-square homogeneity
of >= 40 000 device outcomes (one launch, thousands of call ids) against >= 40 000 ref-mode outcomes, on the ordered
slot contents (each slot's marginal, and ordered pairs of slots), with a negative control showing the statistic
separates different samplers."""
import numpy as np
import pytest
import torch

import orc
from dist_helpers import assert_different_law, assert_same_law
from helpers import load_karate

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
N = 40960


@pytest.fixture(scope="module")
def cabi():
    from tch_geometric import _cabi
    return _cabi


@pytest.fixture(scope="module")
def karate():
    ei, n = load_karate()
    ptrs, idx, _ = orc.to_csc(ei, n)
    rptrs, ridx, _ = orc.to_csr(ei, n)
    return n, ptrs, idx, rptrs, ridx


def _t(a, dtype=torch.int64):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV).to(dtype)


def _device_slots(cabi, ptrs, idx, vertex, k, sampler, weights=None, seed=123):
    """N outcomes of sampling `vertex` with fan-out k: 4096 call ids x 10 seeds in one launch -> [N, k] positions in
    the column (or -1)"""
    nb, B = N // 10, 10
    g = cabi.graph_view(_t(ptrs), _t(idx), weights=_t(weights, torch.float64) if weights is not None else None)
    seeds = torch.full((nb, B), vertex, dtype=torch.int64, device=DEV)
    out = cabi.NsBatchedOut(nb, B, [k], torch.device(DEV))
    cabi.ns_homo_batched(g, seeds, [k], seed, 7, out, sampler=sampler)
    torch.cuda.synchronize()
    cnt = min(k, int(ptrs[vertex + 1] - ptrs[vertex])) if sampler != 1 else k
    assert bool((out.counts[:, 1] == B * cnt).all())
    e = out.edge_index[:, :B * cnt].reshape(nb * B, cnt).cpu().numpy()      # slot order = output order
    return e - int(ptrs[vertex])


def _ref_slots(ptrs, idx, vertex, k, sampler, weights=None):
    """the same N outcomes from the reference's algorithm on the reference's stream (one call, one stream threaded
    through all N vertices, neighbor_sampling.rs:165,206)"""
    rng = orc.rng_ref_child(orc.rng_ref())                                  # utils/random.rs:19-22
    o = orc.ns_homo(ptrs, idx, np.full(N, vertex), [k], rng, sampler=sampler, weights=weights)
    cnt = len(o[3]) // N
    return o[3].reshape(N, cnt) - int(ptrs[vertex])


def _compare_slots(dev, ref, deg, what):
    assert dev.shape == ref.shape
    k = dev.shape[1]
    for s in range(k):                                                      # each ordered slot's marginal
        assert_same_law(dev[:, s], ref[:, s], "%s slot %d" % (what, s), n_bins=deg)
    for a, b in ((0, 1), (0, k - 1), (k // 2, k - 1)) if k > 1 else ():     # ordered pairs of slots
        assert_same_law(dev[:, a] * deg + dev[:, b], ref[:, a] * deg + ref[:, b], "%s slots (%d,%d)" % (what, a, b),
                        n_bins=deg * deg)
    assert_same_law(np.sort(dev, axis=1)[:, 0], np.sort(ref, axis=1)[:, 0], what + " smallest position", n_bins=deg)


def test_uniform_without_replacement_law(cabi, karate):
    """reservoir_sampling (sampling.rs:6-26, quirk j in [0,i) included) vs reservoir by tickets on the device"""
    n, ptrs, idx, _, _ = karate
    deg = int(ptrs[1] - ptrs[0])
    assert deg == 16
    dev, ref = _device_slots(cabi, ptrs, idx, 0, 5, 0), _ref_slots(ptrs, idx, 0, 5, 0)
    assert all(len(set(r)) == 5 for r in dev[:500])                        # without replacement
    _compare_slots(dev, ref, deg, "uniform k=5 of 16")
    # the law is NOT uniform per slot (slot s keeps item s unless a later item hits it), so this is a sharp test:
    # the fraction of outcomes that keep item 0 in slot 0
    assert abs((dev[:, 0] == 0).mean() - (ref[:, 0] == 0).mean()) < 0.01 < (ref[:, 0] == 0).mean() - 1 / 16
    # k close to n, and k = 1
    _compare_slots(_device_slots(cabi, ptrs, idx, 0, 15, 0), _ref_slots(ptrs, idx, 0, 15, 0), deg, "uniform k=15 of 16")
    _compare_slots(_device_slots(cabi, ptrs, idx, 0, 1, 0), _ref_slots(ptrs, idx, 0, 1, 0), deg, "uniform k=1 of 16")


def test_uniform_with_replacement_law(cabi, karate):
    """replacement_sampling (sampling.rs:57-69)"""
    n, ptrs, idx, _, _ = karate
    dev, ref = _device_slots(cabi, ptrs, idx, 0, 5, 1), _ref_slots(ptrs, idx, 0, 5, 1)
    _compare_slots(dev, ref, 16, "with replacement k=5 of 16")
    # negative control: the statistic separates the two samplers
    assert_different_law(dev[:, 0], _ref_slots(ptrs, idx, 0, 5, 0)[:, 0], "replace vs no-replace, slot 0", n_bins=16)


def test_weighted_law(cabi, karate):
    """reservoir_sampling_weighted (sampling.rs:28-55), f64 weights"""
    n, ptrs, idx, _, _ = karate
    rs = np.random.default_rng(3)
    w = rs.uniform(0.2, 3.0, idx.size)
    w[int(ptrs[0]):int(ptrs[1])] = np.arange(1, 17, dtype=np.float64)      # vertex 0's column: weights 1..16
    dev, ref = _device_slots(cabi, ptrs, idx, 0, 5, 2, weights=w), _ref_slots(ptrs, idx, 0, 5, 2, weights=w)
    _compare_slots(dev, ref, 16, "weighted k=5 of 16")
    assert_different_law(dev[:, 4], _ref_slots(ptrs, idx, 0, 5, 0)[:, 4], "weighted vs uniform, slot 4", n_bins=16)


def test_random_walk_step_law(cabi, karate):
    """node2vec rejection step (random_walk.rs:52-66) with p != q: N walkers from vertex 0, three steps"""
    n, _, _, rptrs, ridx = karate
    g = cabi.graph_view(_t(rptrs), _t(ridx))
    start = torch.zeros(N, dtype=torch.int64, device=DEV)
    for p, q in ((0.5, 2.0), (4.0, 0.25)):
        dev = cabi.random_walk(g, start, 3, p, q, 99, 1).cpu().numpy()
        ref = orc.random_walk(rptrs, ridx, np.zeros(N, dtype=np.int64), 3, p, q, orc.rng_ref_child(orc.rng_ref()))
        for step in (1, 2, 3):
            assert_same_law(dev[:, step], ref[:, step], "walk p=%g q=%g step %d" % (p, q, step), n_bins=n)
        assert_same_law(dev[:, 1] * n + dev[:, 2], ref[:, 1] * n + ref[:, 2], "walk p=%g q=%g steps (1,2)" % (p, q),
                        n_bins=n * n)
        # return probability to the start after two steps depends on p: compare directly
        assert abs((dev[:, 2] == 0).mean() - (ref[:, 2] == 0).mean()) < 0.012
    dev_a = cabi.random_walk(g, start, 3, 0.5, 2.0, 99, 1).cpu().numpy()
    ref_b = orc.random_walk(rptrs, ridx, np.zeros(N, dtype=np.int64), 3, 4.0, 0.25, orc.rng_ref_child(orc.rng_ref()))
    assert_different_law(dev_a[:, 2], ref_b[:, 2], "walk (p,q) = (0.5,2) vs (4,0.25), step 2", n_bins=n)


def test_tempo_random_walk_step_law(cabi, karate):
    """time-windowed step (random_walk.rs:118-148): one-slot reservoir over the admissible neighbours, restart on none"""
    n, _, _, rptrs, ridx = karate
    rs = np.random.default_rng(8)
    node_ts = rs.integers(0, 40, n)
    node_ts[rs.random(n) < 0.2] = -1                                        # unknown timestamps are admissible (:131)
    edge_ts = rs.integers(0, 40, ridx.size)
    edge_ts[rs.random(ridx.size) < 0.3] = -1                                # falls back to the node's timestamp
    g = cabi.graph_view(_t(rptrs), _t(ridx))
    start = np.zeros(N, dtype=np.int64)
    start_ts = np.full(N, 10, dtype=np.int64)
    dev_w, dev_t = cabi.tempo_random_walk(g, _t(node_ts), _t(edge_ts), _t(start), _t(start_ts), 4, (-5, 15), 5, 2)
    ref_w, ref_t = orc.tempo_random_walk(rptrs, ridx, node_ts, edge_ts, start, start_ts, 4, (-5, 15),
                                         orc.rng_ref_child(orc.rng_ref()))
    dev_w, dev_t = dev_w.cpu().numpy(), dev_t.cpu().numpy()
    for step in (1, 2, 3):
        assert_same_law(dev_w[:, step] + 1, ref_w[:, step] + 1, "tempo walk step %d" % step, n_bins=n + 1)
        assert_same_law(dev_t[:, step] + 1, ref_t[:, step] + 1, "tempo walk timestamp %d" % step, n_bins=42)
    assert_same_law((dev_w[:, 1] + 1) * (n + 1) + dev_w[:, 2] + 1, (ref_w[:, 1] + 1) * (n + 1) + ref_w[:, 2] + 1,
                    "tempo walk steps (1,2)", n_bins=(n + 1) ** 2)


# ---------------------------------------------------------------- round 4: operators whose philox-mode is not the literal
# loop -- the device against the oracle's ref-mode (graphs and outcome codes shared with the CPU half of the check)
@pytest.fixture(scope="module")
def tg():
    import tch_geometric
    return tch_geometric


def test_hgt_sample_from_law(tg):
    """`BudgetDict::sample_from` (hgt_sampling.rs:104-135: weighted reservoir over score^2; blocked f64 sum on the device)
    through the operator surface, one call per outcome."""
    from test_oracle_law_philox_vs_ref import hgt_law_graph, hgt_outcomes
    n_calls = 8000
    nt, et, P, I = hgt_law_graph()
    Pc = {k: _t(v) for k, v in P.items()}
    Ic = {k: _t(v) for k, v in I.items()}
    inputs = {"a": torch.zeros(1, dtype=torch.int64, device=DEV)}
    tg.seed(77)
    dev = np.empty(n_calls, dtype=np.int64)
    got = []
    for c in range(n_calls):                                                # every call draws with the next call id
        s, _, _, _, _ = tg.hgt_sampling(nt, et, Pc, Ic, None, inputs, None, {"a": [2], "b": [2]}, 1)
        got.append(s["b"])
    got = torch.stack(got).cpu().numpy()
    dev = got[:, 0] * 6 + got[:, 1]
    parent = orc.rng_ref()
    ref = hgt_outcomes(lambda c: orc.rng_ref_child(parent), n_calls)
    assert_same_law(dev, ref, "hgt sample_from ordered pair", n_bins=36)
    assert_same_law(dev // 6, ref // 6, "hgt sample_from slot 0", n_bins=6)
    assert_same_law(dev % 6, ref % 6, "hgt sample_from slot 1", n_bins=6)
    flat = hgt_outcomes(lambda c: orc.rng_ref_child(parent), n_calls, duplicates=False)
    assert_different_law(dev, flat, "hgt scores 9:4:1:1:1:1 vs equal scores", n_bins=36)


def test_biased_walk_first_step_law(cabi):
    """biased_tempo_random_walk (random_walk.rs:258-271): one-slot weighted reservoir in f32 over the bias weights"""
    from test_oracle_law_philox_vs_ref import biased_law_graph
    ptrs, idx, node_ts, edge_ts = biased_law_graph()
    g = cabi.graph_view(_t(ptrs), _t(idx))
    start, start_ts = np.zeros(N, dtype=np.int64), np.full(N, 5, dtype=np.int64)
    first = {}
    for bias in ("uniform", "linear", "exponential"):
        w, _, status = cabi.biased_tempo_random_walk(g, _t(node_ts), _t(edge_ts), _t(start), _t(start_ts), 2, bias, True, 1,
                                                     61, 3, max_degree=12)
        assert int(status.item()) == 0
        ref, _ = orc.biased_tempo_random_walk(ptrs, idx, node_ts, edge_ts, start, start_ts, 2, bias, True, 1,
                                              orc.rng_ref_child(orc.rng_ref()))
        first[bias] = w[:, 1].cpu().numpy()
        assert_same_law(first[bias], ref[:, 1], "biased walk (%s) step 1" % bias, n_bins=13)
    assert_different_law(first["uniform"], first["linear"], "uniform vs linear bias", n_bins=13)
    assert_different_law(first["uniform"], first["exponential"], "uniform vs exponential bias", n_bins=13)


def test_negative_sampling_law(tg, karate):
    """negative_sample_neighbors_homogenous (negative_sampling.rs:31-45): accepted nodes of one input vertex"""
    n, _, _, rptrs, ridx = karate
    inputs = np.zeros(N // 4, dtype=np.int64)
    tg.seed(88)
    s, r, c, _ = tg.negative_sample_neighbors_homogenous(_t(rptrs), _t(ridx), (n, n), _t(inputs), 4, 8)
    dev = s[c].cpu().numpy()
    rs, rr, rc, _ = orc.neg_homo(rptrs, ridx, (n, n), inputs, 4, 8, orc.rng_ref_child(orc.rng_ref()))
    ref = rs[rc]
    assert len(dev) > 0.95 * N and len(ref) > 0.95 * N
    assert_same_law(dev, ref, "negatives of vertex 0", n_bins=n)
    inputs[:] = 33
    os_, or_, oc, _ = orc.neg_homo(rptrs, ridx, (n, n), inputs, 4, 8, orc.rng_ref_child(orc.rng_ref()))
    assert_different_law(dev, os_[oc], "negatives of vertex 0 vs vertex 33", n_bins=n)


def test_budget_sampling_law(tg, karate):
    """budget_sampling's `Budget::sample` (budget_sampling.rs:137-151): reservoir over the node's candidate list"""
    n, ptrs, idx, _, _ = karate
    nt, et = ["a"], [("a", "r", "a")]
    m = N // 2
    inputs = np.zeros(m, dtype=np.int64)
    tg.seed(99)
    _, _, _, cols, eidx = tg.budget_sampling(nt, et, {"a__r__a": _t(ptrs)}, {"a__r__a": _t(idx)}, None, {"a": _t(inputs)},
                                             None, {"a": [5]}, 1, None, False, False)
    dev = eidx["a__r__a"].cpu().numpy()
    assert len(dev) == m * 5 and np.array_equal(cols["a__r__a"].cpu().numpy(), np.repeat(np.arange(m), 5))
    dev = dev.reshape(m, 5)
    o = orc.budget(nt, et, {"a__r__a": ptrs}, {"a__r__a": idx}, None, {"a": inputs}, None, {"a": [5]}, 1,
                   orc.rng_ref_child(orc.rng_ref()))
    ref = o[4]["a__r__a"].reshape(m, 5)
    for s in range(5):
        assert_same_law(dev[:, s], ref[:, s], "budget slot %d" % s, n_bins=16)
    assert_same_law(dev[:, 0] * 16 + dev[:, 3], ref[:, 0] * 16 + ref[:, 3], "budget slots (0,3)", n_bins=256)
    o3 = orc.budget(nt, et, {"a__r__a": ptrs}, {"a__r__a": idx}, None, {"a": inputs}, None, {"a": [3]}, 1,
                    orc.rng_ref_child(orc.rng_ref()))
    assert_different_law(dev[:, 0], o3[4]["a__r__a"].reshape(m, 3)[:, 0], "k = 5 vs k = 3, slot 0", n_bins=16)
