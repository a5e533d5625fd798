"""Ad-hoc widening of the randomised parity sweeps: the test bodies of tests/test_gpu_random_sweep_hetero.py over
hundreds of further random cases (hetero sampling in its 4 variants, HGT, budget, homogeneous sampling, walks, negatives,
the window-ordered launch, the partitioned sampler) against the oracle.  Run on a GPU box
from the repo root: `python tools/stress_parity.py`."""
import os, sys
ROOT = os.getcwd()
for p in ("oracle", "tests", "tch-geometric_amd"):
    sys.path.insert(0, os.path.join(ROOT, p))
import tch_geometric as tg
import test_gpu_random_sweep_hetero as T
bad = 0
MULT = int(os.environ.get("STRESS_MULT", "1"))   # widen every range by this factor
for case in range(1000, 1000 + 400 * MULT):
    try:
        T.test_hetero_neighbor_sampling_random(tg, case)
    except Exception as e:  # noqa
        bad += 1
        print("FAIL hetero case", case, type(e).__name__, str(e)[:200], flush=True)
        if bad > 5:
            break
for case in range(1000, 1000 + 120 * MULT):
    try:
        T.test_hgt_and_budget_random(tg, case)
    except Exception as e:  # noqa
        bad += 1
        print("FAIL hgt/budget case", case, type(e).__name__, str(e)[:200], flush=True)
        if bad > 10:
            break
import test_gpu_random_sweep as S  # noqa: E402
import test_gpu_random_sweep_windowed as W  # noqa: E402
from tch_geometric import _cabi  # noqa: E402
for name, fn, lo, hi in (("neighbor sampling", S.test_neighbor_sampling_random_cases, 1000, 1300),
                         ("walks / negatives", S.test_walks_and_negatives_random_cases, 1000, 1100),
                         ("windowed launch", W.test_windowed_launch_random_cases, 1000, 1150),
                         ("partitioned", W.test_partitioned_random_cases, 1000, 1100)):
    for case in range(lo, lo + (hi - lo) * MULT):
        try:
            fn(_cabi, case)
        except Exception as e:  # noqa
            bad += 1
            print("FAIL", name, "case", case, type(e).__name__, str(e)[:200], flush=True)
            if bad > 20:
                break
    print(name, "done", flush=True)
# ---- the operator surface's homogeneous call in every sampler / filter variant (device-side bookkeeping, one read-back)
import numpy as np  # noqa: E402
import orc  # noqa: E402
import torch  # noqa: E402

for case in range(400 * MULT):
    rs = np.random.default_rng(20_000 + case)
    n = int(rs.integers(3, 3000))
    e = int(rs.integers(0, 40 * n))
    ei = S.random_graph(rs, n, e) if e >= 8 else np.stack([rs.integers(0, n, e), rs.integers(0, n, e)]).astype(np.int64)
    ptrs, idx, _ = orc.to_csc(ei, n)
    hops = int(rs.integers(1, 4))
    fan = [int(rs.integers(1, 41)) for _ in range(hops)]
    seeds = rs.integers(0, n, int(rs.integers(1, 200)))
    variant = case % 5
    sampler, flt, kw = None, None, {}
    cu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    if variant == 1:
        sampler, kw = tg.UniformEdgeSampler(True), dict(sampler=orc.SAMPLER_UNIFORM_REPL)
    if variant in (2, 4):
        w = rs.uniform(0.05, 4.0, len(idx))
        sampler, kw = tg.WeightedEdgeSampler(cu(w)), dict(sampler=orc.SAMPLER_WEIGHTED, weights=w)
    if variant in (3, 4):
        ts, st = rs.integers(0, 30, len(idx)), rs.integers(0, 30, len(seeds))
        mode, fwd = int(rs.integers(0, 3)), bool(rs.integers(0, 2))
        flt = (tg.TemporalEdgeFilter((0, 12), cu(ts), fwd, mode), cu(st))
        kw.update(filter_mode=mode, forward=fwd, window=(0, 12), timestamps=ts, inputs_state=st)
    try:
        tg.seed(case)
        got = tg.neighbor_sampling_homogenous(cu(ptrs), cu(idx), cu(seeds), fan, sampler, flt)
        o = orc.ns_homo(ptrs, idx, seeds, fan, orc.rng_philox(case, 0), **kw)
        assert [tuple(x) for x in got[4]] == o[4]
        for u, v in zip(got[:4], o[:4]):
            assert np.array_equal(u.cpu().numpy(), v)
    except Exception as ex:  # noqa
        bad += 1
        print("FAIL surface homo case", case, variant, type(ex).__name__, str(ex)[:200], flush=True)
        if bad > 30:
            break
print("surface homogeneous done", flush=True)
# ---- many relations under filters / weights: several segmented rounds per hop (<= 16 entries, <= 8 segments each)
from helpers import rel_key  # noqa: E402

for case in range(90 * MULT):
    rs = np.random.default_rng(30_000 + case)
    T_ = int(rs.integers(1, 6))
    node_types = ["t%d" % i for i in range(T_)]
    counts = {t: int(rs.integers(2, 200)) for t in node_types}
    R_ = int(rs.integers(9, 45))
    edge_types, P, I = [], {}, {}
    for r in range(R_):
        s_, d_ = node_types[int(rs.integers(0, T_))], node_types[int(rs.integers(0, T_))]
        et = (s_, "r%d" % r, d_)
        e = int(rs.integers(0, 600)) if rs.random() > 0.15 else 0
        ei = np.stack([rs.integers(0, counts[s_], e), rs.integers(0, counts[d_], e)]).astype(np.int64).reshape(2, e)
        edge_types.append(et)
        P[rel_key(et)], I[rel_key(et)], _ = orc.to_csc(ei, (counts[s_], counts[d_]))
    rels = [rel_key(et) for et in edge_types]
    hops = int(rs.integers(1, 4))
    nn = {k: [int(rs.integers(1, 7)) for _ in range(hops)] for k in rels}
    inputs = {t: rs.integers(0, counts[t], int(rs.integers(1, 20))) for t in node_types if rs.random() > 0.4}
    if not inputs:
        inputs = {node_types[0]: rs.integers(0, counts[node_types[0]], 5)}
    cud = lambda d: {k: cu(v) for k, v in d.items()}
    sampler, flt, kw = None, None, {}
    if case % 3 in (1, 2):
        Wt = {k: rs.uniform(0.1, 4.0, len(I[k])) for k in rels}
        sampler, kw = tg.WeightedEdgeSampler(cud(Wt)), dict(sampler=orc.SAMPLER_WEIGHTED, weights=Wt)
    if case % 3 in (0, 2):
        TS = {k: rs.integers(0, 12, len(I[k])) for k in rels}
        ST = {t: rs.integers(0, 12, len(v)) for t, v in inputs.items()}
        mode, fwd = int(rs.integers(0, 3)), bool(rs.integers(0, 2))
        flt = (tg.TemporalEdgeFilter((0, 6), cud(TS), fwd, mode), cud(ST))
        kw.update(filter_mode=mode, forward=fwd, window=(0, 6), timestamps=TS, inputs_state=ST)
    try:
        tg.seed(case)
        s2, r2, c2, e2, lo2 = tg.neighbor_sampling_heterogenous(node_types, edge_types, cud(P), cud(I), cud(inputs), nn, hops,
                                                                sampler, flt)
        o = orc.ns_hetero(node_types, edge_types, P, I, inputs, nn, hops, orc.rng_philox(case, 0), **kw)
        for t in node_types:
            assert np.array_equal(s2[t].cpu().numpy(), o[0][t]), t
        for k in rels:
            assert np.array_equal(r2[k].cpu().numpy(), o[1][k]) and np.array_equal(c2[k].cpu().numpy(), o[2][k]), k
            assert np.array_equal(e2[k].cpu().numpy(), o[3][k]) and [tuple(x) for x in lo2[k]] == o[4][k], k
    except Exception as ex:  # noqa
        bad += 1
        print("FAIL many relations case", case, type(ex).__name__, str(ex)[:200], flush=True)
        if bad > 40:
            break
print("many relations done", flush=True)
print("done, failures:", bad)
