"""biased_tempo_random_walk on the GPU == the oracle's philox-mode, bit for bit, through the C ABI and the operator
surface: every bias, both directions, LDS and global-slab sorts, denormal softmax weights, panics, restarts."""
import numpy as np
import pytest
import torch

import orc
from helpers import load_karate

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def cabi():
    from tch_geometric import _cabi
    return _cabi


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.int64))).to(DEV)


def _run(cabi, ptrs, idx, nts, ets, start, sts, L, bias, forward, R, seed, call):
    g = cabi.graph_view(_t(ptrs), _t(idx))
    max_deg = int(np.diff(ptrs).max()) if len(ptrs) > 1 else 0
    w, t, status = cabi.biased_tempo_random_walk(g, _t(nts), _t(ets), _t(start), _t(sts), L, bias, forward, R, seed, call,
                                                 max_degree=max_deg)
    torch.cuda.synchronize()
    return w.cpu().numpy(), t.cpu().numpy(), int(status.item())


@pytest.mark.parametrize("bias", ["uniform", "linear", "exponential"])
@pytest.mark.parametrize("forward", [True, False])
def test_karate_equals_oracle(cabi, bias, forward):
    ei, n = load_karate()
    ptrs, idx, _ = orc.to_csr(ei, n)
    rs = np.random.default_rng(1)
    nts, ets = rs.integers(-1, 5, n), rs.integers(-1, 5, len(idx))
    start = rs.integers(0, n, 300)
    sts = rs.integers(-1, 4, 300)
    for L, R in ((10, 10), (1, 3), (2, 1), (70, 4)):
        w, t, st = _run(cabi, ptrs, idx, nts, ets, start, sts, L, bias, forward, R, 21, 3)
        ow, ot = orc.biased_tempo_random_walk(ptrs, idx, nts, ets, start, sts, L, bias, forward, R, orc.rng_philox(21, 3))
        assert st == 0 and np.array_equal(w, ow) and np.array_equal(t, ot), (bias, forward, L, R)


@pytest.mark.parametrize("case", range(6))
def test_random_graphs_equal_oracle(cabi, case):
    rs = np.random.default_rng(100 + case)
    n = int(rs.integers(3, 800))
    e = int(rs.integers(1, 30 * n))
    row, col = rs.integers(0, n, e), rs.integers(0, n, e)
    hub = rs.integers(0, n)
    row[rs.integers(0, e, e // 3)] = hub                    # one heavy row (above 64 candidates, often above 1024)
    ptrs, idx, _ = orc.to_csr(np.stack([row, col]), n)
    span = int(rs.choice([3, 40, 400]))
    nts, ets = rs.integers(-1, span, n), rs.integers(-1, span, len(idx))
    start = rs.integers(0, n, 200)
    start[:20] = hub
    sts = rs.integers(-1, span // 2 + 1, 200)
    for bias in ("uniform", "linear", "exponential"):
        fwd = bool(rs.integers(0, 2))
        L, R = int(rs.integers(2, 12)), int(rs.integers(1, 4))
        try:
            ow, ot = orc.biased_tempo_random_walk(ptrs, idx, nts, ets, start, sts, L, bias, fwd, R, orc.rng_philox(case, 9))
        except RuntimeError:
            w, t, st = _run(cabi, ptrs, idx, nts, ets, start, sts, L, bias, fwd, R, case, 9)
            assert st & 2
            continue
        w, t, st = _run(cabi, ptrs, idx, nts, ets, start, sts, L, bias, fwd, R, case, 9)
        assert st == 0 and np.array_equal(w, ow) and np.array_equal(t, ot), (case, bias)


def _star(times):
    k = len(times)
    return np.array([0, k] + [k] * k), np.arange(1, k + 1), np.zeros(k + 1, dtype=np.int64), np.asarray(times)


@pytest.mark.parametrize("k", [2, 63, 64, 65, 1024, 1025, 5000, 70000])
def test_star_all_sizes(cabi, k):
    rs = np.random.default_rng(k)
    ptrs, idx, nts, ets = _star(rs.integers(0, 50, k))
    start, sts = np.zeros(64, dtype=np.int64), rs.integers(0, 10, 64)
    for bias in ("uniform", "linear", "exponential"):
        w, t, st = _run(cabi, ptrs, idx, nts, ets, start, sts, 2, bias, True, 1, 5, 0)
        ow, ot = orc.biased_tempo_random_walk(ptrs, idx, nts, ets, start, sts, 2, bias, True, 1, orc.rng_philox(5, 0))
        assert st == 0 and np.array_equal(w, ow) and np.array_equal(t, ot), (k, bias)


def test_denormal_softmax_weights_and_panic(cabi):
    # exponents 88..103 give denormal float32 weights; they must survive the division on both sides
    ptrs, idx, nts, ets = _star([100, 12, 5, 0, 3, 100, 1, 99])
    start, sts = np.zeros(512, dtype=np.int64), np.zeros(512, dtype=np.int64)
    w, t, st = _run(cabi, ptrs, idx, nts, ets, start, sts, 2, "exponential", True, 1, 8, 0)
    ow, ot = orc.biased_tempo_random_walk(ptrs, idx, nts, ets, start, sts, 2, "exponential", True, 1, orc.rng_philox(8, 0))
    assert st == 0 and np.array_equal(w, ow) and np.array_equal(t, ot)
    ptrs, idx, nts, ets = _star([0, 0, 1000])
    _, _, st = _run(cabi, ptrs, idx, nts, ets, [0], [0], 2, "exponential", False, 1, 1, 0)
    assert st & 2


def test_restart_quirk_and_surface(cabi):
    import tch_geometric as tg
    ptrs, idx = np.array([0, 2, 3, 3, 3, 3]), np.array([1, 2, 4])
    ets, nts = np.array([5, 7, 6]), np.zeros(5, dtype=np.int64)
    stale = False
    for call in range(40):
        w, t, st = _run(cabi, ptrs, idx, nts, ets, [0], [0], 4, "uniform", True, 2, 2, call)
        ow, ot = orc.biased_tempo_random_walk(ptrs, idx, nts, ets, np.array([0]), np.array([0]), 4, "uniform", True, 2,
                                              orc.rng_philox(2, call))
        assert np.array_equal(w, ow) and np.array_equal(t, ot)
        stale |= bool(w[0, 2] == -1 and t[0, 2] == 6)
    assert stale
    w, t, _ = _run(cabi, ptrs, idx, nts, ets, [0], [0], 4, "uniform", True, 0, 2, 0)
    assert np.all(w == -1) and np.all(t == -1)
    # operator surface: same argument order as python.rs:645-656, seeded call counter
    ei, n = load_karate()
    p, i, _ = orc.to_csr(ei, n)
    rs = np.random.default_rng(4)
    nts, ets = rs.integers(-1, 5, n), rs.integers(-1, 5, len(i))
    start, sts = np.array([0, 1, 2, 3]), np.array([0, -1, 2, 3])
    for where in ("cuda", "cpu"):
        tg.seed(77)
        tg.random_walk(_t(p), _t(i), _t(start), 2, 1.0, 1.0)   # advances the call counter to 1
        a, b = tg.biased_tempo_random_walk(_t(p).to(where), _t(i).to(where), _t(nts).to(where), _t(ets).to(where),
                                           _t(start).to(where), _t(sts).to(where), 10, "exponential", True, 10)
        oa, ob = orc.biased_tempo_random_walk(p, i, nts, ets, start, sts, 10, "exponential", True, 10, orc.rng_philox(77, 1))
        assert a.device.type == where and np.array_equal(a.cpu().numpy(), oa) and np.array_equal(b.cpu().numpy(), ob)
    with pytest.raises(RuntimeError, match="empty range"):
        sp, si, sn, se = _star([0, 0, 1000])
        tg.biased_tempo_random_walk(_t(sp), _t(si), _t(sn), _t(se), _t([0]), _t([0]), 2, "exponential", False, 1)
