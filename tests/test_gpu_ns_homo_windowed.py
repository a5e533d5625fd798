"""The window-ordered form of the many-batch launch (tg_ns_homo_batched_ws, csrc/ns_homo_win.hip) == the fused
per-batch kernel == the oracle, bit for bit: output positions are fixed by per-batch prefix sums
(neighbor_sampling.rs:212-217), so the order in which a hop's gathers are issued cannot show in any output word."""
import numpy as np
import pytest
import torch

import orc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
WINDOWED, FUSED, WINDOWED_WIDE = 1, 2, 3


@pytest.fixture(scope="module")
def cabi():
    from tch_geometric import _cabi
    return _cabi


def _rmat(cabi, scale, shadows):
    dev = torch.device(DEV)
    n = 1 << scale
    row, col = cabi.rmat_edges(scale, n * 16, 0x5EED0000 + scale, dev)
    ptrs, idx, _ = cabi.coo_to_csx(row, col, n, n, True)
    i32 = idx.to(torch.int32) if shadows else None
    p32 = ptrs.to(torch.int32) if shadows else None
    return n, ptrs, idx, cabi.graph_view(ptrs, idx, indices32=i32, ptrs32=p32)


def _both(cabi, g, seeds, fan, seed, call, sampler=0, form=WINDOWED, **kw):
    nb, B = seeds.shape
    a, b = cabi.NsBatchedOut(nb, B, fan, seeds.device), cabi.NsBatchedOut(nb, B, fan, seeds.device)
    for o in (a, b):                                    # poison: a slot the windowed form fails to write shows up
        for t in (o.samples, o.rows, o.cols, o.edge_index):
            t.fill_(-7)
    ws = cabi.ns_homo_workspace(nb, B, fan, seeds.device)
    cabi.ns_homo_batched(g, seeds, fan, seed, call, a, sampler=sampler, ws=ws, form=form, **kw)
    cabi.ns_homo_batched(g, seeds, fan, seed, call, b, sampler=sampler, ws=ws, form=FUSED, **kw)
    torch.cuda.synchronize()
    return a, b


def _assert_equal(a, b):
    assert torch.equal(a.counts, b.counts) and torch.equal(a.layer_offsets, b.layer_offsets)
    c = a.counts.cpu()
    for j in range(a.n_batches):
        x, y = a.batch(j, c), b.batch(j, c)
        assert x[4] == y[4]
        for u, v in zip(x[:4], y[:4]):
            assert torch.equal(u, v), j


@pytest.mark.parametrize("shadows", [False, True])
@pytest.mark.parametrize("sampler", [0, 1])
def test_windowed_equals_fused_and_oracle(cabi, shadows, sampler):
    n, ptrs, idx, g = _rmat(cabi, 14, shadows)
    fan = [15, 10]
    seeds = cabi.seed_batches(0xBA7C4, 3, 37, 200, n, torch.device(DEV))
    a, b = _both(cabi, g, seeds, fan, 5, 3, sampler=sampler)
    _assert_equal(a, b)
    hp, hi, hs = ptrs.cpu().numpy(), idx.cpu().numpy(), seeds.cpu().numpy()
    for j in (0, 17, 36):
        o = orc.ns_homo(hp, hi, hs[j], fan, orc.rng_philox(5, 3 + j), sampler=sampler)
        x = a.batch(j)
        assert x[4] == o[4]
        for u, v in zip(x[:4], o[:4]):
            assert np.array_equal(u.cpu().numpy(), v)


@pytest.mark.parametrize("form", [WINDOWED, WINDOWED_WIDE])
@pytest.mark.parametrize("fan", [[1], [3, 3, 3], [32, 2], [20], [2, 2, 2, 2, 2]])
def test_windowed_other_fanouts(cabi, fan, form):
    n, ptrs, idx, g = _rmat(cabi, 12, True)
    seeds = cabi.seed_batches(0xBA7C4, 11, 9, 70, n, torch.device(DEV))
    _assert_equal(*_both(cabi, g, seeds, fan, 1, 11, form=form))


def test_windowed_wide_items_equal_oracle(cabi):
    n, ptrs, idx, g = _rmat(cabi, 14, False)
    seeds = cabi.seed_batches(3, 0, 5, 300, n, torch.device(DEV))
    a, b = _both(cabi, g, seeds, [15, 10], 8, 2, form=WINDOWED_WIDE)
    _assert_equal(a, b)
    o = orc.ns_homo(ptrs.cpu().numpy(), idx.cpu().numpy(), seeds[4].cpu().numpy(), [15, 10], orc.rng_philox(8, 6))
    x = a.batch(4)
    assert x[4] == o[4] and all(np.array_equal(u.cpu().numpy(), v) for u, v in zip(x[:4], o[:4]))


def test_windowed_edge_cases(cabi):
    """isolated vertices only, duplicate seeds, one batch, id_base / tag of a relation-hop"""
    dev = torch.device(DEV)
    n, ptrs, idx, g = _rmat(cabi, 12, False)
    deg = ptrs[1:] - ptrs[:-1]
    iso = torch.nonzero(deg == 0).reshape(-1)[:64].contiguous()
    assert iso.numel() == 64
    a, b = _both(cabi, g, iso.reshape(2, 32).contiguous(), [4, 4], 0, 0)
    _assert_equal(a, b)
    assert int(a.counts[:, 1].sum()) == 0
    hub = torch.argmax(deg).reshape(1, 1).repeat(3, 16).contiguous()         # the same hub 16 times per batch
    _assert_equal(*_both(cabi, g, hub, [15, 10], 2, 9))
    one = cabi.seed_batches(1, 0, 1, 1024, n, dev)
    _assert_equal(*_both(cabi, g, one, [15, 10], 2, 9))
    _assert_equal(*_both(cabi, g, one, [5], 2, 9, rng_tag=2 | (3 << 8), id_base=1000))


def test_windowed_many_rounds(cabi):
    """a frontier of more than 65 536 slots per batch (several scan rounds per hop)"""
    n, ptrs, idx, g = _rmat(cabi, 13, True)
    seeds = cabi.seed_batches(7, 0, 2, 70000, n, torch.device(DEV))
    _assert_equal(*_both(cabi, g, seeds, [2, 2], 3, 4))


def test_windowed_auto_form_falls_back(cabi):
    """form = auto on a small launch runs the fused kernel; filtered / weighted launches ignore the workspace"""
    dev = torch.device(DEV)
    n, ptrs, idx, g = _rmat(cabi, 12, False)
    seeds = cabi.seed_batches(1, 0, 4, 64, n, dev)
    ws = cabi.ns_homo_workspace(4, 64, [5, 5], dev)
    a, b = cabi.NsBatchedOut(4, 64, [5, 5], dev), cabi.NsBatchedOut(4, 64, [5, 5], dev)
    cabi.ns_homo_batched(g, seeds, [5, 5], 0, 0, a, ws=ws)
    cabi.ns_homo_batched(g, seeds, [5, 5], 0, 0, b)
    torch.cuda.synchronize()
    _assert_equal(a, b)
