"""GPU parity: tg_ns_homo_batched (HIP, through the C ABI) == CPU oracle philox-mode, bit for bit.

Also the synthetic-input generators and device ingest against their CPU twins."""
import numpy as np
import pytest
import torch

import orc
from helpers import load_karate, validate_neighbor_samples

pytestmark = pytest.mark.gpu
SEED = 0x5EED


@pytest.fixture(scope="module")
def cabi():
    from tch_geometric import _cabi
    return _cabi


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


def _gpu_csc(cabi, dev, row, col, n):
    r, c = torch.from_numpy(row).to(dev), torch.from_numpy(col).to(dev)
    return cabi.coo_to_csx(r, c, n, n, True)


def _check_batches(cabi, dev, ptrs, idx, seeds, fanout, sampler=0, call_id=100):
    """runs the GPU on all batches at once and the oracle batch by batch"""
    ptrs_d, idx_d = torch.from_numpy(ptrs).to(dev), torch.from_numpy(idx).to(dev)
    g = cabi.graph_view(ptrs_d, idx_d)
    seeds_d = torch.from_numpy(np.ascontiguousarray(seeds)).to(dev)
    out = cabi.NsBatchedOut(seeds.shape[0], seeds.shape[1], fanout, dev)
    cabi.ns_homo_batched(g, seeds_d, fanout, SEED, call_id, out, sampler=sampler)
    torch.cuda.synchronize()
    counts = out.counts.cpu()
    total_edges = 0
    for b in range(seeds.shape[0]):
        gs, gr, gc, ge, glo = out.batch(b, counts)
        os_, or_, oc, oe, olo = orc.ns_homo(ptrs, idx, seeds[b], fanout, orc.rng_philox(SEED, call_id + b),
                                            sampler=sampler)
        assert glo == olo, (b, glo, olo)
        assert np.array_equal(gs.cpu().numpy(), os_), b
        assert np.array_equal(gr.cpu().numpy(), or_), b
        assert np.array_equal(gc.cpu().numpy(), oc), b
        assert np.array_equal(ge.cpu().numpy(), oe), b
        total_edges += len(oe)
    return total_edges


def test_generators_match_cpu_twins(cabi, dev):
    row, col = cabi.rmat_edges(12, 4096 * 16, 77, dev)
    orow, ocol = orc.rmat_edges(12, 4096 * 16, 77)
    assert np.array_equal(row.cpu().numpy(), orow) and np.array_equal(col.cpu().numpy(), ocol)
    s = cabi.seed_batches(3, 5, 7, 33, 4096, dev)
    assert np.array_equal(s.cpu().numpy(), orc.seed_batches(3, 5, 7, 33, 4096))


def test_device_ingest_matches_oracle(cabi, dev):
    orow, ocol = orc.rmat_edges(10, 1024 * 16, 5)
    for csc in (True, False):
        p, i, perm = cabi.coo_to_csx(torch.from_numpy(orow).to(dev), torch.from_numpy(ocol).to(dev), 1024, 1024, csc)
        op, oi, operm = orc.to_csx(np.stack([orow, ocol]), 1024, csc)
        assert np.array_equal(p.cpu().numpy(), op) and np.array_equal(i.cpu().numpy(), oi)
        assert np.array_equal(perm.cpu().numpy(), operm)
    # storage.rs:152-163 exact vector through the device ind2ptr
    ind = torch.tensor([3, 3, 3, 4, 4, 7, 7, 8, 8], device=dev)
    assert cabi.ind2ptr(ind, 10).cpu().tolist() == [0, 0, 0, 0, 3, 5, 5, 5, 7, 9, 9]
    assert cabi.ind2ptr(torch.zeros(0, dtype=torch.int64, device=dev), 3).cpu().tolist() == [0, 0, 0, 0]


@pytest.mark.parametrize("sampler", [0, 1])
@pytest.mark.parametrize("fanout", [[5, 5], [4, 3], [1], [3, 2, 2], [16, 16]])
def test_karate_matches_oracle(cabi, dev, sampler, fanout):
    """BASELINE cfg1 (karate, [5,5], batch 4) and the reference unit-test configs."""
    ei, n = load_karate()
    ptrs, idx, _ = orc.to_csc(ei, n)
    seeds = np.array([[0, 1, 4, 5], [33, 33, 0, 2], [7, 8, 9, 10]], dtype=np.int64)
    _check_batches(cabi, dev, ptrs, idx, seeds, fanout, sampler=sampler)


def test_karate_invariants_of_the_reference_tests(cabi, dev):
    ei, n = load_karate()
    ptrs, idx, _ = orc.to_csc(ei, n)
    g = cabi.graph_view(torch.from_numpy(ptrs).to(dev), torch.from_numpy(idx).to(dev))
    seeds = torch.tensor([[0, 1, 4, 5]], device=dev)
    out = cabi.NsBatchedOut(1, 4, [4, 3], dev)
    cabi.ns_homo_batched(g, seeds, [4, 3], 1, 2, out, sampler=1)
    s, r, c, e, lo = out.batch(0)
    s, r, c = s.cpu().numpy(), r.cpu().numpy(), c.cpu().numpy()
    validate_neighbor_samples(ptrs, idx, r, c, s, s, lo, [4, 3])     # neighbor_sampling.rs:437-464


@pytest.mark.parametrize("sampler", [0, 1])
def test_rmat14_many_batches(cabi, dev, sampler):
    scale, n = 14, 1 << 14
    orow, ocol = orc.rmat_edges(scale, n * 16, 0x5EED0000 + scale)
    ptrs, idx, _ = orc.to_csc(np.stack([orow, ocol]), n)
    seeds = orc.seed_batches(0xBA7C4, 0, 24, 256, n)
    edges = _check_batches(cabi, dev, ptrs, idx, seeds, [15, 10], sampler=sampler)
    assert edges > 24 * 256 * 10


def test_rmat16_bench_shape_and_narrow_workgroups(cabi, dev):
    scale, n = 16, 1 << 16
    orow, ocol = orc.rmat_edges(scale, n * 16, 0x5EED0000 + scale)
    ptrs, idx, _ = orc.to_csc(np.stack([orow, ocol]), n)
    seeds = orc.seed_batches(0xBA7C4, 0, 3, 1024, n)            # batch 1024, fanout [15,10]: the bench shape
    _check_batches(cabi, dev, ptrs, idx, seeds, [15, 10])
    seeds = orc.seed_batches(0xBA7C4, 9, 600, 8, n)             # >= 512 batches: the 256-thread launch shape
    _check_batches(cabi, dev, ptrs, idx, seeds, [15, 10])
    seeds = orc.seed_batches(0xBA7C4, 2, 2, 64, n)              # KMAX = 32 instantiation
    _check_batches(cabi, dev, ptrs, idx, seeds, [32, 20])
    _check_batches(cabi, dev, ptrs, idx, seeds, [17, 2], sampler=1)


def test_fanout_above_the_register_sampler(cabi, dev):
    """fan-outs > TG_MAX_FANOUT take the LDS-resident ticket form: same results"""
    n = 1 << 13
    orow, ocol = orc.rmat_edges(13, n * 16, 17)
    ptrs, idx, _ = orc.to_csc(np.stack([orow, ocol]), n)
    seeds = orc.seed_batches(3, 0, 3, 40, n)
    _check_batches(cabi, dev, ptrs, idx, seeds, [40, 3])
    _check_batches(cabi, dev, ptrs, idx, seeds, [100])
    _check_batches(cabi, dev, ptrs, idx, seeds, [33, 2], sampler=1)


def test_edge_cases(cabi, dev):
    ptrs = np.array([0, 0, 2, 2, 5], dtype=np.int64)   # 0,2 isolated; 1 <- {0,2}; 3 <- {0,1,3}
    idx = np.array([0, 2, 0, 1, 3], dtype=np.int64)
    _check_batches(cabi, dev, ptrs, idx, np.array([[0, 1, 2, 1, 3]]), [3, 3])
    _check_batches(cabi, dev, ptrs, idx, np.array([[0, 2]]), [2, 2])                 # nothing to sample
    _check_batches(cabi, dev, ptrs, idx, np.zeros((2, 0), dtype=np.int64), [2, 2])   # empty batches
    _check_batches(cabi, dev, ptrs, idx, np.array([[3, 1]]), [])                     # zero hops
    _check_batches(cabi, dev, ptrs, idx, np.array([[3, 3, 3]]), [2, 2, 2, 2], sampler=1)


def test_large_frontier_spans_several_scan_rounds(cabi, dev):
    """frontier > 1024 chunks * 64 slots forces the multi-round offset scan"""
    n = 1 << 12
    orow, ocol = orc.rmat_edges(12, n * 16, 3)
    ptrs, idx, _ = orc.to_csc(np.stack([orow, ocol]), n)
    seeds = orc.seed_batches(1, 0, 1, 70000, n)
    _check_batches(cabi, dev, ptrs, idx, seeds, [2, 2])


def test_errors_are_reported(cabi, dev):
    ptrs = torch.tensor([0, 1], device=dev)
    idx = torch.tensor([0], device=dev)
    g = cabi.graph_view(ptrs, idx)
    out = cabi.NsBatchedOut(1, 1, [2], dev)
    with pytest.raises(cabi.TchGeoError):
        cabi.ns_homo_batched(g, torch.tensor([[0]], device=dev), [300], 0, 0, out)    # above any supported fan-out
    with pytest.raises(cabi.TchGeoError):
        cabi.ns_homo_batched(g, torch.tensor([[0]], device=dev), [3, 3], 0, 0, out)   # slabs too small
