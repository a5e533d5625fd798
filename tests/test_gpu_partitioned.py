"""Range-partitioned neighbor sampling with the real HIP kernels: (a) one rank (the remote-frontier kernel mode
+ reassembly), (b) two ranks sharing the one GPU over gloo (host-staged exchange).  Both must equal the
replicated-graph launch (tg_ns_homo_batched) bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SEED, FANOUT, B = 0xD157, [15, 10], 128


def _graph(dev):
    from tch_geometric import _cabi
    n = 1 << 14
    row, col = _cabi.rmat_edges(14, n * 16, 0xAA, dev)
    ptrs, idx, _ = _cabi.coo_to_csx(row, col, n, n, True)
    return ptrs, idx, n


def _replicated(ptrs, idx, seeds, first_call, sampler):
    from tch_geometric import _cabi
    out = _cabi.NsBatchedOut(seeds.shape[0], seeds.shape[1], FANOUT, seeds.device)
    _cabi.ns_homo_batched(_cabi.graph_view(ptrs, idx), seeds, FANOUT, SEED, first_call, out, sampler=sampler)
    torch.cuda.synchronize()
    return [out.batch(b) for b in range(seeds.shape[0])]


@pytest.mark.parametrize("sampler", [0, 1])
def test_single_rank_partitioned_equals_batched(sampler):
    from tch_geometric import _cabi, partitioned
    dev = torch.device("cuda:0")
    ptrs, idx, n = _graph(dev)
    seeds = _cabi.seed_batches(9, 40, 5, B, n, dev)
    shard = partitioned.CscShard.from_full(ptrs, idx, 0, 1)
    res = partitioned.ns_homo_partitioned(shard, seeds, FANOUT, SEED, 40, sampler=sampler)
    ref = _replicated(ptrs, idx, seeds, 40, sampler)
    for (s, r, c, e, lo), (rs, rr, rc, re_, rlo) in zip(res, ref):
        assert lo == rlo
        assert torch.equal(s, rs) and torch.equal(r, rr) and torch.equal(c, rc) and torch.equal(e, re_)
    # and against the oracle for one batch
    o = orc.ns_homo(ptrs.cpu().numpy(), idx.cpu().numpy(), seeds[2].cpu().numpy(), FANOUT, orc.rng_philox(SEED, 42),
                    sampler=sampler)
    assert np.array_equal(res[2][0].cpu().numpy(), o[0]) and np.array_equal(res[2][3].cpu().numpy(), o[3])
    # the device form of the exchange (csrc/partition.hip) fills ordinary per-batch slabs with the same contents
    # (with one-word packed replies, which this small graph allows, and with pairs)
    # (slot replies, the default here; compact replies as packed words and as pairs)
    for packed, slots in ((None, None), (None, False), (False, False)):
        dout = partitioned.ns_homo_partitioned_device(shard, seeds, FANOUT, SEED, 40, sampler=sampler,
                                                      packed_replies=packed, slot_replies=slots)
        torch.cuda.synchronize()
        for b, (rs, rr, rc, re_, rlo) in enumerate(ref):
            s, r, c, e, lo = dout.batch(b)
            assert lo == rlo and torch.equal(s, rs) and torch.equal(r, rr) and torch.equal(c, rc) and torch.equal(e, re_), b


@pytest.mark.parametrize("sampler", [0, 1])
@pytest.mark.parametrize("packed", [None, False, "slots"])
def test_owner_side_in_window_order(sampler, packed):
    """tg_part_sample_ws / tg_part_sample_slots: the owner sorts a hop's requests by the window of their column and samples
    in that order; the replies are the same words in the same places.  Thresholds lowered so that the small graph takes the ordered path;
    then a launch that takes it by itself (RMAT-22, 512 batches: 7.8 M requests in the second hop)."""
    import ctypes as C
    from tch_geometric import _cabi, partitioned
    dev = torch.device("cuda:0")
    _cabi.lib.tg_part_sample_order_thresholds(C.c_int64(1), C.c_int64(1))
    try:
        ptrs, idx, n = _graph(dev)
        seeds = _cabi.seed_batches(9, 40, 7, B, n, dev)
        seeds[3, :5] = int(torch.argmax(ptrs[1:] - ptrs[:-1]))
        shard = partitioned.CscShard.from_full(ptrs, idx, 0, 1)
        ref = _replicated(ptrs, idx, seeds, 40, sampler)
        slots = packed == "slots"
        dout = partitioned.ns_homo_partitioned_device(shard, seeds, FANOUT, SEED, 40, sampler=sampler,
                                                      packed_replies=None if slots else packed, slot_replies=slots)
        torch.cuda.synchronize()
        for b, (rs, rr, rc, re_, rlo) in enumerate(ref):
            s, r, c, e, lo = dout.batch(b)
            assert lo == rlo and torch.equal(s, rs) and torch.equal(r, rr) and torch.equal(c, rc) and torch.equal(e, re_), b
    finally:
        _cabi.lib.tg_part_sample_order_thresholds(C.c_int64(1 << 21), C.c_int64(1 << 24))
    if sampler == 0 and packed in (None, "slots"):
        n = 1 << 22
        row, col = _cabi.rmat_edges(22, n * 16, 0x5EED0016, dev)
        ptrs, idx, _ = _cabi.coo_to_csx(row, col, n, n, True)
        del row, col
        nb = 512
        seeds = _cabi.seed_batches(0xBA7C4, 0, nb, 1024, n, dev)
        shard = partitioned.CscShard.from_full(ptrs, idx, 0, 1)
        ps = partitioned.PartitionedSampler(shard, nb, 1024, FANOUT, slot_replies=packed == "slots")
        assert ps.slots == (packed == "slots")
        out = ps.sample(seeds, SEED, 0)
        ref = _cabi.NsBatchedOut(nb, 1024, FANOUT, dev)
        _cabi.ns_homo_batched(_cabi.graph_view(ptrs, idx), seeds, FANOUT, SEED, 0, ref)
        torch.cuda.synchronize()
        assert torch.equal(out.counts, ref.counts) and torch.equal(out.layer_offsets, ref.layer_offsets)
        ar_n = torch.arange(out.samples.shape[1], device=dev)[None, :]
        ar_e = torch.arange(out.rows.shape[1], device=dev)[None, :]
        assert bool(((out.samples == ref.samples) | (ar_n >= ref.counts[:, 0:1])).all())
        for a_, b_ in ((out.rows, ref.rows), (out.cols, ref.cols), (out.edge_index, ref.edge_index)):
            assert bool(((a_ == b_) | (ar_e >= ref.counts[:, 1:2])).all())


@pytest.mark.parametrize("fan", [[1], [32, 2], [3, 3, 3], []])
def test_device_form_other_fanouts_and_empty_columns(fan):
    from tch_geometric import _cabi, partitioned
    dev = torch.device("cuda:0")
    ptrs, idx, n = _graph(dev)
    seeds = _cabi.seed_batches(11, 7, 3, 50, n, dev)
    shard = partitioned.CscShard.from_full(ptrs, idx, 0, 1)
    dout = partitioned.ns_homo_partitioned_device(shard, seeds, fan, SEED, 7)
    out = _cabi.NsBatchedOut(3, 50, fan, dev)
    _cabi.ns_homo_batched(_cabi.graph_view(ptrs, idx), seeds, fan, SEED, 7, out)
    torch.cuda.synchronize()
    for b in range(3):
        for x, y in zip(dout.batch(b), out.batch(b)):
            assert (x == y) if isinstance(x, list) else torch.equal(x, y)


def test_sampler_object_reuses_its_buffers():
    """two calls through one PartitionedSampler (buffers allocated once) equal two replicated launches"""
    from tch_geometric import _cabi, partitioned
    dev = torch.device("cuda:0")
    ptrs, idx, n = _graph(dev)
    shard = partitioned.CscShard.from_full(ptrs, idx, 0, 1)
    ps = partitioned.PartitionedSampler(shard, 5, B, FANOUT)
    for first in (40, 900):
        seeds = _cabi.seed_batches(9, first, 5, B, n, dev)
        out = ps.sample(seeds, SEED, first)
        torch.cuda.synchronize()
        ref = _replicated(ptrs, idx, seeds, first, 0)
        for b, (rs, rr, rc, re_, rlo) in enumerate(ref):
            s, r, c, e, lo = out.batch(b)
            assert lo == rlo and torch.equal(s, rs) and torch.equal(r, rr) and torch.equal(c, rc) and torch.equal(e, re_)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    for p in (os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tch-geometric_amd")):
        sys.path.insert(0, p)
    from tch_geometric import _cabi, partitioned
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")          # both ranks share the box's one GPU; a real node gives each its own
    ptrs, idx, n = _graph(dev)
    shard = partitioned.CscShard.from_full(ptrs, idx, rank, world)
    first = 100 + rank * 4
    seeds = _cabi.seed_batches(9, first, 4, B, n, dev)
    res = partitioned.ns_homo_partitioned(shard, seeds, FANOUT, SEED, first)
    ref = _replicated(ptrs, idx, seeds, first, 0)
    ok = all(lo == rlo and torch.equal(s, rs) and torch.equal(r, rr) and torch.equal(c, rc) and torch.equal(e, re_)
             for (s, r, c, e, lo), (rs, rr, rc, re_, rlo) in zip(res, ref))
    dout = partitioned.ns_homo_partitioned_device(shard, seeds, FANOUT, SEED, first)   # device form, same exchange
    torch.cuda.synchronize()
    for b, (rs, rr, rc, re_, rlo) in enumerate(ref):
        s, r, c, e, lo = dout.batch(b)
        ok = ok and lo == rlo and torch.equal(s, rs) and torch.equal(r, rr) and torch.equal(c, rc) and torch.equal(e, re_)
    # the same exchange under a dynamic temporal filter and with weights (filter states travel with the requests)
    aptrs, aidx, _, ts, w = _attr_graph(dev)
    states = torch.randint(0, 60, (4, B), device=dev, generator=torch.Generator(device=dev).manual_seed(rank))
    for case in (FILTER_CASES[2], FILTER_CASES[4]):
        ash = partitioned.CscShard.from_full(aptrs, aidx, rank, world, weights=w, timestamps=ts)
        ps = partitioned.PartitionedSampler(ash, 4, B, [6, 4], sampler=case["sampler"], filter_mode=case["filter_mode"],
                                            forward=case["forward"], window=case["window"])
        got = ps.sample(seeds, SEED, first, seeds_state=states)
        want = _replicated_general(aptrs, aidx, ts, w, seeds, states, [6, 4], first, case)
        c = want.counts.cpu()
        ok = ok and torch.equal(got.counts.cpu(), c)
        for b in range(4):
            x, y = got.batch(b, c), want.batch(b, c)
            ok = ok and x[4] == y[4] and all(torch.equal(u, v) for u, v in zip(x[:4], y[:4]))
            ok = ok and torch.equal(got.states[b, :int(c[b, 0])], want.states[b, :int(c[b, 0])])
    remote = sum(int(((s[B:] // shard.shard_size).clamp(max=world - 1) != rank).sum()) for s, *_ in res)
    q.put((rank, ok, remote, sum(int(r.numel()) for _, r, *_ in res)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_on_one_gpu_over_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, ok, remote, edges in got:
        assert ok, "rank %d: partitioned result differs from the replicated launch" % rank
        assert edges > 0 and remote > 0        # the test really crossed the partition boundary


@pytest.mark.parametrize("packed", [False, True, "slots"])
@pytest.mark.parametrize("world", [3, 8])
def test_device_kernels_with_an_emulated_world(world, packed):
    """The request bucketing / owner sampling / emit kernels for world > 2, with the all-to-alls emulated in one
    process: bucket p of the requests is sampled against shard p, replies are concatenated in bucket order.  Both reply
    formats: (neighbour, edge pointer) pairs and the one-word packed entries."""
    from helpers_part import emulated_world_sample
    from tch_geometric import _cabi, partitioned
    dev = torch.device("cuda:0")
    ptrs, idx, n = _graph(dev)
    nb, fan = 6, [7, 5]
    seeds = _cabi.seed_batches(13, 500, nb, B, n, dev)
    shards = [partitioned.CscShard.from_full(ptrs, idx, r, world) for r in range(world)]
    out, crossed = emulated_world_sample(_cabi, shards, seeds, fan, SEED, 500, packed=packed is True, slots=packed == "slots")
    ref = _cabi.NsBatchedOut(nb, B, fan, dev)
    _cabi.ns_homo_batched(_cabi.graph_view(ptrs, idx), seeds, fan, SEED, 500, ref)
    torch.cuda.synchronize()
    assert crossed > 0
    for b in range(nb):
        for x, y in zip(out.batch(b), ref.batch(b)):
            assert (x == y) if isinstance(x, list) else torch.equal(x, y)


@pytest.mark.parametrize("slots", [False, True])
def test_ranks_without_requests(slots):
    """all seeds in one shard's vertex range, one hop: the other owners receive nothing and may be handed no request
    buffer at all (tg_part_count / tg_part_sample with m = 0)"""
    from helpers_part import emulated_world_sample
    from tch_geometric import _cabi, partitioned
    dev = torch.device("cuda:0")
    ptrs, idx, n = _graph(dev)
    world, fan = 4, [3]
    seeds = torch.randint(0, n // world // 2, (2, 8), device=dev)
    shards = [partitioned.CscShard.from_full(ptrs, idx, r, world) for r in range(world)]
    out, crossed = emulated_world_sample(_cabi, shards, seeds, fan, SEED, 9, slots=slots)
    assert crossed == 0
    ref = _cabi.NsBatchedOut(2, 8, fan, dev)
    _cabi.ns_homo_batched(_cabi.graph_view(ptrs, idx), seeds, fan, SEED, 9, ref)
    torch.cuda.synchronize()
    for b in range(2):
        for x, y in zip(out.batch(b), ref.batch(b)):
            assert (x == y) if isinstance(x, list) else torch.equal(x, y)


def test_rccl_backend_gets_device_tensors_only(monkeypatch):
    """every tensor the device form hands to all_to_all_single under a non-gloo backend lives on the GPU (the
    collective is mocked as a loop-back: this rank's own buffers come back)"""
    from tch_geometric import _cabi, partitioned
    seen = []

    def fake_a2a(out, inp, output_split_sizes=None, input_split_sizes=None, group=None):
        seen.append((out.device.type, inp.device.type))
        out.copy_(inp)

    monkeypatch.setattr(dist, "is_initialized", lambda: True)
    monkeypatch.setattr(dist, "get_world_size", lambda group=None: 2)
    monkeypatch.setattr(dist, "get_rank", lambda group=None: 0)
    monkeypatch.setattr(dist, "get_backend", lambda group=None: "nccl")
    def fake_all_gather(parts, t, group=None):
        seen.append((parts[0].device.type, t.device.type))
        for p in parts:
            p.copy_(t)

    monkeypatch.setattr(dist, "all_to_all_single", fake_a2a)
    monkeypatch.setattr(dist, "all_gather", fake_all_gather)
    dev = torch.device("cuda:0")
    assert partitioned._exchange_counts([3, 4], None, dev) == [3, 4]
    ptrs, idx, n = _graph(dev)
    shard = partitioned.CscShard.from_full(ptrs, idx, 0, 2)
    seeds = _cabi.seed_batches(9, 0, 3, B, n, dev) % shard.v_hi          # keep every seed inside this rank's shard
    # collectives per hop: compact replies 5 (request sizes, requests, reply sizes, counts, replies); slot replies 3
    # (request sizes, requests, slots) + the one all_gather that agrees on the slot format when the sampler is made
    for slots, per_hop, once in ((False, 5, 0), (None, 3, 1)):
        del seen[1:]
        ps = partitioned.PartitionedSampler(shard, 3, B, [4, 3], slot_replies=slots)
        assert ps.slots == (slots is None)
        ps.sample(seeds, SEED, 0, first_call_ids=[0, 0])
        torch.cuda.synchronize()
        assert len(seen) == 1 + once + 2 * per_hop and all(a == "cuda" and b == "cuda" for a, b in seen)


# ---------------------------------------------------------------- temporal filters and weights on a partitioned graph
def test_status_of_every_hop_reaches_the_caller():
    """the owner-side status word used to be zeroed per hop and read after the last one only.  A column of
    zero weights that only a SEED reaches (hop 0 of a 2-hop call; the later hop is healthy) must raise, as the reference
    panics (sampling.rs:49) and as the replicated operator does."""
    from tch_geometric import _cabi, partitioned
    dev = torch.device("cuda:0")
    ptrs, idx, n, ts, w = _attr_graph(dev)
    deg = ptrs[1:] - ptrs[:-1]
    outdeg = torch.bincount(idx, minlength=n)
    cand = ((deg > 6) & (outdeg == 0)).nonzero().reshape(-1)           # never sampled as a neighbour: only a seed reaches it
    assert cand.numel() > 0
    hub = int(cand[0])
    seeds = _cabi.seed_batches(21, 0, 2, B, n, dev)
    seeds[1, 3] = hub
    fan = [6, 4]
    shard = partitioned.CscShard.from_full(ptrs, idx, 0, 1, weights=w, timestamps=ts)
    ps = partitioned.PartitionedSampler(shard, 2, B, fan, sampler=2)
    ps.sample(seeds, SEED, 0)                                          # healthy weights: fine
    w0 = w.clone()
    w0[int(ptrs[hub]):int(ptrs[hub + 1])] = 0.0                        # the hub's whole column weighs nothing
    # vertices sampled FROM the hub's column get expanded in the later hop with healthy weights: only hop 0 can report it
    shard0 = partitioned.CscShard.from_full(ptrs, idx, 0, 1, weights=w0, timestamps=ts)
    ps0 = partitioned.PartitionedSampler(shard0, 2, B, fan, sampler=2)
    with pytest.raises(RuntimeError, match="non-positive running weight sum"):
        ps0.sample(seeds, SEED, 0)
    seeds[1, 3] = seeds[1, 4]
    ps0.sample(seeds, SEED, 0)                                         # the same sampler, hub not reached: fine again


def test_column_group_overflow_repeats_the_call():
    """a too-small column-group workspace is reported in the call's one read-back; the call is repeated with a larger guess
    and the guess is kept -- results equal the replicated launch"""
    from tch_geometric import _cabi, partitioned
    dev = torch.device("cuda:0")
    ptrs, idx, n, ts, w = _attr_graph(dev)
    case = FILTER_CASES[2]
    fan, nb, first = [6, 4], 4, 70
    seeds = _cabi.seed_batches(21, first, nb, B, n, dev)
    states = torch.randint(0, 60, (nb, B), device=dev)
    ref = _replicated_general(ptrs, idx, ts, w, seeds, states, fan, first, case)
    shard = partitioned.CscShard.from_full(ptrs, idx, 0, 1, weights=w, timestamps=ts)
    ps = partitioned.PartitionedSampler(shard, nb, B, fan, sampler=case["sampler"], filter_mode=case["filter_mode"],
                                        forward=case["forward"], window=case["window"])
    ps._group_mult = 1e-6                                              # a guess of ONE group: every hop overflows at first
    out = ps.sample(seeds, SEED, first, seeds_state=states)
    torch.cuda.synchronize()
    assert ps._group_mult > 1e-6
    c = ref.counts.cpu()
    assert torch.equal(out.counts.cpu(), c) and int(c[:, 1].sum()) > 0
    for b in range(nb):
        for u, v in zip(out.batch(b, c)[:4], ref.batch(b, c)[:4]):
            assert torch.equal(u, v)


def _attr_graph(dev):
    ptrs, idx, n = _graph(dev)
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    ts = torch.randint(0, 60, (idx.numel(),), device=dev, generator=g)
    w = torch.rand(idx.numel(), device=dev, generator=g, dtype=torch.float64) + 0.05
    return ptrs, idx, n, ts, w


FILTER_CASES = [dict(sampler=0, filter_mode=0, forward=False, window=(10, 45)),      # static window
                dict(sampler=1, filter_mode=1, forward=True, window=(-20, 5)),       # relative, with replacement
                dict(sampler=0, filter_mode=2, forward=False, window=(0, 30)),       # dynamic: states follow the edges
                dict(sampler=2, filter_mode=-1, forward=False, window=(0, 0)),       # weighted, no filter
                dict(sampler=2, filter_mode=2, forward=True, window=(-15, 15))]      # weighted under a dynamic filter


def _replicated_general(ptrs, idx, ts, w, seeds, states, fan, first, case):
    from tch_geometric import _cabi
    out = _cabi.NsBatchedOut(seeds.shape[0], seeds.shape[1], fan, seeds.device, with_states=case["filter_mode"] != -1)
    g = _cabi.graph_view(ptrs, idx, weights=w if case["sampler"] == 2 else None,
                         timestamps=ts if case["filter_mode"] != -1 else None)
    _cabi.ns_homo_batched(g, seeds, fan, SEED, first, out, sampler=case["sampler"], filter_mode=case["filter_mode"],
                          forward=case["forward"], window=case["window"],
                          seeds_state=states if case["filter_mode"] != -1 else None)
    torch.cuda.synchronize()
    return out


@pytest.mark.parametrize("packed", [False, True])
@pytest.mark.parametrize("case", FILTER_CASES)
@pytest.mark.parametrize("world", [1, 3])
def test_filters_and_weights_equal_the_replicated_launch(case, world, packed):
    """neighbor_sampling.rs:36-77 / :131-158 over a partitioned graph: world 1 through PartitionedSampler, world 3 with
    the exchange emulated (bucket p answered from shard p); both equal tg_ns_homo_batched on the whole graph"""
    from helpers_part import emulated_world_sample
    from tch_geometric import _cabi, partitioned
    dev = torch.device("cuda:0")
    ptrs, idx, n, ts, w = _attr_graph(dev)
    fan, nb, first = [6, 4], 4, 70
    seeds = _cabi.seed_batches(21, first, nb, B, n, dev)
    states = torch.randint(0, 60, (nb, B), device=dev)
    ref = _replicated_general(ptrs, idx, ts, w, seeds, states, fan, first, case)
    assert int(ref.counts[:, 1].sum()) > 0
    filtered = case["filter_mode"] != -1
    if world == 1:
        shard = partitioned.CscShard.from_full(ptrs, idx, 0, 1, weights=w, timestamps=ts)
        ps = partitioned.PartitionedSampler(shard, nb, B, fan, sampler=case["sampler"], filter_mode=case["filter_mode"],
                                            forward=case["forward"], window=case["window"], packed_replies=packed)
        out = ps.sample(seeds, SEED, first, seeds_state=states if filtered else None)
        torch.cuda.synchronize()
    else:
        shards = [partitioned.CscShard.from_full(ptrs, idx, r, world, weights=w, timestamps=ts) for r in range(world)]
        out, crossed = emulated_world_sample(_cabi, shards, seeds, fan, SEED, first, sampler=case["sampler"],
                                             filter_mode=case["filter_mode"], forward=case["forward"],
                                             window=case["window"], seeds_state=states, packed=packed)
        assert crossed > 0
    c = ref.counts.cpu()
    assert torch.equal(out.counts.cpu(), c)
    for b in range(nb):
        x, y = out.batch(b, c), ref.batch(b, c)
        assert x[4] == y[4]
        for u, v in zip(x[:4], y[:4]):
            assert torch.equal(u, v), (case, b)
        if filtered:
            ns = int(c[b, 0])
            assert torch.equal(out.states[b, :ns], ref.states[b, :ns])


# ---------------------------------------------------------------- the multi-rank protocol over RCCL itself, one rank
def _rccl_worker(ports, q):
    for p in (os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tch-geometric_amd")):
        sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    try:
        from tch_geometric import _cabi, partitioned
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        for attempt, port in enumerate(ports):      # a port probed free can be taken again before the store binds it
            os.environ["MASTER_PORT"] = str(port)
            try:
                dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
                break
            except Exception:  # noqa: BLE001
                if attempt + 1 == len(ports):
                    raise
        ok, calls = True, 0
        errs = []

        def chk(cond, label):                 # the first labels that differ travel back in the status string
            nonlocal ok
            if not cond:
                ok = False
                if len(errs) < 6:
                    errs.append(label)

        real = dist.all_to_all_single

        def counted(*a, **k):
            nonlocal calls
            calls += 1
            return real(*a, **k)

        dist.all_to_all_single = counted
        aptrs, aidx, n, ts, w = _attr_graph(dev)
        seeds = _cabi.seed_batches(21, 70, 4, B, n, dev)
        states = torch.randint(0, 60, (4, B), device=dev)
        shard = partitioned.CscShard.from_full(aptrs, aidx, 0, 1, weights=w, timestamps=ts)
        for case in (dict(sampler=0, filter_mode=-1, forward=False, window=(0, 0)), FILTER_CASES[2], FILTER_CASES[4]):
            ps = partitioned.PartitionedSampler(shard, 4, B, [6, 4], sampler=case["sampler"], filter_mode=case["filter_mode"],
                                                forward=case["forward"], window=case["window"], force_exchange=True)
            got = ps.sample(seeds, SEED, 70, seeds_state=states if case["filter_mode"] != -1 else None)
            want = _replicated_general(aptrs, aidx, ts, w, seeds, states, [6, 4], 70, case)
            c = want.counts.cpu()
            chk(torch.equal(got.counts.cpu(), c), ("single", case["sampler"], case["filter_mode"], "counts"))
            for b in range(4):
                x, y = got.batch(b, c), want.batch(b, c)
                chk(x[4] == y[4] and all(torch.equal(u, v) for u, v in zip(x[:4], y[:4])),
                    ("single", case["sampler"], case["filter_mode"], "batch", b))
        # super-batches in flight from ONE host thread over ONE communicator (round 4: PipelinedPartitionedSampler,
        # bench.py --lanes): the steps of two samplers interleave in a fixed order; every lane's slabs are reused
        n_jobs = 5
        job_seeds = [_cabi.seed_batches(33 + i, 900 + 10 * i, 4, B, n, dev) for i in range(n_jobs)]
        plain = dict(sampler=0, filter_mode=-1, forward=False, window=(0, 0))
        pipe = partitioned.PipelinedPartitionedSampler(shard, 4, B, [6, 4], lanes=2, force_exchange=True)
        kept = {}

        def consume(i, out):
            c = out.counts.clone()
            kept[i] = (c, [tuple(t.clone() for t in out.batch(b, c.cpu())[:4]) + (out.batch(b, c.cpu())[4],) for b in range(4)])

        pipe.sample_many(n_jobs, lambda i: job_seeds[i], SEED, lambda i: (900 + 10 * i, [900 + 10 * i]), consume)
        torch.cuda.synchronize()
        for i in range(n_jobs):
            want = _replicated_general(aptrs, aidx, ts, w, job_seeds[i], states, [6, 4], 900 + 10 * i, plain)
            c = want.counts.cpu()
            chk(i in kept and torch.equal(kept[i][0].cpu(), c), ("lanes", i, "counts"))
            for b in range(4):
                y = want.batch(b, c)
                x = kept[i][1][b]
                chk(x[4] == y[4] and all(torch.equal(u, v) for u, v in zip(x[:4], y[:4])), ("lanes", i, "batch", b))
        # ... and under a filter with weights (the status word is read per super-batch)
        case = FILTER_CASES[4]
        pipe_g = partitioned.PipelinedPartitionedSampler(shard, 4, B, [6, 4], lanes=2, force_exchange=True, sampler=case["sampler"],
                                                         filter_mode=case["filter_mode"], forward=case["forward"],
                                                         window=case["window"])
        kept.clear()
        pipe_g.sample_many(3, lambda i: job_seeds[i], SEED, lambda i: (900 + 10 * i, [900 + 10 * i]), consume,
                           seeds_state_of=lambda i: states)
        torch.cuda.synchronize()
        for i in range(3):
            want = _replicated_general(aptrs, aidx, ts, w, job_seeds[i], states, [6, 4], 900 + 10 * i, case)
            c = want.counts.cpu()
            chk(torch.equal(kept[i][0].cpu(), c), ("lanes, filter + weights", i, "counts"))
            for b in range(4):
                y = want.batch(b, c)
                x = kept[i][1][b]
                chk(x[4] == y[4] and all(torch.equal(u, v) for u, v in zip(x[:4], y[:4])), ("lanes, filter + weights", i, "batch", b))
        q.put(("ok" if ok else "mismatch %r" % (errs,), calls, dist.get_backend()))
        dist.destroy_process_group()
    except Exception as e:  # noqa: BLE001
        q.put(("error: %r" % (e,), 0, ""))


def test_protocol_over_rccl_with_one_rank():
    """every collective of the multi-rank protocol (sizes, requests, filter states, counts, reply sizes, replies) through
    RCCL (`backend="nccl"`) with device tensors: one rank exchanging with itself -- the transport a one-GPU box can run"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=([_free_port() for _ in range(3)], q))
    p.start()
    status, calls, backend = q.get(timeout=300)
    p.join(timeout=120)
    assert status == "ok", status
    assert backend == "nccl" and calls >= 3 * 2 * 5      # three configurations x two hops x >= 5 collectives per hop


def test_slot_replies_random_sweep():
    """slot replies over emulated worlds of 1 .. 8 ranks: random fan-outs (1 .. 16, one- and two-chunk slots), both plain
    samplers, graphs with empty columns, a hub, shards that own nothing, seeds that repeat -- each equal to the replicated
    launch; and the format's limits (fan-out 17, > 1 024 bits: no slot format)"""
    import ctypes as C
    from helpers_part import emulated_world_sample
    from tch_geometric import _cabi, partitioned
    dev = torch.device("cuda:0")
    rs = np.random.default_rng(77)
    w = C.c_int32(-1)
    _cabi.check(_cabi.lib.tg_part_slot_words(C.c_int32(17), C.c_int32(20), C.c_int32(10), C.byref(w)))
    assert w.value == 0
    _cabi.check(_cabi.lib.tg_part_slot_words(C.c_int32(16), C.c_int32(32), C.c_int32(32), C.byref(w)))
    assert w.value == 0                                              # 40 + 16 * 64 = 1 064 bits
    _cabi.check(_cabi.lib.tg_part_slot_words(C.c_int32(16), C.c_int32(32), C.c_int32(29), C.byref(w)))
    assert w.value == 32                                             # 1 016 bits: two chunks
    _cabi.check(_cabi.lib.tg_part_slot_words(C.c_int32(10), C.c_int32(24), C.c_int32(19), C.byref(w)))
    assert w.value == 16                                             # RMAT-24's 470 bits: one chunk
    for case in range(14):
        n = int(rs.integers(40, 3000))
        e = int(rs.integers(n, 12 * n))
        row, col = rs.integers(0, n, e), rs.integers(0, max(1, n - n // 5), e)   # the last fifth of the columns is empty
        if case % 3 == 0:
            col[: e // 4] = int(rs.integers(0, n))                   # a hub column
        ptrs_h, idx_h, _ = orc.to_csc(np.stack([row, col]), n)
        ptrs, idx = torch.from_numpy(ptrs_h).to(dev), torch.from_numpy(idx_h).to(dev)
        world = int(rs.integers(1, 9))
        hops = int(rs.integers(1, 4))
        fan = [int(rs.integers(1, 17)) for _ in range(hops)]
        while np.prod(fan) > 600:
            fan[int(np.argmax(fan))] //= 2
        sampler = int(rs.integers(0, 2))
        nb, B_ = int(rs.integers(1, 7)), int(rs.integers(1, 40))
        seeds = torch.from_numpy(rs.integers(0, n, (nb, B_))).to(dev)
        seeds[0, : min(3, B_)] = int(col[0])                         # repeated seeds
        shards = [partitioned.CscShard.from_full(ptrs, idx, r, world) for r in range(world)]
        out, _ = emulated_world_sample(_cabi, shards, seeds, fan, SEED + case, 1000 * case, sampler=sampler, slots=True)
        ref = _cabi.NsBatchedOut(nb, B_, fan, dev)
        _cabi.ns_homo_batched(_cabi.graph_view(ptrs, idx), seeds, fan, SEED + case, 1000 * case, ref, sampler=sampler)
        torch.cuda.synchronize()
        for b in range(nb):
            for x, y in zip(out.batch(b), ref.batch(b)):
                assert (x == y) if isinstance(x, list) else torch.equal(x, y), (case, world, fan, sampler, b)
        if case < 4:                                                  # and the oracle itself, for one batch
            o = orc.ns_homo(ptrs_h, idx_h, seeds[0].cpu().numpy(), fan, orc.rng_philox(SEED + case, 1000 * case), sampler=sampler)
            s, r, c, e_, lo = out.batch(0)
            assert np.array_equal(s.cpu().numpy(), o[0]) and np.array_equal(e_.cpu().numpy(), o[3]) and lo == o[4]
