"""Range-partitioned neighbor sampling (SURVEY 8(e) mode 2) on CPU: two gloo ranks each own half of the
columns and half of the seed batches; the owner-side sampler stand-in is the oracle's philox-mode addressed
with the REQUESTER's (call id, slot).  The union must equal the replicated-graph oracle batch for batch."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SEED, FANOUT, B, PER_RANK = 0xD157, [6, 4], 16, 3


def _paths():
    for p in (os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tch-geometric_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)


def _graph():
    _paths()
    import orc
    n = 1 << 9
    row, col = orc.rmat_edges(9, n * 12, 33)
    ptrs, idx, _ = orc.to_csc(np.stack([row, col]), n)
    return ptrs, idx, n


def _oracle_hop(shard, local_v, call_ids, slot_ids, k, seed, sampler):
    import orc
    ptrs, idx = shard.ptrs.numpy(), shard.indices.numpy()
    cnt, nbr, ep = [], [], []
    for v, c, s in zip(local_v.tolist(), call_ids.tolist(), slot_ids.tolist()):
        r = orc.ns_homo(ptrs, idx, [v], [k], orc.rng_philox(seed, c), sampler=sampler, id_base=s)
        cnt.append(len(r[1]))
        nbr.append(r[0][1:])
        ep.append(r[3])
    t = lambda parts: torch.from_numpy(np.concatenate(parts).astype(np.int64)) if parts else torch.zeros(0, dtype=torch.int64)
    return torch.tensor(cnt, dtype=torch.int64), t(nbr), t(ep)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, sampler, q, pipelined=False):
    _paths()
    import orc
    from tch_geometric import partitioned
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ptrs, idx, n = _graph()
    shard = partitioned.CscShard.from_full(torch.from_numpy(ptrs), torch.from_numpy(idx), rank, world)
    first = rank * PER_RANK
    seeds = torch.from_numpy(orc.seed_batches(5, first, PER_RANK, B, n))
    if not pipelined:
        res = partitioned.ns_homo_partitioned(shard, seeds, FANOUT, SEED, first, sampler=sampler, _hop_fn=_oracle_hop)
    else:
        # every batch a super-batch of its own; two in flight from ONE thread over ONE communicator, in the fixed,
        # rank-independent order of `interleave` (hop h of super-batch i + 1 is enqueued between the request and the reply
        # exchange of super-batch i); a log of the steps shows the schedule really interleaves
        res, log = [None] * PER_RANK, []

        def start(i, lane):
            def run():
                gen = partitioned.ns_homo_partitioned_steps(shard, seeds[i:i + 1], FANOUT, SEED, first + i, sampler=sampler,
                                                            _hop_fn=_oracle_hop)
                while True:
                    try:
                        log.append((i, next(gen)))
                    except StopIteration as done:
                        res[i] = done.value[0]
                        return
                    yield
            return run()

        partitioned.interleave(PER_RANK, 2, start)
        jobs = [i for i, _ in log]
        assert any(a != b for a, b in zip(jobs, jobs[1:])) and jobs != sorted(jobs), "the schedule did not interleave"
        q.put(("log%d" % rank, log))
    q.put((rank, [(s.tolist(), r.tolist(), c.tolist(), e.tolist(), lo) for s, r, c, e, lo in res]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("sampler", [0, 1])
def test_two_rank_partitioned_equals_replicated(sampler):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, sampler, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    import orc
    ptrs, idx, n = _graph()
    for rank in range(world):
        seeds = orc.seed_batches(5, rank * PER_RANK, PER_RANK, B, n)
        for j in range(PER_RANK):
            o = orc.ns_homo(ptrs, idx, seeds[j], FANOUT, orc.rng_philox(SEED, rank * PER_RANK + j), sampler=sampler)
            s, r, c, e, lo = got[rank][j]
            assert lo == o[4]
            assert s == o[0].tolist() and r == o[1].tolist() and c == o[2].tolist() and e == o[3].tolist()


def test_two_rank_pipelined_schedule_equals_replicated():
    """the pipelined schedule that is legal with several ranks -- one host thread, one communicator, a
    fixed enqueue order -- gives the replicated result bit for bit, and both ranks walk the same step sequence."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, 0, q, True)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in range(2 * world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got["log0"] == got["log1"] and len(got["log0"]) == PER_RANK * 2 * len(FANOUT)   # two size exchanges per hop
    import orc
    ptrs, idx, n = _graph()
    for rank in range(world):
        seeds = orc.seed_batches(5, rank * PER_RANK, PER_RANK, B, n)
        for j in range(PER_RANK):
            o = orc.ns_homo(ptrs, idx, seeds[j], FANOUT, orc.rng_philox(SEED, rank * PER_RANK + j))
            s, r, c, e, lo = got[rank][j]
            assert lo == o[4]
            assert s == o[0].tolist() and r == o[1].tolist() and c == o[2].tolist() and e == o[3].tolist()


def test_interleave_order_is_fixed():
    _paths()
    from tch_geometric import partitioned
    order = []

    def start(i, lane):
        def run():
            for step in range(3):
                order.append((i, lane, step))
                yield
        return run()

    done = []
    partitioned.interleave(5, 2, start, lambda i, lane: done.append(i))
    assert done == [0, 1, 2, 3, 4]
    assert order[:6] == [(0, 0, 0), (1, 1, 0), (0, 0, 1), (1, 1, 1), (0, 0, 2), (1, 1, 2)]
    assert [(i, l) for i, l, s in order if s == 0] == [(0, 0), (1, 1), (2, 0), (3, 1), (4, 0)]


def test_shards_tile_the_graph():
    _paths()
    from tch_geometric import partitioned
    ptrs, idx, n = _graph()
    P, I = torch.from_numpy(ptrs), torch.from_numpy(idx)
    for world in (1, 2, 3, 8):
        shards = [partitioned.CscShard.from_full(P, I, r, world) for r in range(world)]
        assert shards[0].v_lo == 0 and shards[-1].v_hi == n
        assert all(a.v_hi == b.v_lo for a, b in zip(shards, shards[1:]))
        assert sum(s.indices.numel() for s in shards) == len(idx)
        for s in shards:
            assert int(s.ptrs[0]) == 0 and int(s.ptrs[-1]) == s.indices.numel()
            assert torch.equal(s.indices, I[s.e_lo:s.e_lo + s.indices.numel()])


def test_size_exchange_never_hands_cpu_tensors_to_rccl(monkeypatch):
    """ProcessGroupNCCL rejects CPU tensors: with a non-gloo backend the split sizes must travel on the compute
    device (and the call must say so when it has none) -- checked here with the collective mocked."""
    import torch.distributed as dist
    from tch_geometric import partitioned
    seen = []

    def fake_a2a(out, inp, output_split_sizes=None, input_split_sizes=None, group=None):
        seen.append((out.device.type, inp.device.type))
        out.copy_(inp)

    monkeypatch.setattr(dist, "is_initialized", lambda: True)
    monkeypatch.setattr(dist, "get_world_size", lambda group=None: 2)
    monkeypatch.setattr(dist, "get_rank", lambda group=None: 0)
    monkeypatch.setattr(dist, "all_to_all_single", fake_a2a)
    monkeypatch.setattr(dist, "get_backend", lambda group=None: "nccl")
    with pytest.raises(ValueError, match="needs the compute device"):
        partitioned._exchange_counts([3, 4], None)
    with pytest.raises(ValueError, match="needs the compute device"):
        partitioned._exchange_counts([3, 4], None, torch.device("cpu"))
    assert not seen
    monkeypatch.setattr(dist, "get_backend", lambda group=None: "gloo")
    assert partitioned._exchange_counts([3, 4], None) == [3, 4] and seen == [("cpu", "cpu")]
