"""N > 1 path on CPU: two gloo ranks shard seed batches the way bench.py does; the sampler stand-in is the
oracle's philox-mode (the same draws the HIP kernels use), so the test pins that (a) the shards tile the
global batch range, (b) a batch's result does not depend on the rank that computes it, (c) the
measurement reduction is MAX over time and SUM over counters."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, per_rank, q):
    for p in (os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tch-geometric_amd")):
        sys.path.insert(0, p)
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "tg_sharding", os.path.join(ROOT, "tch-geometric_amd", "tch_geometric", "sharding.py"))
    sharding = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sharding)
    import orc
    from helpers import load_karate

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ei, n = load_karate()
    ptrs, idx, _ = orc.to_csc(ei, n)
    first, last = sharding.rank_batch_range(rank, world, per_rank)
    seeds = orc.seed_batches(0xBA7C4, first, last - first, 8, n)
    edges, digest = 0, []
    for b in range(first, last):
        s, r, c, e, lo = orc.ns_homo(ptrs, idx, seeds[b - first], [5, 5], orc.rng_philox(0, b))
        edges += len(r)
        digest.append((b, int(s.sum()), int(e.sum()), len(r)))
    sharding.fence(None)
    tmax, tot = sharding.reduce_measurement(0.5 + rank, torch.tensor([edges, last - first], dtype=torch.int64))
    gathered = [None] * world
    dist.all_gather_object(gathered, digest)
    if rank == 0:
        q.put((tmax, tot.tolist(), gathered))
    dist.destroy_process_group()


def test_two_rank_sharding_matches_single_process():
    world, per_rank = 2, 6
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, per_rank, q)) for r in range(world)]
    for p in procs:
        p.start()
    tmax, tot, gathered = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import orc
    from helpers import load_karate
    ei, n = load_karate()
    ptrs, idx, _ = orc.to_csc(ei, n)
    seeds = orc.seed_batches(0xBA7C4, 0, world * per_rank, 8, n)
    expect, edges = [], 0
    for b in range(world * per_rank):
        s, r, c, e, lo = orc.ns_homo(ptrs, idx, seeds[b], [5, 5], orc.rng_philox(0, b))
        expect.append((b, int(s.sum()), int(e.sum()), len(r)))
        edges += len(r)
    assert [d for part in gathered for d in part] == expect      # tiles [0, 12) in order, same results
    assert tmax == 1.5 and tot == [edges, world * per_rank]      # MAX over ranks, SUM of counters


def test_rank_ranges_tile_without_overlap():
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "tg_sharding", os.path.join(ROOT, "tch-geometric_amd", "tch_geometric", "sharding.py"))
    sharding = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sharding)
    for world in (1, 2, 4, 8):
        r = [sharding.rank_batch_range(k, world, 9216) for k in range(world)]
        assert r[0][0] == 0 and all(r[k][1] == r[k + 1][0] for k in range(world - 1)) and r[-1][1] == world * 9216
