"""Throughput of the range-partitioned sampler (SURVEY 8(e) mode 2 / BASELINE cfg5 shape).

  1 GPU :  python tools/bench_partitioned.py --scale 24
  N GPUs:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
               tools/bench_partitioned.py --scale 27            (one rank per GPU, RCCL all-to-all)
Every rank generates the whole R-MAT edge list chunk by chunk but keeps only the columns it owns."""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tch-geometric_amd"))
from tch_geometric import _cabi, partitioned  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--scale", type=int, default=24)
ap.add_argument("--batches", type=int, default=256, help="seed batches per rank per exchange round")
ap.add_argument("--rounds", type=int, default=4)
ap.add_argument("--batch", type=int, default=1024)
ap.add_argument("--form", default="device", choices=["device", "torch"], help="device kernels (partition.hip) or torch-level protocol")
args = ap.parse_args()

world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
torch.cuda.set_device(dev)
if world > 1:
    dist.init_process_group("nccl", device_id=dev)
n = 1 << args.scale
size = partitioned.CscShard.shard_size_for(n, world)
v_lo, v_hi = min(rank * size, n), min((rank + 1) * size, n)
# build only this rank's shard: edges whose column falls into [v_lo, v_hi)
row, col = _cabi.rmat_edges(args.scale, n * 16, 0x5EED0000 + args.scale, dev)
keep = (col >= v_lo) & (col < v_hi)
row, col = row[keep], col[keep] - v_lo
del keep
_, c2 = _cabi.rmat_edges(args.scale, n * 16, 0x5EED0000 + args.scale, dev)
below = int((c2 < v_lo).sum())      # global edge offset of the shard = edges of the columns before v_lo
del c2
ptrs, idx, _ = _cabi.coo_to_csx(row, col, n, v_hi - v_lo, True)
del row, col
shard = partitioned.CscShard(ptrs, idx, v_lo, v_hi, below, n, size)
torch.cuda.synchronize()

fan = [15, 10]
edges = 0
times = []
for rnd in range(args.rounds + 1):
    first = (rnd * world + rank) * args.batches
    seeds = _cabi.seed_batches(0xBA7C4, first, args.batches, args.batch, n, dev)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    if args.form == "device":
        dres = partitioned.ns_homo_partitioned_device(shard, seeds, fan, 0, first)
    else:
        res = partitioned.ns_homo_partitioned(shard, seeds, fan, 0, first)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if rnd:                        # round 0 is warm-up
        times.append(dt)
        edges += int(dres.counts[:, 1].sum()) if args.form == "device" else sum(int(r.numel()) for _, r, *_ in res)
tot = torch.tensor([edges], dtype=torch.float64, device=dev)
tmax = torch.tensor([sum(times)], dtype=torch.float64, device=dev)
if world > 1:
    dist.all_reduce(tot)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
if rank == 0:
    print(json.dumps({"metric": "sampled edges/sec, range-partitioned neighbor_sampling_homogenous", "n_gpus": world,
                      "scale": args.scale, "form": args.form, "batches_per_round_per_rank": args.batches, "rounds": args.rounds,
                      "value": float(tot) / float(tmax), "unit": "edges/s", "seconds": float(tmax)}))
if world > 1:
    dist.destroy_process_group()
