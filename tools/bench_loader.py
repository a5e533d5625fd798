"""End-to-end mini-batch throughput: seeds -> batched sampling -> feature / label gather -> mini-batch views
(tch_geometric.loader.NeighborLoader) on RMAT-24 with a [2^24, D] float32 feature matrix, batch 1024, fanout [15,10].
Prints one JSON object."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tch-geometric_amd"))
from tch_geometric import _cabi  # noqa: E402
from tch_geometric.loader import NeighborLoader  # noqa: E402
from tch_geometric.transforms import Graph  # noqa: E402

dev = torch.device("cuda:0")
scale, D = int(os.environ.get("SCALE", "24")), int(os.environ.get("DIM", "128"))
n = 1 << scale
row, col = _cabi.rmat_edges(scale, n * 16, 0x5EED0000 + scale, dev)
data = Graph(edge_index=torch.stack([row, col]), num_nodes=n)
del row, col
data.x = torch.empty((n, D), dtype=torch.float32, device=dev)
data.x.view(torch.int32)[:] = 1
data.y = torch.zeros(n, dtype=torch.int64, device=dev)
n_batches = int(os.environ.get("BATCHES", "2048"))
seeds = _cabi.seed_batches(0xBA7C4, 0, n_batches, 1024, n, dev).reshape(-1)
res = {"config": "RMAT-%d, x [%d, %d] f32, %d mini-batches of 1024 seeds, fanout [15, 10]" % (scale, n, D, n_batches)}
for prefetch in (1, 16, 256):
    t_build = time.perf_counter()
    loader = NeighborLoader(data, [15, 10], input_nodes=seeds, batch_size=1024, prefetch=prefetch)
    torch.cuda.synchronize()
    t_build = time.perf_counter() - t_build
    it = iter(loader)
    for _ in range(min(prefetch, 32)):
        next(it)                                             # warm-up launch
    torch.cuda.synchronize()
    edges = nodes = nb = 0
    budget = max(prefetch, min(n_batches - prefetch, 64 if prefetch == 1 else 1024))
    t0 = time.perf_counter()
    for b in it:
        edges += b.e_id.numel()
        nodes += b.num_nodes
        nb += 1
        if nb >= budget:
            break
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    res["prefetch_%d" % prefetch] = {"mini_batches_per_s": nb / dt, "sampled_edges_per_s": edges / dt,
                                     "feature_GBps": nodes * D * 4 / dt / 1e9, "ms_per_mini_batch": dt / nb * 1e3,
                                     "loader_setup_s": t_build}
    del loader, it
print(json.dumps(res))
