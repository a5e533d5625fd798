"""Two window-ordered launches in flight on two HIP streams (two sets of slabs) against the same launches one after the
other, inside one process (same placements for both measurements)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tch-geometric_amd"))
from tch_geometric import _cabi  # noqa: E402

dev = torch.device("cuda:0")
G, B, fan, scale = int(os.environ.get("G", 16384)), 1024, [15, 10], 24
n = 1 << scale
sets = [(_cabi.NsBatchedOut(G, B, fan, dev), _cabi.ns_homo_workspace(G, B, fan, dev)) for _ in range(2)]
row, col = _cabi.rmat_edges(scale, n * 16, 0x5EED0000 + scale, dev)
ptrs, idx, _ = _cabi.coo_to_csx(row, col, n, n, True)
del row, col
g = _cabi.graph_view(ptrs, idx, indices32=idx.to(torch.int32), ptrs32=ptrs.to(torch.int32))
seeds = _cabi.seed_batches(0xBA7C4, 0, G, B, n, dev)
streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
res = []
for rep in range(3):
    for mode in ("sequential", "two_streams"):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        n_l = 8
        if mode == "sequential":
            for i in range(n_l):
                out, ws = sets[i % 2]
                _cabi.ns_homo_batched(g, seeds, fan, 0, i * G, out, ws=ws, form=1)
        else:
            for s in streams:
                s.wait_stream(torch.cuda.current_stream(dev))
            for i in range(n_l):
                out, ws = sets[i % 2]
                with torch.cuda.stream(streams[i % 2]):
                    _cabi.ns_homo_batched(g, seeds, fan, 0, i * G, out, ws=ws, form=1)
            for s in streams:
                torch.cuda.current_stream(dev).wait_stream(s)
        e1.record()
        torch.cuda.synchronize()
        res.append((rep, mode, round(e0.elapsed_time(e1) / n_l, 3)))
for r in res:
    print(*r)
