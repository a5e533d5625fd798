"""One window-ordered launch over 16 384 batches against the same batches as TWO (or four) smaller launches in flight on
separate HIP streams, inside one process: is a launch worth splitting into concurrent parts?"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tch-geometric_amd"))
from tch_geometric import _cabi  # noqa: E402

dev = torch.device("cuda:0")
G, B, fan, scale = 16384, 1024, [15, 10], 24
n = 1 << scale
whole = (_cabi.NsBatchedOut(G, B, fan, dev), _cabi.ns_homo_workspace(G, B, fan, dev))
P = int(os.environ.get("PARTS", 2))   # (whole + parts = 200 GB: one split at a time)
parts = {P: [(_cabi.NsBatchedOut(G // P, B, fan, dev), _cabi.ns_homo_workspace(G // P, B, fan, dev)) for _ in range(P)]}
row, col = _cabi.rmat_edges(scale, n * 16, 0x5EED0000 + scale, dev)
ptrs, idx, _ = _cabi.coo_to_csx(row, col, n, n, True)
del row, col
g = _cabi.graph_view(ptrs, idx, indices32=idx.to(torch.int32), ptrs32=ptrs.to(torch.int32))
seeds = _cabi.seed_batches(0xBA7C4, 0, G, B, n, dev)
streams = [torch.cuda.Stream(device=dev) for _ in range(4)]
cur = torch.cuda.current_stream(dev)
for rep in range(3):
    for mode in (1, P):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for it in range(4):
            if mode == 1:
                _cabi.ns_homo_batched(g, seeds, fan, 0, 0, whole[0], ws=whole[1], form=1)
            else:
                gp = G // mode
                for j in range(mode):
                    streams[j].wait_stream(cur)
                    with torch.cuda.stream(streams[j]):
                        _cabi.ns_homo_batched(g, seeds[j * gp:(j + 1) * gp], fan, 0, j * gp, parts[mode][j][0],
                                              ws=parts[mode][j][1], form=1)
                for j in range(mode):
                    cur.wait_stream(streams[j])
        e1.record()
        torch.cuda.synchronize()
        print(rep, "parts", mode, round(e0.elapsed_time(e1) / 4, 3), flush=True)
