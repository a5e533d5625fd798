"""On ONE allocation of the output slabs: does shifting the slabs against each other by a few KB change the time of the
window-ordered launch?  (Separates "bank aliasing between the streams written at equal offsets" from "where the
allocation landed")"""
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
PAD = 1 << 18                                            # int64 elements of slack per slab (2 MB)
base = _cabi.NsBatchedOut(G, B, fan, dev)                # layer_offsets / counts are reused
cn, ce = base.cap_nodes, base.cap_edges
del base.samples, base.rows, base.cols, base.edge_index
flat = [torch.empty(G * c + PAD, dtype=torch.int64, device=dev) for c in (cn, ce, ce, ce)]
ws = _cabi.ns_homo_workspace(G, B, fan, dev)
row, col = _cabi.rmat_edges(scale, n * 16, 0x5EED0000 + scale, dev)
ptrs, idx, _ = _cabi.coo_to_csx(row, col, n, n, True)
del row, col
g = _cabi.graph_view(ptrs, idx, indices32=idx.to(torch.int32), ptrs32=ptrs.to(torch.int32))
seeds = _cabi.seed_batches(0xBA7C4, 0, G, B, n, dev)
res = []
for rep in range(2):
    for skew in (0, 4352, 8192 + 512, 65536 + 256, 2048, 1 << 20):
        s = skew // 8
        base.samples = flat[0][0:G * cn].view(G, cn)
        base.rows = flat[1][1 * s:1 * s + G * ce].view(G, ce)
        base.cols = flat[2][2 * s:2 * s + G * ce].view(G, ce)
        base.edge_index = flat[3][(3 * s) % PAD:(3 * s) % PAD + G * ce].view(G, ce)
        for _ in range(2):
            _cabi.ns_homo_batched(g, seeds, fan, 0, 0, base, ws=ws, form=1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            _cabi.ns_homo_batched(g, seeds, fan, 0, 0, base, ws=ws, form=1)
        e1.record()
        torch.cuda.synchronize()
        res.append({"rep": rep, "skew_bytes": skew, "ms_per_launch": e0.elapsed_time(e1) / 4})
print(json.dumps(res))
