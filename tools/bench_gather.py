"""Throughput of tg_gather_rows (the feature gather after sampling) on the node ids one cfg2 launch produces:
RMAT-24, 64 batches x 1024 seeds, fanout [15,10]; feature matrix [2^24, D].  Algorithmic bytes per gathered row =
2 * row_bytes + 8 (read the row, write the row, read the index).  Prints one JSON object."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tch-geometric_amd"))
from tch_geometric import _cabi  # noqa: E402

dev = torch.device("cuda:0")
scale = int(os.environ.get("SCALE", "24"))
n = 1 << scale
row, col = _cabi.rmat_edges(scale, n * 16, 0x5EED0000 + scale, dev)
ptrs, idx, _ = _cabi.coo_to_csx(row, col, n, n, True)
del row, col
nb, B, fan = int(os.environ.get("BATCHES", "64")), 1024, [15, 10]
seeds = _cabi.seed_batches(0xBA7C4, 0, nb, B, n, dev)
out = _cabi.NsBatchedOut(nb, B, fan, dev)
_cabi.ns_homo_batched(_cabi.graph_view(ptrs, idx), seeds, fan, 0, 0, out)
counts = out.counts.cpu()
index = torch.cat([out.samples[b, :int(counts[b, 0])] for b in range(nb)]).contiguous()
del ptrs, idx, out
res = {"config": "RMAT-%d node ids of %d batches x %d seeds, fanout %s" % (scale, nb, B, fan), "rows": index.numel(),
       "distinct_rows": int(torch.unique(index).numel())}
for name, dtype, d in (("f32_x128", torch.float32, 128), ("bf16_x128", torch.bfloat16, 128), ("f32_x32", torch.float32, 32),
                       ("f32_x602", torch.float32, 602)):
    feat = torch.empty((n, d), dtype=dtype, device=dev)
    feat.view(torch.uint8)[:] = 1
    row_bytes = d * feat.element_size()
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    for _ in range(3):
        _cabi.gather_rows(feat, index, status)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        _cabi.gather_rows(feat, index, status)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    # torch's own index_select on the same inputs, for scale
    for _ in range(3):
        feat.index_select(0, index)
    e0.record()
    for _ in range(reps):
        feat.index_select(0, index)
    e1.record()
    torch.cuda.synchronize()
    ms_t = e0.elapsed_time(e1) / reps
    alg = index.numel() * (2 * row_bytes + 8)
    res[name] = {"row_bytes": row_bytes, "ms": ms, "algorithmic_GBps": alg / ms / 1e6, "frac_of_8TBps": alg / ms / 1e6 / 8000.0,
                 "rows_per_s": index.numel() / ms * 1e3, "torch_index_select_ms": ms_t}
    del feat
print(json.dumps(res, indent=1))
