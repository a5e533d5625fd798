"""Does the time of the window-ordered launch depend on WHICH allocation the slabs live in, inside one process?
Allocates several sets of output slabs + workspace (as many as fit), times the same launch on each (HIP events), prints
ms per launch per set.  (the launch time is two-valued from process to process.)"""
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
sets = []
for j in range(int(os.environ.get("SETS", 2))):
    try:
        sets.append((_cabi.NsBatchedOut(G, B, fan, dev), _cabi.ns_homo_workspace(G, B, fan, dev)))
    except torch.OutOfMemoryError:
        break
row, col = _cabi.rmat_edges(scale, n * 16, 0x5EED0000 + scale, dev)
ptrs, idx, _ = _cabi.coo_to_csx(row, col, n, n, True)
del row, col
g = _cabi.graph_view(ptrs, idx, indices32=idx.to(torch.int32), ptrs32=ptrs.to(torch.int32))
seeds = _cabi.seed_batches(0xBA7C4, 0, G, B, n, dev)
res = []
for rep in range(2):
    for j, (out, ws) in enumerate(sets):
        for _ in range(2):
            _cabi.ns_homo_batched(g, seeds, fan, 0, 0, out, ws=ws, form=1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            _cabi.ns_homo_batched(g, seeds, fan, 0, 0, out, ws=ws, form=1)
        e1.record()
        torch.cuda.synchronize()
        res.append({"rep": rep, "set": j, "ms_per_launch": e0.elapsed_time(e1) / 4,
                    "samples_ptr": hex(out.samples.data_ptr()), "rows_ptr": hex(out.rows.data_ptr())})
print(json.dumps(res, indent=1))
