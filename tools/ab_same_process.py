"""A/B of library builds INSIDE one process, on the same buffers: `python tools/ab_same_process.py libA.so libB.so ...`
(paths relative to the repo root; "-" = the default build).  Where a process's arenas land decides +-5 % of the launch
time, so builds compared across processes drown in that; on one allocation the launch time repeats to
0.01 ms.  Prints ms per launch of the default bench launch (RMAT-24, 16 384 batches) per build, three alternating rounds.
Environment knobs of the library are read once per library instance, so `ENV=...` prefixes apply to all builds alike."""
import ctypes as C
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
out = _cabi.NsBatchedOut(G, B, fan, dev)
row, col = _cabi.rmat_edges(scale, n * 16, 0x5EED0000 + scale, dev)
ptrs, idx, _ = _cabi.coo_to_csx(row, col, n, n, True)
del row, col
g = _cabi.graph_view(ptrs, idx, indices32=idx.to(torch.int32), ptrs32=ptrs.to(torch.int32), max_degree="auto")
ws = _cabi.ns_homo_workspace(G, B, fan, dev, staged=True, graph=g)   # sized for the staged pipeline (what the launch takes at this size)
ws = torch.empty(ws.numel() + (128 << 20), dtype=torch.int64, device=dev)   # slack: a variant may lay its workspace out larger
seeds = _cabi.seed_batches(0xBA7C4, 0, G, B, n, dev)
default = _cabi.lib
libs = []
for path in sys.argv[1:]:
    if path == "-":
        libs.append(("default", default))
    else:
        h = C.CDLL(os.path.join(ROOT, path))
        h.tg_version.restype = C.c_char_p
        h.tg_last_error.restype = C.c_char_p
        libs.append((os.path.basename(path), h))
res = []
for rnd in range(3):
    for name, h in libs:
        _cabi.lib = h
        for _ in range(2):
            _cabi.ns_homo_batched(g, seeds, fan, 0, 0, out, ws=ws, form=1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            _cabi.ns_homo_batched(g, seeds, fan, 0, 0, out, ws=ws, form=1)
        e1.record()
        torch.cuda.synchronize()
        res.append((rnd, name, round(e0.elapsed_time(e1) / 4, 3)))
_cabi.lib = default
for r in res:
    print(*r)
