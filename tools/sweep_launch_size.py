"""The launch at the sizes a loader uses: ms and roofline fraction of tg_ns_homo_batched_ws by batches per launch and form
(fused per-batch kernel, window-ordered push / staged pipelines, what AUTO takes), RMAT-24, 1 024 seeds, [15, 10].
  python tools/sweep_launch_size.py [batches per launch ...]      (default 256 1024 4096)
Prints one JSON line per size."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tch-geometric_amd"))
from tch_geometric import _cabi  # noqa: E402

dev = torch.device("cuda:0")
B, fan, scale = 1024, [15, 10], 24
n = 1 << scale
sizes = [int(x) for x in sys.argv[1:]] or [256, 1024, 4096]
row, col = _cabi.rmat_edges(scale, n * 16, 0x5EED0000 + scale, dev)
ptrs, idx, _ = _cabi.coo_to_csx(row, col, n, n, True)
del row, col
g = _cabi.graph_view(ptrs, idx, indices32=idx.to(torch.int32), ptrs32=ptrs.to(torch.int32), max_degree="auto")
base = _cabi.ns_win_tuning()
VARIANTS = [("fused", 2, {}), ("push", 1, dict(staged=0)), ("staged", 1, dict(staged=1, stage_parts=1)),
            ("auto", 0, {})]


def timed(fn, reps):
    fn()
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for G in sizes:
    out = _cabi.NsBatchedOut(G, B, fan, dev)
    ws = _cabi.ns_homo_workspace(G, B, fan, dev, staged=True, graph=g)
    seeds = _cabi.seed_batches(0xBA7C4, 0, G, B, n, dev)
    _cabi.ns_homo_batched(g, seeds, fan, 0, 0, out, form=2)
    torch.cuda.synchronize()
    edges = int(out.counts[:, 1].sum())
    slots = int(out.layer_offsets[:, len(fan) - 1, 0].sum())
    alg = 24 * slots + 40 * edges + 16 * B * G
    res = {"batches_per_launch": G, "sampled_edges": edges, "algorithmic_MB": alg / 1e6, "forms": {}}
    for name, form, kv in VARIANTS:
        _cabi.ns_win_tuning_set(**kv)
        try:
            ms = min(timed(lambda: _cabi.ns_homo_batched(g, seeds, fan, 0, 0, out, ws=ws, form=form),
                           max(4, 16384 // G)) for _ in range(3))
            taken = _cabi.ns_homo_batched_form(g, out, G, B, fan, ws=ws, form=form)[0]
            staged = bool(_cabi.ns_homo_batched_staged(g, out, G, B, fan, ws=ws, form=form))
        finally:
            _cabi.ns_win_tuning_set(**base)
        res["forms"][name] = {"ms": round(ms, 4), "G_edges_per_s": round(edges / ms / 1e6, 2),
                              "roofline_frac": round(alg / ms / 1e6 / 8000, 4),
                              "takes": "fused" if taken == 2 else ("staged" if staged else "push")}
    print(json.dumps(res), flush=True)
    del out, ws, seeds
    torch.cuda.empty_cache()
