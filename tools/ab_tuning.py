"""Same-process A/B of TUNING VARIANTS of the window-ordered launch (tg_ns_win_tuning_set) on the bench launch (RMAT-24,
G batches of 1 024 seeds, [15, 10], u32 shadows), alternating, on one set of slabs: ms per launch and per stage (HIP
events between the kernels, tg_ns_win_stage_timing).  Also the speed of light of the output contract (tg_probe_ns_sol).
  python tools/ab_tuning.py "name:key=val,key=val" ...      (no variants: the default tuning only)
Prints JSON lines; `G` in the environment = batches per launch (default 16 384)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tch-geometric_amd"))
from tch_geometric import _cabi  # noqa: E402

dev = torch.device("cuda:0")
G, B, fan, scale = int(os.environ.get("G", 16384)), 1024, [15, 10], 24
ROUNDS = int(os.environ.get("ROUNDS", 3))
n = 1 << scale
out = _cabi.NsBatchedOut(G, B, fan, dev)
out2 = _cabi.NsBatchedOut(G, B, fan, dev) if os.environ.get("SOL", "1") == "1" else None
row, col = _cabi.rmat_edges(scale, n * 16, 0x5EED0000 + scale, dev)
ptrs, idx, _ = _cabi.coo_to_csx(row, col, n, n, True)
del row, col
g = _cabi.graph_view(ptrs, idx, indices32=idx.to(torch.int32), ptrs32=ptrs.to(torch.int32), max_degree="auto")
ws = _cabi.ns_homo_workspace(G, B, fan, dev, staged=True, graph=g)   # sized for both pipelines
seeds = _cabi.seed_batches(0xBA7C4, 0, G, B, n, dev)

variants = [("default", {})]
for spec in sys.argv[1:]:
    name, _, kv = spec.partition(":")
    variants.append((name, {k: int(v) for k, v in (x.split("=") for x in kv.split(",") if x)}))


def launch():
    _cabi.ns_homo_batched(g, seeds, fan, 0, 0, out, ws=ws, form=1)


def timed(reps=4):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        launch()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


base = _cabi.ns_win_tuning()
launch()
torch.cuda.synchronize()
edges = int(out.counts[:, 1].sum())
slots = int(out.layer_offsets[:, len(fan) - 1, 0].sum())
alg = 24 * slots + 40 * edges + 16 * B * G
print(json.dumps({"batches": G, "sampled_edges": edges, "frontier_slots": slots, "algorithmic_GB": alg / 1e9,
                  "max_degree": g.max_degree, "workspace_GB": ws.numel() * 8 / 1e9, "tuning": base}), flush=True)
res = {}
for rnd in range(ROUNDS):
    for name, kv in variants:
        _cabi.ns_win_tuning_set(**kv)
        try:
            launch()
            launch()
            ms = timed()
            _cabi.ns_win_stage_timing(True)
            launch()
            st = _cabi.ns_win_stage_times()
            _cabi.ns_win_stage_timing(False)
        finally:
            _cabi.ns_win_tuning_set(**base)
        res.setdefault(name, []).append(ms)
        print(json.dumps({"round": rnd, "variant": name, "knobs": kv, "ms_per_launch": round(ms, 3),
                          "roofline_frac": round(alg / ms / 1e6 / 8000, 4),
                          "stages_ms": {k: round(v, 3) for k, v in st}}), flush=True)
if out2 is not None:   # speed of light of the output contract: the algorithmic bytes as pure streams, other slabs
    launch()
    torch.cuda.synchronize()
    for rep in range(2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            _cabi.probe_ns_sol(out, out2, seeds, len(fan))
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 4
    print(json.dumps({"speed_of_light_of_the_output_contract_ms": round(ms, 3), "GBps": round(alg / ms / 1e6, 1),
                      "frac_of_8TBps": round(alg / ms / 1e6 / 8000, 4)}), flush=True)
print(json.dumps({"summary_ms_min": {k: round(min(v), 3) for k, v in res.items()}}))
