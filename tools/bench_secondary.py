"""Secondary configs of BASELINE.json on one MI355X (not the headline bench): cfg3 random_walk
(node2vec p=q=1, walk_len 80, 1M starts, RMAT-24) and the general-(p,q) variant.  Prints one JSON object."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tch-geometric_amd"))
from tch_geometric import _cabi  # noqa: E402

dev = torch.device("cuda:0")
scale = int(os.environ.get("SCALE", "24"))
n = 1 << scale
row, col = _cabi.rmat_edges(scale, n * 16, 0x5EED0000 + scale, dev)
ptrs, idx, perm = _cabi.coo_to_csx(row, col, n, n, False)      # CSR for walks
del row, col, perm
g = _cabi.graph_view(ptrs, idx)
shadows = os.environ.get("SHADOWS", "1") == "1"        # u32 shadows of the CSR for the node2vec walk (same results)
g_walk = _cabi.graph_view(ptrs, idx, indices32=idx.to(torch.int32), ptrs32=ptrs.to(torch.int32)) if shadows else g
n_walkers, L = 1 << 20, 80
start = _cabi.seed_batches(0x57A27, 0, 1, n_walkers, n, dev)[0].contiguous()
res = {}
es, es_ms = None, None
for name, p, q in (("p1_q1", 1.0, 1.0), ("p1_q1.5", 1.0, 1.5), ("p1_q1.5_edge_set", 1.0, 1.5)) if os.environ.get("NODE2VEC", "1") == "1" else ():
    if name.endswith("edge_set"):          # has_edge as a hash probe: built once per graph
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        es = _cabi.edge_set(g_walk, dev)
        torch.cuda.synchronize()
        es_ms = (time.perf_counter() - t0) * 1e3
    _cabi.random_walk(g_walk, start, L, p, q, 0, 0, edge_set=es)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 3
    e0.record()
    for r in range(reps):
        w = _cabi.random_walk(g_walk, start, L, p, q, 0, r + 1, edge_set=es)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    steps = int((w[:, 1:] >= 0).sum().item())
    cells = n_walkers * (L + 1)
    alg = 32 * steps + 8 * n_walkers + 8 * (cells - steps)       # 32 B per executed step, 8 B per -1 cell
    res[name] = {"ms": ms, "executed_steps": steps, "steps_per_s": steps / ms * 1e3,
                 "algorithmic_GBps": alg / ms / 1e6, "frac_of_8TBps": alg / ms / 1e6 / 8000.0}
    if es is not None:
        res[name]["edge_set_build_ms"] = es_ms
        res[name]["edge_set_GB"] = es.numel() * 8 / 1e9
        res[name]["same_walks_as_without"] = bool(torch.equal(w, _cabi.random_walk(g_walk, start, L, p, q, 0, reps)))
# ---- temporal walks on the same graph: edge timestamps in [0, 100), start timestamps in [0, 50)
gen = torch.Generator(device=dev)
gen.manual_seed(3)
ets = torch.randint(0, 100, (idx.numel(),), device=dev, generator=gen)
nts = torch.full((n,), -1, dtype=torch.int64, device=dev)
sts = torch.randint(0, 50, (n_walkers,), device=dev, generator=gen)
max_deg = int((ptrs[1:] - ptrs[:-1]).max().item())
Lt = 20


def timed(fn, reps=2):
    fn(0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for r in range(reps):
        out = fn(r + 1)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, out


if os.environ.get("TEMPORAL", "1") == "1":
    ms, (w, t) = timed(lambda c: _cabi.tempo_random_walk(g, nts, ets, start, sts, Lt, (0, 30), 0, c))
    steps = int((w[:, 1:] >= 0).sum().item())
    deg = ptrs[1:] - ptrs[:-1]
    src = w[:, :-1]
    inspected = int(deg[src[src >= 0]].sum().item())         # every position but the last is expanded
    res["tempo_random_walk_len20_window30"] = {"ms": ms, "steps": steps, "steps_per_s": steps / ms * 1e3,
                                               "inspected_edges": inspected,
                                               # SURVEY 8(d) prices an inspected edge at 16 B (index + edge timestamp); since
                                               # round 3 the kernel reads the timestamp only (8 B) and one index per step
                                               "algorithmic_GBps_at_16B_per_edge": 16 * inspected / ms / 1e6,
                                               "moved_GBps": (8 * inspected + 8 * steps) / ms / 1e6,
                                               "moved_frac_of_8TBps": (8 * inspected + 8 * steps) / ms / 1e6 / 8000.0}
    for bias in os.environ.get("BIASES", "uniform,exponential,linear").split(","):
        ms, (w, t, st) = timed(lambda c: _cabi.biased_tempo_random_walk(g, nts, ets, start, sts, Lt, bias, True, 2, 0, c,
                                                                      max_degree=max_deg))
        steps = int((w[:, 1:] >= 0).sum().item())
        res["biased_tempo_random_walk_len20_%s" % bias] = {"ms": ms, "steps": steps, "steps_per_s": steps / ms * 1e3,
                                                          "status": int(st.item())}
print(json.dumps({"config": "random_walk walk_length=80, %d starts, RMAT-%d CSR" % (n_walkers, scale), **res}))
