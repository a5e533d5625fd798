"""Throughput of the scan path (temporal filter / weighted sampler) of neighbor sampling on RMAT-24: every frontier
vertex's whole column is streamed (timestamps, and weights), so the algorithmic bytes are 8 B (16 B) per inspected
edge -- full cache lines, unlike the gathers of the unfiltered path.  Prints one JSON object."""
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
E = idx.numel()
g = torch.Generator(device=dev)
g.manual_seed(1)
ts = torch.randint(0, 100, (E,), device=dev, generator=g)
w = torch.rand(E, device=dev, generator=g, dtype=torch.float64) + 0.1
deg = ptrs[1:] - ptrs[:-1]
nb, B, fan = int(os.environ.get("BATCHES", "64")), 1024, [15, 10]
seeds = _cabi.seed_batches(0xBA7C4, 0, nb, B, n, dev)
res = {"config": "RMAT-%d, %d batches x %d seeds, fanout %s" % (scale, nb, B, fan)}
VARIANTS = (
    ("temporal_static_window_half", dict(filter_mode=0, window=(0, 49), ts=True), 8),
    ("temporal_relative_backward", dict(filter_mode=1, forward=False, window=(0, 30), ts=True), 8),
    ("weighted", dict(sampler=2, wts=True), 8),
    ("weighted_temporal", dict(sampler=2, filter_mode=0, window=(0, 49), ts=True, wts=True), 16))
only = os.environ.get("ONLY")
for name, kw, per_edge in VARIANTS:
    if only and name not in only.split(","):
        continue
    graph = _cabi.graph_view(ptrs, idx, w if kw.get("wts") else None, ts if kw.get("ts") else None)
    out = _cabi.NsBatchedOut(nb, B, fan, dev, with_states=kw.get("filter_mode", -1) != -1)
    st = torch.full((nb, B), 50, dtype=torch.int64, device=dev) if kw.get("filter_mode", -1) != -1 else None

    def run(call):
        _cabi.ns_homo_batched(graph, seeds, fan, 0, call, out, sampler=kw.get("sampler", 0),
                              filter_mode=kw.get("filter_mode", -1), forward=kw.get("forward", False),
                              window=kw.get("window", (0, 0)), seeds_state=st)
    run(0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    run(0)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    # inspected edges = sum of deg over every expanded vertex (seeds + hop-1 samples)
    inspected = 0
    for b in range(nb):
        front = int(out.layer_offsets[b, len(fan) - 1, 0])
        inspected += int(deg[out.samples[b, :front]].sum())
    edges = int(out.counts[:, 1].sum())
    res[name] = {"ms": ms, "inspected_edges": inspected, "sampled_edges": edges,
                 "algorithmic_GBps": per_edge * inspected / ms / 1e6,
                 "frac_of_8TBps": per_edge * inspected / ms / 1e6 / 8000.0, "sampled_edges_per_s": edges / ms * 1e3}
# one filtered call through the operator surface (whole-device flat hops, tg_ns_hop_scan)
import time  # noqa: E402
import tch_geometric as tg  # noqa: E402
flt = (tg.TemporalEdgeFilter((0, 49), ts, False, tg.TEMPORAL_SAMPLE_STATIC), torch.full((B,), 50, dtype=torch.int64, device=dev))
tg.seed(0)
for _ in range(3):
    o = tg.neighbor_sampling_homogenous(ptrs, idx, seeds[0], fan, None, flt)
torch.cuda.synchronize()
t0 = time.perf_counter()
for j in range(10):
    o = tg.neighbor_sampling_homogenous(ptrs, idx, seeds[j % nb], fan, None, flt)
torch.cuda.synchronize()
res["temporal_static_single_call_through_surface"] = {"ms_per_call": (time.perf_counter() - t0) / 10 * 1e3,
                                                       "sampled_edges": int(o[1].numel())}
wsampler = tg.WeightedEdgeSampler(w)
for _ in range(2):
    o = tg.neighbor_sampling_homogenous(ptrs, idx, seeds[0], fan, wsampler)
torch.cuda.synchronize()
t0 = time.perf_counter()
for j in range(5):
    o = tg.neighbor_sampling_homogenous(ptrs, idx, seeds[j % nb], fan, wsampler)
torch.cuda.synchronize()
res["weighted_single_call_through_surface"] = {"ms_per_call": (time.perf_counter() - t0) / 5 * 1e3,
                                                "sampled_edges": int(o[1].numel())}
print(json.dumps(res))
