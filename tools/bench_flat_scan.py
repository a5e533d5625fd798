"""Many-batch sampling under a temporal filter / with weights through the WHOLE-DEVICE flat hops (PartitionedSampler with
one rank: tg_part_requests -> tg_part_unpack -> tg_ns_hop_scan / tg_ns_hop_weighted -> tg_part_pack -> tg_part_emit)
against the one-workgroup-per-batch launch (tg_ns_homo_batched) on RMAT-24: same outputs, bit for bit.  Prints one JSON."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tch-geometric_amd"))
from tch_geometric import _cabi, partitioned  # noqa: E402

dev = torch.device("cuda:0")
scale = int(os.environ.get("SCALE", "24"))
G = int(os.environ.get("BATCHES", "64"))
n = 1 << scale
row, col = _cabi.rmat_edges(scale, n * 16, 0x5EED0000 + scale, dev)
ptrs, idx, _ = _cabi.coo_to_csx(row, col, n, n, True)
del row, col
gen = torch.Generator(device=dev)
gen.manual_seed(5)
ts = torch.randint(0, 100, (idx.numel(),), device=dev, generator=gen)
w = torch.rand(idx.numel(), device=dev, generator=gen, dtype=torch.float64) + 0.1
fan, B = [15, 10], 1024
seeds = _cabi.seed_batches(0xBA7C4, 0, G, B, n, dev)
states = torch.full((G, B), 50, dtype=torch.int64, device=dev)
res = {"config": "RMAT-%d, %d batches x %d seeds, fanout %s" % (scale, G, B, fan)}


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, out


for name, kw in (("temporal_static_window_half", dict(filter_mode=_cabi.FILTER_STATIC, window=(0, 49))),
                 ("weighted", dict(sampler=_cabi.SAMPLER_WEIGHTED)),
                 ("weighted_temporal", dict(sampler=_cabi.SAMPLER_WEIGHTED, filter_mode=_cabi.FILTER_STATIC, window=(0, 49)))):
    filtered = "filter_mode" in kw
    shard = partitioned.CscShard.from_full(ptrs, idx, 0, 1, weights=w if "sampler" in kw else None,
                                           timestamps=ts if filtered else None)
    ps = partitioned.PartitionedSampler(shard, G, B, fan, **kw)
    ms, out = timed(lambda: ps.sample(seeds, 7, 100, seeds_state=states if filtered else None))
    edges = int(out.counts[:, 1].sum().item())
    g = _cabi.graph_view(ptrs, idx, weights=w if "sampler" in kw else None, timestamps=ts if filtered else None)
    ref = _cabi.NsBatchedOut(G, B, fan, dev, with_states=filtered)
    ms_b, _ = timed(lambda: _cabi.ns_homo_batched(g, seeds, fan, 7, 100, ref, sampler=kw.get("sampler", _cabi.SAMPLER_UNIFORM),
                                                   filter_mode=kw.get("filter_mode", _cabi.FILTER_NONE), window=kw.get("window", (0, 0)),
                                                   seeds_state=states if filtered else None))
    same = bool(torch.equal(out.counts, ref.counts))
    # round 4: the C ABI's own switch -- tg_ns_homo_batched_ws with the workspace of tg_ns_homo_batched_workspace_bytes
    ws = _cabi.ns_homo_batched_workspace(g, G, B, fan, dev, sampler=kw.get("sampler", _cabi.SAMPLER_UNIFORM),
                                         filter_mode=kw.get("filter_mode", _cabi.FILTER_NONE))
    own = _cabi.NsBatchedOut(G, B, fan, dev, with_states=filtered)
    ms_o, _ = timed(lambda: _cabi.ns_homo_batched(g, seeds, fan, 7, 100, own, sampler=kw.get("sampler", _cabi.SAMPLER_UNIFORM),
                                                   filter_mode=kw.get("filter_mode", _cabi.FILTER_NONE), window=kw.get("window", (0, 0)),
                                                   seeds_state=states if filtered else None, ws=ws)) if ws is not None else (None, None)
    same_own = None
    if ws is not None:   # used prefixes of batch 0 and the last batch, and every count
        cnt = ref.counts.cpu()
        same_own = bool(torch.equal(own.counts, ref.counts)) and all(
            all(torch.equal(x, y) for x, y in zip(own.batch(j, cnt)[:4], ref.batch(j, cnt)[:4])) for j in (0, G - 1))
    res[name] = {"flat_hops_ms": ms, "per_batch_workgroups_ms": ms_b, "c_abi_own_switch_ms": ms_o, "sampled_edges": edges,
                 "flat_G_edges_per_s": edges / ms / 1e6, "same_counts": same, "own_switch_same_output": same_own}
    print(json.dumps({name: res[name]}), file=sys.stderr, flush=True)
    del ps, ref
print(json.dumps(res))
