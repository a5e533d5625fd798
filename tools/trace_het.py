"""cfg4 under a temporal filter / with weights: a fixed number of neighbor_sampling_heterogenous calls, for
`rocprofv3 --kernel-trace --stats -- python3 tools/trace_het.py [filtered|weighted|hgt] [calls]` (kernels per call and
their busy time against the wall time printed here)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tch-geometric_amd"))
import tch_geometric as tg  # noqa: E402
from tch_geometric import _cabi  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "filtered"
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 50
dev = torch.device("cuda:0")
scales = {"A": 23, "B": 22, "C": 22}
edge_types = [("A", "e0", "A"), ("A", "e1", "B"), ("B", "e2", "A"), ("B", "e3", "C"), ("C", "e4", "A")]
E = int(os.environ.get("EDGES", 20_000_000))
P, I = {}, {}
for r, (s, _, d) in enumerate(edge_types):
    row, col = _cabi.rmat_edges_rect(scales[s], scales[d], E, 0xC0F4 + r, dev)
    key = "%s__%s__%s" % (s, edge_types[r][1], d)
    P[key], I[key], _ = _cabi.coo_to_csx(row, col, 1 << scales[s], 1 << scales[d], True)
node_types = ["A", "B", "C"]
nn = {k: [15, 10] for k in P}
g = torch.Generator(device=dev)
g.manual_seed(3)
hts = {k: torch.randint(0, 100, (I[k].numel(),), device=dev, generator=g) for k in P}
hw = {k: torch.rand(I[k].numel(), device=dev, generator=g, dtype=torch.float64) + 0.1 for k in P}
tg.seed(1)
seeds = [_cabi.seed_batches(0xBA7C4, c, 1, 1024, 1 << 23, dev)[0].contiguous() for c in range(calls + 1)]
ns = {t: [512, 512] for t in node_types}


if what == "homo_weighted":   # the homogeneous operator with weights on RMAT-24 (hub columns of 10^5 edges)
    row, col = _cabi.rmat_edges(24, 16 << 24, 0x5EED0000 + 24, dev)
    HP, HI, _ = _cabi.coo_to_csx(row, col, 1 << 24, 1 << 24, True)
    del row, col
    HW = torch.rand(HI.numel(), device=dev, generator=g, dtype=torch.float64) + 0.1
    seeds = [_cabi.seed_batches(0xBA7C4, c, 1, 1024, 1 << 24, dev)[0].contiguous() for c in range(calls + 1)]


def call(sd):
    if what == "homo_weighted":
        return tg.neighbor_sampling_homogenous(HP, HI, sd, [15, 10], tg.WeightedEdgeSampler(HW))
    if what == "filtered":
        flt_h = (tg.TemporalEdgeFilter((0, 49), hts, False, tg.TEMPORAL_SAMPLE_STATIC), {"A": torch.full_like(sd, 50)})
        return tg.neighbor_sampling_heterogenous(node_types, edge_types, P, I, {"A": sd}, nn, 2, None, flt_h)
    if what == "weighted":
        return tg.neighbor_sampling_heterogenous(node_types, edge_types, P, I, {"A": sd}, nn, 2, tg.WeightedEdgeSampler(hw))
    if what == "hgt":
        return tg.hgt_sampling(node_types, edge_types, P, I, None, {"A": sd}, None, ns, 2)
    return tg.neighbor_sampling_heterogenous(node_types, edge_types, P, I, {"A": sd}, nn, 2)


call(seeds[0])
torch.cuda.synchronize()
print("TRACE_BEGIN", flush=True)
t0 = time.perf_counter()
for c in range(calls):
    call(seeds[c + 1])
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("%s: %d calls, %.3f ms per call (wall)" % (what, calls, dt / calls * 1e3), flush=True)
