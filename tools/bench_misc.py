"""Measurements of the remaining operators on one MI355X: negative sampling (homogeneous, RMAT-24 CSR), device ingest
(to_csc of RMAT-24 / RMAT-26), budget_sampling and heterogeneous negative sampling on the cfg4 graph.  One JSON object."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tch-geometric_amd"))
import tch_geometric as tg  # noqa: E402
from tch_geometric import _cabi  # noqa: E402

dev = torch.device("cuda:0")
res = {}


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps, out


# ---- ingest: tg_coo_to_csx alone (outputs and workspace allocated beforehand, HIP events around the call)
import ctypes as C  # noqa: E402
for scale in (24, 26):
    n = 1 << scale
    nnz = n * 16
    row, col = _cabi.rmat_edges(scale, nnz, 0x5EED0000 + scale, dev)
    o = dict(dtype=torch.int64, device=dev)
    ptrs_o, idx_o, perm_o = torch.empty(n + 1, **o), torch.empty(nnz, **o), torch.empty(nnz, **o)
    nbytes = C.c_int64(0)
    _cabi.check(_cabi.lib.tg_coo_to_csx_workspace_bytes(C.c_int64(nnz), C.c_int64(n), C.c_int64(n), C.byref(nbytes)))
    ws = torch.empty(nbytes.value // 8 + 1, **o)

    def ingest():
        _cabi.check(_cabi.lib.tg_coo_to_csx(_cabi.ptr(row), _cabi.ptr(col), C.c_int64(nnz), C.c_int64(n), C.c_int64(n),
                                            C.c_int32(1), _cabi.ptr(ptrs_o), _cabi.ptr(idx_o), _cabi.ptr(perm_o),
                                            _cabi.ptr(ws), C.c_int64(nbytes.value), _cabi.stream_ptr(dev)))
    ingest()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        ingest()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    res["to_csc_rmat%d" % scale] = {"edges": nnz, "ms": ms, "edges_per_s": nnz / ms * 1e3, "workspace_GB": nbytes.value / 1e9}
    del row, col, ptrs_o, idx_o, perm_o, ws
# ---- negative sampling, homogeneous (CSR of RMAT-24)
n = 1 << 24
row, col = _cabi.rmat_edges(24, n * 16, 0x5EED0000 + 24, dev)
ptrs, idx, _ = _cabi.coo_to_csx(row, col, n, n, False)
del row, col
tg.seed(1)
for n_in in (1024, 1 << 20):
    inputs = _cabi.seed_batches(0x4E47, 0, 1, n_in, n, dev)[0].contiguous()
    dt, out = timed(lambda: tg.negative_sample_neighbors_homogenous(ptrs, idx, (n, n), inputs, 5, 5))
    res["negative_homogeneous_%d_inputs_x5" % n_in] = {"ms_per_call": dt * 1e3, "negatives": int(out[1].numel()),
                                                       "negatives_per_s": int(out[1].numel()) / dt}
del ptrs, idx
# ---- cfg4 graph: budget sampling and heterogeneous negatives
scales = {"A": 23, "B": 22, "C": 22}
edge_types = [("A", "e0", "A"), ("A", "e1", "B"), ("B", "e2", "A"), ("B", "e3", "C"), ("C", "e4", "A")]
node_types = ["A", "B", "C"]
P, I, PR, IR, sizes = {}, {}, {}, {}, {}
for r, (s, nm, d) in enumerate(edge_types):
    rw, cl = _cabi.rmat_edges_rect(scales[s], scales[d], 20_000_000, 0xC0F4 + r, dev)
    key = "%s__%s__%s" % (s, nm, d)
    P[key], I[key], _ = _cabi.coo_to_csx(rw, cl, 1 << scales[s], 1 << scales[d], True)
    PR[key], IR[key], _ = _cabi.coo_to_csx(rw, cl, 1 << scales[s], 1 << scales[d], False)
    sizes[key] = (1 << scales[s], 1 << scales[d])
seeds = _cabi.seed_batches(0xBA7C4, 1, 1, 1024, 1 << 23, dev)[0].contiguous()
nn = {t: [15, 10] for t in node_types}
dt, out = timed(lambda: tg.budget_sampling(node_types, edge_types, P, I, None, {"A": seeds}, None, nn, 2, None, False, False))
res["budget_sampling_cfg4"] = {"ms_per_call": dt * 1e3, "nodes": sum(int(v.numel()) for v in out[0].values()),
                               "edges": sum(int(v.numel()) for v in out[2].values())}
dt, out = timed(lambda: tg.negative_sample_neighbors_heterogenous(node_types, edge_types, PR, IR, sizes, {"A": seeds}, 5, 5, False))
res["negative_heterogeneous_cfg4_1024_inputs_x5"] = {"ms_per_call": dt * 1e3,
                                                     "negatives": sum(int(v.numel()) for v in out[1].values())}
print(json.dumps(res))
