"""Measures the ceiling for independent random 8-byte gathers (calibration for DESIGN.md / profiles/)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tch-geometric_amd"))
from tch_geometric import _cabi  # noqa: E402

dev = torch.device("cuda:0")
n_table = 1 << 28  # 2 GiB of int64, same size as RMAT-24's `indices`
table = torch.arange(n_table, dtype=torch.int64, device=dev)
res = {}
for waves_per_cu in (8, 16, 32):
    n_threads = 256 * waves_per_cu * 64
    per_thread = 256
    _cabi.probe_random_gather(table, n_threads, per_thread)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    reps = 5
    for r in range(reps):
        _cabi.probe_random_gather(table, n_threads, per_thread, seed=r + 2)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    g = n_threads * per_thread
    res["waves_per_cu_%d" % waves_per_cu] = {"gathers": g, "ms": ms, "Ggathers_per_s": g / ms / 1e6,
                                             "useful_GBps": 8 * g / ms / 1e6, "sector64_GBps": 64 * g / ms / 1e6}
print(json.dumps(res, indent=1))
