"""Random 8-byte gather rate by table size (which level of the hierarchy serves the lines): calibration for the
window-partitioned gather of ns_homo."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tch-geometric_amd"))
from tch_geometric import _cabi  # noqa: E402

dev = torch.device("cuda:0")
big = torch.arange(1 << 28, dtype=torch.int64, device=dev)
res = {}
sizes = [int(x) for x in sys.argv[1:]] or [18, 20, 21, 22, 23, 24, 26, 27, 28, 30, 31]   # log2 of the table's bytes
for log2_bytes in sizes:
    table = big[:1 << (log2_bytes - 3)]
    n_threads, per_thread = 256 * 16 * 64, 256
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
    res["table_%g_KiB" % ((1 << log2_bytes) / 1024)] = {"gathers": g, "ms": ms, "Ggathers_per_s": g / ms / 1e6}
print(json.dumps(res, indent=1))
