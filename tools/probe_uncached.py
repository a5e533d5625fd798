"""Does the memory type of the gathered table change the HBM request size of a random 8-byte gather?
Tables: torch allocation (coarse-grained, cached), hipExtMallocWithFlags fine-grained (1) and uncached (3).
Prints gathers/s per table type; run under rocprofv3 --pmc TCC_EA0_RDREQ_* to see the request sizes."""
import ctypes as C
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tch-geometric_amd"))
from tch_geometric import _cabi  # noqa: E402

hip = C.CDLL("libamdhip64.so")
dev = torch.device("cuda:0")
n_table = 1 << 28
src = torch.arange(n_table, dtype=torch.int64, device=dev)
n_threads, per_thread = 256 * 16 * 64, 256
sink = torch.empty(n_threads, dtype=torch.int64, device=dev)
stream = _cabi.stream_ptr(dev)
res = {}
only = os.environ.get("ONLY")
for name, flags in (("torch_default", None), ("finegrained", 1), ("uncached", 3)):
    if only and name not in only.split(","):
        continue
    if flags is None:
        p = C.c_void_p(src.data_ptr())
    else:
        p = C.c_void_p()
        rc = hip.hipExtMallocWithFlags(C.byref(p), C.c_size_t(n_table * 8), C.c_uint(flags))
        if rc != 0:
            res[name] = {"error": "hipExtMallocWithFlags rc=%d" % rc}
            continue
        rc = hip.hipMemcpy(p, C.c_void_p(src.data_ptr()), C.c_size_t(n_table * 8), C.c_int(3))  # device to device
        assert rc == 0, rc
    def run(seed):
        _cabi.check(_cabi.lib.tg_probe_random_gather(p, C.c_int64(n_table), C.c_int64(n_threads), C.c_int64(per_thread),
                                                     C.c_uint64(seed), _cabi.ptr(sink), stream))
    run(1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for r in range(5):
        run(r + 2)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    g = n_threads * per_thread
    res[name] = {"gathers": g, "ms": ms, "Ggathers_per_s": g / ms / 1e6, "checksum": int(sink.sum().item())}
    if flags is not None:
        hip.hipFree(p)
print(json.dumps(res, indent=1))
