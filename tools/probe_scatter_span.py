"""74 M scattered 80-byte row writes (what K4 does with its sample runs) into destinations of different spans:
does the SPAN of the scatter target, at equal bytes written, change the rate?  torch.index_copy_ over rows of 10 int64."""
import json
import time

import torch

dev = torch.device("cuda:0")
n = 74_000_000
src = torch.ones((n, 10), dtype=torch.int64, device=dev)
res = []
for span_gb in (3.3, 5.5, 11, 22, 44):
    rows = int(span_gb * 1e9 / 80)
    dst = torch.empty((rows, 10), dtype=torch.int64, device=dev)
    idx = torch.randint(0, rows, (n,), device=dev)
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        dst.index_copy_(0, idx, src)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    res.append({"span_GB": span_gb, "ms": round(dt * 1e3, 3), "GBps_written": round(n * 80 / dt / 1e9, 1)})
    del dst, idx
print(json.dumps(res))
