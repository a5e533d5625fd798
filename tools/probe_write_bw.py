"""The chip's streaming WRITE rate (what bounds the emit kernel of the window-ordered launch): torch
fill_ / copy_ over buffers far larger than the Infinity Cache."""
import json
import torch

dev = torch.device("cuda:0")
res = {}
for gb in (1, 4, 10):
    n = gb * (1 << 30) // 8
    a = torch.empty(n, dtype=torch.int64, device=dev)
    b = torch.empty(n, dtype=torch.int64, device=dev)
    for name, fn, bytes_moved in (("fill", lambda: a.fill_(7), n * 8), ("copy", lambda: b.copy_(a), n * 16),
                                  ("arange_like_add", lambda: torch.add(a, 1, out=b), n * 16)):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        res["%s_%dGB" % (name, gb)] = {"ms": ms, "TBps": bytes_moved / ms / 1e9}
    del a, b
print(json.dumps(res, indent=1))
