import os, sys, torch, time
sys.path.insert(0, os.path.join(os.getcwd(), "tch-geometric_amd"))
from tch_geometric import _cabi
dev = torch.device("cuda:0")
scale = 24; n = 1 << scale
row, col = _cabi.rmat_edges(scale, n * 16, 0x5EED0000 + scale, dev)
ptrs, idx, perm = _cabi.coo_to_csx(row, col, n, n, False)
del row, col, perm
g = _cabi.graph_view(ptrs, idx)
gen = torch.Generator(device=dev); gen.manual_seed(3)
ets = torch.randint(0, 100, (idx.numel(),), device=dev, generator=gen)
nts = torch.randint(0, 100, (n,), device=dev, generator=gen)
for nw in (1 << 16, 1 << 18, 1 << 19, 1 << 20):
    start = _cabi.seed_batches(0x57A27, 0, 1, nw, n, dev)[0].contiguous()
    sts = torch.randint(0, 50, (nw,), device=dev, generator=gen)
    _cabi.tempo_random_walk(g, nts, ets, start, sts, 20, (0, 30), 0, 0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for r in range(3): _cabi.tempo_random_walk(g, nts, ets, start, sts, 20, (0, 30), 0, r + 1)
    torch.cuda.synchronize(); print(nw, (time.perf_counter() - t0) / 3 * 1e3, "ms", flush=True)
