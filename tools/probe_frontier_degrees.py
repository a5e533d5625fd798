import sys, os, torch
sys.path.insert(0, "/root/repo/tch-geometric_amd")
from tch_geometric import _cabi
dev = torch.device("cuda:0")
n = 1 << 24
row, col = _cabi.rmat_edges(24, n * 16, 0x5EED0000 + 24, dev)
ptrs, idx, _ = _cabi.coo_to_csx(row, col, n, n, True)
del row, col
g = _cabi.graph_view(ptrs, idx)
G, B, fan = 1024, 1024, [15, 10]
seeds = _cabi.seed_batches(0xBA7C4, 0, G, B, n, dev)
out = _cabi.NsBatchedOut(G, B, fan, dev)
_cabi.ns_homo_batched(g, seeds, fan, 0, 0, out)
torch.cuda.synchronize()
lo = out.layer_offsets  # [G, 2, 3]
n1 = lo[:, 1, 0] - B   # hop-1 samples per batch
ar = torch.arange(out.samples.shape[1], device=dev)[None, :]
mask = (ar >= B) & (ar < (B + n1)[:, None])
front = out.samples[mask]
deg = ptrs[front + 1] - ptrs[front]
w = deg.clamp(max=10).double()
print("items", front.numel(), "gathers", int(w.sum()), "mean deg", float(deg.double().mean()))
for t in (16, 64, 256, 1024, 2048, 4096, 8192, 16384, 32768, 65536, 131072, 1 << 20):
    print(t, "items<=", round(float((deg <= t).double().mean()), 4), "gathers<=", round(float(w[deg <= t].sum() / w.sum()), 4))
