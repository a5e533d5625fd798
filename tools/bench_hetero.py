"""BASELINE cfg4: neighbor_sampling_heterogenous and hgt_sampling through the operator surface on a synthetic
3-node-type / 5-edge-type graph (A = 2^23, B = 2^22, C = 2^22 nodes; 20 M rectangular R-MAT edges per relation),
1024 seeds of type A, 2 hops.  Per-call latency (the surface is one call per mini-batch).  Prints one JSON object."""
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
scales = {"A": 23, "B": 22, "C": 22}
edge_types = [("A", "e0", "A"), ("A", "e1", "B"), ("B", "e2", "A"), ("B", "e3", "C"), ("C", "e4", "A")]
E = int(os.environ.get("EDGES", 20_000_000))
P, I = {}, {}
for r, (s, _, d) in enumerate(edge_types):
    row, col = _cabi.rmat_edges_rect(scales[s], scales[d], E, 0xC0F4 + r, dev)
    key = "%s__%s__%s" % (s, edge_types[r][1], d)
    P[key], I[key], _ = _cabi.coo_to_csx(row, col, 1 << scales[s], 1 << scales[d], True)
node_types = ["A", "B", "C"]
res = {"config": "3 ntypes (2^23, 2^22, 2^22), 5 etypes x %d edges, 1024 seeds of type A, 2 hops" % E}


def timeit(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = None
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps, out


tg.seed(1)
calls = [0]


def seeds():
    calls[0] += 1
    return _cabi.seed_batches(0xBA7C4, calls[0], 1, 1024, 1 << 23, dev)[0].contiguous()


nn = {k: [15, 10] for k in P}
dt, out = timeit(lambda: tg.neighbor_sampling_heterogenous(node_types, edge_types, P, I, {"A": seeds()}, nn, 2))
edges = sum(int(v.numel()) for v in out[1].values())
res["neighbor_sampling_heterogenous"] = {"ms_per_call": dt * 1e3, "sampled_edges_per_call": edges,
                                         "edges_per_s": edges / dt}
ns = {t: [512, 512] for t in node_types}
dt, out = timeit(lambda: tg.hgt_sampling(node_types, edge_types, P, I, None, {"A": seeds()}, None, ns, 2))
nodes = sum(int(v.numel()) for v in out[0].values())
edges = sum(int(v.numel()) for v in out[2].values())
res["hgt_sampling"] = {"ms_per_call": dt * 1e3, "nodes_per_call": nodes, "edges_per_call": edges,
                       "nodes_plus_edges_per_s": (nodes + edges) / dt}
dt, out = timeit(lambda: tg.neighbor_sampling_homogenous(P["A__e0__A"], I["A__e0__A"], seeds(), [15, 10]))
res["neighbor_sampling_homogenous_single_call"] = {"ms_per_call": dt * 1e3, "sampled_edges_per_call": int(out[1].numel()),
                                                   "edges_per_s": int(out[1].numel()) / dt}
g = torch.Generator(device=dev)
g.manual_seed(3)
ts = torch.randint(0, 100, (I["A__e0__A"].numel(),), device=dev, generator=g)
flt = lambda sd: (tg.TemporalEdgeFilter((0, 49), ts, False, tg.TEMPORAL_SAMPLE_STATIC), torch.full_like(sd, 50))
def filtered():
    sd = seeds()
    return tg.neighbor_sampling_homogenous(P["A__e0__A"], I["A__e0__A"], sd, [15, 10], None, flt(sd))
dt, out = timeit(filtered)
res["neighbor_sampling_homogenous_temporal_single_call"] = {"ms_per_call": dt * 1e3,
                                                            "sampled_edges_per_call": int(out[1].numel())}
# ---- heterogeneous sampling under a temporal filter / with weights: flat (hop, relation) steps, device-driven
hts = {k: torch.randint(0, 100, (I[k].numel(),), device=dev, generator=g) for k in P}
hw = {k: torch.rand(I[k].numel(), device=dev, generator=g, dtype=torch.float64) + 0.1 for k in P}
def het_filtered():
    sd = seeds()
    flt_h = (tg.TemporalEdgeFilter((0, 49), hts, False, tg.TEMPORAL_SAMPLE_STATIC), {"A": torch.full_like(sd, 50)})
    return tg.neighbor_sampling_heterogenous(node_types, edge_types, P, I, {"A": sd}, nn, 2, None, flt_h)
dt, out = timeit(het_filtered)
res["neighbor_sampling_heterogenous_temporal"] = {"ms_per_call": dt * 1e3,
                                                  "sampled_edges_per_call": sum(int(v.numel()) for v in out[1].values())}
def het_weighted():
    return tg.neighbor_sampling_heterogenous(node_types, edge_types, P, I, {"A": seeds()}, nn, 2, tg.WeightedEdgeSampler(hw))
dt, out = timeit(het_weighted)
res["neighbor_sampling_heterogenous_weighted"] = {"ms_per_call": dt * 1e3,
                                                  "sampled_edges_per_call": sum(int(v.numel()) for v in out[1].values())}
# ---- the batched C ABI (tg_ns_hetero_batched): many seed batches of type A in one launch, all hops and relations fused
tix = {t: i for i, t in enumerate(node_types)}
rels_c = [(tix[s_], tix[d_], P["%s__%s__%s" % (s_, n_, d_)], I["%s__%s__%s" % (s_, n_, d_)], [15, 10]) for (s_, n_, d_) in edge_types]
res["tg_ns_hetero_batched"] = {}
for nbatch in (1, 64, 512):
    sd = _cabi.seed_batches(0xBA7C4, 5000, nbatch, 1024, 1 << 23, dev)
    hb = _cabi.NsHeteroBatched(3, rels_c, [sd, None, None], 2, nbatch, dev)
    hb.run(0, 0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for rep in range(5):
        hb.run(0, rep * nbatch)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    ne = int(hb.counts[:, 3:].sum().item())
    res["tg_ns_hetero_batched"]["%d_batches" % nbatch] = {"ms_per_launch": ms, "sampled_edges": ne, "edges_per_s": ne / ms * 1e3}
    del hb
# ---- worker threads, each on its own HIP stream (the blocking size read-backs release the GIL): throughput of the
# latency-bound per-call operators when a DataLoader keeps several mini-batches in flight
import threading  # noqa: E402


def threaded(make_call, n_threads, per_thread=20):
    seeds_per = [[_cabi.seed_batches(0xBA7C4, 1000 + t * per_thread + j, 1, 1024, 1 << 23, dev)[0].contiguous()
                  for j in range(per_thread + 1)] for t in range(n_threads)]
    torch.cuda.synchronize()
    done = [0] * n_threads
    barrier = threading.Barrier(n_threads + 1)

    def work(t):
        st = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(st):
            make_call(seeds_per[t][0])          # warm-up
            st.synchronize()
            barrier.wait()
            for j in range(per_thread):
                out = make_call(seeds_per[t][j + 1])
                done[t] += out
            st.synchronize()
        barrier.wait()

    ths = [threading.Thread(target=work, args=(t,)) for t in range(n_threads)]
    for th in ths:
        th.start()
    barrier.wait()
    t0 = time.perf_counter()
    barrier.wait()
    dt = time.perf_counter() - t0
    for th in ths:
        th.join()
    return dt, sum(done), n_threads * per_thread


def het_call(sd):
    o = tg.neighbor_sampling_heterogenous(node_types, edge_types, P, I, {"A": sd}, nn, 2)
    return sum(int(v.numel()) for v in o[1].values())


def hgt_call(sd):
    o = tg.hgt_sampling(node_types, edge_types, P, I, None, {"A": sd}, None, ns, 2)
    return sum(int(v.numel()) for v in o[0].values()) + sum(int(v.numel()) for v in o[2].values())


def homo_call(sd):
    return int(tg.neighbor_sampling_homogenous(P["A__e0__A"], I["A__e0__A"], sd, [15, 10])[1].numel())


res["threads"] = {}
for name, fn in (("neighbor_sampling_heterogenous", het_call), ("hgt_sampling", hgt_call),
                 ("neighbor_sampling_homogenous", homo_call)):
    res["threads"][name] = {}
    for nt in (1, 4, 8, 16):
        dt, units, calls_done = threaded(fn, nt)
        res["threads"][name]["%d_threads" % nt] = {"calls_per_s": calls_done / dt, "units_per_s": units / dt}
print(json.dumps(res))
