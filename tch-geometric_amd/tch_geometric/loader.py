"""Mini-batch loader over the batched C ABI: the caller side of the sampling path.

The reference's surface is one call per mini-batch (examples/neighbor_sampling.py:18-24: sample, then PyG's
`filter_data`).  On a GPU that shape is latency-bound, so this loader keeps `prefetch` mini-batches in flight per
launch: one `tg_ns_homo_batched` launch samples them all (batch j of the epoch draws with call id `call_id0 + j`, so
every mini-batch equals what `neighbor_sampling_homogenous` returns for that (seed, call id) -- and the oracle), one
`tg_gather_rows` launch per attribute fetches the feature rows of all their nodes, and the mini-batches are handed
out as lazily built views of that `SuperBatch` (or the super-batch itself: `NeighborLoader.super_batches()`).
Everything stays in HBM; the host learns only the per-batch sizes (one read-back per launch, taken from pinned memory
behind a launch that runs one super-batch ahead on a side stream).
"""
from typing import Iterator, List, Optional

import torch
from torch import Tensor

from . import _cabi
from . import tch_geometric as _host
from .transforms import Graph, HeteroGraph, _attr_kind, _num_nodes, _tensor_items, rel_key, to_csc, to_hetero_csc


def _checked_inputs(nodes: Tensor, n_nodes: int) -> Tensor:
    """int64, flat, and inside [0, n_nodes): the kernels index `ptrs[w]` unchecked (the reference panics on such
    an id, neighbor_sampling.rs:197)."""
    nodes = nodes.reshape(-1).to(torch.int64)
    if nodes.numel() and (int(nodes.min()) < 0 or int(nodes.max()) >= n_nodes):
        raise IndexError("input_nodes outside [0, %d)" % n_nodes)
    return nodes


class SuperBatch:
    """`prefetch` mini-batches sampled by ONE launch, as flat batch-major device tensors plus per-batch offsets -- the unit
    the loader really works in.  `n_id` [sum nodes], `edge_index` [2, sum edges] (batch-local numbering), `e_id`
    [sum edges] (COO edge ids of the source graph), one flat tensor per node / edge attribute, `node_ptr` / `edge_ptr`
    (python lists, length G + 1), `layer_offsets` [G][hops] triples, `call_id0`.  Iterating (or indexing) yields the
    mini-batches as `MiniBatch` views; a consumer that can take the whole super-batch (a model over a batch of
    sub-graphs with `ptr` offsets) pays no per-mini-batch host work at all."""
    __slots__ = ("n_id", "edge_index", "_e_id", "_e_ptr", "_perm", "node_attrs", "edge_attrs", "node_ptr", "edge_ptr",
                 "layer_offsets", "batch_size", "call_id0", "n_hops", "_views", "_index")

    @property
    def e_id(self):
        """COO edge ids of the sampled edges: a gather through the ingest permutation (one random 8-byte read per edge),
        made when first asked for -- a consumer that only needs n_id / edge_index / x does not pay for it"""
        if self._e_id is None:
            self._e_id = _cabi.gather_rows(self._perm, self._e_ptr)[0]
        return self._e_id

    def views_of(self, j):
        """every tensor view of mini-batch j, cut by the host module in one call (BatchViews): (n_id, edge_index, node
        attributes..., edge attributes...); e_id is cut apart, when asked for"""
        v = getattr(self, "_views", None)
        if v is None:
            names = ["n_id", "edge_index"] + list(self.node_attrs) + list(self.edge_attrs)
            bases = [self.n_id, self.edge_index] + list(self.node_attrs.values()) + list(self.edge_attrs.values())
            dims = [0, 1] + [0] * (len(bases) - 2)
            kinds = [0, 1] + [0] * len(self.node_attrs) + [1] * len(self.edge_attrs)
            self._index = {k: i for i, k in enumerate(names)}
            v = self._views = _host.BatchViews(bases, dims, kinds, self.node_ptr, self.edge_ptr)
        return v.at(j)

    def __len__(self):
        return len(self.node_ptr) - 1

    def __getitem__(self, j):
        if j < 0:
            j += len(self)
        if not 0 <= j < len(self):
            raise IndexError(j)
        return MiniBatch(self, j)

    def __iter__(self):
        for j in range(len(self.node_ptr) - 1):
            yield MiniBatch(self, j)

    @property
    def num_nodes(self):
        return self.node_ptr[-1]

    @property
    def num_edges(self):
        return self.edge_ptr[-1]


class MiniBatch:
    """Mini-batch j of a SuperBatch.  Nothing is built until it is asked for: sizes are python ints, tensors are views
    (torch.narrow) made on first access -- handing a mini-batch out costs well under a microsecond of host time instead
    of the ~20 us an eagerly built container with a dozen attributes took (profiles/r01/loader_end_to_end.json)."""
    __slots__ = ("_sb", "_j", "_cache", "__dict__")   # __dict__: a consumer may hang its own attributes on a mini-batch

    def __init__(self, sb, j):
        self._sb, self._j, self._cache = sb, j, None

    num_nodes = property(lambda self: self._sb.node_ptr[self._j + 1] - self._sb.node_ptr[self._j])
    num_edges = property(lambda self: self._sb.edge_ptr[self._j + 1] - self._sb.edge_ptr[self._j])
    batch_size = property(lambda self: self._sb.batch_size)
    call_id = property(lambda self: self._sb.call_id0 + self._j)
    layer_offsets = property(lambda self: [tuple(x) for x in self._sb.layer_offsets[self._j][:self._sb.n_hops]])

    def _all(self):  # the views of this mini-batch: one call into the host module on first use, then a tuple
        c = self._cache
        if c is None:
            c = self._cache = self._sb.views_of(self._j)
        return c

    n_id = property(lambda self: self._all()[0])
    edge_index = property(lambda self: self._all()[1])

    @property
    def e_id(self):
        sb, j = self._sb, self._j
        a = sb.edge_ptr[j]
        return sb.e_id.narrow(0, a, sb.edge_ptr[j + 1] - a)

    def __getattr__(self, name):  # node / edge attributes of the source graph (x, y, edge_attr, ...)
        sb = object.__getattribute__(self, "_sb")
        if name in sb.node_attrs or name in sb.edge_attrs:
            views = self._all()
            return views[sb._index[name]]
        raise AttributeError(name)

    def tensor_items(self):
        views = self._all()
        return [(k, views[i]) for k, i in self._sb._index.items()] + [("e_id", self.e_id)]


class NeighborLoader:
    """One launch samples `prefetch` mini-batches; while the caller consumes super-batch i, super-batch i + 1 is already
    being sampled on a side stream into the other of two slab sets (its sizes travel to pinned host memory behind the
    kernel), so neither the launch nor its one read-back sits on the consumer's path.  Launches of >= 2 048 mini-batches
    take the window-ordered form (workspace kept by the loader)."""

    def __init__(self, data, num_neighbors: List[int], input_nodes: Optional[Tensor] = None, batch_size: int = 1024,
                 prefetch: int = 16, replace: bool = False, shuffle: bool = False, drop_last: bool = False,
                 seed: int = 0, call_id0: int = 0, device="cuda"):
        self.data, self.fanout = data, [int(k) for k in num_neighbors]
        self.device = torch.device(device)
        self.batch_size, self.prefetch = int(batch_size), max(1, int(prefetch))
        self.sampler = _cabi.SAMPLER_UNIFORM_REPL if replace else _cabi.SAMPLER_UNIFORM
        self.shuffle, self.drop_last, self.seed, self.call_id0 = shuffle, drop_last, int(seed), int(call_id0)
        self.n_nodes = _num_nodes(data)
        self.col_ptrs, self.row_indices, self.perm = to_csc(data, self.device)
        # u32 shadows halve the bytes per gathered line; ids and offsets fit below 2^31 here
        small = self.n_nodes < 2 ** 31 and self.row_indices.numel() < 2 ** 31
        self._idx32 = self.row_indices.to(torch.int32) if small else None
        self._ptr32 = self.col_ptrs.to(torch.int32) if small else None
        self._graph = _cabi.graph_view(self.col_ptrs, self.row_indices, indices32=self._idx32, ptrs32=self._ptr32)
        nodes = torch.arange(self.n_nodes, device=self.device) if input_nodes is None else input_nodes.to(self.device)
        self.input_nodes = _checked_inputs(nodes, self.n_nodes)
        self._n_edges = int(data.edge_index.shape[1])
        self._node_attrs, self._edge_attrs = [], []
        for key, value in _tensor_items(data):
            kind = _attr_kind(key, value, self.n_nodes, self._n_edges)
            if kind == "node":
                self._node_attrs.append((key, value.to(self.device)))
            elif kind == "edge":
                self._edge_attrs.append((key, value.to(self.device)))
        self._pool = []                     # slab sets of finished iterators (each live iterator owns its own)
        self._ws = None
        self._side = None
        self.epoch = 0

    def __len__(self) -> int:
        n = self.input_nodes.numel()
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    # ---- stage 1 (side stream): sample G mini-batches into slab set `which`, sizes -> pinned host memory
    def _sample(self, slabs, which, seeds: Tensor, first_batch: int):
        """`slabs` is the iterator's own set: slots 0 / 1 for full launches (one consumed, one sampled), "ragged" for the
        epoch's last, narrower mini-batch -- so that one neither replaces the big slabs nor is replaced by them"""
        G, B = seeds.shape
        H = len(self.fanout)
        slab = slabs.get(which)
        if slab is None or slab["out"].n_batches < G or slab["out"].n_seeds != B:
            cap = G if which == "ragged" else max(G, min(self.prefetch, len(self)))
            slab = {"out": _cabi.NsBatchedOut(cap, B, self.fanout, self.device),
                    "counts": torch.empty((cap, 2), dtype=torch.int64).pin_memory(),
                    "lo": torch.empty((cap, max(H, 1), 3), dtype=torch.int64).pin_memory(),
                    "free": None}
            slabs[which] = slab
        if G >= 2048 and self._ws is None:   # many batches per launch: the window-ordered form pays
            self._ws = _cabi.ns_homo_workspace(max(G, min(self.prefetch, len(self))), B, self.fanout, self.device)
        cur = torch.cuda.current_stream(self.device)
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.device)
        side = self._side
        side.wait_stream(cur)                # the seeds (and an earlier consumer of this slab set) are ahead of us
        if slab["free"] is not None:
            side.wait_event(slab["free"])
        with torch.cuda.stream(side):
            seeds = seeds.contiguous()
            out = slab["out"]
            _cabi.ns_homo_batched(self._graph, seeds, self.fanout, self.seed, self.call_id0 + first_batch, out,
                                  sampler=self.sampler, ws=self._ws if G >= 2048 else None)
            slab["counts"][:G].copy_(out.counts[:G], non_blocking=True)
            slab["lo"][:G].copy_(out.layer_offsets[:G], non_blocking=True)
            done = torch.cuda.Event()
            done.record(side)
        seeds.record_stream(side)
        return (slab, G, B, first_batch, done)

    # ---- stage 2 (caller's stream): flatten, gather the attribute rows
    def _finish(self, ticket) -> SuperBatch:
        slab, G, B, first_batch, done = ticket
        out = slab["out"]
        done.synchronize()                   # the host needs the sizes; the device work it waits for is long done
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(done)
        counts = slab["counts"][:G]
        n_nodes, n_edges = counts[:, 0].tolist(), counts[:, 1].tolist()
        n_id, edge_index, e_ptr = _cabi.ns_homo_compact(out, G, counts, stacked=True)   # copies: the slabs go back to the sampler
        free = torch.cuda.Event()
        free.record(cur)
        slab["free"] = free
        rows_of = lambda table, index: _cabi.gather_rows(table, index)[0]   # ids come from the sampler: no range read-back
        sb = SuperBatch()
        sb.n_id, sb.edge_index = n_id, edge_index
        sb._e_id, sb._e_ptr, sb._perm = None, e_ptr, self.perm
        sb.node_attrs = {k: rows_of(v, n_id) for k, v in self._node_attrs}
        sb.edge_attrs = {k: rows_of(v, sb.e_id) for k, v in self._edge_attrs}
        ptr_n, ptr_e, a, e = [0], [0], 0, 0
        for x, y in zip(n_nodes, n_edges):
            a += x
            e += y
            ptr_n.append(a)
            ptr_e.append(e)
        sb.node_ptr, sb.edge_ptr = ptr_n, ptr_e
        sb.layer_offsets = slab["lo"][:G].tolist()
        sb.batch_size, sb.call_id0, sb.n_hops = B, self.call_id0 + first_batch, len(self.fanout)
        return sb

    def super_batches(self) -> Iterator[SuperBatch]:
        """The epoch as super-batches of up to `prefetch` mini-batches; the next one is sampled while this one is used."""
        nodes = self.input_nodes
        epoch = self.epoch                                      # captured: a second iterator is the next epoch
        self.epoch += 1
        if self.shuffle:
            gen = torch.Generator(device=self.device)
            gen.manual_seed(self.seed * 1000003 + epoch)
            nodes = nodes[torch.randperm(nodes.numel(), device=self.device, generator=gen)]
        B, n = self.batch_size, nodes.numel()
        n_full = n // B
        # every epoch draws afresh, as the reference's global stream does (utils/random.rs:19-22): mini-batch j of
        # epoch e uses call id call_id0 + e * len(self) + j -- reproducible for a fixed (seed, epoch)
        batch0 = epoch * len(self)
        work = []
        for start in range(0, n_full, self.prefetch):
            G = min(self.prefetch, n_full - start)
            work.append((nodes[start * B:(start + G) * B].reshape(G, B), batch0 + start))
        if not self.drop_last and n_full * B < n:               # ragged last mini-batch: its own launch
            work.append((nodes[n_full * B:].reshape(1, -1), batch0 + n_full))
        # this iterator's own slabs (two iterators alive at once -- zip(loader, loader), a restart after an abandoned
        # epoch -- must not sample into each other's); they go back to the pool when the iterator ends or is dropped
        slabs = self._pool.pop() if self._pool else {}
        which_of = lambda i: "ragged" if work[i][0].shape[1] != B else i & 1
        pending = None
        try:
            for i in range(len(work)):
                if pending is None:
                    pending = self._sample(slabs, which_of(i), *work[i])
                ticket = pending
                pending = self._sample(slabs, which_of(i + 1), *work[i + 1]) if i + 1 < len(work) else None
                yield self._finish(ticket)
        finally:
            if pending is not None:          # abandoned with a launch in flight: it still writes into these slabs
                pending[4].synchronize()
            self._pool.append(slabs)

    def __iter__(self) -> Iterator[MiniBatch]:
        for sb in self.super_batches():
            yield from sb


class HeteroNeighborLoader:
    """The heterogeneous counterpart: seeds of ONE node type, `prefetch` mini-batches per tg_ns_hetero_batched launch
    (all hops and relations fused, default samplers), per-type / per-relation slabs flattened by tg_compact_rows, node
    attributes gathered per type, edge attributes per relation through the ingest permutation.  Mini-batch j of the
    epoch equals neighbor_sampling_heterogenous for (seed, call_id0 + j)."""

    def __init__(self, data, num_neighbors: List[int], input_type: str, input_nodes: Optional[Tensor] = None,
                 batch_size: int = 1024, prefetch: int = 16, replace: bool = False, drop_last: bool = False, seed: int = 0,
                 call_id0: int = 0, device="cuda"):
        self.data, self.fanout, self.device = data, [int(k) for k in num_neighbors], torch.device(device)
        self.node_types, self.edge_types = list(data.node_types), list(data.edge_types)
        self.input_type, self.batch_size, self.prefetch = input_type, int(batch_size), max(1, int(prefetch))
        self.sampler = _cabi.SAMPLER_UNIFORM_REPL if replace else _cabi.SAMPLER_UNIFORM
        self.drop_last, self.seed, self.call_id0 = drop_last, int(seed), int(call_id0)
        self.col_ptrs, self.row_indices, self.perm = to_hetero_csc(data, self.device)
        tix = {t: i for i, t in enumerate(self.node_types)}
        self._tix = tix
        self._rels = [(tix[et[0]], tix[et[2]], self.col_ptrs[rel_key(et)], self.row_indices[rel_key(et)], self.fanout)
                      for et in self.edge_types]
        n_in = _num_nodes(data[input_type])
        nodes = torch.arange(n_in, device=self.device) if input_nodes is None else input_nodes.to(self.device)
        self.input_nodes = _checked_inputs(nodes, n_in)
        self.epoch = 0
        self._node_attrs = {t: [(k, v.to(self.device)) for k, v in _tensor_items(data[t])
                                if v.dim() > 0 and v.shape[0] == _num_nodes(data[t])] for t in self.node_types}
        self._edge_attrs = {}
        for et in self.edge_types:
            n_e = int(data[et].edge_index.shape[1])
            self._edge_attrs[et] = [(k, v.to(self.device)) for k, v in _tensor_items(data[et])
                                    if k != "edge_index" and v.dim() > 0 and v.shape[0] == n_e]

    def __len__(self) -> int:
        n = self.input_nodes.numel()
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def _emit(self, seeds: Tensor, first_batch: int) -> Iterator[HeteroGraph]:
        G, B = seeds.shape
        T, R, H = len(self.node_types), len(self.edge_types), len(self.fanout)
        inputs = [None] * T
        inputs[self._tix[self.input_type]] = seeds.contiguous()
        hb = _cabi.NsHeteroBatched(T, self._rels, inputs, H, G, self.device, sampler=self.sampler)
        hb.run(self.seed, self.call_id0 + first_batch)
        counts = hb.counts.cpu()                                # the launch's only read-back
        lo = hb.layer_offsets.cpu().tolist()
        rows_of = lambda table, index: _cabi.gather_rows(table, index)[0]
        node_parts, attr_parts = {}, {}
        for t, nt in enumerate(self.node_types):
            lens = counts[:, t].tolist()
            flat = _cabi.compact_rows(hb.samples[t], hb.counts[:, t], sum(lens))
            node_parts[nt] = (torch.split(flat, lens), lens)
            attr_parts[nt] = {k: torch.split(rows_of(v, flat), lens) for k, v in self._node_attrs[nt]}
        edge_parts = {}
        for r, et in enumerate(self.edge_types):
            lens = counts[:, T + r].tolist()
            tot = sum(lens)
            fr = _cabi.compact_rows(hb.rows[r], hb.counts[:, T + r], tot)
            fc = _cabi.compact_rows(hb.cols[r], hb.counts[:, T + r], tot)
            fe = rows_of(self.perm[rel_key(et)], _cabi.compact_rows(hb.edge_index[r], hb.counts[:, T + r], tot))
            edge_parts[et] = (torch.split(torch.stack([fr, fc]), lens, dim=1), torch.split(fe, lens),
                              {k: torch.split(rows_of(v, fe), lens) for k, v in self._edge_attrs[et]})
        for b in range(G):
            g = HeteroGraph()
            for nt in self.node_types:
                st = g[nt]
                st.n_id, st.num_nodes = node_parts[nt][0][b], node_parts[nt][1][b]
                for k, parts in attr_parts[nt].items():
                    setattr(st, k, parts[b])
            g[self.input_type].batch_size = B
            for r, et in enumerate(self.edge_types):
                st = g[et]
                st.edge_index, st.e_id = edge_parts[et][0][b], edge_parts[et][1][b]
                st.layer_offsets = [tuple(x) for x in lo[b][r][:H]]
                for k, parts in edge_parts[et][2].items():
                    setattr(st, k, parts[b])
            g.call_id = self.call_id0 + first_batch + b
            yield g

    def __iter__(self) -> Iterator[HeteroGraph]:
        nodes, B = self.input_nodes, self.batch_size
        batch0 = self.epoch * len(self)                         # fresh draws every epoch (see NeighborLoader)
        self.epoch += 1
        n_full = nodes.numel() // B
        for start in range(0, n_full, self.prefetch):
            G = min(self.prefetch, n_full - start)
            yield from self._emit(nodes[start * B:(start + G) * B].reshape(G, B), batch0 + start)
        if not self.drop_last and n_full * B < nodes.numel():
            yield from self._emit(nodes[n_full * B:].reshape(1, -1), batch0 + n_full)
