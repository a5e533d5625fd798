"""Mini-batch loader over the batched C ABI: the caller side of the sampling path.

The reference's surface is one call per mini-batch (examples/neighbor_sampling.py:18-24: sample, then PyG's
`filter_data`).  On a GPU that shape is latency-bound, so this loader keeps `prefetch` mini-batches in flight per
launch: one `tg_ns_homo_batched` launch samples them all (batch j of the epoch draws with call id `call_id0 + j`, so
every mini-batch equals what `neighbor_sampling_homogenous` returns for that (seed, call id) -- and the oracle), one
`tg_gather_rows` launch per attribute fetches the feature rows of all their nodes, and the mini-batches are handed
out as views.  Everything stays in HBM; the host learns only the per-batch sizes (one read-back per launch).
"""
from typing import Iterator, List, Optional

import torch
from torch import Tensor

from . import _cabi
from .transforms import Graph, HeteroGraph, _attr_kind, _num_nodes, _tensor_items, rel_key, to_csc, to_hetero_csc


def _checked_inputs(nodes: Tensor, n_nodes: int) -> Tensor:
    """int64, flat, and inside [0, n_nodes): the kernels index `ptrs[w]` unchecked (the reference panics on such
    an id, neighbor_sampling.rs:197)."""
    nodes = nodes.reshape(-1).to(torch.int64)
    if nodes.numel() and (int(nodes.min()) < 0 or int(nodes.max()) >= n_nodes):
        raise IndexError("input_nodes outside [0, %d)" % n_nodes)
    return nodes


class NeighborLoader:
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
        self._out = None
        self.epoch = 0

    def __len__(self) -> int:
        n = self.input_nodes.numel()
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def _emit(self, seeds: Tensor, first_batch: int) -> Iterator[Graph]:
        """seeds: [G, B] -- sample G mini-batches in one launch, gather their attributes, yield them"""
        G, B = seeds.shape
        if self._out is None or self._out.n_batches < G or self._out.n_seeds != B:
            self._out = _cabi.NsBatchedOut(max(G, min(self.prefetch, len(self))), B, self.fanout, self.device)
        out = self._out
        _cabi.ns_homo_batched(self._graph, seeds.contiguous(), self.fanout, self.seed, self.call_id0 + first_batch, out,
                              sampler=self.sampler)
        counts = out.counts[:G].cpu()                      # the launch's only read-back
        n_nodes, n_edges = counts[:, 0].tolist(), counts[:, 1].tolist()
        lo = out.layer_offsets[:G].cpu().tolist()
        n_id, rows, cols, e_ptr = _cabi.ns_homo_compact(out, G, counts)  # copies: the slabs are reused by the next launch
        rows_of = lambda table, index: _cabi.gather_rows(table, index)[0]   # ids come from the sampler: no range read-back
        e_id = rows_of(self.perm, e_ptr)                    # COO edge ids of the source graph
        node_parts = {k: torch.split(rows_of(v, n_id), n_nodes) for k, v in self._node_attrs}
        edge_parts = {k: torch.split(rows_of(v, e_id), n_edges) for k, v in self._edge_attrs}
        n_parts, e_parts = torch.split(n_id, n_nodes), torch.split(e_id, n_edges)
        ei_parts = torch.split(torch.stack([rows, cols]), n_edges, dim=1)    # [2, E_b] views of one [2, sum E] tensor
        for b in range(G):
            g = Graph(num_nodes=n_nodes[b], n_id=n_parts[b], e_id=e_parts[b], batch_size=B,
                      edge_index=ei_parts[b],
                      layer_offsets=[tuple(x) for x in lo[b][:len(self.fanout)]], call_id=self.call_id0 + first_batch + b)
            for k, parts in node_parts.items():
                setattr(g, k, parts[b])
            for k, parts in edge_parts.items():
                setattr(g, k, parts[b])
            yield g

    def __iter__(self) -> Iterator[Graph]:
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
        for start in range(0, n_full, self.prefetch):
            G = min(self.prefetch, n_full - start)
            yield from self._emit(nodes[start * B:(start + G) * B].reshape(G, B), batch0 + start)
        if not self.drop_last and n_full * B < n:               # ragged last mini-batch: its own launch
            yield from self._emit(nodes[n_full * B:].reshape(1, -1), batch0 + n_full)


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
