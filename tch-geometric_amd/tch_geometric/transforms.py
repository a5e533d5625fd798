"""Mini-batch materialisation on the device: the step AFTER the sampling path.

The reference's examples name a `tch_geometric.transforms` package (examples/neighbor_sampling_typed.py:5,
examples/hgt_sampling.py:5-6, examples/negative_sampling.py:5) that its tree does not contain, and hand the sampled
index tensors to PyG's `filter_data` (examples/neighbor_sampling.py:24).  This module provides those names over the
operator surface with the feature / attribute row gather done by `tg_gather_rows` (csrc/gather.hip) -- nothing
here touches host memory, and nothing needs torch_geometric.

Graph containers are duck-typed:
  homogeneous    an object with `.edge_index` i64[2, E] and `.num_nodes` (or `.x`); every other tensor attribute whose
                 first dimension is num_nodes is a node attribute, E an edge attribute (PyG's rule in
                 torch_geometric.loader.utils.filter_data);
  heterogeneous  an object with `.node_types`, `.edge_types` and `data[node_type]` / `data[edge_type]` stores that look
                 like the homogeneous object (`HeteroGraph` below is the minimal one).
"""
from types import SimpleNamespace
from typing import Dict, List, Optional, Tuple

import torch
from torch import Tensor

from . import _cabi
from . import tch_geometric as native

EdgeType = Tuple[str, str, str]


def rel_key(edge_type: EdgeType) -> str:
    """relation dictionary key of the operator surface (neighbor_sampling.rs:257)"""
    return "__".join(edge_type)


def gather_rows(src: Tensor, index: Tensor) -> Tensor:
    """src[index] over dim 0 on the device; IndexError when an index is out of range."""
    if not src.is_cuda:
        raise ValueError("gather_rows runs on the HIP device; got a %s tensor" % src.device)
    if index.dtype != torch.int64:
        raise ValueError("Tensor must be a is of invalid type. Expected Int64 but got %s" % index.dtype)
    out, status = _cabi.gather_rows(src, index.to(src.device))
    if int(status.item()) != 0:
        raise IndexError("gather_rows: index out of range for %d rows" % src.shape[0])
    return out


class Graph(SimpleNamespace):
    """Minimal homogeneous container: Graph(edge_index=..., num_nodes=..., x=..., edge_attr=...)."""

    def tensor_items(self):
        return [(k, v) for k, v in vars(self).items() if isinstance(v, Tensor)]


class HeteroGraph:
    """Minimal heterogeneous container: stores keyed by node type (str) or edge type (src, rel, dst)."""

    def __init__(self):
        self._nodes: Dict[str, Graph] = {}
        self._edges: Dict[EdgeType, Graph] = {}

    def __getitem__(self, key):
        table = self._nodes if isinstance(key, str) else self._edges
        if key not in table:
            table[key] = Graph()
        return table[key]

    @property
    def node_types(self) -> List[str]:
        return list(self._nodes)

    @property
    def edge_types(self) -> List[EdgeType]:
        return list(self._edges)


def _tensor_items(store):
    if hasattr(store, "tensor_items"):
        return store.tensor_items()
    if hasattr(store, "items"):  # PyG storages
        return [(k, v) for k, v in store.items() if isinstance(v, Tensor)]
    return [(k, v) for k, v in vars(store).items() if isinstance(v, Tensor)]


def _num_nodes(store) -> int:
    n = getattr(store, "num_nodes", None)
    if n is not None:
        return int(n)
    return int(store.x.shape[0])


def _attr_kind(key: str, value: Tensor, n_nodes: int, n_edges: int) -> Optional[str]:
    """PyG's attribute rule: names containing "edge" (and per-edge timestamps / weights) are edge attributes when
    their first dimension is E; otherwise first dimension N -> node attribute, E -> edge attribute."""
    if key == "edge_index" or value.dim() == 0:
        return None
    edge_named = "edge" in key or key in ("timestamps", "weights")
    if edge_named and value.shape[0] == n_edges:
        return "edge"
    if value.shape[0] == n_nodes and not edge_named:
        return "node"
    return "edge" if value.shape[0] == n_edges else None


def to_csc(data, device=None):
    """(col_ptrs, row_indices, perm) of a homogeneous container (what the examples call thg.loader.to_csc)."""
    ei = data.edge_index if device is None else data.edge_index.to(device)
    return native.to_csc(ei, _num_nodes(data))


def to_hetero_csc(data, device=None):
    ptrs, idx, perm = {}, {}, {}
    for et in data.edge_types:
        ei = data[et].edge_index if device is None else data[et].edge_index.to(device)
        size = (_num_nodes(data[et[0]]), _num_nodes(data[et[2]]))
        ptrs[rel_key(et)], idx[rel_key(et)], perm[rel_key(et)] = native.to_csc(ei, size)
    return ptrs, idx, perm


def filter_data(data, samples: Tensor, rows: Tensor, cols: Tensor, edge_index: Tensor, perm: Optional[Tensor] = None):
    """Sub-graph container of one sampled batch: node attributes gathered by `samples`, edge attributes by
    `perm[edge_index]` (COO order of the source graph), `edge_index` = [rows; cols] in batch-local numbering."""
    n_nodes = _num_nodes(data)
    n_edges = int(data.edge_index.shape[1])
    edge = gather_rows(perm, edge_index) if perm is not None else edge_index
    out = Graph(num_nodes=int(samples.numel()), edge_index=torch.stack([rows, cols]), n_id=samples, e_id=edge)
    for key, value in _tensor_items(data):
        kind = _attr_kind(key, value, n_nodes, n_edges)
        if kind == "node":
            setattr(out, key, gather_rows(value.to(samples.device), samples))
        elif kind == "edge":
            setattr(out, key, gather_rows(value.to(samples.device), edge))
    return out


def filter_hetero_data(data, samples: Dict[str, Tensor], rows, cols, edge_index, perm: Optional[Dict[str, Tensor]] = None):
    out = HeteroGraph()
    for nt in data.node_types:
        s = samples[nt]
        store = out[nt]
        store.num_nodes, store.n_id = int(s.numel()), s
        n_nodes = _num_nodes(data[nt])
        for key, value in _tensor_items(data[nt]):
            if value.dim() > 0 and value.shape[0] == n_nodes:
                setattr(store, key, gather_rows(value.to(s.device), s))
    for et in data.edge_types:
        k = rel_key(et)
        store = out[et]
        edge = gather_rows(perm[k], edge_index[k]) if perm is not None else edge_index[k]
        store.edge_index, store.e_id = torch.stack([rows[k], cols[k]]), edge
        n_edges = int(data[et].edge_index.shape[1])
        for key, value in _tensor_items(data[et]):
            if key != "edge_index" and value.dim() > 0 and value.shape[0] == n_edges:
                setattr(store, key, gather_rows(value.to(edge.device), edge))
    return out


def _is_hetero(data) -> bool:
    return hasattr(data, "node_types") and hasattr(data, "edge_types")


class NeighborSamplerTransform:
    """examples/neighbor_sampling_typed.py:16-17, :26-27: `transform = NeighborSamplerTransform(data, num_neighbors)`;
    `batch = transform(inputs)` with a tensor (homogeneous) or a dict of tensors (heterogeneous)."""

    def __init__(self, data, num_neighbors: List[int], sampler=None, filter=None, device="cuda"):
        self.data, self.num_neighbors, self.sampler, self.filter = data, list(num_neighbors), sampler, filter
        self.hetero, self.device = _is_hetero(data), torch.device(device)
        if self.hetero:
            self.col_ptrs, self.row_indices, self.perm = to_hetero_csc(data, self.device)
        else:
            self.col_ptrs, self.row_indices, self.perm = to_csc(data, self.device)

    def __call__(self, inputs, inputs_state=None):
        flt = (self.filter, inputs_state) if self.filter is not None else None
        if not self.hetero:
            s, r, c, e, offsets = native.neighbor_sampling_homogenous(self.col_ptrs, self.row_indices,
                                                                      inputs.to(self.device), self.num_neighbors,
                                                                      self.sampler, flt)
            batch = filter_data(self.data, s, r, c, e, self.perm)
            batch.layer_offsets, batch.batch_size = offsets, int(inputs.numel())
            return batch
        inputs = {k: v.to(self.device) for k, v in inputs.items()}
        fan = {rel_key(et): self.num_neighbors for et in self.data.edge_types}
        s, r, c, e, offsets = native.neighbor_sampling_heterogenous(list(self.data.node_types), list(self.data.edge_types),
                                                                    self.col_ptrs, self.row_indices, inputs, fan,
                                                                    len(self.num_neighbors), self.sampler, flt)
        batch = filter_hetero_data(self.data, s, r, c, e, self.perm)
        batch.layer_offsets = offsets
        return batch


class HGTSamplerTransform:
    """examples/hgt_sampling.py:24-25, :30-31: HGT budget sampling over a heterogeneous container; with
    temporal=True the edge stores' int64 `timestamps` and per-call input timestamps / time range are used."""

    def __init__(self, data, num_samples: List[int], temporal: bool = False, device="cuda"):
        self.data, self.num_samples, self.temporal = data, list(num_samples), temporal
        self.device = torch.device(device)
        self.col_ptrs, self.row_indices, self.perm = to_hetero_csc(data, self.device)
        self.row_timestamps = None
        if temporal:  # timestamps follow the CSC edge order
            self.row_timestamps = {rel_key(et): gather_rows(data[et].timestamps.to(self.device), self.perm[rel_key(et)])
                                   for et in data.edge_types}

    def __call__(self, inputs: Dict[str, Tensor], inputs_timestamps=None, timerange=None):
        inputs = {k: v.to(self.device) for k, v in inputs.items()}
        if inputs_timestamps is not None:
            inputs_timestamps = {k: v.to(self.device) for k, v in inputs_timestamps.items()}
        num = {nt: self.num_samples for nt in self.data.node_types}
        s, ts, r, c, e = native.hgt_sampling(list(self.data.node_types), list(self.data.edge_types), self.col_ptrs,
                                             self.row_indices, self.row_timestamps, inputs, inputs_timestamps, num,
                                             len(self.num_samples), timerange)
        batch = filter_hetero_data(self.data, s, r, c, e, self.perm)
        batch.samples_timestamps = ts
        return batch


class NegativeSamplerTransform:
    """examples/negative_sampling.py:16-17: negatives for every input node; the returned container carries the
    gathered node attributes of `samples` and the negative edges (rows = input slot, cols = batch-local id)."""

    def __init__(self, data, num_neg: int, try_count: int, inbound: bool = False, device="cuda"):
        self.data, self.num_neg, self.try_count, self.inbound = data, num_neg, try_count, inbound
        self.hetero, self.device = _is_hetero(data), torch.device(device)
        if self.hetero:
            self.sizes, self.row_ptrs, self.col_indices = {}, {}, {}
            for et in data.edge_types:
                size = (_num_nodes(data[et[0]]), _num_nodes(data[et[2]]))
                k = rel_key(et)
                self.sizes[k] = size
                self.row_ptrs[k], self.col_indices[k], _ = native.to_csr(data[et].edge_index.to(self.device), size)
        else:
            n = _num_nodes(data)
            self.size = (n, n)
            self.row_ptrs, self.col_indices, _ = native.to_csr(data.edge_index.to(self.device), n)

    def __call__(self, inputs):
        if not self.hetero:
            s, r, c, n_in = native.negative_sample_neighbors_homogenous(self.row_ptrs, self.col_indices, self.size,
                                                                        inputs.to(self.device), self.num_neg,
                                                                        self.try_count)
            out = Graph(num_nodes=int(s.numel()), n_id=s, neg_edge_index=torch.stack([r, c]), batch_size=n_in)
            n_nodes = _num_nodes(self.data)
            for key, value in _tensor_items(self.data):
                if key != "edge_index" and value.dim() > 0 and value.shape[0] == n_nodes:
                    setattr(out, key, gather_rows(value.to(self.device), s))
            return out
        inputs = {k: v.to(self.device) for k, v in inputs.items()}
        s, r, c, counts = native.negative_sample_neighbors_heterogenous(list(self.data.node_types),
                                                                        list(self.data.edge_types), self.row_ptrs,
                                                                        self.col_indices, self.sizes, inputs,
                                                                        self.num_neg, self.try_count, self.inbound)
        out = HeteroGraph()
        for nt in self.data.node_types:
            store = out[nt]
            store.n_id, store.num_nodes, store.batch_size = s[nt], int(s[nt].numel()), counts[nt]
            n_nodes = _num_nodes(self.data[nt])
            for key, value in _tensor_items(self.data[nt]):
                if value.dim() > 0 and value.shape[0] == n_nodes:
                    setattr(store, key, gather_rows(value.to(self.device), s[nt]))
        for et in self.data.edge_types:
            out[et].neg_edge_index = torch.stack([r[rel_key(et)], c[rel_key(et)]])
        return out
