"""Type stubs of the native module (operator surface of the reference: src/python.rs:785-803).

Tensors are int64 unless noted; they may live on a HIP device (adjacency stays resident in HBM) or on
the CPU (uploaded per call).  Relation dicts are keyed "src__rel__dst" (neighbor_sampling.rs:257).

Limits the reference does not have: per-hop fan-out at most 4096 for the unweighted, unfiltered samplers and at most
1024 under a temporal filter or the weighted sampler (ValueError beyond; the C ABI's batched launch tg_ns_homo_batched
takes fan-outs up to 255 and 8 hops, TG_MAX_HOPS: the operators route larger fan-outs to the flat per-hop kernels and
deeper calls hop by hop); heterogeneous sampling at most 8 node types / 16 relations in the fused launch (more fall back
to one launch per relation and hop)."""
from typing import Dict, List, Optional, Tuple, Union

from torch import Tensor

from .utils import TemporalEdgeFilter, UniformEdgeSampler, WeightedEdgeSampler

NodeType = str
RelType = str
EdgeType = Tuple[str, str, str]
LayerOffset = Tuple[int, int, int]
EdgeSampler = Union[UniformEdgeSampler, WeightedEdgeSampler]
HomoFilter = Tuple[TemporalEdgeFilter, Tensor]
HeteroFilter = Tuple[TemporalEdgeFilter, Dict[NodeType, Tensor]]

def seed(seed: int) -> None: ...                      # additive: the reference cannot be seeded from Python
def rng_state() -> Tuple[int, int]: ...
def set_rng_state(seed: int, call_counter: int) -> None: ...
def backend_version() -> str: ...
class PanicException(RuntimeError): ...               # raised where the reference panics (pyo3_runtime.PanicException there)
def graph_cache_info() -> Dict[str, int]: ...         # additive: CPU-resident adjacency tensors are uploaded once and
def graph_cache_clear() -> None: ...                  # kept on the device (keyed on storage identity + content version);
                                                      # also the edge sets that answer has_edge for random_walk with
                                                      # p != q (built by the first call of >= 2^20 walker steps on a graph)

def to_csc(row_col: Tensor, size: Union[int, Tuple[int, int]]) -> Tuple[Tensor, Tensor, Tensor]: ...
def to_csr(row_col: Tensor, size: Union[int, Tuple[int, int]]) -> Tuple[Tensor, Tensor, Tensor]: ...

def neighbor_sampling_homogenous(
    col_ptrs: Tensor, row_indices: Tensor, inputs: Tensor, num_neighbors: List[int],
    sampler: Optional[EdgeSampler] = None, filter: Optional[HomoFilter] = None,
) -> Tuple[Tensor, Tensor, Tensor, Tensor, List[LayerOffset]]: ...

def neighbor_sampling_heterogenous(
    node_types: List[NodeType], edge_types: List[EdgeType], col_ptrs: Dict[RelType, Tensor],
    row_indices: Dict[RelType, Tensor], inputs: Dict[NodeType, Tensor], num_neighbors: Dict[RelType, List[int]],
    num_hops: int, sampler: Optional[EdgeSampler] = None, filter: Optional[HeteroFilter] = None,
) -> Tuple[Dict[NodeType, Tensor], Dict[RelType, Tensor], Dict[RelType, Tensor], Dict[RelType, Tensor],
           Dict[RelType, List[LayerOffset]]]: ...

def hgt_sampling(
    node_types: List[NodeType], edge_types: List[EdgeType], col_ptrs: Dict[RelType, Tensor],
    row_indices: Dict[RelType, Tensor], row_timestamps: Optional[Dict[RelType, Tensor]],
    inputs: Dict[NodeType, Tensor], input_timestamps: Optional[Dict[NodeType, Tensor]],
    num_samples: Dict[NodeType, List[int]], num_hops: int, timerange: Optional[Tuple[int, int]] = None,
) -> Tuple[Dict[NodeType, Tensor], Dict[NodeType, Tensor], Dict[RelType, Tensor], Dict[RelType, Tensor],
           Dict[RelType, Tensor]]: ...

def random_walk(row_ptrs: Tensor, col_indices: Tensor, start: Tensor, walk_length: int, p: float,
                q: float) -> Tensor: ...

def tempo_random_walk(
    row_ptrs: Tensor, col_indices: Tensor, node_timestamps: Tensor, edge_timestamps: Tensor, start: Tensor,
    start_timestamps: Tensor, walk_length: int, window: Tuple[int, int],
) -> Tuple[Tensor, Tensor]: ...

def negative_sample_neighbors_homogenous(
    row_ptrs: Tensor, col_indices: Tensor, graph_size: Tuple[int, int], inputs: Tensor, num_neg: int, try_count: int,
) -> Tuple[Tensor, Tensor, Tensor, int]: ...

def negative_sample_neighbors_heterogenous(
    node_types: List[NodeType], edge_types: List[EdgeType], row_ptrs: Dict[RelType, Tensor],
    col_indices: Dict[RelType, Tensor], sizes: Dict[RelType, Tuple[int, int]], inputs: Dict[NodeType, Tensor],
    num_neg: int, try_count: int, inbound: bool,
) -> Tuple[Dict[NodeType, Tensor], Dict[RelType, Tensor], Dict[RelType, Tensor], Dict[NodeType, int]]: ...

def budget_sampling(
    node_types: List[NodeType], edge_types: List[EdgeType], col_ptrs: Dict[RelType, Tensor],
    row_indices: Dict[RelType, Tensor], row_timestamps: Optional[Dict[RelType, Tensor]],
    inputs: Dict[NodeType, Tensor], input_timestamps: Optional[Dict[NodeType, Tensor]],
    num_neighbors: Dict[NodeType, List[int]], num_hops: int, window: Optional[Tuple[int, int]], forward: bool,
    relative: bool,
) -> Tuple[Dict[NodeType, Tensor], Dict[NodeType, Tensor], Dict[RelType, Tensor], Dict[RelType, Tensor],
           Dict[RelType, Tensor]]: ...

def biased_tempo_random_walk(
    row_ptrs: Tensor, col_indices: Tensor, node_timestamps: Tensor, edge_timestamps: Tensor, start: Tensor,
    start_timestamps: Tensor, walk_length: int, bias_type: str, forward: bool, retry_count: int,
) -> Tuple[Tensor, Tensor]: ...
