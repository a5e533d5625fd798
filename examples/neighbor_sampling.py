"""The reference's examples/neighbor_sampling.py on the MI355X backend: standard, weighted and temporally filtered
neighbor sampling through the operator surface, mini-batches materialised on the device (filter_data)."""
import torch

from _data import fake_dataset
import tch_geometric as thg
from tch_geometric.transforms import filter_data, to_csc

samples_per_node, num_neighbors = 4, [4, 3]
data = fake_dataset()
col_ptrs, row_indices, perm = to_csc(data)                       # thg.to_csc(edge_index, num_nodes)
start = torch.arange(8, device="cuda").repeat(samples_per_node)

# standard sampling
samples, rows, cols, edge_index, layer_offsets = thg.neighbor_sampling_homogenous(col_ptrs, row_indices, start, num_neighbors)
batch = filter_data(data, samples, rows, cols, edge_index, perm)
print("uniform :", batch.x.shape, batch.edge_index.shape, layer_offsets)

# weighted sampling
weights = torch.rand(row_indices.shape, dtype=torch.double, device="cuda")
out = thg.neighbor_sampling_homogenous(col_ptrs, row_indices, start, num_neighbors, thg.WeightedEdgeSampler(weights))
print("weighted:", out[0].shape, out[1].shape)

# temporal filtering: keep edges whose timestamp lies in the window, relative to each seed's own time
timestamps = torch.randint(0, 5, row_indices.shape, device="cuda")
initial = torch.randint(0, 5, start.shape, device="cuda")
flt = (thg.TemporalEdgeFilter((0, 3), timestamps, False, thg.TEMPORAL_SAMPLE_RELATIVE), initial)
out = thg.neighbor_sampling_homogenous(col_ptrs, row_indices, start, num_neighbors, None, flt)
print("temporal:", out[0].shape, out[1].shape)
