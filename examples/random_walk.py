"""The reference's examples/random_walk.py: node2vec walks (and the temporal variants) on the device."""
import torch

from _data import fake_dataset
import tch_geometric as thg

walk_length, walks_per_node, p, q = 10, 4, 1.0, 1.5
data = fake_dataset()
row_ptrs, col_indices, perm = thg.to_csr(data.edge_index, data.num_nodes)
start = torch.arange(8, device="cuda").repeat(walks_per_node)
pos_rw = thg.random_walk(row_ptrs, col_indices, start, walk_length - 1, p, q)
print("node2vec walks:", pos_rw.shape)

node_ts = torch.full((data.num_nodes,), -1, device="cuda")
edge_ts = torch.randint(0, 50, col_indices.shape, device="cuda")
start_ts = torch.randint(0, 10, start.shape, device="cuda")
walks, walk_ts = thg.tempo_random_walk(row_ptrs, col_indices, node_ts, edge_ts, start, start_ts, walk_length, (0, 20))
print("temporal walks:", walks.shape, walk_ts.shape)
walks, walk_ts = thg.biased_tempo_random_walk(row_ptrs, col_indices, node_ts, edge_ts, start, start_ts, walk_length,
                                              "exponential", True, 10)
print("biased temporal walks:", walks.shape, int((walks >= 0).sum()))
