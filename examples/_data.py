"""Synthetic stand-ins for the PyG datasets the reference's examples download (FakeDataset / FakeHeteroDataset):
random graphs with features, in the duck-typed containers of tch_geometric.transforms.  No network, no torch_geometric."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tch-geometric_amd"))
from tch_geometric.transforms import Graph, HeteroGraph  # noqa: E402


def fake_dataset(num_nodes=1000, avg_degree=10, channels=64, seed=0, device="cuda"):
    g = torch.Generator().manual_seed(seed)
    e = num_nodes * avg_degree
    ei = torch.randint(0, num_nodes, (2, e), generator=g)
    return Graph(edge_index=ei.to(device), num_nodes=num_nodes, x=torch.randn(num_nodes, channels, generator=g).to(device),
                 y=torch.randint(0, 10, (num_nodes,), generator=g).to(device))


def fake_hetero_dataset(num_node_types=3, num_edge_types=6, avg_num_nodes=1000, avg_degree=10, channels=64, seed=0,
                        device="cuda"):
    g = torch.Generator().manual_seed(seed)
    data = HeteroGraph()
    counts = {}
    for t in range(num_node_types):
        n = int(avg_num_nodes * (0.5 + torch.rand(1, generator=g).item()))
        counts["v%d" % t] = n
        data["v%d" % t].x = torch.randn(n, channels, generator=g).to(device)
        data["v%d" % t].num_nodes = n
    for r in range(num_edge_types):
        s, d = "v%d" % (r % num_node_types), "v%d" % ((r * 2 + 1) % num_node_types)
        e = counts[d] * avg_degree
        ei = torch.stack([torch.randint(0, counts[s], (e,), generator=g), torch.randint(0, counts[d], (e,), generator=g)])
        data[(s, "e%d" % r, d)].edge_index = ei.to(device)
        data[(s, "e%d" % r, d)].timestamps = torch.randint(0, 100, (e,), generator=g).to(device)
    return data
