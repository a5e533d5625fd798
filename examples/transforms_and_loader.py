"""The reference's examples/neighbor_sampling_typed.py, hgt_sampling.py and negative_sampling.py: the transform
classes they import, plus the prefetching NeighborLoader."""
import torch

from _data import fake_dataset, fake_hetero_dataset
from tch_geometric.loader import NeighborLoader
from tch_geometric.transforms import HGTSamplerTransform, NegativeSamplerTransform, NeighborSamplerTransform

data, hetero = fake_dataset(), fake_hetero_dataset()
inputs = torch.arange(10)

batch = NeighborSamplerTransform(data, num_neighbors=[4, 3])(inputs)
print("homogeneous :", batch.x.shape, batch.edge_index.shape)
hb = NeighborSamplerTransform(hetero, num_neighbors=[4, 3])({"v0": inputs})
print("heterogeneous:", {t: tuple(hb[t].x.shape) for t in hetero.node_types})

hgt = HGTSamplerTransform(hetero, num_samples=[4, 3])({"v0": inputs})
print("hgt          :", {t: hgt[t].num_nodes for t in hetero.node_types})
hgt_t = HGTSamplerTransform(hetero, num_samples=[4, 3], temporal=True)(
    {"v0": inputs}, {"v0": torch.randint(0, 100, (10,))}, (0, 50))
print("hgt temporal :", {t: hgt_t[t].num_nodes for t in hetero.node_types})

neg = NegativeSamplerTransform(data, 5, 5, inbound=False)(torch.arange(data.num_nodes))
print("negatives    :", neg.neg_edge_index.shape)

loader = NeighborLoader(data, [4, 3], batch_size=128, prefetch=4, shuffle=True)
seen = sum(b.batch_size for b in loader)
print("loader       : %d mini-batches, %d seeds" % (len(loader), seen))
