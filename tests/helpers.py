"""Shared test helpers: fixtures and the reference's structural validators
re-expressed in numpy (the reference asserts only invariants)."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_karate():
    d = np.load(os.path.join(GOLDEN, "karate.npz"), allow_pickle=False)
    return d["edge_index"].astype(np.int64), int(d["num_nodes"])


def load_fake_hetero():
    """-> node_counts {ntype: n}, edges {(src, rel, dst): edge_index[2,E]} (io.rs:21-65 key scheme)."""
    d = np.load(os.path.join(GOLDEN, "fakehetero_edges.npz"), allow_pickle=False)
    counts, edges = {}, {}
    for key in d.files:
        if key.startswith("num_nodes_"):
            counts[key[len("num_nodes_"):]] = int(d[key])
        elif key.startswith("edge_"):
            s, r, t = key.split("_")[1].split("-")
            edges[(s, r, t)] = d[key].astype(np.int64)
    return counts, edges


def has_edge(ptrs, indices, x, y):
    row = indices[ptrs[x]:ptrs[x + 1]]
    i = np.searchsorted(row, y)
    return bool(i < row.size and row[i] == y)


def validate_neighbor_samples(ptrs, indices, rows, cols, samples_src, samples_dst, layer_offsets, num_neighbors):
    """neighbor_sampling.rs:370-401 validate_neighbor_samples."""
    for j, i in zip(rows, cols):
        v, w = samples_src[j], samples_dst[i]
        assert has_edge(ptrs, indices, w, v), (j, i, v, w)
    counts = np.zeros(len(samples_dst), dtype=np.int64)
    np.add.at(counts, cols, 1)
    begin = 0
    for l, (_, _, dst_end) in enumerate(layer_offsets):
        assert np.all(counts[begin:dst_end] <= num_neighbors[l])
        begin = dst_end


def roots_of(cols, n_inputs, n_samples):
    """root (input slot) of every sample of the forest (homogeneous: rows[e] = n_inputs + e)."""
    root = np.arange(n_samples, dtype=np.int64)
    for e, i in enumerate(cols):
        root[n_inputs + e] = root[i]
    return root


def rel_key(et):
    return "%s__%s__%s" % tuple(et)


def load_fake_dataset():
    """the reference's tests/fakedataset.npz (edge list only): 1144 nodes, 22648 edges"""
    d = np.load(os.path.join(GOLDEN, "fakedataset_edges.npz"), allow_pickle=False)
    return d["edge_index"].astype(np.int64), int(d["num_nodes"])
