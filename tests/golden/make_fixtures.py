"""Regenerates the data fixtures under tests/golden/ from the reference's own
test data files (plain numpy .npz, loaded with allow_pickle=False).

  karate.npz              edge_index i64[2,156], num_nodes          (tests/karate.npz)
  fakehetero_edges.npz    the six edge_index arrays + node counts   (tests/fakeheterodataset.npz;
                          the x/y feature arrays are never read by any sampler test, io.rs:21-65)

  fakedataset_edges.npz   edge_index i64[2,22648], num_nodes (tests/fakedataset.npz: held by the reference, read by none
                          of its tests; used here as a mid-size homogeneous graph)

Run in the build container only (needs /root/reference): python tests/golden/make_fixtures.py
"""
import os
import numpy as np

REF = "/root/reference/tests"
HERE = os.path.dirname(os.path.abspath(__file__))

k = np.load(os.path.join(REF, "karate.npz"), allow_pickle=False)
np.savez_compressed(os.path.join(HERE, "karate.npz"), edge_index=k["edge_index"],
                    num_nodes=np.int64(k["x"].shape[0]))

h = np.load(os.path.join(REF, "fakeheterodataset.npz"), allow_pickle=False)
out = {}
for key in h.files:
    if key.startswith("node_") and key.endswith("_x"):
        out["num_nodes_" + key.split("_")[1]] = np.int64(h[key].shape[0])
    elif key.startswith("edge_"):
        out[key] = h[key]
np.savez_compressed(os.path.join(HERE, "fakehetero_edges.npz"), **out)
print({k_: (v.shape, v.dtype) for k_, v in out.items()})

f = np.load(os.path.join(REF, "fakedataset.npz"), allow_pickle=False)
np.savez_compressed(os.path.join(HERE, "fakedataset_edges.npz"), edge_index=f["edge_index"],
                    num_nodes=np.int64(f["x"].shape[0]))
