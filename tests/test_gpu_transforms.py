"""tg_gather_rows and the mini-batch transforms built on it (the step after the sampling path): the row gather is
checked against numpy fancy indexing (bit-exact: bytes are moved, never converted), the transforms against the
sampled index tensors they wrap."""
import numpy as np
import pytest
import torch

from helpers import load_fake_hetero, load_karate, rel_key

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def tr():
    from tch_geometric import transforms
    return transforms


@pytest.mark.parametrize("dtype,shape", [
    (np.float32, (1000, 128)),   # 512-byte rows: 16-byte vectors, 32 lanes per row
    (np.float32, (1000, 600)),   # 2400-byte rows: one wavefront per row, several passes
    (np.float16, (777, 7)),      # 14-byte rows: byte path
    (np.float64, (513, 3)),      # 24-byte rows: 8-byte vectors
    (np.int32, (300, 5)),        # 20-byte rows: 4-byte vectors
    (np.int64, (4096,)),         # 1-D (perm composition)
    (np.uint8, (100, 2, 3, 5)),  # trailing dims
    (np.float32, (5, 4)),        # 16-byte rows, fewer rows than one tile
])
def test_gather_rows_equals_numpy(tr, dtype, shape):
    rs = np.random.default_rng(7)
    src = (rs.standard_normal(shape) * 100).astype(dtype)
    for n in (0, 1, 63, 5000):
        idx = rs.integers(0, shape[0], n)
        got = tr.gather_rows(torch.from_numpy(src).to(DEV), torch.from_numpy(idx).to(DEV))
        assert got.dtype == torch.from_numpy(src).dtype and tuple(got.shape) == (n,) + tuple(shape[1:])
        assert np.array_equal(got.cpu().numpy().view(np.uint8), src[idx].view(np.uint8))


def test_gather_rows_strided_source_and_errors(tr):
    rs = np.random.default_rng(8)
    src = rs.standard_normal((200, 96)).astype(np.float32)
    t = torch.from_numpy(src).to(DEV)
    idx = rs.integers(0, 200, 1000)
    ti = torch.from_numpy(idx).to(DEV)
    for view, ref in ((t[:, :64], src[:, :64]), (t[:, 3:20], src[:, 3:20]), (t[::2], src[::2]), (t.t()[:50], src.T[:50])):
        i2 = ti % view.shape[0]
        assert np.array_equal(tr.gather_rows(view, i2).cpu().numpy(), ref[idx % ref.shape[0]])
    with pytest.raises(IndexError):
        tr.gather_rows(t, torch.tensor([0, 200], device=DEV))
    with pytest.raises(IndexError):
        tr.gather_rows(t, torch.tensor([-1], device=DEV))
    with pytest.raises(ValueError):
        tr.gather_rows(t, torch.tensor([0], device=DEV, dtype=torch.int32))
    with pytest.raises(ValueError):
        tr.gather_rows(t.cpu(), torch.tensor([0]))


def _karate_graph(tr):
    ei, n = load_karate()
    rs = np.random.default_rng(3)
    x = rs.standard_normal((n, 16)).astype(np.float32)
    y = rs.integers(0, 4, n)
    ea = rs.standard_normal((ei.shape[1], 3))
    g = tr.Graph(edge_index=torch.from_numpy(ei).to(DEV), num_nodes=n, x=torch.from_numpy(x).to(DEV),
                 y=torch.from_numpy(y).to(DEV), edge_attr=torch.from_numpy(ea).to(DEV))
    return g, ei, x, y, ea


def test_neighbor_sampler_transform_homogeneous(tr):
    import tch_geometric as tg
    g, ei, x, y, ea = _karate_graph(tr)
    t = tr.NeighborSamplerTransform(g, [4, 3])
    tg.seed(11)
    inputs = torch.arange(10)
    b = t(inputs)
    s = b.n_id.cpu().numpy()
    r, c = b.edge_index.cpu().numpy()
    e = b.e_id.cpu().numpy()
    assert b.batch_size == 10 and b.num_nodes == len(s) and np.array_equal(s[:10], np.arange(10))
    assert np.array_equal(b.x.cpu().numpy(), x[s]) and np.array_equal(b.y.cpu().numpy(), y[s])
    assert np.array_equal(b.edge_attr.cpu().numpy(), ea[e])
    # e_id names the source COO edge (neighbour -> parent) of every sampled edge
    assert np.array_equal(ei[0, e], s[r]) and np.array_equal(ei[1, e], s[c])
    # the same call through the operator surface gives the same batch
    tg.seed(11)
    s2, r2, c2, e2, lo = tg.neighbor_sampling_homogenous(t.col_ptrs, t.row_indices, inputs.to(DEV), [4, 3])
    assert np.array_equal(s2.cpu().numpy(), s) and lo == b.layer_offsets
    assert np.array_equal(t.perm[e2].cpu().numpy(), e)


def _hetero_graph(tr):
    counts, edges = load_fake_hetero()
    rs = np.random.default_rng(5)
    g = tr.HeteroGraph()
    feats = {}
    for nt, n in counts.items():
        feats[nt] = rs.standard_normal((n, 8)).astype(np.float32)
        g[nt].x, g[nt].num_nodes = torch.from_numpy(feats[nt]).to(DEV), n
    for et, ei in edges.items():
        g[et].edge_index = torch.from_numpy(ei).to(DEV)
        g[et].timestamps = torch.from_numpy(rs.integers(0, 10, ei.shape[1])).to(DEV)
    return g, counts, edges, feats


def test_neighbor_sampler_transform_heterogeneous(tr):
    g, counts, edges, feats = _hetero_graph(tr)
    t = tr.NeighborSamplerTransform(g, [4, 3])
    nt0 = g.node_types[0]
    b = t({nt0: torch.arange(10)})
    for nt in g.node_types:
        s = b[nt].n_id.cpu().numpy()
        assert np.array_equal(b[nt].x.cpu().numpy(), feats[nt][s])
    total = 0
    for et, ei in edges.items():
        r, c = b[et].edge_index.cpu().numpy()
        e = b[et].e_id.cpu().numpy()
        total += len(e)
        assert np.array_equal(ei[0, e], b[et[0]].n_id.cpu().numpy()[r])
        assert np.array_equal(ei[1, e], b[et[2]].n_id.cpu().numpy()[c])
        assert np.array_equal(b[et].timestamps.cpu().numpy(), g[et].timestamps.cpu().numpy()[e])
    assert total > 0


def test_hgt_and_negative_transforms(tr):
    g, counts, edges, feats = _hetero_graph(tr)
    nt0 = g.node_types[0]
    for temporal in (False, True):
        t = tr.HGTSamplerTransform(g, [4, 3], temporal=temporal)
        if temporal:
            b = t({nt0: torch.arange(10)}, {nt0: torch.full((10,), 5, dtype=torch.int64)}, (0, 6))
        else:
            b = t({nt0: torch.arange(10)})
        for nt in g.node_types:
            assert np.array_equal(b[nt].x.cpu().numpy(), feats[nt][b[nt].n_id.cpu().numpy()])
        for et, ei in edges.items():
            r, c = b[et].edge_index.cpu().numpy()
            e = b[et].e_id.cpu().numpy()
            assert np.array_equal(ei[0, e], b[et[0]].n_id.cpu().numpy()[r])
            assert np.array_equal(ei[1, e], b[et[2]].n_id.cpu().numpy()[c])
    kg, ei, x, y, ea = _karate_graph(tr)
    nb = tr.NegativeSamplerTransform(kg, 5, 5)(torch.arange(34))
    s = nb.n_id.cpu().numpy()
    r, c = nb.neg_edge_index.cpu().numpy()
    assert nb.batch_size == 34 and np.array_equal(nb.x.cpu().numpy(), x[s])
    existing = set(zip(ei[0].tolist(), ei[1].tolist()))
    assert all((int(s[a]), int(s[b])) not in existing for a, b in zip(r, c))
    hb = tr.NegativeSamplerTransform(g, 3, 4)({nt0: torch.arange(10)})
    for nt in g.node_types:
        assert np.array_equal(hb[nt].x.cpu().numpy(), feats[nt][hb[nt].n_id.cpu().numpy()])
