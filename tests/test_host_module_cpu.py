"""Host module on a box without a GPU: it builds, exposes the reference's surface, and refuses to compute
(no CPU fallback)."""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def tg():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tch-geometric_amd"), "-s"])
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tch-geometric_amd", "host", "build_host.py")])
    import tch_geometric
    return tch_geometric


REFERENCE_SURFACE = ["to_csc", "to_csr", "neighbor_sampling_homogenous", "neighbor_sampling_heterogenous",
                     "hgt_sampling", "budget_sampling", "random_walk", "tempo_random_walk",
                     "biased_tempo_random_walk", "negative_sample_neighbors_homogenous",
                     "negative_sample_neighbors_heterogenous"]          # src/python.rs:56-59, 785-797


def test_surface_names(tg):
    missing = [n for n in REFERENCE_SURFACE if not hasattr(tg, n)]
    assert missing == [], missing                                        # all 11 names of python.rs:785-797
    for n in ("seed", "rng_state", "set_rng_state", "UniformEdgeSampler", "WeightedEdgeSampler",
              "TemporalEdgeFilter", "TEMPORAL_SAMPLE_STATIC", "TEMPORAL_SAMPLE_RELATIVE", "TEMPORAL_SAMPLE_DYNAMIC"):
        assert hasattr(tg, n)
    assert (tg.TEMPORAL_SAMPLE_STATIC, tg.TEMPORAL_SAMPLE_RELATIVE, tg.TEMPORAL_SAMPLE_DYNAMIC) == (0, 1, 2)


def test_seed_state(tg):
    tg.seed(42)
    assert tg.rng_state() == (42, 0)
    tg.set_rng_state(7, 9)
    assert tg.rng_state() == (7, 9)


@pytest.mark.skipif(torch.cuda.is_available(), reason="only meaningful without a GPU")
def test_no_cpu_fallback(tg):
    P, I = torch.tensor([0, 1, 2]), torch.tensor([1, 0])
    with pytest.raises(RuntimeError, match="no CPU path"):
        tg.neighbor_sampling_homogenous(P, I, torch.tensor([0]), [2])
    with pytest.raises(RuntimeError, match="no CPU path"):
        tg.random_walk(P, I, torch.tensor([0]), 3, 1.0, 1.0)
    with pytest.raises(RuntimeError, match="no CPU path"):
        tg.to_csc(torch.tensor([[0, 1], [1, 0]]), 2)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "tch-geometric_amd")
    for base, _, files in os.walk(pkg):
        if os.path.basename(base) in ("build", "lib", "__pycache__"):
            continue
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                src = open(os.path.join(base, f), errors="ignore").read()
                assert "import orc" not in src and "oracle/" not in src and "tg_oracle" not in src, f


def test_transform_helpers_without_a_gpu():
    """attribute classification and containers of tch_geometric.transforms are plain host logic"""
    import torch
    from tch_geometric import transforms as tr
    assert tr.rel_key(("a", "to", "b")) == "a__to__b"
    n, e = 5, 7
    assert tr._attr_kind("x", torch.zeros(n, 3), n, e) == "node"
    assert tr._attr_kind("edge_attr", torch.zeros(e, 2), n, e) == "edge"
    assert tr._attr_kind("edge_index", torch.zeros(2, e), n, e) is None
    assert tr._attr_kind("timestamps", torch.zeros(n), n, n) == "edge"       # N == E: the name decides
    assert tr._attr_kind("y", torch.zeros(n), n, n) == "node"
    assert tr._attr_kind("other", torch.zeros(3), n, e) is None and tr._attr_kind("s", torch.tensor(1.0), n, e) is None
    g = tr.HeteroGraph()
    g["a"].x = torch.zeros(4, 2)
    g[("a", "to", "b")].edge_index = torch.zeros(2, 3, dtype=torch.int64)
    g["b"].num_nodes = 9
    assert g.node_types == ["a", "b"] and g.edge_types == [("a", "to", "b")]
    assert tr._num_nodes(g["a"]) == 4 and tr._num_nodes(g["b"]) == 9 and tr._is_hetero(g) and not tr._is_hetero(g["a"])
    import pytest
    with pytest.raises(ValueError):                                           # the row gather is a device kernel
        tr.gather_rows(torch.zeros(3, 2), torch.tensor([0]))
