"""Oracle ingest (src/data/storage.rs) against the reference's exact-value tests."""
import numpy as np

import orc
from helpers import load_karate


def test_ind2ptr_reference_vector():
    # storage.rs:152-163 test_ind2ptr
    out = orc.ind2ptr([3, 3, 3, 4, 4, 7, 7, 8, 8], 10)
    assert out.tolist() == [0, 0, 0, 0, 3, 5, 5, 5, 7, 9, 9]


def test_ind2ptr_empty():
    assert orc.ind2ptr(np.zeros(0, dtype=np.int64), 4).tolist() == [0] * 5


def test_to_csc_reference_vector():
    # storage.rs:165-184 test_to_csc
    ei = np.array([[1, 2, 3, 4, 9, 5, 6, 7], [0, 0, 0, 1, 4, 1, 2, 2]], dtype=np.int64)
    ptrs, idx, perm = orc.to_csc(ei, (10, 10))
    deg = np.diff(ptrs)
    assert (deg[0], deg[1], deg[4], deg[2]) == (3, 2, 1, 2)
    assert idx[ptrs[0]:ptrs[1]].tolist() == [1, 2, 3]
    assert idx[ptrs[1]:ptrs[2]].tolist() == [4, 5]
    assert ei[0][perm].tolist() == idx.tolist()


def test_to_csx_matches_numpy_stable_argsort_on_karate():
    ei, n = load_karate()
    for csc in (True, False):
        ptrs, idx, perm = orc.to_csx(ei, n, csc)
        key = ei[1] * n + ei[0] if csc else ei[0] * n + ei[1]
        p = np.argsort(key, kind="stable")
        assert perm.tolist() == p.tolist()
        assert idx.tolist() == (ei[0] if csc else ei[1])[p].tolist()
        major = (ei[1] if csc else ei[0])[p]
        assert ptrs.tolist() == np.concatenate([[0], np.cumsum(np.bincount(major, minlength=n))]).tolist()


def test_karate_csc_prefix_survey_app_b():
    ei, n = load_karate()
    ptrs, idx, _ = orc.to_csc(ei, n)
    assert ptrs[:8].tolist() == [0, 16, 25, 35, 41, 44, 48, 52]
    assert idx[:16].tolist() == [1, 2, 3, 4, 5, 6, 7, 8, 10, 11, 12, 13, 17, 19, 21, 31]


def test_rectangular():
    ei = np.array([[0, 2, 1, 2], [4, 0, 4, 4]], dtype=np.int64)  # 3 x 5
    ptrs, idx, perm = orc.to_csc(ei, (3, 5))
    assert ptrs.tolist() == [0, 1, 1, 1, 1, 4] and idx.tolist() == [2, 0, 1, 2]
    ptrs, idx, perm = orc.to_csr(ei, (3, 5))
    assert ptrs.tolist() == [0, 1, 2, 4] and idx.tolist() == [4, 4, 0, 4]
