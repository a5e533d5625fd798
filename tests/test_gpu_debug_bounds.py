"""The -DTG_DEBUG_BOUNDS build (`make dbg`, lib/libtchgeo_hip_dbg.so): frontier ids are range-checked before they index
`ptrs` in the multi-hop neighbor-sampling kernels; an offender raises the registered flag and vertex 0 is sampled in its
place.  This is the build work-dropping EXPERIMENTS run on (round 2 lost a box to an experiment whose
second hop indexed `ptrs` with never-written frontier slots).  The offending id used here is n_major itself -- one past
the last vertex, still inside the offset table's allocation -- so that nothing reads out of bounds even unguarded."""
import ctypes as C
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DBG = os.path.join(ROOT, "tch-geometric_amd", "lib", "libtchgeo_hip_dbg.so")


@pytest.fixture()
def dbg_cabi():
    from tch_geometric import _cabi
    if not os.path.exists(DBG):
        pytest.skip("debug library not built (make -C tch-geometric_amd dbg)")
    h = C.CDLL(DBG)
    h.tg_version.restype = C.c_char_p
    h.tg_last_error.restype = C.c_char_p
    regular = _cabi.lib
    _cabi.lib = h
    try:
        yield _cabi, regular
    finally:
        h.tg_debug_bounds_set_flag(None)
        _cabi.lib = regular


@pytest.mark.parametrize("form,knobs", [(2, {}), (1, dict(staged=0)), (1, dict(staged=1))])
def test_out_of_range_frontier_id_is_flagged_not_followed(dbg_cabi, form, knobs):
    cabi, regular = dbg_cabi
    dev = torch.device("cuda:0")
    n = 1 << 12
    row, col = cabi.rmat_edges(12, n * 16, 0x5EED000C, dev)
    ptrs, idx, _ = cabi.coo_to_csx(row, col, n, n, True)
    deg = ptrs[1:] - ptrs[:-1]
    hub = int(torch.argmax(deg))
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    assert regular.tg_debug_bounds_set_flag(C.c_void_p(flag.data_ptr())) == 3          # TG_ERR_UNSUPPORTED: no checks there
    assert cabi.lib.tg_debug_bounds_set_flag(C.c_void_p(flag.data_ptr())) == 0
    seeds = cabi.seed_batches(0xBA7C4, 0, 6, 64, n, dev)
    seeds[:, 0] = hub
    fan = [15, 10]
    prev = cabi.ns_win_tuning_set(**knobs) if knobs else None
    try:
        for corrupt in (False, True):
            idx_c = idx.clone()
            if corrupt:
                idx_c[int(ptrs[hub]):int(ptrs[hub + 1])] = n                           # every neighbour of the hub: id n_major
            g = cabi.graph_view(ptrs, idx_c, indices32=idx_c.to(torch.int32), ptrs32=ptrs.to(torch.int32))
            out = cabi.NsBatchedOut(6, 64, fan, dev)
            ws = cabi.ns_homo_workspace(6, 64, fan, dev, staged=True)
            flag.zero_()
            cabi.ns_homo_batched(g, seeds, fan, 0, 0, out, ws=ws if form != 2 else None, form=form)
            torch.cuda.synchronize()
            assert int(flag[0]) == int(corrupt)
            if corrupt:                                                                 # the hub's samples ARE the bad id ...
                s, r, c, e, lo = out.batch(0)
                h1 = lo[1][1]
                assert bool((s[64:64 + h1][c[:h1] == 0] == n).all())
                # ... and their expansion sampled vertex 0's column in their place (edge pointers inside column 0)
                kids = (c[h1:] >= 64) & (s[c[h1:]] == n)
                assert bool(((e[h1:][kids] >= ptrs[0]) & (e[h1:][kids] < ptrs[1])).all())
    finally:
        if prev:
            cabi.ns_win_tuning_set(**prev)
