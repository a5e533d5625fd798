"""The launch form bench.py times, tested at the configuration it is timed on.

bench.py's step is tg_ns_homo_batched_ws(form=auto) over thousands of 1 024-seed batches of RMAT-24 with the u32 shadows:
the window-ordered form with ~2 056 windows, every one of the 8 XCD queues populated, several iterations of the bucket
base scan.  Here that exact shape (2 048 batches; the launch code does not change with the batch count beyond it) is
compared WORD FOR WORD, on the device, with the fused per-batch kernel (used prefixes of all four arrays, counts, layer
offsets; the slabs are poison-filled first and everything beyond the used prefix must still be poison), and batches of
it are replayed by the CPU oracle (neighbor_sampling.rs:188-223).  A mid-size case forces > 1 024 windows through the
tuning hook so that the multi-window sort and the multi-iteration scans also run in the quick suite, and launches with an
ODD slab pitch cover the 16-byte pair stores of the emit kernel on 8-byte-aligned slabs."""
import numpy as np
import pytest
import torch

import orc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
AUTO, WINDOWED, FUSED, WINDOWED_WIDE = 0, 1, 2, 3
POISON = -7


@pytest.fixture(scope="module")
def cabi():
    from tch_geometric import _cabi
    return _cabi


def _rmat(cabi, scale, shadows=True):
    dev = torch.device(DEV)
    n = 1 << scale
    row, col = cabi.rmat_edges(scale, n * 16, 0x5EED0000 + scale, dev)
    ptrs, idx, _ = cabi.coo_to_csx(row, col, n, n, True)
    del row, col
    i32 = idx.to(torch.int32) if shadows else None
    p32 = ptrs.to(torch.int32) if shadows else None
    return n, ptrs, idx, cabi.graph_view(ptrs, idx, indices32=i32, ptrs32=p32)


def _poisoned(cabi, nb, B, fan):
    o = cabi.NsBatchedOut(nb, B, fan, torch.device(DEV))
    for t in (o.samples, o.rows, o.cols, o.edge_index):
        t.fill_(POISON)
    return o


def assert_equal_on_device(a, b):
    """every batch: counts, layer offsets, the used prefix of the four arrays equal; beyond it `a` is untouched"""
    assert torch.equal(a.counts, b.counts) and torch.equal(a.layer_offsets, b.layer_offsets)
    dev = a.samples.device
    for name, col in (("samples", 0), ("rows", 1), ("cols", 1), ("edge_index", 1)):
        x, y = getattr(a, name), getattr(b, name)
        step = 256                                            # batches per comparison: bounds the temporaries
        ar = torch.arange(x.shape[1], device=dev)[None, :]
        for lo in range(0, x.shape[0], step):
            used = ar < a.counts[lo:lo + step, col:col + 1]
            xs, ys = x[lo:lo + step], y[lo:lo + step]
            assert bool(((xs == ys) | ~used).all()), "%s differs in batches [%d, %d)" % (name, lo, lo + step)
            assert bool(((xs == POISON) | used).all()), "%s written beyond its used prefix" % name


def assert_oracle(cabi, out, ptrs, idx, seeds, fan, seed, call0, batches, sampler=0):
    hp, hi = ptrs.cpu().numpy(), idx.cpu().numpy()
    for j in batches:
        o = orc.ns_homo(hp, hi, seeds[j].cpu().numpy(), fan, orc.rng_philox(seed, call0 + j), sampler=sampler)
        x = out.batch(j)
        assert x[4] == o[4]
        for u, v in zip(x[:4], o[:4]):
            assert np.array_equal(u.cpu().numpy(), v), j


def test_bench_shape_rmat24_windowed_equals_fused_and_oracle(cabi):
    """RMAT-24, u32 shadows, 2 048 batches x 1 024 seeds, [15, 10], form = auto: what bench.py launches"""
    n, ptrs, idx, g = _rmat(cabi, 24)
    nb, B, fan = 2048, 1024, [15, 10]
    dev = torch.device(DEV)
    seeds = cabi.seed_batches(0xBA7C4, 4096, nb, B, n, dev)
    a, b = _poisoned(cabi, nb, B, fan), _poisoned(cabi, nb, B, fan)
    ws = cabi.ns_homo_workspace(nb, B, fan, dev, staged=True)
    taken, n_win = cabi.ns_homo_batched_form(g, a, nb, B, fan, ws=ws, form=AUTO)
    assert taken == WINDOWED and n_win >= 2048, (taken, n_win)       # the 16-byte-item windowed form, ~2 056 windows
    assert cabi.ns_win_tuning()["window_bytes"] == 512 * 1024
    cabi.ns_homo_batched(g, seeds, fan, 0, 4096, a, ws=ws, form=AUTO)
    cabi.ns_homo_batched(g, seeds, fan, 0, 4096, b, form=FUSED)
    torch.cuda.synchronize()
    assert int(a.counts[:, 1].sum()) > nb * 20000                     # ~29.5 K sampled edges per batch
    assert_equal_on_device(a, b)
    assert_oracle(cabi, a, ptrs, idx, seeds, fan, 0, 4096, (0, 1023, 2047))
    # the other item form and the un-fused first hops on the same launch
    c = _poisoned(cabi, nb, B, fan)
    cabi.ns_homo_batched(g, seeds, fan, 0, 4096, c, ws=ws, form=WINDOWED_WIDE)
    torch.cuda.synchronize()
    assert_equal_on_device(c, b)
    # variants of the push form (separate histogram pass, un-fused first hops: round 2's pipeline) and the staged form
    # (gather first into 64-byte stage slots, emit afterwards), in one stream and in parts on two
    for knobs in (dict(fold_hist=0), dict(fuse_first_hops=0), dict(staged=1, stage_parts=1), dict(staged=1, stage_parts=2),
                  dict(staged=1, stage_split=1), dict(staged=1, stage_fine=0), dict(staged=1, stage_parts=2, stage_concurrent=1),
                  dict(staged=1, stage_gather_threads=1024, stage_gather_blocks=256), dict(staged=1, stage_fine_sub_bits=4)):
        before = cabi.ns_win_tuning_set(**knobs)
        try:
            d = _poisoned(cabi, nb, B, fan)
            assert cabi.ns_homo_batched_staged(g, d, nb, B, fan, ws=ws, form=AUTO) == bool(knobs.get("staged"))
            cabi.ns_homo_batched(g, seeds, fan, 0, 4096, d, ws=ws, form=AUTO)
            torch.cuda.synchronize()
            assert_equal_on_device(d, b)
        finally:
            cabi.ns_win_tuning_set(**before)


@pytest.mark.parametrize("sampler", [0, 1])
@pytest.mark.parametrize("shadows", [True, False])
def test_many_windows_mid_size(cabi, sampler, shadows):
    """RMAT-16 with 2-KiB (1-KiB without shadows... 4 KiB of i64) windows: > 2 048 windows, >= 256 per XCD queue, three
    iterations of the base scan, 33 blocks of the column scan -- the code paths of the RMAT-24 launch in the quick suite"""
    n, ptrs, idx, g = _rmat(cabi, 16, shadows)
    nb, B, fan = 300, 256, [15, 10]
    dev = torch.device(DEV)
    seeds = cabi.seed_batches(0xBA7C4, 77, nb, B, n, dev)
    before = cabi.ns_win_tuning_set(window_bytes=2048 if shadows else 4096)
    try:
        a, b = _poisoned(cabi, nb, B, fan), _poisoned(cabi, nb, B, fan)
        ws = cabi.ns_homo_workspace(nb, B, fan, dev, staged=True)
        taken, n_win = cabi.ns_homo_batched_form(g, a, nb, B, fan, ws=ws, form=WINDOWED, sampler=sampler)
        assert taken == WINDOWED and n_win > 2048, (taken, n_win)
        cabi.ns_homo_batched(g, seeds, fan, 9, 77, a, sampler=sampler, ws=ws, form=WINDOWED)
        cabi.ns_homo_batched(g, seeds, fan, 9, 77, b, sampler=sampler, form=FUSED)
        torch.cuda.synchronize()
        assert_equal_on_device(a, b)
        assert_oracle(cabi, a, ptrs, idx, seeds, fan, 9, 77, (0, 150, 299), sampler=sampler)
        for knobs in (dict(direct_hop0=0), dict(fuse_first_hops=0), dict(fold_hist=0), dict(emit_blocks=7),
                      dict(emit_blocks=1000, emit_threads=128), dict(gather_blocks=64, gather_threads=128),
                      dict(staged=1), dict(staged=1, stage_round_chunks=1),
                      dict(staged=1, stage_round_chunks=16, stage_emit_threads=128),
                      dict(staged=1, stage_gather_threads=128, stage_gather_blocks=8), dict(staged=1, emit_threads=128),
                      dict(staged=1, stage_parts=3, stage_part_min_batches=8),
                      dict(staged=1, stage_parts=16, stage_part_min_batches=1), dict(staged=1, stage_parts=1),
                      dict(staged=1, stage_split=1), dict(staged=1, stage_split=1, emit_threads=128, stage_fine=0),
                      dict(staged=1, stage_parts=4, stage_part_min_batches=2, stage_concurrent=1),
                      dict(staged=1, stage_sort_blocks=3), dict(staged=1, stage_sort_blocks=1024),
                      dict(staged=1, stage_gather_threads=1024, stage_gather_blocks=16),
                      dict(staged=1, stage_fine_sub_bits=4), dict(staged=1, stage_fine_sub_bits=5, stage_fine_blocks=8),
                      dict(staged=1, stage_fine_sub_bits=6, stage_parts=3, stage_part_min_batches=8)):
            prev = cabi.ns_win_tuning_set(**knobs)
            try:
                c = _poisoned(cabi, nb, B, fan)
                assert cabi.ns_homo_batched_staged(g, c, nb, B, fan, ws=ws, form=WINDOWED, sampler=sampler) == \
                    bool(knobs.get("staged")), knobs
                cabi.ns_homo_batched(g, seeds, fan, 9, 77, c, sampler=sampler, ws=ws, form=WINDOWED)
                torch.cuda.synchronize()
                assert_equal_on_device(c, b)
            finally:
                cabi.ns_win_tuning_set(**prev)
    finally:
        cabi.ns_win_tuning_set(**before)
    assert cabi.ns_win_tuning() == before


@pytest.mark.parametrize("fan,B", [([3], 3), ([3, 3, 3], 3), ([5], 7), ([1, 1, 1], 5), ([15, 9, 3], 1)])
@pytest.mark.parametrize("direct", [1, 0, 2])
def test_odd_slab_pitch(cabi, fan, B, direct):
    """odd cap_edges: the slabs of odd batches start on an odd element, i.e. 8-byte-aligned; the emit kernel's 16-byte pair
    stores take their alignment from the address (ns_homo_win.hip)"""
    n, ptrs, idx, g = _rmat(cabi, 12)
    nb = 9
    dev = torch.device(DEV)
    assert cabi.ns_homo_capacity(B, fan)[1] % 2 == 1
    seeds = cabi.seed_batches(0xBA7C4, 5, nb, B, n, dev)
    seeds[:, 0] = int(torch.argmax(ptrs[1:] - ptrs[:-1]))            # a hub in every batch: columns longer than the fan-out
    before = cabi.ns_win_tuning_set(direct_hop0=min(direct, 1), staged=int(direct == 2))   # 2: the staged form
    try:
        a, b = _poisoned(cabi, nb, B, fan), _poisoned(cabi, nb, B, fan)
        ws = cabi.ns_homo_workspace(nb, B, fan, dev, staged=True)
        assert cabi.ns_homo_batched_form(g, a, nb, B, fan, ws=ws, form=WINDOWED)[0] == WINDOWED
        # the staged pipeline needs a second hop and ordered fan-outs <= 30: the others take the push form
        assert cabi.ns_homo_batched_staged(g, a, nb, B, fan, ws=ws, form=WINDOWED) == (direct == 2 and len(fan) > 1)
        cabi.ns_homo_batched(g, seeds, fan, 2, 5, a, ws=ws, form=WINDOWED)
        cabi.ns_homo_batched(g, seeds, fan, 2, 5, b, form=FUSED)
        torch.cuda.synchronize()
        assert_equal_on_device(a, b)
        assert_oracle(cabi, a, ptrs, idx, seeds, fan, 2, 5, range(nb))
    finally:
        cabi.ns_win_tuning_set(**before)


def test_form_query(cabi):
    n, ptrs, idx, g = _rmat(cabi, 12)
    dev = torch.device(DEV)
    out = cabi.NsBatchedOut(4, 64, [15, 10], dev)
    ws = cabi.ns_homo_workspace(4, 64, [15, 10], dev)
    assert cabi.ns_homo_batched_form(g, out, 4, 64, [15, 10], ws=None, form=WINDOWED)[0] == FUSED     # no workspace
    assert cabi.ns_homo_batched_form(g, out, 4, 64, [15, 10], ws=ws, form=AUTO)[0] == FUSED          # too small to pay
    assert cabi.ns_homo_batched_form(g, out, 4, 64, [15, 10], ws=ws, form=WINDOWED)[0] == WINDOWED
    assert cabi.ns_homo_batched_form(g, out, 4, 64, [15, 10], ws=ws, form=WINDOWED_WIDE)[0] == WINDOWED_WIDE
    assert cabi.ns_homo_batched_form(g, out, 4, 64, [15, 10], ws=ws, form=WINDOWED, sampler=2)[0] == FUSED
    assert cabi.ns_homo_batched_form(g, out, 4, 64, [15, 10], ws=ws[:8], form=WINDOWED)[0] == FUSED


def test_push_sized_workspace_takes_the_push_pipeline(cabi):
    n, ptrs, idx, g = _rmat(cabi, 12)
    dev = torch.device(DEV)
    nb, B, fan = 8, 64, [15, 10]
    seeds = cabi.seed_batches(0xBA7C4, 0, nb, B, n, dev)
    small, big = cabi.ns_homo_workspace(nb, B, fan, dev, staged=False), cabi.ns_homo_workspace(nb, B, fan, dev, staged=True)
    assert big.numel() > small.numel()
    before = cabi.ns_win_tuning_set(staged=1)
    try:
        a, b = _poisoned(cabi, nb, B, fan), _poisoned(cabi, nb, B, fan)
        assert not cabi.ns_homo_batched_staged(g, a, nb, B, fan, ws=small, form=WINDOWED)
        assert cabi.ns_homo_batched_staged(g, a, nb, B, fan, ws=big, form=WINDOWED)
        cabi.ns_homo_batched(g, seeds, fan, 0, 0, a, ws=small, form=WINDOWED)
        cabi.ns_homo_batched(g, seeds, fan, 0, 0, b, ws=big, form=WINDOWED)
        torch.cuda.synchronize()
        assert_equal_on_device(a, b)
    finally:
        cabi.ns_win_tuning_set(**before)


def test_stage_times(cabi):
    n, ptrs, idx, g = _rmat(cabi, 14)
    dev = torch.device(DEV)
    nb, B, fan = 64, 128, [15, 10]
    seeds = cabi.seed_batches(0xBA7C4, 0, nb, B, n, dev)
    out, ws = cabi.NsBatchedOut(nb, B, fan, dev), cabi.ns_homo_workspace(nb, B, fan, dev, staged=True)
    cabi.ns_win_stage_timing(True)
    try:
        cabi.ns_homo_batched(g, seeds, fan, 0, 0, out, ws=ws, form=WINDOWED)
        st = cabi.ns_win_stage_times()
    finally:
        cabi.ns_win_stage_timing(False)
    names = [s for s, _ in st]
    assert names[0].startswith("first_hops") and names[-1] == "gather.h1" and all(ms >= 0 for _, ms in st)  # the push form
    before = cabi.ns_win_tuning_set(staged=1)
    cabi.ns_win_stage_timing(True)
    try:
        cabi.ns_homo_batched(g, seeds, fan, 0, 0, out, ws=ws, form=WINDOWED)
        st = cabi.ns_win_stage_times()
    finally:
        cabi.ns_win_stage_timing(False)
        cabi.ns_win_tuning_set(**before)
    names = [s for s, _ in st]
    assert names[0] == "first.h0" and names[-1] == "emit.h1"                                         # the staged form


# ---------------------------------------------------------------- round 4: packed stage slots carrying the positions
@pytest.mark.parametrize("fan,hint", [([4, 20], None), ([3, 30], None), ([15, 10], "auto"), ([15, 10], 1 << 30),
                                      ([6, 16], "auto"), ([2, 3, 4], "auto")])
@pytest.mark.parametrize("sampler", [0, 1])
def test_staged_slot_formats(cabi, fan, hint, sampler):
    """one-chunk and two-chunk slots (fan-outs 17..30 and wide bit fields take two), with and without the max-degree hint:
    the same outputs as the fused kernel and the oracle"""
    dev = torch.device(DEV)
    n = 1 << 12
    row, col = cabi.rmat_edges(12, n * 16, 0x5EED0000 + 12, dev)
    ptrs, idx, _ = cabi.coo_to_csx(row, col, n, n, True)
    g = cabi.graph_view(ptrs, idx, indices32=idx.to(torch.int32), ptrs32=ptrs.to(torch.int32), max_degree=hint)
    if hint == "auto":
        assert g.max_degree == int((ptrs[1:] - ptrs[:-1]).max())
    nb, B = 24, 48
    seeds = cabi.seed_batches(0xBA7C4, 3, nb, B, n, dev)
    seeds[:, 0] = int(torch.argmax(ptrs[1:] - ptrs[:-1]))
    before = cabi.ns_win_tuning_set(staged=1, window_bytes=2048)
    try:
        for parts, split, sub in ((1, 0, 7), (3, 0, 7), (1, 1, 7), (1, 0, 4), (3, 0, 5)):
            cabi.ns_win_tuning_set(stage_parts=parts, stage_part_min_batches=4, stage_split=split, stage_fine_sub_bits=sub)
            a, b = _poisoned(cabi, nb, B, fan), _poisoned(cabi, nb, B, fan)
            ws = cabi.ns_homo_workspace(nb, B, fan, dev, staged=True, graph=g)
            assert cabi.ns_homo_batched_staged(g, a, nb, B, fan, ws=ws, form=WINDOWED, sampler=sampler)
            cabi.ns_homo_batched(g, seeds, fan, 9, 40, a, ws=ws, form=WINDOWED, sampler=sampler)
            cabi.ns_homo_batched(g, seeds, fan, 9, 40, b, form=FUSED, sampler=sampler)
            torch.cuda.synchronize()
            assert_equal_on_device(a, b)
            assert_oracle(cabi, a, ptrs, idx, seeds, fan, 9, 40, (0, nb - 1), sampler=sampler)
    finally:
        cabi.ns_win_tuning_set(**before)


def test_workspace_for_a_graph_is_smaller_when_the_pairs_fit_one_chunk(cabi):
    n, ptrs, idx, g = _rmat(cabi, 12)
    dev = torch.device(DEV)
    nb, B, fan = 64, 64, [15, 10]
    free_of_graph = cabi.ns_homo_workspace(nb, B, fan, dev, staged=True)
    for_graph = cabi.ns_homo_workspace(nb, B, fan, dev, staged=True, graph=g)       # 12 + 16 bits per pair: one chunk
    push_only = cabi.ns_homo_workspace(nb, B, fan, dev, staged=False)
    assert push_only.numel() < for_graph.numel() < free_of_graph.numel()
    # one 64-byte chunk per frontier slot of the widest ordered hop (nb * B * 15 slots) instead of two
    assert abs((free_of_graph.numel() - for_graph.numel()) * 8 - nb * B * 15 * 64) <= 512


def test_slot_draw_fallback_runs_on_the_device(cabi):
    """A rejected 32-bit word (Lemire's test; probability (2^32 mod range) / 2^32) is replaced by the slot's 64-bit fallback
    draw.  A hub column of 3 000 001 edges (2^32 mod 3 000 000 = 1 967 296) makes ~1 draw in 2 000 fall back: the fused
    kernel, the window-ordered forms and the flat hop must take exactly the oracle's fallbacks."""
    dev = torch.device(DEV)
    n, hub = 512, 3_000_001
    rs = np.random.default_rng(4)
    deg = np.full(n, 4, dtype=np.int64)
    deg[0] = hub
    ptrs_h = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(deg, out=ptrs_h[1:])
    idx_h = rs.integers(0, n, int(ptrs_h[-1]))
    idx_h[:hub:7] = 0                                                # the hub is its own frequent neighbour: hop 2 draws from it too
    ptrs, idx = torch.from_numpy(ptrs_h).to(dev), torch.from_numpy(idx_h).to(dev)
    g = cabi.graph_view(ptrs, idx, indices32=idx.to(torch.int32), ptrs32=ptrs.to(torch.int32), max_degree="auto")
    assert g.max_degree == hub
    nb, B, fan = 16, 256, [15, 10]
    seeds = torch.zeros((nb, B), dtype=torch.int64, device=dev)    # every seed is the hub
    for sampler in (0, 1):
        f = _poisoned(cabi, nb, B, fan)
        cabi.ns_homo_batched(g, seeds, fan, 5, 100, f, form=FUSED, sampler=sampler)
        torch.cuda.synchronize()
        orc.slot_fallback_count(reset=True)
        assert_oracle(cabi, f, ptrs, idx, seeds, fan, 5, 100, range(4), sampler=sampler)
        assert orc.slot_fallback_count() >= 5, "the case does not reach the fallback"
        for staged in (0, 1):
            before = cabi.ns_win_tuning_set(staged=staged)
            try:
                w = _poisoned(cabi, nb, B, fan)
                ws = cabi.ns_homo_workspace(nb, B, fan, dev, staged=True, graph=g)
                cabi.ns_homo_batched(g, seeds, fan, 5, 100, w, ws=ws, form=WINDOWED, sampler=sampler)
                torch.cuda.synchronize()
                assert bool(cabi.ns_homo_batched_staged(g, w, nb, B, fan, ws=ws, form=WINDOWED, sampler=sampler)) == bool(staged)
                assert_equal_on_device(w, f)
            finally:
                cabi.ns_win_tuning_set(**before)
    # the flat hop (lane-per-slot chains, any fan-out) on the same column
    verts = torch.zeros(2048, dtype=torch.int64, device=dev)
    for k, sampler in ((15, 0), (200, 0), (15, 1)):
        cnt, offsets, nbr, ep, _ = cabi.ns_hop(g, verts, k, 5, call_id=7, sampler=sampler)
        torch.cuda.synchronize()
        orc.slot_fallback_count(reset=True)
        o = orc.ns_homo(ptrs_h, idx_h, np.zeros(2048, dtype=np.int64), [k], orc.rng_philox(5, 7), sampler=sampler)
        assert orc.slot_fallback_count() >= 3
        m = int(offsets[-1])
        assert m == len(o[3]) and np.array_equal(ep[:m].cpu().numpy(), o[3]) and np.array_equal(nbr[:m].cpu().numpy(), o[0][2048:])


@pytest.mark.parametrize("align64", [1, 0])
def test_frontiers_of_several_scatter_tiles_and_both_store_alignments(cabi, align64):
    """sort level 1 works through LDS tiles of 4 096 items: seeds on the graph's heaviest columns give every batch a second
    hop's frontier of > 12 000 items (three to four tiles per batch, the last one ragged); the emit passes' store
    instructions start on 64-byte boundaries (store_align64 = 1, the default) or on 16-byte ones -- the same outputs as
    the fused kernel and the oracle either way, odd pitches included (B = 1 023: slabs of odd batches start on an odd
    element)"""
    dev = torch.device(DEV)
    n, ptrs, idx, g = _rmat(cabi, 16)
    deg = ptrs[1:] - ptrs[:-1]
    heavy = torch.argsort(deg, descending=True)[:4096]
    nb, fan = 12, [15, 10]
    gen = torch.Generator(device=dev).manual_seed(5)
    for B in (1024, 1023):
        seeds = heavy[torch.randint(0, heavy.numel(), (nb, B), device=dev, generator=gen)].contiguous()
        before = cabi.ns_win_tuning_set(staged=1, window_bytes=1 << 16, store_align64=align64)
        try:
            a, b = _poisoned(cabi, nb, B, fan), _poisoned(cabi, nb, B, fan)
            ws = cabi.ns_homo_workspace(nb, B, fan, dev, staged=True, graph=g)
            assert cabi.ns_homo_batched_staged(g, a, nb, B, fan, ws=ws, form=WINDOWED)
            cabi.ns_homo_batched(g, seeds, fan, 3, 700, a, ws=ws, form=WINDOWED)
            cabi.ns_homo_batched(g, seeds, fan, 3, 700, b, form=FUSED)
            torch.cuda.synchronize()
            assert int(a.layer_offsets[:, 1, 0].min()) - B > 12000      # every batch's second frontier spans several tiles
            assert_equal_on_device(a, b)
            assert_oracle(cabi, a, ptrs, idx, seeds, fan, 3, 700, (0, nb - 1))
        finally:
            cabi.ns_win_tuning_set(**before)


def test_pipeline_is_chosen_by_launch_size_by_default(cabi):
    """tg_ns_win_tuning.staged = 2 (the default): the staged pipeline from 2 048 batches on when the stage slots are one
    chunk, the push pipeline below that and for two-chunk slots; the workspace query follows the same rule; both equal the
    fused kernel"""
    dev = torch.device(DEV)
    n, ptrs, idx, g = _rmat(cabi, 12)
    assert cabi.ns_win_tuning()["staged"] == 2
    B = 4
    for nb, fan, want in ((2048, [15, 10], True), (2047, [15, 10], False), (2048, [4, 20], False)):
        ws = cabi.ns_homo_workspace(nb, B, fan, dev, graph=g)                     # sized as the rule says
        push_only = cabi.ns_homo_workspace(nb, B, fan, dev, staged=False, graph=g)
        assert (ws.numel() > push_only.numel()) == want
        a, b = _poisoned(cabi, nb, B, fan), _poisoned(cabi, nb, B, fan)
        assert bool(cabi.ns_homo_batched_staged(g, a, nb, B, fan, ws=ws, form=WINDOWED)) == want
        seeds = cabi.seed_batches(0xBA7C4, 0, nb, B, n, dev)
        cabi.ns_homo_batched(g, seeds, fan, 1, 0, a, ws=ws, form=WINDOWED)
        cabi.ns_homo_batched(g, seeds, fan, 1, 0, b, form=FUSED)
        torch.cuda.synchronize()
        assert_equal_on_device(a, b)
        del a, b, ws, push_only
