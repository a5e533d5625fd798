"""ctypes front-end of the CPU oracle (oracle/tg_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by the shipped package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# ORC_LIB=asan selects the AddressSanitizer/UBSan build (`make -C oracle asan`; run the CPU tests with
# LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0)
_LIB_PATH = os.path.join(_HERE, "_build", "libtg_oracle_asan.so" if os.environ.get("ORC_LIB") == "asan" else "libtg_oracle.so")

RNG_REF, RNG_PHILOX = 0, 1
RES_TICKETS, RES_LITERAL, RES_CHUNKED, RES_AUTO = 0, 1, 2, -1
SAMPLER_UNIFORM, SAMPLER_UNIFORM_REPL, SAMPLER_WEIGHTED = 0, 1, 2
FILTER_NONE, FILTER_STATIC, FILTER_RELATIVE, FILTER_DYNAMIC = -1, 0, 1, 2
TAG_NS_HOMO, TAG_NS_HETERO, TAG_RW, TAG_RW_TEMPO, TAG_NEG_HOMO, TAG_NEG_HETERO, TAG_HGT = 1, 2, 3, 4, 5, 6, 7


class Rng(C.Structure):
    _fields_ = [("mode", C.c_int32), ("_pad", C.c_int32), ("s", C.c_uint64 * 4), ("seed", C.c_uint64),
                ("call_id", C.c_uint64), ("raw_draws", C.c_uint64)]


class NsCfg(C.Structure):
    _fields_ = [("sampler", C.c_int32), ("filter_mode", C.c_int32), ("forward", C.c_int32),
                ("reservoir_algo", C.c_int32), ("win_lo", C.c_int64), ("win_hi", C.c_int64),
                ("weights", C.c_void_p), ("timestamps", C.c_void_p), ("id_base", C.c_int64)]


def build(force=False):
    """Compile the oracle with the committed Makefile (gcc only)."""
    if force or not os.path.exists(_LIB_PATH):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_rng_next_u64.restype = C.c_uint64
        _lib.orc_rng_gen_range_u64.restype = C.c_uint64
        _lib.orc_rng_gen_range_u64.argtypes = [C.c_void_p, C.c_uint64]
        _lib.orc_rng_gen_range_f32.restype = C.c_float
        _lib.orc_rng_gen_range_f32.argtypes = [C.c_void_p, C.c_float]
        _lib.orc_rng_gen_range_f64.restype = C.c_double
        _lib.orc_rng_gen_range_f64.argtypes = [C.c_void_p, C.c_double, C.c_double]
        _lib.orc_rng_philox.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64]
        _lib.orc_reservoir_positions.restype = C.c_int64
        _lib.orc_reservoir_positions.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64, C.c_int64, C.c_int64, C.c_int,
                                                 C.c_void_p]
        _lib.orc_philox_named_draw.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint32,
                                               C.c_uint32, C.c_void_p]
        _lib.orc_ns_hetero.restype = C.c_void_p
        _lib.orc_neg_hetero.restype = C.c_void_p
        _lib.orc_het_num_samples.restype = C.c_int64
        _lib.orc_het_num_samples.argtypes = [C.c_void_p, C.c_int]
        _lib.orc_het_num_edges.restype = C.c_int64
        _lib.orc_het_num_edges.argtypes = [C.c_void_p, C.c_int]
        _lib.orc_het_copy_samples.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        _lib.orc_het_copy_edges.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.orc_het_copy_layer_offsets.argtypes = [C.c_void_p, C.c_void_p]
        _lib.orc_het_free.argtypes = [C.c_void_p]
        _lib.orc_neg_homo.restype = C.c_int64
        _lib.orc_bench_ns_homo.restype = C.c_double
        if hasattr(_lib, "orc_hgt"):
            _lib.orc_hgt.restype = C.c_void_p
    return _lib


def _p(a):
    return C.c_void_p(a.ctypes.data) if a is not None else C.c_void_p(0)


def _i64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.int64))


# ---------------------------------------------------------------- RNG
def rng_ref(seed32=bytes(32)):
    """rand 0.8.5 SmallRng::from_seed(seed32) -- the reference tests use [0;32]."""
    r = Rng()
    lib().orc_rng_ref_from_seed(C.byref(r), (C.c_uint8 * 32).from_buffer_copy(bytes(seed32)))
    return r


def rng_ref_state(s4):
    r = Rng()
    lib().orc_rng_ref_from_state(C.byref(r), (C.c_uint64 * 4)(*s4))
    return r


def rng_ref_child(parent):
    r = Rng()
    lib().orc_rng_ref_child(C.byref(r), C.byref(parent))
    return r


def rng_philox(seed, call_id=0):
    r = Rng()
    lib().orc_rng_philox(C.byref(r), seed, call_id)
    return r


def philox_raw(ctr4, key2):
    out = (C.c_uint32 * 4)()
    lib().orc_philox_raw((C.c_uint32 * 4)(*ctr4), (C.c_uint32 * 2)(*key2), out)
    return list(out)


def philox_named_draw(seed, call_id, tag, id_, d0, d1):
    out = (C.c_uint32 * 4)()
    lib().orc_philox_named_draw(seed, call_id, tag, id_, d0, d1, out)
    return list(out)


def bounded_word(w, range_):
    """Lemire's bounded integer from one 32-bit word -> (value, accepted)."""
    ok = C.c_int(0)
    f = lib().orc_test_bounded_word
    f.restype = C.c_uint32
    v = f(C.c_uint32(w), C.c_uint32(range_), C.byref(ok))
    return int(v), bool(ok.value)


def slot_draw(rng, tag, id_, d1, s, range_):
    """philox-mode's bounded draw of slot s -> (value, fell back to the 64-bit draw)."""
    fb = C.c_int(0)
    f = lib().orc_test_slot_draw
    f.restype = C.c_uint64
    v = f(C.byref(rng), C.c_uint32(tag), C.c_uint64(id_), C.c_uint32(d1), C.c_int64(s), C.c_uint64(range_), C.byref(fb))
    return int(v), bool(fb.value)


def slot_fallback_count(reset=False):
    """64-bit fallback draws the philox-mode slot draw has taken so far (a rejected 32-bit word each)."""
    f = lib().orc_slot_fallback_count
    f.restype = C.c_uint64
    return int(f(C.c_int(int(reset))))


def reservoir_positions(rng, n, k, algo=RES_TICKETS, tag=TAG_NS_HOMO, id_=0):
    dst = np.zeros(max(k, 1), dtype=np.int64)
    cnt = lib().orc_reservoir_positions(C.byref(rng), tag, id_, n, k, algo, _p(dst))
    return dst[:cnt].copy()


# ---------------------------------------------------------------- ingest
def ind2ptr(ind, m):
    ind = _i64(ind)
    out = np.empty(m + 1, dtype=np.int64)
    lib().orc_ind2ptr(_p(ind), C.c_int64(ind.size), C.c_int64(m), _p(out))
    return out


def to_csx(row_col, size, csc):
    row_col = _i64(row_col)
    row, col = np.ascontiguousarray(row_col[0]), np.ascontiguousarray(row_col[1])
    size0, size1 = (size, size) if np.isscalar(size) else size
    nnz = row.size
    ptrs = np.empty((size1 if csc else size0) + 1, dtype=np.int64)
    indices = np.empty(nnz, dtype=np.int64)
    perm = np.empty(nnz, dtype=np.int64)
    lib().orc_to_csx(_p(row), _p(col), C.c_int64(nnz), C.c_int64(size0), C.c_int64(size1), C.c_int(int(csc)),
                     _p(ptrs), _p(indices), _p(perm))
    return ptrs, indices, perm


def to_csc(row_col, size):
    return to_csx(row_col, size, True)


def to_csr(row_col, size):
    return to_csx(row_col, size, False)


# ---------------------------------------------------------------- neighbor sampling
def _mk_cfg(sampler, weights, filter_mode, forward, window, timestamps, reservoir_algo, keep):
    cfg = NsCfg()
    cfg.sampler = sampler
    cfg.filter_mode = filter_mode
    cfg.forward = int(bool(forward))
    cfg.reservoir_algo = reservoir_algo
    cfg.win_lo, cfg.win_hi = (window if window is not None else (0, 0))
    if weights is not None:
        w = np.ascontiguousarray(np.asarray(weights, dtype=np.float64))
        keep.append(w)
        cfg.weights = w.ctypes.data
    if timestamps is not None:
        t = _i64(timestamps)
        keep.append(t)
        cfg.timestamps = t.ctypes.data
    return cfg


def ns_capacity(B, fanout):
    cap, layer = B, B
    for k in fanout:
        layer *= int(k)
        cap += layer
    return cap


def ns_homo(ptrs, indices, inputs, fanout, rng, sampler=SAMPLER_UNIFORM, weights=None, filter_mode=FILTER_NONE,
            forward=False, window=None, timestamps=None, inputs_state=None, reservoir_algo=RES_AUTO, id_base=0):
    """neighbor_sampling_homogenous -> (samples, rows, cols, edge_index, layer_offsets)."""
    ptrs, indices, inputs = _i64(ptrs), _i64(indices), _i64(inputs)
    fan = _i64(fanout)
    keep = []
    cfg = _mk_cfg(sampler, weights, filter_mode, forward, window, timestamps, reservoir_algo, keep)
    cfg.id_base = id_base
    B, H = inputs.size, fan.size
    cap = ns_capacity(B, fanout)
    samples = np.empty(cap, dtype=np.int64)
    rows = np.empty(cap, dtype=np.int64)
    cols = np.empty(cap, dtype=np.int64)
    eidx = np.empty(cap, dtype=np.int64)
    lo = np.zeros(3 * max(H, 1), dtype=np.int64)
    counts = np.zeros(2, dtype=np.int64)
    st = _i64(inputs_state) if inputs_state is not None else None
    rc = lib().orc_ns_homo(_p(ptrs), _p(indices), _p(inputs), C.c_int64(B), _p(fan), C.c_int32(H), C.byref(cfg),
                           _p(st), C.byref(rng), _p(samples), C.c_int64(cap), _p(rows), _p(cols), _p(eidx),
                           C.c_int64(cap), _p(lo), _p(counts))
    if rc != 0:
        raise RuntimeError("oracle ns_homo failed rc=%d (reference would panic)" % rc)
    ns, ne = int(counts[0]), int(counts[1])
    layer_offsets = [tuple(int(x) for x in lo[3 * h:3 * h + 3]) for h in range(H)]
    return samples[:ns].copy(), rows[:ne].copy(), cols[:ne].copy(), eidx[:ne].copy(), layer_offsets


def _ptr_array(arrs):
    return (C.c_void_p * len(arrs))(*[a.ctypes.data if a is not None else None for a in arrs])


def _rel_key(et):
    return "%s__%s__%s" % tuple(et)


def ns_hetero(node_types, edge_types, col_ptrs, row_indices, inputs, num_neighbors, num_hops, rng,
              sampler=SAMPLER_UNIFORM, weights=None, filter_mode=FILTER_NONE, forward=False, window=None,
              timestamps=None, inputs_state=None, reservoir_algo=RES_AUTO):
    """neighbor_sampling_heterogenous in canonical (edge_types) relation order.

    dicts are keyed like python.rs: relations by "src__rel__dst", node types by name."""
    T, R, H = len(node_types), len(edge_types), num_hops
    tix = {t: i for i, t in enumerate(node_types)}
    rels = [_rel_key(e) for e in edge_types]
    rel_src = (C.c_int32 * R)(*[tix[e[0]] for e in edge_types])
    rel_dst = (C.c_int32 * R)(*[tix[e[2]] for e in edge_types])
    P = [_i64(col_ptrs[r]) for r in rels]
    I = [_i64(row_indices[r]) for r in rels]
    IN = [_i64(inputs[t]) if t in inputs else np.zeros(0, dtype=np.int64) for t in node_types]
    n_in = _i64([a.size for a in IN])
    fan = _i64([[num_neighbors[r][h] for h in range(H)] for r in rels]).reshape(-1)
    keep = []
    cfgs = (NsCfg * R)()
    for i, r in enumerate(rels):
        cfgs[i] = _mk_cfg(sampler, weights[r] if weights is not None else None, filter_mode, forward, window,
                          timestamps[r] if timestamps is not None else None, reservoir_algo, keep)
    ST = None
    if inputs_state is not None:
        ST = [_i64(inputs_state[t]) if t in inputs_state else None for t in node_types]
    status = C.c_int32(0)
    h = lib().orc_ns_hetero(C.c_int32(T), C.c_int32(R), rel_src, rel_dst, _ptr_array(P), _ptr_array(I),
                            _ptr_array(IN), _p(n_in), _p(fan), C.c_int32(H), cfgs,
                            _ptr_array(ST) if ST is not None else C.c_void_p(0), C.byref(rng), C.byref(status))
    try:
        if status.value != 0:
            raise RuntimeError("oracle ns_hetero failed")
        return _unpack_het(h, node_types, rels, H)
    finally:
        lib().orc_het_free(C.c_void_p(h))


def _unpack_het(h, node_types, rels, H, with_eidx=True):
    L = lib()
    h = C.c_void_p(h)
    samples, rows, cols, eidx, los = {}, {}, {}, {}, {}
    for t, name in enumerate(node_types):
        n = L.orc_het_num_samples(h, t)
        a = np.empty(n, dtype=np.int64)
        L.orc_het_copy_samples(h, t, _p(a))
        samples[name] = a
    lo = np.zeros(max(len(rels) * H * 3, 1), dtype=np.int64)
    if H:
        L.orc_het_copy_layer_offsets(h, _p(lo))
    for r, name in enumerate(rels):
        n = L.orc_het_num_edges(h, r)
        a, b, c = (np.empty(n, dtype=np.int64) for _ in range(3))
        L.orc_het_copy_edges(h, r, _p(a), _p(b), _p(c))
        rows[name], cols[name], eidx[name] = a, b, c
        los[name] = [tuple(int(x) for x in lo[(r * H + l) * 3:(r * H + l) * 3 + 3]) for l in range(H)]
    return samples, rows, cols, eidx, los


# ---------------------------------------------------------------- walks
def random_walk(row_ptrs, col_indices, start, walk_length, p, q, rng):
    ptrs, indices, start = _i64(row_ptrs), _i64(col_indices), _i64(start)
    walks = np.empty((start.size, walk_length + 1), dtype=np.int64)
    lib().orc_random_walk(_p(ptrs), _p(indices), _p(start), C.c_int64(start.size), C.c_int64(walk_length),
                          C.c_float(p), C.c_float(q), C.byref(rng), _p(walks))
    return walks


def tempo_random_walk(row_ptrs, col_indices, node_ts, edge_ts, start, start_ts, walk_length, window, rng,
                      reservoir_algo=RES_AUTO):
    ptrs, indices, start = _i64(row_ptrs), _i64(col_indices), _i64(start)
    node_ts, edge_ts, start_ts = _i64(node_ts), _i64(edge_ts), _i64(start_ts)
    walks = np.empty((start.size, walk_length), dtype=np.int64)
    wts = np.empty((start.size, walk_length), dtype=np.int64)
    lib().orc_tempo_random_walk(_p(ptrs), _p(indices), _p(node_ts), _p(edge_ts), _p(start), _p(start_ts),
                                C.c_int64(start.size), C.c_int64(walk_length), C.c_int64(window[0]),
                                C.c_int64(window[1]), C.c_int32(reservoir_algo), C.byref(rng), _p(walks), _p(wts))
    return walks, wts


BIAS = {"uniform": 0, "linear": 1, "exponential": 2}


def biased_tempo_random_walk(row_ptrs, col_indices, node_ts, edge_ts, start, start_ts, walk_length, bias_type, forward,
                             retry_count, rng):
    """random_walk.rs:184-288; raises RuntimeError where the reference panics."""
    ptrs, indices, start = _i64(row_ptrs), _i64(col_indices), _i64(start)
    node_ts, edge_ts, start_ts = _i64(node_ts), _i64(edge_ts), _i64(start_ts)
    walks = np.empty((start.size, walk_length), dtype=np.int64)
    wts = np.empty((start.size, walk_length), dtype=np.int64)
    rc = lib().orc_biased_tempo_random_walk(_p(ptrs), _p(indices), _p(node_ts), _p(edge_ts), _p(start), _p(start_ts),
                                            C.c_int64(start.size), C.c_int64(walk_length), C.c_int32(BIAS[bias_type]),
                                            C.c_int32(int(forward)), C.c_int64(retry_count), C.byref(rng), _p(walks),
                                            _p(wts))
    if rc != 0:
        raise RuntimeError("cannot sample empty range (sampling.rs:49)")
    return walks, wts


# ---------------------------------------------------------------- negative sampling
def neg_homo(row_ptrs, col_indices, graph_size, inputs, num_neg, try_count, rng):
    ptrs, indices, inputs = _i64(row_ptrs), _i64(col_indices), _i64(inputs)
    B = inputs.size
    samples = np.empty(B * (1 + num_neg), dtype=np.int64)
    rows = np.empty(max(B * num_neg, 1), dtype=np.int64)
    cols = np.empty(max(B * num_neg, 1), dtype=np.int64)
    counts = np.zeros(2, dtype=np.int64)
    sc = lib().orc_neg_homo(_p(ptrs), _p(indices), C.c_int64(graph_size[1]), _p(inputs), C.c_int64(B),
                            C.c_int64(num_neg), C.c_int64(try_count), C.byref(rng), _p(samples), _p(rows), _p(cols),
                            _p(counts))
    ns, ne = int(counts[0]), int(counts[1])
    return samples[:ns].copy(), rows[:ne].copy(), cols[:ne].copy(), int(sc)


def neg_hetero(node_types, edge_types, row_ptrs, col_indices, sizes, inputs, num_neg, try_count, inbound, rng):
    T, R = len(node_types), len(edge_types)
    tix = {t: i for i, t in enumerate(node_types)}
    rels = [_rel_key(e) for e in edge_types]
    rel_src = (C.c_int32 * R)(*[tix[e[0]] for e in edge_types])
    rel_dst = (C.c_int32 * R)(*[tix[e[2]] for e in edge_types])
    P = [_i64(row_ptrs[r]) for r in rels]
    I = [_i64(col_indices[r]) for r in rels]
    SZ = _i64([list(sizes[r]) for r in rels]).reshape(-1)
    IN = [_i64(inputs[t]) if t in inputs else np.zeros(0, dtype=np.int64) for t in node_types]
    n_in = _i64([a.size for a in IN])
    has = (C.c_int32 * T)(*[int(t in inputs) for t in node_types])
    sc = np.zeros(T, dtype=np.int64)
    status = C.c_int32(0)
    h = lib().orc_neg_hetero(C.c_int32(T), C.c_int32(R), rel_src, rel_dst, _ptr_array(P), _ptr_array(I), _p(SZ),
                             _ptr_array(IN), _p(n_in), has, C.c_int64(num_neg), C.c_int64(try_count),
                             C.c_int32(int(inbound)), C.byref(rng), _p(sc), C.byref(status))
    try:
        if status.value != 0:
            raise RuntimeError("oracle neg_hetero: the reference would panic (inbound has_edge row out of range)")
        samples, rows, cols, _, _ = _unpack_het(h, node_types, rels, 0)
    finally:
        lib().orc_het_free(C.c_void_p(h))
    return samples, rows, cols, {t: int(sc[i]) for i, t in enumerate(node_types)}


# ---------------------------------------------------------------- CPU baseline
PORTABLE_CFLAGS = "-O3 -march=x86-64-v2 -ffp-contract=off -std=gnu11"   # oracle/Makefile: the build that travels to the GPU box
NATIVE_CFLAGS = "-O3 -march=native -ffp-contract=off -std=gnu11"          # the timed CPU baseline
_native = None


def native_lib():
    """The oracle compiled ON THIS HOST with -O3 -march=native into a temporary directory -- only for
    timing the CPU baseline: a -march=native object built in the build container must not travel to another CPU, which is
    why the committed Makefile builds the checker with -march=x86-64-v2.  -> (CDLL, cc, cflags) or None if gcc fails."""
    global _native
    if _native is None:
        import tempfile
        out = os.path.join(tempfile.mkdtemp(prefix="tg_oracle_native_"), "libtg_oracle_native.so")
        cc = os.environ.get("CC", "gcc")
        cmd = [cc] + NATIVE_CFLAGS.split() + ["-fPIC", "-shared", "-o", out, os.path.join(_HERE, "tg_oracle.c"),
                                               "-lpthread", "-lm"]
        try:
            subprocess.check_call(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            h = C.CDLL(out)
            h.orc_bench_ns_homo.restype = C.c_double
            ver = subprocess.run([cc, "--version"], capture_output=True, text=True).stdout.splitlines()[0]
            _native = (h, ver, NATIVE_CFLAGS)
        except Exception:  # noqa: BLE001
            _native = False
    return _native or None


def bench_ns_homo(ptrs, indices, seeds, fanout, n_threads, handle=None):
    """seeds: [n_batches, B] int64.  Returns (seconds, sampled_edges).  handle: a CDLL of another build of the oracle
    (native_lib), default the checker's own."""
    ptrs, indices, seeds, fan = _i64(ptrs), _i64(indices), _i64(seeds), _i64(fanout)
    edges = C.c_int64(0)
    sec = (handle or lib()).orc_bench_ns_homo(_p(ptrs), _p(indices), _p(seeds), C.c_int64(seeds.shape[1]),
                                  C.c_int64(seeds.shape[0]), _p(fan), C.c_int32(fan.size), C.c_int32(n_threads),
                                  C.byref(edges))
    return float(sec), int(edges.value)


# ---------------------------------------------------------------- synthetic inputs
def rmat_edges(scale, n_edges, seed):
    row = np.empty(n_edges, dtype=np.int64)
    col = np.empty(n_edges, dtype=np.int64)
    lib().orc_rmat_edges(C.c_int32(scale), C.c_int64(n_edges), C.c_uint64(seed), _p(row), _p(col))
    return row, col


def seed_batches(seed, first_batch, n_batches, n_seeds, n_nodes):
    out = np.empty((n_batches, n_seeds), dtype=np.int64)
    lib().orc_seed_batches(C.c_uint64(seed), C.c_int64(first_batch), C.c_int64(n_batches), C.c_int64(n_seeds),
                           C.c_int64(n_nodes), _p(out))
    return out


# ---------------------------------------------------------------- HGT sampling
def hgt(node_types, edge_types, col_ptrs, row_indices, row_timestamps, inputs, input_timestamps, num_samples,
        num_hops, rng, timerange=None, edge_reservoir_algo=RES_AUTO):
    """hgt_sampling -> (samples, sample_ts, rows, cols, edge_index) dicts (canonical node/edge type order)."""
    L = lib()
    L.orc_hgt.restype = C.c_void_p
    L.orc_het_copy_sample_ts.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    T, R, H = len(node_types), len(edge_types), num_hops
    tix = {t: i for i, t in enumerate(node_types)}
    rels = [_rel_key(e) for e in edge_types]
    rel_src = (C.c_int32 * R)(*[tix[e[0]] for e in edge_types])
    rel_dst = (C.c_int32 * R)(*[tix[e[2]] for e in edge_types])
    P = [_i64(col_ptrs[r]) for r in rels]
    I = [_i64(row_indices[r]) for r in rels]
    TS = None
    if row_timestamps is not None:
        TS = [_i64(row_timestamps[r]) if r in row_timestamps else None for r in rels]
    IN = [_i64(inputs[t]) if t in inputs else None for t in node_types]
    n_in = _i64([a.size if a is not None else -1 for a in IN])
    ITS = None
    if input_timestamps is not None:
        ITS = [_i64(input_timestamps[t]) if t in inputs else None for t in node_types]
    ns = _i64([[num_samples[t][h] if t in num_samples else -1 for h in range(H)] for t in node_types]).reshape(-1)
    status = C.c_int32(0)
    tr = timerange if timerange is not None else (0, 0)
    h = L.orc_hgt(C.c_int32(T), C.c_int32(R), rel_src, rel_dst, _ptr_array(P), _ptr_array(I),
                  _ptr_array(TS) if TS is not None else C.c_void_p(0), _ptr_array(IN), _p(n_in),
                  _ptr_array(ITS) if ITS is not None else C.c_void_p(0), _p(ns), C.c_int32(H),
                  C.c_int32(int(timerange is not None)), C.c_int64(tr[0]), C.c_int64(tr[1]),
                  C.c_int32(edge_reservoir_algo), C.byref(rng), C.byref(status))
    try:
        if status.value != 0:
            raise RuntimeError("oracle hgt: the reference would panic")
        samples, rows, cols, eidx, _ = _unpack_het(h, node_types, rels, 0)
        ts = {}
        for t, name in enumerate(node_types):
            a = np.empty(len(samples[name]), dtype=np.int64)
            L.orc_het_copy_sample_ts(C.c_void_p(h), t, _p(a))
            ts[name] = a
        return samples, ts, rows, cols, eidx
    finally:
        L.orc_het_free(C.c_void_p(h))


# ---------------------------------------------------------------- budget sampling
class BudgetFilter(C.Structure):
    _fields_ = [("enabled", C.c_int32), ("forward", C.c_int32), ("relative", C.c_int32), ("_pad", C.c_int32),
                ("lo", C.c_int64), ("hi", C.c_int64)]


def budget(node_types, edge_types, col_ptrs, row_indices, row_timestamps, inputs, input_timestamps, num_neighbors,
           num_hops, rng, window=None, forward=False, relative=False):
    """budget_sampling -> (samples, sample_ts, rows, cols, edge_index) dicts (canonical relation order)."""
    L = lib()
    L.orc_budget_sampling.restype = C.c_void_p
    L.orc_het_copy_sample_ts.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    T, R, H = len(node_types), len(edge_types), num_hops
    tix = {t: i for i, t in enumerate(node_types)}
    rels = [_rel_key(e) for e in edge_types]
    rel_src = (C.c_int32 * R)(*[tix[e[0]] for e in edge_types])
    rel_dst = (C.c_int32 * R)(*[tix[e[2]] for e in edge_types])
    P = [_i64(col_ptrs[r]) for r in rels]
    I = [_i64(row_indices[r]) for r in rels]
    TS = None
    if row_timestamps is not None:
        TS = [_i64(row_timestamps[r]) if r in row_timestamps else None for r in rels]
    IN = [_i64(inputs[t]) if t in inputs else None for t in node_types]
    n_in = _i64([a.size if a is not None else -1 for a in IN])
    ITS = None
    if input_timestamps is not None:
        ITS = [_i64(input_timestamps[t]) if t in input_timestamps else None for t in node_types]
    nn = _i64([[num_neighbors[t][h] if t in num_neighbors else -1 for h in range(H)] for t in node_types]).reshape(-1)
    flt = BudgetFilter()
    if window is not None:
        flt.enabled, flt.forward, flt.relative = 1, int(bool(forward)), int(bool(relative))
        flt.lo, flt.hi = window
    status = C.c_int32(0)
    h = L.orc_budget_sampling(C.c_int32(T), C.c_int32(R), rel_src, rel_dst, _ptr_array(P), _ptr_array(I),
                     _ptr_array(TS) if TS is not None else C.c_void_p(0), _ptr_array(IN), _p(n_in),
                     _ptr_array(ITS) if ITS is not None else C.c_void_p(0), _p(nn), C.c_int32(H), C.byref(flt),
                     C.byref(rng), C.byref(status))
    try:
        if status.value != 0:
            raise RuntimeError("oracle budget: the reference would panic")
        samples, rows, cols, eidx, _ = _unpack_het(h, node_types, rels, 0)
        ts = {}
        for t, name in enumerate(node_types):
            a = np.empty(len(samples[name]), dtype=np.int64)
            L.orc_het_copy_sample_ts(C.c_void_p(h), t, _p(a))
            ts[name] = a
        return samples, ts, rows, cols, eidx
    finally:
        L.orc_het_free(C.c_void_p(h))
