"""ctypes binding of the C ABI declared in include/tchgeo.h (libtchgeo_hip.so).

There is no CPU fallback: if the gfx950 library is missing this module raises
at import, and every wrapper raises on a non-zero status code.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TCHGEO_LIB") or os.path.join(os.path.dirname(_HERE), "lib", "libtchgeo_hip.so")  # TCHGEO_LIB: A/B builds

TG_OK = 0
TG_MAX_HOPS = 8
TG_MAX_FANOUT = 32
SAMPLER_UNIFORM, SAMPLER_UNIFORM_REPL, SAMPLER_WEIGHTED = 0, 1, 2
PART_REPLY_PACKED, PART_REPLY_PAIRS, PART_REPLY_TRIPLES, PART_REPLY_PACKED_STATE = 1, 2, 3, 4   # tchgeo.h TG_PART_REPLY_*
FILTER_NONE, FILTER_STATIC, FILTER_RELATIVE, FILTER_DYNAMIC = -1, 0, 1, 2

EXPORTS = ["tg_version", "tg_last_error", "tg_ns_homo_capacity", "tg_ns_homo_batched", "tg_random_walk",
           "tg_edge_set_bytes", "tg_edge_set_build", "tg_random_walk_es",
           "tg_tempo_random_walk", "tg_rmat_edges", "tg_seed_batches", "tg_ind2ptr", "tg_probe_random_gather",
           "tg_neg_workspace_bytes", "tg_neg_sample", "tg_hgt_workspace_bytes", "tg_hgt_sample", "tg_ns_hop_workspace_bytes", "tg_ns_hop", "tg_rmat_edges_rect",
           "tg_coo_to_csx_workspace_bytes", "tg_coo_to_csx", "tg_budget_layer", "tg_check_range",
           "tg_ns_hop_scan_workspace_bytes", "tg_ns_hop_scan", "tg_ns_hop_weighted", "tg_ns_hop_weighted_groups", "tg_ns_hop_weighted_workspace_bytes", "tg_gather_rows",
           "tg_biased_walk_workspace_bytes", "tg_biased_tempo_random_walk", "tg_ns_hetero_capacity",
           "tg_ns_hetero_batched", "tg_ns_homo_compact", "tg_part_workspace_bytes", "tg_part_begin",
           "tg_sanitize_range", "tg_ns_hop_segments", "tg_het_hop_begin_all", "tg_het_hop_end_all", "tg_part_requests", "tg_part_count", "tg_part_scan_workspace_bytes", "tg_part_sample", "tg_part_emit", "tg_part_slot_words", "tg_part_sample_slots", "tg_part_emit_slots", "tg_part_unpack", "tg_part_pack", "tg_compact_rows", "tg_budget_capacity",
           "tg_budget_workspace_bytes", "tg_budget_sample", "tg_ns_homo_workspace_bytes", "tg_ns_homo_batched_ws", "tg_het_meta_words", "tg_het_step_begin",
           "tg_het_step_end", "tg_het_hop_end", "tg_ns_homo_batched_form", "tg_ns_win_tuning_get", "tg_ns_win_tuning_set",
           "tg_ns_win_stage_timing", "tg_ns_win_stage_times", "tg_probe_ns_sol",
           "tg_debug_bounds_set_flag", "tg_part_sample_workspace_bytes", "tg_part_sample_ws",
           "tg_part_sample_order_thresholds", "tg_ns_homo_batched_pipeline", "tg_graph_max_degree",
           "tg_ns_homo_workspace_bytes_for", "tg_ns_homo_batched_workspace_bytes"]


class TgGraph(C.Structure):
    _fields_ = [("ptrs", C.c_void_p), ("indices", C.c_void_p), ("weights", C.c_void_p), ("timestamps", C.c_void_p),
                ("n_major", C.c_int64), ("n_edges", C.c_int64), ("indices32", C.c_void_p), ("ptrs32", C.c_void_p),
                ("max_degree", C.c_int64)]


class TgRng(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("call_id", C.c_uint64)]


class TgNsConfig(C.Structure):
    _fields_ = [("sampler", C.c_int32), ("filter_mode", C.c_int32), ("forward", C.c_int32), ("rng_tag", C.c_uint32),
                ("win_lo", C.c_int64), ("win_hi", C.c_int64), ("seeds_state", C.c_void_p), ("id_base", C.c_int64),
                ("seed_ids", C.c_void_p), ("seed_call_ids", C.c_void_p)]


class TgNsOut(C.Structure):
    _fields_ = [("samples", C.c_void_p), ("rows", C.c_void_p), ("cols", C.c_void_p), ("edge_index", C.c_void_p),
                ("layer_offsets", C.c_void_p), ("counts", C.c_void_p), ("states", C.c_void_p),
                ("cap_nodes", C.c_int64), ("cap_edges", C.c_int64)]


if not os.path.exists(LIB_PATH):
    raise ImportError("tch_geometric: %s is missing -- build it with `make -C tch-geometric_amd` "
                      "(hipcc --offload-arch=gfx950); there is no CPU fallback" % LIB_PATH)
lib = C.CDLL(LIB_PATH)
lib.tg_version.restype = C.c_char_p
lib.tg_last_error.restype = C.c_char_p


class TgHopIn(C.Structure):
    _fields_ = [("vertices", C.c_void_p), ("ids", C.c_void_p), ("call_ids", C.c_void_p), ("m", C.c_int64),
                ("id_base", C.c_int64), ("fanout", C.c_int32), ("sampler", C.c_int32), ("rng_tag", C.c_uint32),
                ("_reserved", C.c_uint32)]


class TgHopOut(C.Structure):
    _fields_ = [("cnt", C.c_void_p), ("offsets", C.c_void_p), ("neighbors", C.c_void_p), ("edge_ptrs", C.c_void_p),
                ("parents", C.c_void_p)]


class TgHopSegment(C.Structure):
    _fields_ = [("graph", C.c_void_p), ("begin", C.c_int64), ("fanout", C.c_int32), ("rng_tag", C.c_uint32)]


class TgHopFilter(C.Structure):
    _fields_ = [("filter_mode", C.c_int32), ("forward", C.c_int32), ("win_lo", C.c_int64), ("win_hi", C.c_int64),
                ("states", C.c_void_p)]


class TchGeoError(RuntimeError):
    pass


def check(rc):
    if rc != TG_OK:
        raise TchGeoError("tchgeo error %d: %s" % (rc, lib.tg_last_error().decode()))


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def stream_ptr(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def graph_max_degree(g, device):
    """The longest column (row) of a graph view, computed on the device (tg_graph_max_degree); one read-back."""
    out = torch.zeros(1, dtype=torch.int64, device=device)
    check(lib.tg_graph_max_degree(C.byref(g), ptr(out), stream_ptr(device)))
    return int(out.item())


def graph_view(ptrs, indices, weights=None, timestamps=None, indices32=None, ptrs32=None, max_degree=None):
    """max_degree: None = unknown (0 in the struct: the window-ordered launch then assumes n_edges), "auto" = computed on the
    device here (one read-back, once per view), or the caller's number."""
    g = TgGraph()
    g.ptrs, g.indices = ptrs.data_ptr(), indices.data_ptr()
    g.weights = weights.data_ptr() if weights is not None else None
    g.timestamps = timestamps.data_ptr() if timestamps is not None else None
    g.n_major, g.n_edges = ptrs.numel() - 1, indices.numel()
    g.indices32 = indices32.data_ptr() if indices32 is not None else None
    g.ptrs32 = ptrs32.data_ptr() if ptrs32 is not None else None
    g._keep = (ptrs, indices, weights, timestamps, indices32, ptrs32)  # the struct only borrows the device memory
    g.max_degree = 0
    if max_degree == "auto":
        g.max_degree = graph_max_degree(g, ptrs.device) if ptrs.is_cuda else int((ptrs[1:] - ptrs[:-1]).max()) if ptrs.numel() > 1 else 0
    elif max_degree is not None:
        g.max_degree = int(max_degree)
    return g


def graph_sizing(n_major, n_edges, max_degree=0):
    """A tg_graph that carries only sizes (no arrays): enough for ns_homo_workspace(graph=...) before the graph exists."""
    g = TgGraph()
    g.n_major, g.n_edges, g.max_degree = int(n_major), int(n_edges), int(max_degree)
    return g


def ns_homo_capacity(n_seeds, fanout):
    cn, ce = C.c_int64(0), C.c_int64(0)
    fan = (C.c_int64 * max(len(fanout), 1))(*fanout)
    check(lib.tg_ns_homo_capacity(C.c_int64(n_seeds), fan, C.c_int32(len(fanout)), C.byref(cn), C.byref(ce)))
    return cn.value, ce.value


class NsBatchedOut:
    """Per-batch output slabs (device) of tg_ns_homo_batched."""

    def __init__(self, n_batches, n_seeds, fanout, device, with_states=False):
        self.n_batches, self.n_seeds, self.n_hops = n_batches, n_seeds, len(fanout)
        self.cap_nodes, self.cap_edges = ns_homo_capacity(n_seeds, fanout)
        o = dict(dtype=torch.int64, device=device)
        self.samples = torch.empty((n_batches, max(self.cap_nodes, 1)), **o)
        self.rows = torch.empty((n_batches, max(self.cap_edges, 1)), **o)
        self.cols = torch.empty((n_batches, max(self.cap_edges, 1)), **o)
        self.edge_index = torch.empty((n_batches, max(self.cap_edges, 1)), **o)
        self.layer_offsets = torch.zeros((n_batches, max(self.n_hops, 1), 3), **o)
        self.counts = torch.zeros((n_batches, 2), **o)
        self.states = torch.empty((n_batches, max(self.cap_nodes, 1)), **o) if with_states else None

    def struct(self):
        s = TgNsOut()
        s.samples, s.rows, s.cols = self.samples.data_ptr(), self.rows.data_ptr(), self.cols.data_ptr()
        s.edge_index, s.layer_offsets = self.edge_index.data_ptr(), self.layer_offsets.data_ptr()
        s.counts = self.counts.data_ptr()
        s.states = self.states.data_ptr() if self.states is not None else None
        s.cap_nodes, s.cap_edges = self.samples.shape[1], self.rows.shape[1]
        return s

    def batch(self, b, counts=None):
        """-> (samples, rows, cols, edge_index, layer_offsets) of batch b, trimmed (host sync)."""
        c = (counts if counts is not None else self.counts.cpu())[b]
        ns, ne = int(c[0]), int(c[1])
        lo = [tuple(int(x) for x in row) for row in self.layer_offsets[b, :self.n_hops].cpu()]
        return self.samples[b, :ns], self.rows[b, :ne], self.cols[b, :ne], self.edge_index[b, :ne], lo


def ns_homo_workspace(n_batches, n_seeds, fanout, device, staged=None, graph=None):
    """Workspace of tg_ns_homo_batched_ws (the window-ordered gather of many-batch launches), as an int64 tensor.
    staged: True = sized for the staged pipeline too (its stage slots), False = push pipeline only, None = as the current
    tuning says.  graph: size the stage slots for this graph's bit widths (else the larger, graph-free size)."""
    nbytes = C.c_int64(0)
    fan = (C.c_int64 * max(len(fanout), 1))(*fanout)
    prev = ns_win_tuning_set(staged=int(staged)) if staged is not None else None
    try:
        check(lib.tg_ns_homo_workspace_bytes_for(C.byref(graph) if graph is not None else None, C.c_int64(n_batches),
                                                 C.c_int64(n_seeds), fan, C.c_int32(len(fanout)), C.byref(nbytes)))
    finally:
        if prev is not None:
            ns_win_tuning_set(staged=prev["staged"])
    return torch.empty(nbytes.value // 8 + 1, dtype=torch.int64, device=device)


def ns_homo_batched_workspace(graph, n_batches, n_seeds, fanout, device, sampler=SAMPLER_UNIFORM, filter_mode=FILTER_NONE):
    """The workspace tg_ns_homo_batched_ws wants for THIS configuration (tg_ns_homo_batched_workspace_bytes): the
    window-ordered form's for the plain samplers, the flat path's for few filtered / weighted batches; None when the launch
    takes none."""
    cfg = TgNsConfig()
    cfg.sampler, cfg.filter_mode = sampler, filter_mode
    nbytes = C.c_int64(0)
    fan = (C.c_int64 * max(len(fanout), 1))(*fanout)
    check(lib.tg_ns_homo_batched_workspace_bytes(C.byref(graph), C.c_int64(n_batches), C.c_int64(n_seeds), fan,
                                                 C.c_int32(len(fanout)), C.byref(cfg), C.byref(nbytes)))
    return torch.empty(nbytes.value // 8 + 1, dtype=torch.int64, device=device) if nbytes.value > 0 else None


def ns_homo_batched(graph, seeds, fanout, seed, call_id, out, sampler=SAMPLER_UNIFORM, filter_mode=FILTER_NONE,
                    forward=False, window=(0, 0), seeds_state=None, rng_tag=0, id_base=0, seed_ids=None,
                    seed_call_ids=None, ws=None, form=0):
    """seeds: [n_batches, n_seeds] int64 on the graph's device; `out` an NsBatchedOut; `ws` (ns_homo_workspace) lets a
    many-batch launch take the window-ordered form (same outputs); form: 0 auto, 1 windowed when applicable, 2 fused."""
    assert seeds.dtype == torch.int64 and seeds.is_contiguous() and seeds.dim() == 2
    cfg = TgNsConfig()
    cfg.sampler, cfg.filter_mode, cfg.forward = sampler, filter_mode, int(bool(forward))
    cfg.win_lo, cfg.win_hi = window
    cfg.seeds_state = seeds_state.data_ptr() if seeds_state is not None else None
    cfg.rng_tag, cfg.id_base = rng_tag, id_base
    cfg.seed_ids = seed_ids.data_ptr() if seed_ids is not None else None
    cfg.seed_call_ids = seed_call_ids.data_ptr() if seed_call_ids is not None else None
    rng = TgRng(seed, call_id)
    fan = (C.c_int64 * max(len(fanout), 1))(*fanout)
    so = out.struct()
    if ws is not None:
        check(lib.tg_ns_homo_batched_ws(C.byref(graph), ptr(seeds), C.c_int64(seeds.shape[0]),
                                        C.c_int64(seeds.shape[1]), fan, C.c_int32(len(fanout)), C.byref(cfg),
                                        C.byref(rng), C.byref(so), ptr(ws), C.c_int64(ws.numel() * 8), C.c_int32(form),
                                        stream_ptr(seeds.device)))
        return out
    check(lib.tg_ns_homo_batched(C.byref(graph), ptr(seeds), C.c_int64(seeds.shape[0]), C.c_int64(seeds.shape[1]),
                                 fan, C.c_int32(len(fanout)), C.byref(cfg), C.byref(rng), C.byref(so),
                                 stream_ptr(seeds.device)))
    return out


def ns_homo_batched_form(graph, out, n_batches, n_seeds, fanout, ws=None, form=0, sampler=SAMPLER_UNIFORM,
                         filter_mode=FILTER_NONE):
    """Which form ns_homo_batched(..., ws=ws, form=form) runs -> (form taken: 1 windowed / 2 fused / 3 windowed with wide
    items, number of windows).  A query: nothing is launched."""
    cfg = TgNsConfig()
    cfg.sampler, cfg.filter_mode = sampler, filter_mode
    fan = (C.c_int64 * max(len(fanout), 1))(*fanout)
    so = out.struct()
    taken, n_win = C.c_int32(0), C.c_int32(0)
    check(lib.tg_ns_homo_batched_form(C.byref(graph), C.c_int64(n_batches), C.c_int64(n_seeds), fan,
                                      C.c_int32(len(fanout)), C.byref(cfg), C.byref(so),
                                      C.c_int64(ws.numel() * 8 if ws is not None else 0), C.c_int32(form),
                                      C.byref(taken), C.byref(n_win)))
    return taken.value, n_win.value


def ns_homo_batched_staged(graph, out, n_batches, n_seeds, fanout, ws=None, form=0, sampler=SAMPLER_UNIFORM):
    """True if ns_homo_batched(..., ws=ws, form=form) would take the STAGED pipeline of the window-ordered form under the
    current tuning (it needs the workspace sized while `staged` was on: ns_homo_workspace(..., staged=True))."""
    cfg = TgNsConfig()
    cfg.sampler, cfg.filter_mode = sampler, FILTER_NONE
    fan = (C.c_int64 * max(len(fanout), 1))(*fanout)
    so = out.struct()
    st = C.c_int32(0)
    check(lib.tg_ns_homo_batched_pipeline(C.byref(graph), C.c_int64(n_batches), C.c_int64(n_seeds), fan,
                                          C.c_int32(len(fanout)), C.byref(cfg), C.byref(so),
                                          C.c_int64(ws.numel() * 8 if ws is not None else 0), C.c_int32(form), C.byref(st)))
    return bool(st.value)


class TgNsWinTuning(C.Structure):
    _fields_ = [("window_bytes", C.c_int64), ("gather_blocks", C.c_int32), ("gather_threads", C.c_int32),
                ("emit_threads", C.c_int32), ("direct_hop0", C.c_int32), ("fuse_first_hops", C.c_int32),
                ("fold_hist", C.c_int32), ("emit_blocks", C.c_int32), ("staged", C.c_int32),
                ("stage_round_chunks", C.c_int32), ("stage_gather_threads", C.c_int32), ("stage_gather_blocks", C.c_int32),
                ("stage_emit_threads", C.c_int32), ("stage_parts", C.c_int32),
                ("stage_part_min_batches", C.c_int32), ("stage_sort_blocks", C.c_int32), ("stage_fine", C.c_int32),
                ("stage_concurrent", C.c_int32), ("stage_split", C.c_int32),
                ("stage_split_round_chunks", C.c_int32), ("store_align64", C.c_int32),
                ("stage_fine_sub_bits", C.c_int32),
                ("stage_fine_blocks", C.c_int32)]


def ns_win_tuning():
    t = TgNsWinTuning()
    check(lib.tg_ns_win_tuning_get(C.byref(t)))
    return {k: getattr(t, k) for k, _ in TgNsWinTuning._fields_}


def ns_win_tuning_set(**kw):
    """Process-wide tuning of the window-ordered launch (outputs never depend on it); -> the previous values."""
    before = ns_win_tuning()
    t = TgNsWinTuning(0, 0, 0, 0, -1, -1, -1, 0, -1, 0, 0, 0, 0, 0, 0, 0, -1, -1, -1, 0, -1, 0, 0)
    for k, v in kw.items():
        assert k in before, k
        setattr(t, k, int(v))
    check(lib.tg_ns_win_tuning_set(C.byref(t)))
    return before


def ns_win_stage_timing(enable):
    check(lib.tg_ns_win_stage_timing(C.c_int32(int(bool(enable)))))


def ns_win_stage_times():
    """Stage durations (ms) of the last window-ordered launch recorded under ns_win_stage_timing(True); waits for it."""
    cap = 80
    ms, names, n = (C.c_float * cap)(), C.create_string_buffer(24 * cap), C.c_int32(0)
    check(lib.tg_ns_win_stage_times(ms, names, C.c_int32(cap), C.byref(n)))
    return [(names.raw[24 * i:24 * i + 24].split(b"\0", 1)[0].decode(), float(ms[i])) for i in range(min(n.value, cap))]


def edge_set(graph, device):
    """hash set of the CSR's edges for tg_random_walk_es (has_edge as one probe): a uint64 tensor"""
    nbytes = C.c_int64(0)
    check(lib.tg_edge_set_bytes(C.byref(graph), C.byref(nbytes)))
    es = torch.empty(nbytes.value // 8, dtype=torch.int64, device=device)
    check(lib.tg_edge_set_build(C.byref(graph), ptr(es), nbytes, stream_ptr(device)))
    return es


def random_walk(graph, start, walk_length, p, q, seed, call_id, edge_set=None):
    walks = torch.empty((start.numel(), walk_length + 1), dtype=torch.int64, device=start.device)
    rng = TgRng(seed, call_id)
    if edge_set is None:
        check(lib.tg_random_walk(C.byref(graph), ptr(start), C.c_int64(start.numel()), C.c_int64(walk_length),
                                 C.c_float(p), C.c_float(q), C.byref(rng), ptr(walks), stream_ptr(start.device)))
    else:
        check(lib.tg_random_walk_es(C.byref(graph), ptr(edge_set), C.c_int64(edge_set.numel() * 8), ptr(start),
                                    C.c_int64(start.numel()), C.c_int64(walk_length), C.c_float(p), C.c_float(q),
                                    C.byref(rng), ptr(walks), stream_ptr(start.device)))
    return walks


def tempo_random_walk(graph, node_ts, edge_ts, start, start_ts, walk_length, window, seed, call_id):
    walks = torch.empty((start.numel(), walk_length), dtype=torch.int64, device=start.device)
    wts = torch.empty((start.numel(), walk_length), dtype=torch.int64, device=start.device)
    rng = TgRng(seed, call_id)
    check(lib.tg_tempo_random_walk(C.byref(graph), ptr(node_ts), ptr(edge_ts), ptr(start), ptr(start_ts),
                                   C.c_int64(start.numel()), C.c_int64(walk_length), C.c_int64(window[0]),
                                   C.c_int64(window[1]), C.byref(rng), ptr(walks), ptr(wts),
                                   stream_ptr(start.device)))
    return walks, wts


class TgHetProblem(C.Structure):
    _fields_ = [("n_types", C.c_int32), ("n_rels", C.c_int32), ("n_hops", C.c_int32), ("sampler", C.c_int32),
                ("rel_src", C.POINTER(C.c_int32)), ("rel_dst", C.POINTER(C.c_int32)), ("graphs", C.POINTER(TgGraph)),
                ("fanout", C.POINTER(C.c_int64)), ("inputs", C.POINTER(C.c_void_p)), ("n_inputs", C.POINTER(C.c_int64))]


class TgHetOut(C.Structure):
    _fields_ = [("samples", C.POINTER(C.c_void_p)), ("cap_nodes", C.POINTER(C.c_int64)),
                ("rows", C.POINTER(C.c_void_p)), ("cols", C.POINTER(C.c_void_p)), ("edge_index", C.POINTER(C.c_void_p)),
                ("cap_edges", C.POINTER(C.c_int64)), ("layer_offsets", C.c_void_p), ("counts", C.c_void_p)]


class NsHeteroBatched:
    """Problem description + output slabs of tg_ns_hetero_batched.  rels: list of (src type index, dst type index,
    ptrs, indices, [fanout per hop]); inputs: list over node types of [n_batches, n_inputs] tensors or None."""

    def __init__(self, n_types, rels, inputs, n_hops, n_batches, device, sampler=SAMPLER_UNIFORM):
        T, R = n_types, len(rels)
        self.T, self.R, self.H, self.nb, self.dev = T, R, n_hops, n_batches, device
        self._keep = [rels, inputs]
        self.rel_src = (C.c_int32 * max(R, 1))(*[r[0] for r in rels])
        self.rel_dst = (C.c_int32 * max(R, 1))(*[r[1] for r in rels])
        self.graphs = (TgGraph * max(R, 1))()
        for i, r in enumerate(rels):
            self.graphs[i] = graph_view(r[2], r[3])
        flat = [int(k) for r in rels for k in r[4]]
        self.fanout = (C.c_int64 * max(len(flat), 1))(*flat)
        self.n_inputs = (C.c_int64 * T)(*[0 if x is None else int(x.shape[1]) for x in inputs])
        self.inputs = (C.c_void_p * T)(*[None if x is None else x.data_ptr() for x in inputs])
        self.problem = TgHetProblem(T, R, n_hops, sampler, self.rel_src, self.rel_dst, self.graphs, self.fanout,
                                    self.inputs, self.n_inputs)
        self.cap_nodes, self.cap_edges = (C.c_int64 * T)(), (C.c_int64 * max(R, 1))()
        check(lib.tg_ns_hetero_capacity(C.byref(self.problem), self.cap_nodes, self.cap_edges))
        o = dict(dtype=torch.int64, device=device)
        self.samples = [torch.empty((n_batches, max(self.cap_nodes[t], 1)), **o) for t in range(T)]
        self.rows = [torch.empty((n_batches, max(self.cap_edges[r], 1)), **o) for r in range(R)]
        self.cols = [torch.empty((n_batches, max(self.cap_edges[r], 1)), **o) for r in range(R)]
        self.edge_index = [torch.empty((n_batches, max(self.cap_edges[r], 1)), **o) for r in range(R)]
        self.layer_offsets = torch.zeros((n_batches, max(R, 1), max(n_hops, 1), 3), **o)
        self.counts = torch.zeros((n_batches, T + R), **o)
        # the slabs are allocated with at least one column; tell the library their true pitch
        self.cap_nodes_alloc = (C.c_int64 * T)(*[max(self.cap_nodes[t], 1) for t in range(T)])
        self.cap_edges_alloc = (C.c_int64 * max(R, 1))(*[max(self.cap_edges[r], 1) for r in range(R)])
        vp = lambda ts: (C.c_void_p * max(len(ts), 1))(*[t.data_ptr() for t in ts])
        self._ptrs = [vp(self.samples), vp(self.rows), vp(self.cols), vp(self.edge_index)]
        self.out = TgHetOut(self._ptrs[0], self.cap_nodes_alloc, self._ptrs[1], self._ptrs[2], self._ptrs[3],
                            self.cap_edges_alloc, self.layer_offsets.data_ptr(), self.counts.data_ptr())

    def run(self, seed, call_id):
        rng = TgRng(seed, call_id)
        check(lib.tg_ns_hetero_batched(C.byref(self.problem), C.c_int64(self.nb), C.byref(rng), C.byref(self.out),
                                       stream_ptr(self.dev)))


BIAS = {"uniform": 0, "linear": 1, "exponential": 2}


def biased_tempo_random_walk(graph, node_ts, edge_ts, start, start_ts, walk_length, bias, forward, retry_count, seed,
                             call_id, max_degree=None):
    """-> (walks, walks_ts, status word)."""
    dev, n = start.device, start.numel()
    walks = torch.empty((n, walk_length), dtype=torch.int64, device=dev)
    wts = torch.empty((n, walk_length), dtype=torch.int64, device=dev)
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    nbytes = C.c_int64(0)
    if max_degree is None:
        max_degree = 0
    check(lib.tg_biased_walk_workspace_bytes(C.c_int64(n), C.c_int64(max_degree), C.c_int32(BIAS[bias]), C.byref(nbytes)))
    ws = torch.empty(nbytes.value // 8 + 1, dtype=torch.int64, device=dev)
    rng = TgRng(seed, call_id)
    check(lib.tg_biased_tempo_random_walk(C.byref(graph), ptr(node_ts), ptr(edge_ts), ptr(start), ptr(start_ts),
                                          C.c_int64(n), C.c_int64(walk_length), C.c_int32(BIAS[bias]),
                                          C.c_int32(int(forward)), C.c_int64(retry_count), C.c_int64(max_degree),
                                          C.byref(rng), ptr(walks), ptr(wts), ptr(status), ptr(ws),
                                          C.c_int64(nbytes.value), stream_ptr(dev)))
    return walks, wts, status


def rmat_edges(scale, n_edges, seed, device):
    row = torch.empty(n_edges, dtype=torch.int64, device=device)
    col = torch.empty(n_edges, dtype=torch.int64, device=device)
    check(lib.tg_rmat_edges(C.c_int32(scale), C.c_int64(n_edges), C.c_uint64(seed), ptr(row), ptr(col),
                            stream_ptr(device)))
    return row, col


def rmat_edges_rect(row_scale, col_scale, n_edges, seed, device):
    row = torch.empty(n_edges, dtype=torch.int64, device=device)
    col = torch.empty(n_edges, dtype=torch.int64, device=device)
    check(lib.tg_rmat_edges_rect(C.c_int32(row_scale), C.c_int32(col_scale), C.c_int64(n_edges), C.c_uint64(seed),
                                 ptr(row), ptr(col), stream_ptr(device)))
    return row, col


def seed_batches(seed, first_batch, n_batches, n_seeds, n_nodes, device):
    out = torch.empty((n_batches, n_seeds), dtype=torch.int64, device=device)
    check(lib.tg_seed_batches(C.c_uint64(seed), C.c_int64(first_batch), C.c_int64(n_batches), C.c_int64(n_seeds),
                              C.c_int64(n_nodes), ptr(out), stream_ptr(device)))
    return out


def ind2ptr(ind, m):
    out = torch.empty(m + 1, dtype=torch.int64, device=ind.device)
    check(lib.tg_ind2ptr(ptr(ind), C.c_int64(ind.numel()), C.c_int64(m), ptr(out), stream_ptr(ind.device)))
    return out


def coo_to_csx(row, col, size0, size1, csc):
    """Device ingest (tg_coo_to_csx): stable radix sort by the reference's key (storage.rs:112,119), then
    ptrs / indices.  Returns (ptrs, indices, perm)."""
    nnz, dev = row.numel(), row.device
    o = dict(dtype=torch.int64, device=dev)
    m = size1 if csc else size0
    ptrs, indices, perm = torch.empty(m + 1, **o), torch.empty(nnz, **o), torch.empty(nnz, **o)
    nbytes = C.c_int64(0)
    check(lib.tg_coo_to_csx_workspace_bytes(C.c_int64(nnz), C.c_int64(size0), C.c_int64(size1), C.byref(nbytes)))
    ws = torch.empty(nbytes.value // 8 + 1, **o)
    row, col = row.contiguous(), col.contiguous()
    check(lib.tg_coo_to_csx(ptr(row) if nnz else None, ptr(col) if nnz else None, C.c_int64(nnz), C.c_int64(size0),
                            C.c_int64(size1), C.c_int32(int(csc)), ptr(ptrs), ptr(indices), ptr(perm), ptr(ws),
                            C.c_int64(nbytes.value), stream_ptr(dev)))
    return ptrs, indices, perm


def ns_homo_compact(out, n_batches, counts_host, stacked=False):
    """Flat batch-major (samples, rows, cols, edge_index) of the first n_batches of an NsBatchedOut; counts_host: the
    [n_batches, 2] counts already read back (sizes the flat arrays).  stacked: rows and cols are the two rows of ONE
    [2, E] tensor (what a loader hands out as edge_index) -> (samples, [rows; cols], edge_index)."""
    dev = out.samples.device
    total_n, total_e = int(counts_host[:, 0].sum()), int(counts_host[:, 1].sum())
    c = out.counts[:n_batches]
    off = torch.zeros((2, n_batches + 1), dtype=torch.int64, device=dev)
    off[:, 1:] = torch.cumsum(c.t(), dim=1)
    o = dict(dtype=torch.int64, device=dev)
    fs, fe = torch.empty(total_n, **o), torch.empty(total_e, **o)
    rc = torch.empty((2, total_e), **o)
    fr, fc = rc[0], rc[1]
    so = out.struct()
    check(lib.tg_ns_homo_compact(C.byref(so), C.c_int64(n_batches), ptr(off[0]), ptr(off[1]), ptr(fs), ptr(fr), ptr(fc),
                                 ptr(fe), stream_ptr(dev)))
    return (fs, rc, fe) if stacked else (fs, fr, fc, fe)


def compact_rows(slab, lens, total):
    """Flat concatenation of slab[r, :lens[r]] (slab: [n_rows, pitch] int64; lens: device view, any stride; total: their
    sum, known to the host)."""
    n_rows = slab.shape[0]
    off = torch.cumsum(lens, 0) - lens
    dst = torch.empty(int(total), dtype=torch.int64, device=slab.device)
    check(lib.tg_compact_rows(ptr(slab), C.c_int64(slab.stride(0)), ptr(lens), C.c_int64(lens.stride(0) if lens.dim() else 1),
                              ptr(off), C.c_int64(n_rows), ptr(dst), stream_ptr(slab.device)))
    return dst


def gather_rows(src, index, status=None):
    """dst[i] = src[index[i]] over dim 0 (tg_gather_rows).  `src` may be strided over dim 0 only.  Returns
    (dst, status): status is a device int32 word, 1 when an index fell outside [0, src.shape[0])."""
    dev = src.device
    if src.dim() == 0:
        raise ValueError("gather_rows needs at least one dimension")
    inner = src[0].is_contiguous() if src.shape[0] else True
    if not inner or (src.dim() > 1 and src.shape[0] > 1 and src.stride(0) < src[0].numel()):
        src = src.contiguous()
    row_elems = 1
    for d in src.shape[1:]:
        row_elems *= d
    item = src.element_size()
    stride = src.stride(0) * item if src.shape[0] > 1 else row_elems * item
    index = index.contiguous()
    dst = torch.empty((index.numel(),) + tuple(src.shape[1:]), dtype=src.dtype, device=dev)
    if status is None:
        status = torch.zeros(1, dtype=torch.int32, device=dev)
    check(lib.tg_gather_rows(C.c_void_p(src.data_ptr()), C.c_int64(src.shape[0]), C.c_int64(row_elems * item),
                             C.c_int64(max(stride, row_elems * item)), ptr(index), C.c_int64(index.numel()),
                             C.c_void_p(dst.data_ptr()), C.c_void_p(status.data_ptr()), stream_ptr(dev)))
    return dst, status


def probe_random_gather(table, n_threads, per_thread, seed=1):
    sink = torch.empty(n_threads, dtype=torch.int64, device=table.device)
    check(lib.tg_probe_random_gather(ptr(table), C.c_int64(table.numel()), C.c_int64(n_threads),
                                     C.c_int64(per_thread), C.c_uint64(seed), ptr(sink), stream_ptr(table.device)))
    return sink


def probe_ns_sol(src, dst, seeds, n_hops):
    """The algorithmic bytes of the finished launch `src` moved as pure streams into `dst` (tg_probe_ns_sol)."""
    sink = torch.zeros(seeds.shape[0], dtype=torch.int64, device=seeds.device)
    a, b = src.struct(), dst.struct()
    check(lib.tg_probe_ns_sol(C.byref(a), C.byref(b), ptr(seeds), C.c_int64(seeds.shape[0]), C.c_int64(seeds.shape[1]),
                              C.c_int32(n_hops), ptr(sink), stream_ptr(seeds.device)))
    return sink


def ns_hop(graph, vertices, fanout, seed, call_id=0, sampler=SAMPLER_UNIFORM, ids=None, call_ids=None, id_base=0,
           rng_tag=0):
    """One flat hop (tg_ns_hop).  -> (cnt[m], offsets[m+1], neighbors, edge_ptrs, parents), the last three sized
    m*fanout with offsets[m] valid entries; no host synchronisation."""
    m, dev = vertices.numel(), vertices.device
    o = dict(dtype=torch.int64, device=dev)
    cnt, offsets = torch.empty(max(m, 1), **o), torch.empty(m + 1, **o)
    nbr, ep, par = (torch.empty(max(m * fanout, 1), **o) for _ in range(3))
    hin, hout = TgHopIn(), TgHopOut()
    hin.vertices = vertices.data_ptr() if m else None
    hin.ids = ids.data_ptr() if ids is not None else None
    hin.call_ids = call_ids.data_ptr() if call_ids is not None else None
    hin.m, hin.id_base, hin.fanout, hin.sampler, hin.rng_tag = m, id_base, fanout, sampler, rng_tag
    hout.cnt, hout.offsets = cnt.data_ptr(), offsets.data_ptr()
    hout.neighbors, hout.edge_ptrs, hout.parents = nbr.data_ptr(), ep.data_ptr(), par.data_ptr()
    nbytes = C.c_int64(0)
    check(lib.tg_ns_hop_workspace_bytes(C.c_int64(m), C.byref(nbytes)))
    ws = torch.empty(nbytes.value // 8 + 1, **o)
    rng = TgRng(seed, call_id)
    check(lib.tg_ns_hop(C.byref(graph), C.byref(hin), C.byref(rng), C.byref(hout), ptr(ws), C.c_int64(nbytes.value),
                        stream_ptr(dev)))
    return cnt[:m], offsets, nbr, ep, par


def ns_hop_segments(segments, vertices, states, seed, filter_mode=FILTER_NONE, window=(0, 0), forward=False, call_id=0,
                    sampler=SAMPLER_UNIFORM, ids=None, call_ids=None, id_base=0, layout=None, group_cap=None):
    """One flat hop over a frontier made of segments (tg_ns_hop_segments).  segments: list of (graph view, begin,
    fanout, rng_tag); layout: optional device tensor [len(segments) + 1] = real segment starts, then the real length.
    -> (cnt[m], offsets[m+1], neighbors, edge_ptrs, parents, states_out, status) -- no host synchronisation."""
    m, dev = vertices.numel(), vertices.device
    o = dict(dtype=torch.int64, device=dev)
    kmax = max(f for _, _, f, _ in segments)
    if group_cap is None:
        group_cap = max(1024, sum(g.n_edges for g, _, _, _ in segments) // 512 + 2 * m + 2)
    cnt, offsets = torch.empty(max(m, 1), **o), torch.empty(m + 1, **o)
    nbr, ep, par, st_out = (torch.empty(max(m * kmax, 1), **o) for _ in range(4))
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    segs = (TgHopSegment * len(segments))()
    for j, (g, begin, fanout, tag) in enumerate(segments):
        segs[j].graph, segs[j].begin, segs[j].fanout, segs[j].rng_tag = C.addressof(g), begin, fanout, tag
    hin, hout, flt = TgHopIn(), TgHopOut(), TgHopFilter()
    hin.vertices = vertices.data_ptr() if m else None
    hin.ids = ids.data_ptr() if ids is not None else None
    hin.call_ids = call_ids.data_ptr() if call_ids is not None else None
    hin.m, hin.id_base, hin.fanout, hin.sampler, hin.rng_tag = m, id_base, kmax, sampler, 0
    hout.cnt, hout.offsets = cnt.data_ptr(), offsets.data_ptr()
    hout.neighbors, hout.edge_ptrs, hout.parents = nbr.data_ptr(), ep.data_ptr(), par.data_ptr()
    flt.filter_mode, flt.forward = filter_mode, int(bool(forward))
    flt.win_lo, flt.win_hi = window
    flt.states = states.data_ptr() if (m and states is not None) else None
    nbytes = C.c_int64(0)
    size_of = lib.tg_ns_hop_weighted_workspace_bytes if sampler == SAMPLER_WEIGHTED else lib.tg_ns_hop_scan_workspace_bytes
    check(size_of(C.c_int64(m), C.c_int32(kmax), C.c_int64(group_cap), C.byref(nbytes)))
    ws = torch.empty(nbytes.value // 8 + 1, **o)
    rng = TgRng(seed, call_id)
    check(lib.tg_ns_hop_segments(segs, C.c_int32(len(segments)), C.byref(hin), ptr(layout) if layout is not None else None,
                                 C.byref(flt), C.byref(rng), C.byref(hout), ptr(st_out), ptr(status), ptr(ws),
                                 C.c_int64(nbytes.value), C.c_int64(group_cap), stream_ptr(dev)))
    return cnt[:m], offsets, nbr, ep, par, st_out, status


def ns_hop_scan(graph, vertices, states, fanout, seed, filter_mode, window, forward=False, call_id=0,
                sampler=SAMPLER_UNIFORM, ids=None, call_ids=None, id_base=0, rng_tag=0, group_cap=None):
    """One flat hop under a temporal filter (tg_ns_hop_scan).
    -> (cnt[m], offsets[m+1], neighbors, edge_ptrs, parents, states_out, status) -- no host synchronisation."""
    m, dev = vertices.numel(), vertices.device
    o = dict(dtype=torch.int64, device=dev)
    if group_cap is None:
        group_cap = max(1024, graph.n_edges // 512 + 2 * m + 2)
    cnt, offsets = torch.empty(max(m, 1), **o), torch.empty(m + 1, **o)
    nbr, ep, par, st_out = (torch.empty(max(m * fanout, 1), **o) for _ in range(4))
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    hin, hout, flt = TgHopIn(), TgHopOut(), TgHopFilter()
    hin.vertices = vertices.data_ptr() if m else None
    hin.ids = ids.data_ptr() if ids is not None else None
    hin.call_ids = call_ids.data_ptr() if call_ids is not None else None
    hin.m, hin.id_base, hin.fanout, hin.sampler, hin.rng_tag = m, id_base, fanout, sampler, rng_tag
    hout.cnt, hout.offsets = cnt.data_ptr(), offsets.data_ptr()
    hout.neighbors, hout.edge_ptrs, hout.parents = nbr.data_ptr(), ep.data_ptr(), par.data_ptr()
    flt.filter_mode, flt.forward = filter_mode, int(bool(forward))
    flt.win_lo, flt.win_hi = window
    flt.states = states.data_ptr() if m else None
    nbytes = C.c_int64(0)
    check(lib.tg_ns_hop_scan_workspace_bytes(C.c_int64(m), C.c_int32(fanout), C.c_int64(group_cap), C.byref(nbytes)))
    ws = torch.empty(nbytes.value // 8 + 1, **o)
    rng = TgRng(seed, call_id)
    check(lib.tg_ns_hop_scan(C.byref(graph), C.byref(hin), C.byref(flt), C.byref(rng), C.byref(hout), ptr(st_out),
                             ptr(status), ptr(ws), C.c_int64(nbytes.value), C.c_int64(group_cap), stream_ptr(dev)))
    return cnt[:m], offsets, nbr, ep, par, st_out, status
