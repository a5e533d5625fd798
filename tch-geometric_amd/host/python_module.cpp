// Host side of the MI355X backend: the CPython module `tch_geometric.tch_geometric`.
//
// Mirrors the reference's PyO3 module (src/python.rs:785-803): same function
// names, positional order, duck-typed sampler / filter arguments
// (python.rs:107-168), return tuples and ValueError behaviour
// (src/utils/tensor.rs:11-45).  The reference's host is Rust; no Rust toolchain
// exists in this pipeline, so the host is C++ (pybind11).  It owns nothing but
// argument handling and buffer allocation: every sampler runs in the gfx950
// library behind include/tchgeo.h -- there is no CPU path here.
//
// Device rule (relaxed from the reference's "must be on Cpu", tensor.rs:50-52):
// tensors may live on a HIP device (adjacency stays resident in HBM, zero copy)
// or on the CPU (uploaded for the call); results come back on the device of
// the `inputs` / `start` tensor.
//
// Randomness: the reference draws from a process-global SmallRng seeded from
// entropy that Python cannot reseed (utils/random.rs:8-22).  Here the global
// state is (seed, call counter); `seed(s)` (additive to the surface) makes
// results reproducible, and equal to the CPU oracle's philox-mode.
#include <optional>

#include "host_common.h"

using namespace tghost;

namespace tghost {
RngState &rng_state() {
    static RngState st;
    return st;
}
} // namespace tghost

namespace {

// ---------------------------------------------------------------- sampler / filter extraction (python.rs:107-168)
struct SamplerArg {
    int kind = TG_SAMPLER_UNIFORM; // python.rs:215 default
    py::object weights;            // Tensor (homogeneous) or dict[str, Tensor]
};
SamplerArg parse_sampler(const py::object &o) {
    SamplerArg s;
    if (o.is_none()) return s;
    if (py::hasattr(o, "with_replacement")) { // tried first, python.rs:132-135
        s.kind = o.attr("with_replacement").cast<bool>() ? TG_SAMPLER_UNIFORM_REPL : TG_SAMPLER_UNIFORM;
    } else if (py::hasattr(o, "weights")) {
        s.kind = TG_SAMPLER_WEIGHTED;
        s.weights = o.attr("weights");
    } else {
        throw py::type_error("sampler must expose `.with_replacement` or `.weights`");
    }
    return s;
}
struct FilterArg {
    int mode = TG_FILTER_NONE;
    bool forward = false;
    int64_t win_lo = 0, win_hi = 0;
    py::object timestamps; // Tensor or dict
    py::object state;      // Tensor or dict
};
FilterArg parse_filter(const py::object &o) {
    FilterArg f;
    if (o.is_none()) return f;
    py::tuple t = o.cast<py::tuple>();
    if (t.size() != 2) throw py::type_error("filter must be a (TemporalEdgeFilter, initial_state) tuple");
    py::object ft = t[0];
    auto window = ft.attr("window").cast<std::pair<int64_t, int64_t>>();
    const int64_t mode = ft.attr("mode").cast<int64_t>();
    f.forward = ft.attr("forward").cast<bool>();
    if (mode < 0 || mode > 2) return f; // python.rs:249: any other mode falls through to IdentityFilter
    f.mode = (int)mode;
    if (mode == TG_FILTER_STATIC) f.forward = true; // python.rs:219-224 builds <true, STATIC> for both
    f.win_lo = window.first;
    f.win_hi = window.second;
    f.timestamps = ft.attr("timestamps");
    f.state = t[1];
    return f;
}
Tensor as_homogeneous(const py::object &o) { // MixedData::build_homogenous python.rs:86-92
    if (py::isinstance<py::dict>(o)) throw py::value_error("Unknown error: \"data must be homogenous\"");
    return o.cast<Tensor>();
}
Tensor from_dict(const py::object &o, const std::string &key) { // MixedData::build_heterogenous python.rs:94-104
    if (!py::isinstance<py::dict>(o)) throw py::value_error("Unknown error: \"data must be heterogenous\"");
    py::dict d = o.cast<py::dict>();
    if (!d.contains(py::str(key))) throw py::key_error(key);
    return d[py::str(key)].cast<Tensor>();
}

// ---------------------------------------------------------------- one batched launch of the neighbor sampler
struct NsResult {
    Tensor samples, rows, cols, edge_index, states;
    std::vector<std::tuple<int64_t, int64_t, int64_t>> layer_offsets;
    int64_t n_samples = 0, n_edges = 0;
};
// workspace of one whole-device hop under a filter / with weights (the weighted sampler's group form needs more per group)
static inline int hop_workspace_bytes(bool weighted, int64_t m, int32_t k, int64_t group_cap, int64_t *bytes) {
    return weighted ? tg_ns_hop_weighted_workspace_bytes(m, k, group_cap, bytes) : tg_ns_hop_scan_workspace_bytes(m, k, group_cap, bytes);
}

// The same operator with NO read-back between hops: sample list, frontier slice and edge count stay on the device
// (csrc/het_steps.hip with one node type and one relation: tg_het_hop_begin_all packs the frontier, tg_ns_hop_segments
// samples it with the homogeneous draw tag, tg_het_hop_end_all appends), one array read when the call ends.  Worst-case
// buffers, so only for calls whose fan-out product stays small; `unchecked_seeds` as in run_ns.
bool run_ns_filtered_device(NsResult &r, const c10::Device &dev, const tg_graph &g, const Tensor &seeds,
                            const Tensor &seeds_state, const std::vector<int64_t> &fanout, const SamplerArg &s,
                            const FilterArg &f, const tg_rng &rng, uint32_t tag, const char *unchecked_seeds) {
    const int64_t n_seeds = seeds.numel();
    const bool weighted = s.kind == TG_SAMPLER_WEIGHTED, filtered = f.mode != TG_FILTER_NONE;
    const int H = (int)fanout.size();
    if (H < 1 || H > TG_MAX_HOPS || n_seeds == 0) return false;
    std::vector<int64_t> cap_f((size_t)H);
    int64_t cap_list = n_seeds, cap_e = 0, fr = n_seeds, max_m = 1, max_o = 1, max_k = 1;
    for (int h = 0; h < H; ++h) {
        const int64_t k = fanout[(size_t)h];
        if (k > 1024 || fr > ((int64_t)1 << 17) || fr * k > ((int64_t)1 << 22)) return false;
        cap_f[(size_t)h] = fr;
        max_m = std::max(max_m, fr);
        max_o = std::max(max_o, fr * k);
        max_k = std::max(max_k, k);
        fr *= k;
        cap_list += fr;
        cap_e += fr;
    }
    int64_t meta_words = 0;
    check_rc(tg_het_meta_words(1, 1, H, &meta_words));
    Tensor meta_init = at::zeros({meta_words}, at::TensorOptions().dtype(at::kLong));
    meta_init.data_ptr<int64_t>()[0] = n_seeds; // len
    meta_init.data_ptr<int64_t>()[2] = n_seeds; // fend (fbeg = 0)
    // lengths and layer offsets | status | "a seed was out of range": what the call reads back when it ends
    Tensor flags = at::zeros({2}, i64(dev));
    Tensor status = at::zeros({1}, at::TensorOptions().dtype(at::kInt).device(dev));
    const Tensor seeds_ok =
        unchecked_seeds ? sanitized_ids(seeds, g.n_major, flags.data_ptr<int64_t>() + 1, dev, unchecked_seeds) : seeds;
    Tensor list = at::empty({cap_list}, i64(dev)), st_list;
    list.narrow(0, 0, n_seeds).copy_(seeds_ok);
    if (filtered) {
        st_list = at::empty({cap_list}, i64(dev));
        st_list.narrow(0, 0, n_seeds).copy_(seeds_state);
    }
    Tensor RW = at::empty({std::max<int64_t>(cap_e, 1)}, i64(dev)), CL = at::empty_like(RW), EI = at::empty_like(RW);
    Tensor F = at::empty({max_m}, i64(dev)), ids = at::empty({max_m}, i64(dev)), fst = at::empty({max_m}, i64(dev));
    Tensor cnt = at::empty({max_m}, i64(dev)), offsets = at::empty({max_m + 1}, i64(dev));
    Tensor nbr = at::empty({max_o}, i64(dev)), ep = at::empty({max_o}, i64(dev)), par = at::empty({max_o}, i64(dev));
    Tensor st_out = at::empty({max_o}, i64(dev));
    Tensor layout = at::empty({2}, i64(dev));
    tg_hop_filter flt{};
    flt.filter_mode = f.mode;
    flt.forward = f.forward ? 1 : 0;
    flt.win_lo = f.win_lo;
    flt.win_hi = f.win_hi;
    flt.states = filtered ? fst.data_ptr<int64_t>() : nullptr;
    tg_hop_out out{cnt.data_ptr<int64_t>(), offsets.data_ptr<int64_t>(), nbr.data_ptr<int64_t>(), ep.data_ptr<int64_t>(),
                   par.data_ptr<int64_t>()};
    Tensor mh;
    int64_t status_h = 0;
    for (int64_t group_mult = 1;; group_mult *= 8) { // a retry only when the column-group guess was too low
        auto groups = [&](int h) {
            return group_mult * std::max<int64_t>(1024, g.n_edges / 512 + 2 * cap_f[(size_t)h] + 2); // weighted: group form
        };
        int64_t ws_max = 0;
        for (int h = 0; h < H; ++h) {
            int64_t b = 0;
            check_rc(hop_workspace_bytes(weighted, cap_f[(size_t)h], (int32_t)fanout[(size_t)h], groups(h), &b));
            ws_max = std::max(ws_max, b);
        }
        Tensor ws = at::empty({ws_max / 8 + 1}, i64(dev));
        Tensor meta = meta_init.to(dev);
        status.zero_();
        for (int h = 0; h < H; ++h) {
            tg_het_entry e{};
            e.segment = 0;
            e.cap = cap_f[(size_t)h];
            e.list_dst = e.list_src = list.data_ptr<int64_t>();
            e.state_dst = e.state_src = filtered ? st_list.data_ptr<int64_t>() : nullptr;
            e.cap_list_src = cap_list;
            e.rows = RW.data_ptr<int64_t>();
            e.cols = CL.data_ptr<int64_t>();
            e.edge_index = EI.data_ptr<int64_t>();
            e.cap_edges = RW.numel();
            int64_t *lay = groups(h) <= ((int64_t)1 << 20) ? layout.data_ptr<int64_t>() : nullptr;
            check_rc(tg_het_hop_begin_all(&e, 1, meta.data_ptr<int64_t>(), 1, 1, H, e.cap, F.data_ptr<int64_t>(),
                                          filtered ? fst.data_ptr<int64_t>() : nullptr, ids.data_ptr<int64_t>(), lay,
                                          stream_of(dev)));
            tg_hop_segment sg{};
            sg.graph = &g;
            sg.fanout = (int32_t)fanout[(size_t)h];
            sg.rng_tag = tag;
            tg_hop_in in{};
            in.vertices = F.data_ptr<int64_t>();
            in.ids = ids.data_ptr<int64_t>();
            in.m = e.cap;
            in.fanout = sg.fanout;
            in.sampler = s.kind;
            int64_t ws_bytes = 0;
            check_rc(hop_workspace_bytes(weighted, e.cap, sg.fanout, groups(h), &ws_bytes));
            check_rc(tg_ns_hop_segments(&sg, 1, &in, lay, &flt, &rng, &out, st_out.data_ptr<int64_t>(),
                                        status.data_ptr<int32_t>(), ws.data_ptr<int64_t>(), ws_bytes, groups(h),
                                        stream_of(dev)));
            check_rc(tg_het_hop_end_all(&e, 1, &out, filtered ? st_out.data_ptr<int64_t>() : nullptr, e.cap,
                                        e.cap * sg.fanout, lay, meta.data_ptr<int64_t>(), 1, 1, H, h, 1,
                                        status.data_ptr<int32_t>(), stream_of(dev)));
        }
        mh = to_host(at::cat({meta, status.to(at::kLong), flags})); // the call's only synchronisation
        status_h = mh.data_ptr<int64_t>()[meta_words];
        if (unchecked_seeds) raise_if_flagged(mh.data_ptr<int64_t>()[meta_words + 2], unchecked_seeds);
        if ((status_h & 1) && group_mult < 4096) continue;
        break;
    }
    if (status_h & 1) throw std::runtime_error("neighbor sampling: column-group workspace overflow");
    if (status_h & 2) // sampling.rs:49: gen_range over an empty float range panics in the reference
        throw PanicError("weighted sampling met a non-positive running weight sum (the reference panics here)");
    if (status_h & 4) throw std::runtime_error("neighbor sampling: internal capacity error");
    const int64_t *m = mh.data_ptr<int64_t>();
    r.n_samples = m[0];
    r.n_edges = m[3];
    r.samples = list;
    if (filtered) r.states = st_list;
    r.rows = RW;
    r.cols = CL;
    r.edge_index = EI;
    for (int h = 0; h < H; ++h) r.layer_offsets.emplace_back(m[4 + 3 * h], m[4 + 3 * h + 1], m[4 + 3 * h + 2]);
    return true;
}

// Uniform samplers under a temporal filter: hop by hop through tg_ns_hop_scan, which spreads the column scans
// over the whole device (the one-workgroup-per-batch kernel needs many batches in one launch to do that).
NsResult run_ns_filtered_flat(const c10::Device &dev, const Tensor &ptrs, const Tensor &indices, const Tensor &weights,
                              const Tensor &timestamps, const Tensor &seeds, const Tensor &seeds_state,
                              const std::vector<int64_t> &fanout, const SamplerArg &s, const FilterArg &f,
                              const tg_rng &rng, uint32_t tag, int64_t id_base, const char *unchecked_seeds = nullptr) {
    NsResult r;
    const int64_t n_seeds = seeds.numel();
    const bool weighted = s.kind == TG_SAMPLER_WEIGHTED, filtered = f.mode != TG_FILTER_NONE;
    tg_graph g{};
    g.ptrs = ptrs.data_ptr<int64_t>();
    g.indices = indices.numel() ? indices.data_ptr<int64_t>() : nullptr;
    g.timestamps = filtered ? timestamps.data_ptr<int64_t>() : nullptr;
    g.weights = weighted ? weights.data_ptr<double>() : nullptr;
    g.n_major = ptrs.numel() - 1;
    g.n_edges = indices.numel();
    if ((weighted || filtered) && id_base == 0 &&
        run_ns_filtered_device(r, dev, g, seeds, seeds_state, fanout, s, f, rng, tag, unchecked_seeds))
        return r;
    if (unchecked_seeds) {
        RangeCheck rc(dev);
        rc.add(seeds, ptrs.numel() - 1);
        rc.verify(unchecked_seeds);
    }
    std::vector<Tensor> samples{seeds}, states, rows, cols, eidx;
    if (filtered) states.push_back(seeds_state);
    Tensor frontier = seeds, fstate = seeds_state;
    int64_t ne = 0, slot0 = 0;
    for (int64_t k : fanout) {
        r.layer_offsets.emplace_back(n_seeds + ne, ne, n_seeds + ne); // neighbor_sampling.rs:193
        const int64_t m = frontier.numel();
        if (m == 0) continue;
        if (k > 1024 && (weighted || filtered))
            throw py::value_error("num_neighbors above 1024 is not supported with a temporal filter or weights");
        Tensor cnt = at::empty({m}, i64(dev)), offsets = at::empty({m + 1}, i64(dev));
        Tensor nbr = at::empty({m * k}, i64(dev)), ep = at::empty({m * k}, i64(dev)), par = at::empty({m * k}, i64(dev));
        Tensor st_out = at::empty({m * k}, i64(dev));
        tg_hop_in in{};
        in.vertices = frontier.data_ptr<int64_t>();
        in.m = m;
        in.id_base = id_base + slot0;
        in.fanout = (int32_t)k;
        in.sampler = s.kind;
        in.rng_tag = tag;
        tg_hop_filter flt{};
        flt.filter_mode = f.mode;
        flt.forward = f.forward ? 1 : 0;
        flt.win_lo = f.win_lo;
        flt.win_hi = f.win_hi;
        flt.states = filtered ? fstate.data_ptr<int64_t>() : nullptr;
        tg_hop_out out{cnt.data_ptr<int64_t>(), offsets.data_ptr<int64_t>(), nbr.data_ptr<int64_t>(),
                       ep.data_ptr<int64_t>(), par.data_ptr<int64_t>()};
        int64_t group_cap = std::max<int64_t>(1024, indices.numel() / 512 + 2 * m + 2), total = 0;
        if (!weighted && !filtered) { // plain uniform samplers over a large frontier: the whole-device hop of ns_hop.hip
            int64_t ws_bytes = 0;
            check_rc(tg_ns_hop_workspace_bytes(m, &ws_bytes));
            Tensor ws = at::empty({ws_bytes / 8 + 1}, i64(dev));
            check_rc(tg_ns_hop(&g, &in, &rng, &out, ws.data_ptr<int64_t>(), ws_bytes, stream_of(dev)));
            total = read_scalar<int64_t>(offsets[m]);
        } else if (weighted) { // sampling.rs:28-55 in the group form: flat over the 512-edge groups of the frontier's columns
            for (;;) {
                Tensor status = at::zeros({1}, at::TensorOptions().dtype(at::kInt).device(dev));
                int64_t ws_bytes = 0;
                check_rc(hop_workspace_bytes(weighted, m, (int32_t)k, group_cap, &ws_bytes));
                Tensor ws = at::empty({ws_bytes / 8 + 1}, i64(dev));
                check_rc(tg_ns_hop_weighted_groups(&g, &in, &flt, &rng, &out, st_out.data_ptr<int64_t>(),
                                                   status.data_ptr<int32_t>(), ws.data_ptr<int64_t>(), ws_bytes, group_cap,
                                                   stream_of(dev)));
                total = read_scalar<int64_t>(offsets[m]);
                const int32_t st_h = read_scalar<int32_t>(status);
                if (st_h & 2) // sampling.rs:49: gen_range over an empty float range panics
                    throw PanicError("weighted sampling met a non-positive running weight sum (the reference panics here)");
                if (!(st_h & 1)) break;
                group_cap *= 8; // the frontier's columns need more groups than guessed
            }
        } else
        for (;;) { // the frontier's columns need sum(ceil(deg/512)) groups; grow the workspace if the guess was low
            Tensor status = at::zeros({1}, at::TensorOptions().dtype(at::kInt).device(dev));
            int64_t ws_bytes = 0;
            check_rc(hop_workspace_bytes(weighted, m, (int32_t)k, group_cap, &ws_bytes));
            Tensor ws = at::empty({ws_bytes / 8 + 1}, i64(dev));
            check_rc(tg_ns_hop_scan(&g, &in, &flt, &rng, &out, st_out.data_ptr<int64_t>(), status.data_ptr<int32_t>(),
                                    ws.data_ptr<int64_t>(), ws_bytes, group_cap, stream_of(dev)));
            total = read_scalar<int64_t>(offsets[m]); // the hop's size read-back
            if (read_scalar<int32_t>(status) == 0) break;
            group_cap *= 8;
        }
        if (total > 0) {
            samples.push_back(nbr.narrow(0, 0, total));
            if (filtered) states.push_back(st_out.narrow(0, 0, total));
            rows.push_back(at::arange(n_seeds + ne, n_seeds + ne + total, i64(dev)));
            cols.push_back(par.narrow(0, 0, total) + slot0);
            eidx.push_back(ep.narrow(0, 0, total));
        }
        frontier = nbr.narrow(0, 0, total).contiguous();
        fstate = st_out.narrow(0, 0, total).contiguous();
        slot0 = n_seeds + ne;
        ne += total;
    }
    auto cat_or_empty = [&](const std::vector<Tensor> &v) { return v.empty() ? at::empty({0}, i64(dev)) : at::cat(v); };
    r.samples = at::cat(samples);
    if (filtered) r.states = at::cat(states);
    r.rows = cat_or_empty(rows);
    r.cols = cat_or_empty(cols);
    r.edge_index = cat_or_empty(eidx);
    r.n_samples = n_seeds + ne;
    r.n_edges = ne;
    return r;
}

NsResult run_ns(const c10::Device &dev, const Tensor &ptrs, const Tensor &indices, const Tensor &weights,
                const Tensor &timestamps, const Tensor &seeds, const Tensor &seeds_state,
                const std::vector<int64_t> &fanout, const SamplerArg &s, const FilterArg &f, const tg_rng &rng,
                uint32_t tag, int64_t id_base, const char *unchecked_seeds = nullptr) {
    // unchecked_seeds: the seeds have not been range-checked yet (the string names the operator in the error); the
    // one-launch path checks them without a read-back of its own (host_common.h sanitized_ids)
    // whole-device flat hops: always for filters / weights (column scans), and for the plain samplers when ONE call
    // brings more seeds than a workgroup should walk alone (the batched kernel gives a batch one workgroup) or a
    // fan-out above 128 (the batched kernel keeps its ticket strips in LDS; tg_ns_hop takes up to 4096)
    int64_t max_k = 1;
    for (int64_t k : fanout) max_k = std::max(max_k, k);
    // ... or more hops than the one-launch kernel's TG_MAX_HOPS (the reference takes any number: neighbor_sampling.rs:188)
    if (f.mode != TG_FILTER_NONE || s.kind == TG_SAMPLER_WEIGHTED || seeds.numel() > 2048 || max_k > 128 ||
        fanout.size() > (size_t)TG_MAX_HOPS) {
        return run_ns_filtered_flat(dev, ptrs, indices, weights, timestamps, seeds, seeds_state, fanout, s, f, rng, tag,
                                    id_base, unchecked_seeds);
    }
    const int32_t H = (int32_t)fanout.size();
    int64_t cap_nodes = 0, cap_edges = 0;
    check_rc(tg_ns_homo_capacity(seeds.numel(), fanout.data(), H, &cap_nodes, &cap_edges));
    NsResult r;
    { // one allocation for the four output lists (the views handed back keep it alive)
        const int64_t cn = std::max<int64_t>(cap_nodes, 1), ce = std::max<int64_t>(cap_edges, 1);
        Tensor arena = at::empty({cn + 3 * ce}, i64(dev));
        r.samples = arena.narrow(0, 0, cn);
        r.rows = arena.narrow(0, cn, ce);
        r.cols = arena.narrow(0, cn + ce, ce);
        r.edge_index = arena.narrow(0, cn + 2 * ce, ce);
    }
    // layer offsets | counts | "a seed was out of range": ONE array, one read-back when the call ends
    const int64_t lo_words = (int64_t)std::max<int32_t>(H, 1) * 3;
    Tensor tail = at::zeros({lo_words + 3}, i64(dev));
    int64_t *tail_p = tail.data_ptr<int64_t>();
    const Tensor seeds_ok =
        unchecked_seeds ? sanitized_ids(seeds, ptrs.numel() - 1, tail_p + lo_words + 2, dev, unchecked_seeds) : seeds;
    if (f.mode != TG_FILTER_NONE) r.states = at::empty({std::max<int64_t>(cap_nodes, 1)}, i64(dev));

    tg_graph g{};
    g.ptrs = ptrs.data_ptr<int64_t>();
    g.indices = indices.data_ptr<int64_t>();
    g.weights = weights.defined() ? weights.data_ptr<double>() : nullptr;
    g.timestamps = timestamps.defined() ? timestamps.data_ptr<int64_t>() : nullptr;
    g.n_major = ptrs.numel() - 1;
    g.n_edges = indices.numel();
    tg_ns_config cfg{};
    cfg.sampler = s.kind;
    cfg.filter_mode = f.mode;
    cfg.forward = f.forward ? 1 : 0;
    cfg.rng_tag = tag;
    cfg.win_lo = f.win_lo;
    cfg.win_hi = f.win_hi;
    cfg.seeds_state = seeds_state.defined() ? seeds_state.data_ptr<int64_t>() : nullptr;
    cfg.id_base = id_base;
    tg_ns_out out{};
    out.samples = r.samples.data_ptr<int64_t>();
    out.rows = r.rows.data_ptr<int64_t>();
    out.cols = r.cols.data_ptr<int64_t>();
    out.edge_index = r.edge_index.data_ptr<int64_t>();
    out.layer_offsets = tail_p;
    out.counts = tail_p + lo_words;
    out.states = r.states.defined() ? r.states.data_ptr<int64_t>() : nullptr;
    out.cap_nodes = r.samples.numel();
    out.cap_edges = r.rows.numel();
    check_rc(tg_ns_homo_batched(&g, seeds_ok.numel() ? seeds_ok.data_ptr<int64_t>() : nullptr, 1, seeds_ok.numel(),
                                fanout.data(), H, &cfg, &rng, &out, stream_of(dev)));
    Tensor th = to_host(tail); // the only synchronisation of the call
    const int64_t *l = th.data_ptr<int64_t>();
    if (unchecked_seeds) raise_if_flagged(l[lo_words + 2], unchecked_seeds);
    r.n_samples = l[lo_words];
    r.n_edges = l[lo_words + 1];
    if (r.n_samples < 0) // sampling.rs:49: gen_range over an empty float range panics in the reference
        throw PanicError("weighted sampling met a non-positive running weight sum (the reference panics here)");
    for (int h = 0; h < H; ++h) r.layer_offsets.emplace_back(l[3 * h], l[3 * h + 1], l[3 * h + 2]);
    return r;
}

void validate_fanout(const std::vector<int64_t> &f) {
    for (int64_t k : f)
        if (k < 1) throw py::value_error("num_neighbors entries must be >= 1 (the reference panics on 0)");
}

// ---------------------------------------------------------------- python.rs:27-53 to_csc / to_csr
std::tuple<int64_t, int64_t> graph_size(const py::object &size) { // GraphSize python.rs:12-25
    if (py::isinstance<py::int_>(size)) {
        const int64_t n = size.cast<int64_t>();
        return {n, n};
    }
    auto p = size.cast<std::pair<int64_t, int64_t>>();
    return {p.first, p.second};
}
std::tuple<Tensor, Tensor, Tensor> to_csx(const Tensor &row_col, const py::object &size, bool csc) {
    auto [size0, size1] = graph_size(size);
    const c10::Device dev = compute_device({&row_col});
    DeviceGuard guard(dev);
    Tensor rc = on(row_col, dev, at::kLong);
    if (rc.dim() != 2 || rc.size(0) != 2) throw py::value_error("row_col must have shape [2, E]");
    Tensor row = rc.select(0, 0).contiguous(), col = rc.select(0, 1).contiguous();
    const int64_t nnz = row.numel(), m = csc ? size1 : size0;
    if (size0 < 1 || size1 < 1) throw py::value_error("graph size must be positive");
    {   // an id outside the graph would be mis-sorted silently (the sort keys only the bits a valid key needs); the
        // reference's ind2ptr writes out of bounds and panics there (storage.rs:67-101)
        RangeCheck rc(dev);
        rc.add(row, size0);
        rc.add(col, size1);
        rc.verify("to_csc/to_csr row_col");
    }
    Tensor ptrs = at::empty({m + 1}, i64(dev)), indices = at::empty({nnz}, i64(dev)), perm = at::empty({nnz}, i64(dev));
    int64_t ws_bytes = 0;
    check_rc(tg_coo_to_csx_workspace_bytes(nnz, size0, size1, &ws_bytes));
    Tensor ws = at::empty({ws_bytes / 8 + 1}, i64(dev));
    // storage.rs:112,119: perm = argsort(major * size_minor + minor); stable, so duplicate edges keep input order
    check_rc(tg_coo_to_csx(nnz ? row.data_ptr<int64_t>() : nullptr, nnz ? col.data_ptr<int64_t>() : nullptr, nnz, size0,
                           size1, csc ? 1 : 0, ptrs.data_ptr<int64_t>(), nnz ? indices.data_ptr<int64_t>() : nullptr,
                           nnz ? perm.data_ptr<int64_t>() : nullptr, ws.data_ptr<int64_t>(), ws_bytes, stream_of(dev)));
    const c10::Device out_dev = row_col.device();
    return {back(ptrs, out_dev), back(indices, out_dev), back(perm, out_dev)};
}

// ---------------------------------------------------------------- python.rs:187-271
py::tuple neighbor_sampling_homogenous(const Tensor &col_ptrs, const Tensor &row_indices, const Tensor &inputs,
                                       const std::vector<int64_t> &num_neighbors, const py::object &sampler,
                                       const py::object &filter) {
    const SamplerArg s = parse_sampler(sampler);
    const FilterArg f = parse_filter(filter);
    validate_fanout(num_neighbors);
    const c10::Device dev = compute_device({&col_ptrs, &row_indices, &inputs});
    DeviceGuard guard(dev);
    Tensor ptrs = on_graph(col_ptrs, dev, at::kLong), idx = on_graph(row_indices, dev, at::kLong);
    Tensor seeds = on(inputs, dev, at::kLong).reshape({-1});
    Tensor w, ts, st;
    if (s.kind == TG_SAMPLER_WEIGHTED) w = on_graph(as_homogeneous(s.weights), dev, at::kDouble); // python.rs:214
    if (f.mode != TG_FILTER_NONE) {
        ts = on_graph(as_homogeneous(f.timestamps), dev, at::kLong); // python.rs:149
        st = on(as_homogeneous(f.state), dev, at::kLong).reshape({-1});
        if (st.numel() != seeds.numel()) throw py::value_error("filter state must have one entry per input");
    }
    NsResult r;
    { // nothing in here touches Python: worker threads (each on its own HIP stream) overlap their calls
        NoGil nogil;
        if (num_neighbors.size() > 1)
            check_graph_ids(idx, ptrs.numel() - 1, dev, "neighbor_sampling_homogenous row_indices");
        r = run_ns(dev, ptrs, idx, w, ts, seeds, st, num_neighbors, s, f, next_rng(), 0, 0,
                   "neighbor_sampling_homogenous inputs");
    }
    const c10::Device out_dev = inputs.device();
    return py::make_tuple(back(r.samples.narrow(0, 0, r.n_samples), out_dev),
                          back(r.rows.narrow(0, 0, r.n_edges), out_dev), back(r.cols.narrow(0, 0, r.n_edges), out_dev),
                          back(r.edge_index.narrow(0, 0, r.n_edges), out_dev), r.layer_offsets);
}

// ---------------------------------------------------------------- python.rs:275-395
// neighbor_sampling.rs:233-356 driven one (hop, relation) at a time on the device; relations are visited in
// `edge_types` order (the reference's HashMap order is not reproducible).
py::tuple neighbor_sampling_heterogenous(const std::vector<std::string> &node_types,
                                         const std::vector<std::tuple<std::string, std::string, std::string>> &edge_types,
                                         const py::dict &col_ptrs, const py::dict &row_indices, const py::dict &inputs,
                                         const py::dict &num_neighbors, int64_t num_hops, const py::object &sampler,
                                         const py::object &filter) {
    const SamplerArg s = parse_sampler(sampler);
    const FilterArg f = parse_filter(filter);
    const size_t T = node_types.size();
    std::map<std::string, size_t> tix;
    for (size_t t = 0; t < T; ++t) tix[node_types[t]] = t;

    const Tensor *first = nullptr;
    Tensor first_holder;
    for (auto item : col_ptrs) {
        first_holder = item.second.cast<Tensor>();
        first = &first_holder;
        break;
    }
    const c10::Device dev = compute_device({first});
    DeviceGuard guard(dev);
    const tg_rng rng = next_rng();
    const bool has_state = f.mode != TG_FILTER_NONE;

    struct Rel {
        std::string key;
        size_t src, dst;
        Tensor ptrs, idx, w, ts;
        std::vector<int64_t> fanout;
        bool active;
        std::vector<Tensor> rows, cols, eidx;
        int64_t n_edges = 0;
        std::vector<std::tuple<int64_t, int64_t, int64_t>> layer_offsets;
    };
    std::vector<Rel> rels;
    for (const auto &et : edge_types) {
        Rel r;
        r.key = rel_key(et);
        if (!col_ptrs.contains(py::str(r.key))) continue; // graphs are keyed by col_ptrs (python.rs:294)
        r.src = tix.at(std::get<0>(et));
        r.dst = tix.at(std::get<2>(et));
        r.ptrs = on_graph(col_ptrs[py::str(r.key)].cast<Tensor>(), dev, at::kLong);
        r.idx = on_graph(row_indices[py::str(r.key)].cast<Tensor>(), dev, at::kLong);
        r.active = num_neighbors.contains(py::str(r.key)); // the hop loop iterates num_neighbors (:294)
        if (r.active) {
            r.fanout = num_neighbors[py::str(r.key)].cast<std::vector<int64_t>>();
            validate_fanout(r.fanout);
            if ((int64_t)r.fanout.size() < num_hops) throw py::index_error("num_neighbors[" + r.key + "] is shorter than num_hops");
        }
        if (s.kind == TG_SAMPLER_WEIGHTED) r.w = on_graph(from_dict(s.weights, r.key), dev, at::kDouble);
        if (has_state) r.ts = on_graph(from_dict(f.timestamps, r.key), dev, at::kLong);
        rels.push_back(std::move(r));
    }

    // per node type: every chunk appended so far, the current frontier and its global begin index
    std::vector<std::vector<Tensor>> chunks(T), st_chunks(T), new_chunks(T), new_st(T);
    std::vector<Tensor> frontier(T), frontier_st(T);
    std::vector<int64_t> len(T, 0), fbegin(T, 0);
    c10::Device out_dev = dev;
    bool out_dev_set = false;
    for (size_t t = 0; t < T; ++t) { // :264-278
        if (inputs.contains(py::str(node_types[t]))) {
            Tensor in = inputs[py::str(node_types[t])].cast<Tensor>();
            if (!out_dev_set) {
                out_dev = in.device();
                out_dev_set = true;
            }
            frontier[t] = on(in, dev, at::kLong).reshape({-1});
            if (has_state) frontier_st[t] = on(from_dict(f.state, node_types[t]), dev, at::kLong).reshape({-1});
        } else {
            frontier[t] = at::empty({0}, i64(dev));
            if (has_state) frontier_st[t] = at::empty({0}, i64(dev));
        }
        chunks[t].push_back(frontier[t]);
        if (has_state) st_chunks[t].push_back(frontier_st[t]);
        len[t] = frontier[t].numel();
    }
    // ---- default samplers without a filter: all hops and relations in ONE launch (tg_ns_hetero_batched), one read-back
    bool fused = !has_state && s.kind != TG_SAMPLER_WEIGHTED && T <= TG_HET_MAX_TYPES && rels.size() <= TG_HET_MAX_RELS &&
                 num_hops >= 0 && num_hops <= TG_MAX_HOPS;
    for (const Rel &r : rels)
        for (int64_t h = 0; fused && r.active && h < num_hops; ++h) fused = r.fanout[(size_t)h] <= TG_MAX_FANOUT;
    int64_t total_inputs = 0; // the fused kernel gives the call ONE workgroup: large calls go hop by hop over the device
    for (size_t t = 0; t < T; ++t) total_inputs += frontier[t].numel();
    fused = fused && total_inputs <= 4096;
    // an input of type t indexes the columns of every relation whose dst is t.  The one-launch form and the
    // device-driven steps check without a read-back of their own (sanitized_ids); the host-driven loop checks first.
    auto input_bound = [&](size_t t) {
        int64_t bound = -1;
        for (const Rel &rl : rels)
            if (rl.dst == t) bound = bound < 0 ? rl.ptrs.numel() - 1 : std::min<int64_t>(bound, rl.ptrs.numel() - 1);
        return bound;
    };
    auto verify_inputs_now = [&]() {
        RangeCheck rc(dev);
        for (const Rel &r : rels) rc.add(frontier[r.dst], r.ptrs.numel() - 1);
        rc.verify("neighbor_sampling_heterogenous inputs");
    };
    {
        if (num_hops > 1) // samples of relation a (src type s) are the next hop's frontier of every relation into s
            for (const Rel &a : rels)
                for (const Rel &b : rels)
                    if (b.dst == a.src)
                        check_graph_ids(a.idx, b.ptrs.numel() - 1, dev, "neighbor_sampling_heterogenous row_indices");
    }
    if (fused) {
        std::optional<NoGil> nogil; // no Python between here and the result dicts
        nogil.emplace();
        const int R = (int)rels.size(), H = (int)num_hops;
        std::vector<int32_t> rel_src((size_t)std::max(R, 1)), rel_dst((size_t)std::max(R, 1));
        std::vector<tg_graph> graphs((size_t)std::max(R, 1));
        std::vector<int64_t> fan((size_t)std::max(R * H, 1), 0), n_in(T, 0), cap_n(T, 0), cap_e((size_t)std::max(R, 1), 0);
        std::vector<const int64_t *> in_ptr(T, nullptr);
        for (int r = 0; r < R; ++r) {
            rel_src[(size_t)r] = (int32_t)rels[(size_t)r].src;
            rel_dst[(size_t)r] = (int32_t)rels[(size_t)r].dst;
            tg_graph g{};
            g.ptrs = rels[(size_t)r].ptrs.data_ptr<int64_t>();
            g.indices = rels[(size_t)r].idx.numel() ? rels[(size_t)r].idx.data_ptr<int64_t>() : nullptr;
            g.n_major = rels[(size_t)r].ptrs.numel() - 1;
            g.n_edges = rels[(size_t)r].idx.numel();
            graphs[(size_t)r] = g;
            for (int h = 0; h < H && rels[(size_t)r].active; ++h) fan[(size_t)(r * H + h)] = rels[(size_t)r].fanout[(size_t)h];
        }
        // counts | layer offsets | "an input was out of range" in one tensor: one read-back for the call
        const int64_t meta_words = (int64_t)T + R + (int64_t)std::max(R * H, 1) * 3;
        Tensor meta = at::zeros({meta_words + 1}, i64(dev));
        std::vector<Tensor> checked(T);
        for (size_t t = 0; t < T; ++t) {
            n_in[t] = frontier[t].numel();
            const int64_t bound = input_bound(t);
            checked[t] = bound < 0 ? frontier[t]
                                   : sanitized_ids(frontier[t], bound, meta.data_ptr<int64_t>() + meta_words, dev,
                                                   "neighbor_sampling_heterogenous inputs");
            in_ptr[t] = n_in[t] ? checked[t].data_ptr<int64_t>() : nullptr;
        }
        tg_het_problem pb{};
        pb.n_types = (int32_t)T;
        pb.n_rels = R;
        pb.n_hops = H;
        pb.sampler = s.kind;
        pb.rel_src = rel_src.data();
        pb.rel_dst = rel_dst.data();
        pb.graphs = graphs.data();
        pb.fanout = fan.data();
        pb.inputs = in_ptr.data();
        pb.n_inputs = n_in.data();
        check_rc(tg_ns_hetero_capacity(&pb, cap_n.data(), cap_e.data()));
        std::vector<Tensor> S(T), RW((size_t)R), CL((size_t)R), EI((size_t)R);
        std::vector<int64_t *> s_ptr(T), r_ptr((size_t)std::max(R, 1)), c_ptr((size_t)std::max(R, 1)), e_ptr((size_t)std::max(R, 1));
        // one allocation for every output list of the call (the views handed back keep it alive): an allocation costs
        // a few microseconds, and there would be T + 3R of them in a call whose kernel takes ~150
        int64_t arena_words = 0;
        for (size_t t = 0; t < T; ++t) arena_words += std::max<int64_t>(cap_n[t], 1);
        for (int r = 0; r < R; ++r) arena_words += 3 * std::max<int64_t>(cap_e[(size_t)r], 1);
        Tensor arena = at::empty({arena_words}, i64(dev));
        int64_t at_word = 0;
        auto carve = [&](int64_t words) {
            Tensor v = arena.narrow(0, at_word, words);
            at_word += words;
            return v;
        };
        for (size_t t = 0; t < T; ++t) {
            S[t] = carve(std::max<int64_t>(cap_n[t], 1));
            s_ptr[t] = S[t].data_ptr<int64_t>();
        }
        for (int r = 0; r < R; ++r) {
            const int64_t c = std::max<int64_t>(cap_e[(size_t)r], 1);
            RW[(size_t)r] = carve(c);
            CL[(size_t)r] = carve(c);
            EI[(size_t)r] = carve(c);
            r_ptr[(size_t)r] = RW[(size_t)r].data_ptr<int64_t>();
            c_ptr[(size_t)r] = CL[(size_t)r].data_ptr<int64_t>();
            e_ptr[(size_t)r] = EI[(size_t)r].data_ptr<int64_t>();
        }
        tg_het_out out{};
        out.samples = s_ptr.data();
        out.cap_nodes = cap_n.data();
        out.rows = r_ptr.data();
        out.cols = c_ptr.data();
        out.edge_index = e_ptr.data();
        out.cap_edges = cap_e.data();
        out.counts = meta.data_ptr<int64_t>();
        out.layer_offsets = meta.data_ptr<int64_t>() + T + R;
        check_rc(tg_ns_hetero_batched(&pb, 1, &rng, &out, stream_of(dev)));
        Tensor m = to_host(meta); // the call's only synchronisation
        const int64_t *mh = m.data_ptr<int64_t>();
        nogil.reset();
        raise_if_flagged(mh[meta_words], "neighbor_sampling_heterogenous inputs");
        py::dict samples, rows, cols, eidx, los;
        for (size_t t = 0; t < T; ++t) samples[py::str(node_types[t])] = back(S[t].narrow(0, 0, mh[t]), out_dev);
        for (int r = 0; r < R; ++r) {
            const Rel &rl = rels[(size_t)r];
            const int64_t ne = mh[T + (size_t)r];
            rows[py::str(rl.key)] = back(RW[(size_t)r].narrow(0, 0, ne), out_dev);
            cols[py::str(rl.key)] = back(CL[(size_t)r].narrow(0, 0, ne), out_dev);
            eidx[py::str(rl.key)] = back(EI[(size_t)r].narrow(0, 0, ne), out_dev);
            std::vector<std::tuple<int64_t, int64_t, int64_t>> lo;
            for (int h = 0; h < H && rl.active; ++h) {
                const int64_t *q = mh + T + R + ((size_t)r * H + (size_t)h) * 3;
                lo.emplace_back(q[0], q[1], q[2]);
            }
            los[py::str(rl.key)] = lo;
        }
        return py::make_tuple(samples, rows, cols, eidx, los);
    }

    // ---- every other case: flat (hop, relation) steps.  Device-driven when the worst-case buffers are affordable: list
    // lengths, frontier slices and edge counts stay on the device (csrc/het_steps.hip), the whole call issues its
    // launches without a read-back in between and reads one small array at the end.
    if (T <= 64 && num_hops >= 1 && num_hops <= TG_MAX_HOPS && dev.is_cuda()) {
        const int R = (int)rels.size(), H = (int)num_hops;
        std::vector<int64_t> cap_list(T), fsz(T), cap_e((size_t)std::max(R, 1), 0);
        std::vector<std::vector<int64_t>> cap_f((size_t)H, std::vector<int64_t>((size_t)std::max(R, 1), 0));
        for (size_t t = 0; t < T; ++t) cap_list[t] = fsz[t] = frontier[t].numel();
        int64_t max_f = 1, max_out = 1, max_k = 1;
        double words = 0;
        bool affordable = true;
        for (int h = 0; h < H && affordable; ++h) {
            std::vector<int64_t> fresh(T, 0);
            for (int r = 0; r < R; ++r) {
                const Rel &rl = rels[(size_t)r];
                if (!rl.active) continue;
                const int64_t fr = fsz[rl.dst], k = rl.fanout[(size_t)h];
                if (fr > 0 && k > ((int64_t)1 << 40) / fr) {
                    affordable = false;
                    break;
                }
                cap_f[(size_t)h][(size_t)r] = fr;
                cap_e[(size_t)r] += fr * k;
                fresh[rl.src] += fr * k;
                max_f = std::max(max_f, fr);
                max_out = std::max(max_out, fr * k);
                max_k = std::max(max_k, k);
            }
            for (size_t t = 0; t < T; ++t) {
                fsz[t] = fresh[t];
                cap_list[t] += fresh[t];
            }
        }
        for (size_t t = 0; t < T; ++t) words += (double)cap_list[t] * (has_state ? 2 : 1);
        for (int r = 0; r < R; ++r) words += (double)cap_e[(size_t)r] * 3;
        const bool weighted = s.kind == TG_SAMPLER_WEIGHTED;
        // (with a filter or weights the relations of a hop share one frontier buffer: up to R times the widest one)
        words += ((double)max_f * 5 + (double)max_out * 4) * ((weighted || has_state) ? std::max(R, 1) : 1);
        if ((weighted || has_state) && max_k > 1024)
            throw py::value_error("num_neighbors above 1024 is not supported with a temporal filter or weights");
        if (affordable && words * 8 <= 8e9 && (weighted || has_state || max_k <= 4096)) {
            std::optional<NoGil> nogil; // no Python between here and the result dicts
            nogil.emplace();
            int64_t meta_words = 0;
            check_rc(tg_het_meta_words((int32_t)T, R, H, &meta_words));
            Tensor meta_init = at::zeros({meta_words}, at::TensorOptions().dtype(at::kLong));
            int64_t *mi = meta_init.data_ptr<int64_t>();
            for (size_t t = 0; t < T; ++t) {
                mi[t] = frontier[t].numel();          // len
                mi[2 * T + t] = frontier[t].numel();  // fend (fbeg = 0)
            }
            std::vector<Tensor> lists(T), st_lists(T), RW((size_t)R), CL((size_t)R), EI((size_t)R);
            Tensor id_flag = at::zeros({1}, i64(dev)); // "an input was out of range": read back with the lengths
            for (size_t t = 0; t < T; ++t) {
                lists[t] = at::empty({std::max<int64_t>(cap_list[t], 1)}, i64(dev));
                const int64_t bound = input_bound(t);
                if (frontier[t].numel())
                    lists[t].narrow(0, 0, frontier[t].numel())
                        .copy_(bound < 0 ? frontier[t]
                                         : sanitized_ids(frontier[t], bound, id_flag.data_ptr<int64_t>(), dev,
                                                         "neighbor_sampling_heterogenous inputs"));
                if (has_state) {
                    st_lists[t] = at::empty({std::max<int64_t>(cap_list[t], 1)}, i64(dev));
                    if (frontier_st[t].numel()) st_lists[t].narrow(0, 0, frontier_st[t].numel()).copy_(frontier_st[t]);
                }
            }
            for (int r = 0; r < R; ++r) {
                const int64_t c = std::max<int64_t>(cap_e[(size_t)r], 1);
                RW[(size_t)r] = at::empty({c}, i64(dev));
                CL[(size_t)r] = at::empty({c}, i64(dev));
                EI[(size_t)r] = at::empty({c}, i64(dev));
            }
            // filters / weights: ALL relations of a hop go through one set of launches (concatenated frontier,
            // tg_ns_hop_segments); the plain samplers beyond the fused launch keep one flat hop per relation
            const bool all_at_once = weighted || has_state;
            int64_t max_m = max_f, max_o = max_out;
            std::vector<int64_t> hop_m((size_t)H, 0), hop_groups((size_t)H, 0), hop_k((size_t)H, 1);
            for (int h = 0; h < H && all_at_once; ++h) {
                int64_t o = 0;
                for (int r = 0; r < R; ++r) {
                    const Rel &rl = rels[(size_t)r];
                    const int64_t cf = cap_f[(size_t)h][(size_t)r];
                    if (!rl.active || cf == 0) continue;
                    hop_m[(size_t)h] += cf;
                    hop_groups[(size_t)h] += rl.idx.numel() / 512;
                    hop_k[(size_t)h] = std::max(hop_k[(size_t)h], rl.fanout[(size_t)h]);
                    o += cf * rl.fanout[(size_t)h];
                }
                max_m = std::max(max_m, hop_m[(size_t)h]);
                max_o = std::max(max_o, o);
            }
            Tensor F = at::empty({max_m}, i64(dev)), ids = at::empty({max_m}, i64(dev)), fst = at::empty({max_m}, i64(dev));
            Tensor cnt = at::empty({max_m}, i64(dev)), offsets = at::empty({max_m + 1}, i64(dev));
            Tensor nbr = at::empty({max_o}, i64(dev)), ep = at::empty({max_o}, i64(dev)), par = at::empty({max_o}, i64(dev));
            Tensor st_out = at::empty({max_o}, i64(dev));
            Tensor status = at::zeros({1}, at::TensorOptions().dtype(at::kInt).device(dev));
            Tensor meta, mh, ws;
            Tensor layout = at::empty({TG_HOP_MAX_SEGMENTS + 1}, i64(dev));
            int32_t status_h = 0;
            std::vector<tg_graph> graphs((size_t)std::max(R, 1));
            for (int r = 0; r < R; ++r) {
                const Rel &rl = rels[(size_t)r];
                tg_graph g{};
                if (rl.active) {
                    g.ptrs = rl.ptrs.data_ptr<int64_t>();
                    g.indices = rl.idx.numel() ? rl.idx.data_ptr<int64_t>() : nullptr;
                    g.timestamps = has_state ? rl.ts.data_ptr<int64_t>() : nullptr;
                    g.weights = weighted ? rl.w.data_ptr<double>() : nullptr;
                    g.n_major = rl.ptrs.numel() - 1;
                    g.n_edges = rl.idx.numel();
                }
                graphs[(size_t)r] = g;
            }
            tg_hop_filter flt{};
            flt.filter_mode = f.mode;
            flt.forward = f.forward ? 1 : 0;
            flt.win_lo = f.win_lo;
            flt.win_hi = f.win_hi;
            flt.states = has_state ? fst.data_ptr<int64_t>() : nullptr;
            tg_hop_out out{cnt.data_ptr<int64_t>(), offsets.data_ptr<int64_t>(), nbr.data_ptr<int64_t>(),
                           ep.data_ptr<int64_t>(), par.data_ptr<int64_t>()};
            for (int64_t group_mult = 1;; group_mult *= 8) { // a retry only when a column-group guess was too low
                auto groups_of_hop = [&](int h) {
                    return group_mult * std::max<int64_t>(1024, hop_groups[(size_t)h] + 2 * hop_m[(size_t)h] + 2);
                };
                int64_t ws_max = 0; // one workspace for every step of the call
                for (int h = 0; h < H; ++h) {
                    if (all_at_once) {
                        int64_t b = 0;
                        if (hop_m[(size_t)h])
                            check_rc(hop_workspace_bytes(weighted, hop_m[(size_t)h], (int32_t)hop_k[(size_t)h], groups_of_hop(h), &b));
                        ws_max = std::max(ws_max, b);
                        continue;
                    }
                    for (int r = 0; r < R; ++r) {
                        const Rel &rl = rels[(size_t)r];
                        const int64_t cf = cap_f[(size_t)h][(size_t)r];
                        if (!rl.active || cf == 0) continue;
                        int64_t b = 0;
                        check_rc(tg_ns_hop_workspace_bytes(cf, &b));
                        ws_max = std::max(ws_max, b);
                    }
                }
                ws = at::empty({ws_max / 8 + 1}, i64(dev));
                meta = meta_init.to(dev);
                status.zero_();
                for (int h = 0; h < H; ++h) {
                    if (all_at_once) {
                        std::vector<tg_het_entry> ent;
                        std::vector<tg_hop_segment> segs;
                        int64_t m_round = 0, o_round = 0;
                        bool any = false;
                        auto flush = [&](bool last) {
                            if (ent.empty()) return;
                            // the frontier without its padding whenever the single-workgroup scans apply (tchgeo.h)
                            int64_t *lay = (m_round <= ((int64_t)1 << 17) && groups_of_hop(h) <= ((int64_t)1 << 20))
                                               ? layout.data_ptr<int64_t>()
                                               : nullptr;
                            check_rc(tg_het_hop_begin_all(ent.data(), (int32_t)ent.size(), meta.data_ptr<int64_t>(), (int32_t)T, R,
                                                          H, m_round, F.data_ptr<int64_t>(),
                                                          has_state ? fst.data_ptr<int64_t>() : nullptr, ids.data_ptr<int64_t>(),
                                                          lay, stream_of(dev)));
                            if (!segs.empty()) {
                                tg_hop_in in{};
                                in.vertices = F.data_ptr<int64_t>();
                                in.ids = ids.data_ptr<int64_t>();
                                in.m = m_round;
                                in.fanout = (int32_t)hop_k[(size_t)h];
                                in.sampler = s.kind;
                                int64_t ws_bytes = 0;
                                check_rc(hop_workspace_bytes(weighted, m_round, (int32_t)hop_k[(size_t)h], groups_of_hop(h), &ws_bytes));
                                check_rc(tg_ns_hop_segments(segs.data(), (int32_t)segs.size(), &in, lay, &flt, &rng, &out,
                                                            st_out.data_ptr<int64_t>(), status.data_ptr<int32_t>(),
                                                            ws.data_ptr<int64_t>(), ws_bytes, groups_of_hop(h), stream_of(dev)));
                            }
                            check_rc(tg_het_hop_end_all(ent.data(), (int32_t)ent.size(), &out,
                                                        has_state ? st_out.data_ptr<int64_t>() : nullptr, m_round, o_round, lay,
                                                        meta.data_ptr<int64_t>(), (int32_t)T, R, H, h, last ? 1 : 0,
                                                        status.data_ptr<int32_t>(), stream_of(dev)));
                            ent.clear();
                            segs.clear();
                            m_round = o_round = 0;
                        };
                        int last_active = -1;
                        for (int r = 0; r < R; ++r)
                            if (rels[(size_t)r].active) last_active = r;
                        for (int r = 0; r < R; ++r) {
                            Rel &rl = rels[(size_t)r];
                            if (!rl.active) continue;
                            any = true;
                            const int64_t cf = cap_f[(size_t)h][(size_t)r];
                            if (ent.size() == TG_HET_HOP_MAX_ENTRIES || (cf > 0 && segs.size() == TG_HOP_MAX_SEGMENTS)) flush(false);
                            tg_het_entry e{};
                            e.rel = r;
                            e.src = (int32_t)rl.src;
                            e.dst = (int32_t)rl.dst;
                            e.segment = -1;
                            e.list_dst = lists[rl.dst].data_ptr<int64_t>();
                            e.state_dst = has_state ? st_lists[rl.dst].data_ptr<int64_t>() : nullptr;
                            e.list_src = lists[rl.src].data_ptr<int64_t>();
                            e.state_src = has_state ? st_lists[rl.src].data_ptr<int64_t>() : nullptr;
                            e.cap_list_src = lists[rl.src].numel();
                            e.rows = RW[(size_t)r].data_ptr<int64_t>();
                            e.cols = CL[(size_t)r].data_ptr<int64_t>();
                            e.edge_index = EI[(size_t)r].data_ptr<int64_t>();
                            e.cap_edges = RW[(size_t)r].numel();
                            if (cf > 0) {
                                e.segment = (int32_t)segs.size();
                                e.begin = m_round;
                                e.cap = cf;
                                tg_hop_segment sg{};
                                sg.graph = &graphs[(size_t)r];
                                sg.begin = m_round;
                                sg.fanout = (int32_t)rl.fanout[(size_t)h];
                                sg.rng_tag = TG_TAG_NS_HETERO | ((uint32_t)r << 8);
                                segs.push_back(sg);
                                m_round += cf;
                                o_round += cf * rl.fanout[(size_t)h];
                            }
                            ent.push_back(e);
                            if (r == last_active) flush(true);
                        }
                        if (!any) check_rc(tg_het_hop_end(meta.data_ptr<int64_t>(), (int32_t)T, R, H, stream_of(dev)));
                        continue;
                    }
                    for (int r = 0; r < R; ++r) {
                        Rel &rl = rels[(size_t)r];
                        const int64_t cf = cap_f[(size_t)h][(size_t)r];
                        if (!rl.active) continue;
                        if (cf == 0) { // nothing can be in the frontier; the layer offset is still recorded (:314)
                            check_rc(tg_het_step_begin(lists[rl.dst].data_ptr<int64_t>(), nullptr, meta.data_ptr<int64_t>(),
                                                       (int32_t)T, R, H, (int32_t)rl.src, (int32_t)rl.dst, r, h, 1,
                                                       F.data_ptr<int64_t>(), nullptr, ids.data_ptr<int64_t>(), stream_of(dev)));
                            continue;
                        }
                        const int64_t k = rl.fanout[(size_t)h];
                        check_rc(tg_het_step_begin(lists[rl.dst].data_ptr<int64_t>(), nullptr, meta.data_ptr<int64_t>(),
                                                   (int32_t)T, R, H, (int32_t)rl.src, (int32_t)rl.dst, r, h, cf,
                                                   F.data_ptr<int64_t>(), nullptr, ids.data_ptr<int64_t>(), stream_of(dev)));
                        tg_hop_in in{};
                        in.vertices = F.data_ptr<int64_t>();
                        in.ids = ids.data_ptr<int64_t>();
                        in.m = cf;
                        in.fanout = (int32_t)k;
                        in.sampler = s.kind;
                        in.rng_tag = TG_TAG_NS_HETERO | ((uint32_t)r << 8);
                        int64_t ws_bytes = 0;
                        check_rc(tg_ns_hop_workspace_bytes(cf, &ws_bytes));
                        check_rc(tg_ns_hop(&graphs[(size_t)r], &in, &rng, &out, ws.data_ptr<int64_t>(), ws_bytes, stream_of(dev)));
                        check_rc(tg_het_step_end(&out, nullptr, cf, (int32_t)k, meta.data_ptr<int64_t>(), (int32_t)T, R, H,
                                                 (int32_t)rl.src, r, lists[rl.src].data_ptr<int64_t>(), nullptr,
                                                 lists[rl.src].numel(), RW[(size_t)r].data_ptr<int64_t>(),
                                                 CL[(size_t)r].data_ptr<int64_t>(), EI[(size_t)r].data_ptr<int64_t>(),
                                                 RW[(size_t)r].numel(), status.data_ptr<int32_t>(), stream_of(dev)));
                    }
                    check_rc(tg_het_hop_end(meta.data_ptr<int64_t>(), (int32_t)T, R, H, stream_of(dev)));
                }
                // the call's only synchronisation: lengths, layer offsets and the status word in one array
                Tensor both = at::cat({meta, status.to(at::kLong), id_flag});
                mh = to_host(both);
                raise_if_flagged(mh.data_ptr<int64_t>()[meta_words + 1], "neighbor_sampling_heterogenous inputs");
                status_h = (int32_t)mh.data_ptr<int64_t>()[meta_words];
                if ((status_h & 1) && group_mult < 4096) continue;
                break;
            }
            nogil.reset();
            if (status_h & 1) throw std::runtime_error("neighbor_sampling_heterogenous: column-group workspace overflow");
            if (status_h & 2) // sampling.rs:49: gen_range over an empty float range panics in the reference
                throw PanicError("weighted sampling met a non-positive running weight sum (the reference panics here)");
            if (status_h & 4) throw std::runtime_error("neighbor_sampling_heterogenous: internal capacity error");
            const int64_t *m = mh.data_ptr<int64_t>();
            py::dict samples, rows, cols, eidx, los;
            for (size_t t = 0; t < T; ++t) samples[py::str(node_types[t])] = back(lists[t].narrow(0, 0, m[t]), out_dev);
            for (int r = 0; r < R; ++r) {
                const Rel &rl = rels[(size_t)r];
                const int64_t ne = m[3 * T + (size_t)r];
                rows[py::str(rl.key)] = back(RW[(size_t)r].narrow(0, 0, ne), out_dev);
                cols[py::str(rl.key)] = back(CL[(size_t)r].narrow(0, 0, ne), out_dev);
                eidx[py::str(rl.key)] = back(EI[(size_t)r].narrow(0, 0, ne), out_dev);
                std::vector<std::tuple<int64_t, int64_t, int64_t>> lo;
                for (int h = 0; h < H && rl.active; ++h) {
                    const int64_t *q = m + 3 * T + R + ((size_t)r * H + (size_t)h) * 3;
                    lo.emplace_back(q[0], q[1], q[2]);
                }
                los[py::str(rl.key)] = lo;
            }
            return py::make_tuple(samples, rows, cols, eidx, los);
        }
    }

    verify_inputs_now();
    for (int64_t ell = 0; ell < num_hops; ++ell) { // :292
        for (size_t t = 0; t < T; ++t) {
            new_chunks[t].clear();
            new_st[t].clear();
        }
        for (size_t ri = 0; ri < rels.size(); ++ri) { // :294, canonical order
            Rel &r = rels[ri];
            if (!r.active) continue;
            r.layer_offsets.emplace_back(len[r.src], r.n_edges, len[r.dst]); // :314-315
            const Tensor &F = frontier[r.dst];
            if (F.numel() == 0) continue;
            NsResult o = run_ns(dev, r.ptrs, r.idx, r.w, r.ts, F, has_state ? frontier_st[r.dst] : Tensor(),
                                {r.fanout[(size_t)ell]}, s, f, rng, TG_TAG_NS_HETERO | ((uint32_t)ri << 8),
                                fbegin[r.dst]);
            const int64_t cnt = o.n_edges, nF = F.numel();
            if (cnt == 0) continue;
            Tensor fresh = o.samples.narrow(0, nF, cnt); // :338
            r.rows.push_back(at::arange(len[r.src], len[r.src] + cnt, i64(dev))); // :335,340 j
            r.cols.push_back(o.cols.narrow(0, 0, cnt) + fbegin[r.dst]);            // :340 i
            r.eidx.push_back(o.edge_index.narrow(0, 0, cnt));
            r.n_edges += cnt;
            chunks[r.src].push_back(fresh);
            new_chunks[r.src].push_back(fresh);
            if (has_state) {
                Tensor fs = o.states.narrow(0, nF, cnt); // :339
                st_chunks[r.src].push_back(fs);
                new_st[r.src].push_back(fs);
            }
            len[r.src] += cnt;
        }
        for (size_t t = 0; t < T; ++t) { // :345-348
            fbegin[t] += frontier[t].numel();
            frontier[t] = new_chunks[t].empty() ? at::empty({0}, i64(dev)) : at::cat(new_chunks[t]);
            if (has_state) frontier_st[t] = new_st[t].empty() ? at::empty({0}, i64(dev)) : at::cat(new_st[t]);
        }
    }

    py::dict samples, rows, cols, eidx, los;
    for (size_t t = 0; t < T; ++t) samples[py::str(node_types[t])] = back(at::cat(chunks[t]), out_dev);
    auto cat_or_empty = [&](const std::vector<Tensor> &v) { return v.empty() ? at::empty({0}, i64(dev)) : at::cat(v); };
    for (Rel &r : rels) {
        rows[py::str(r.key)] = back(cat_or_empty(r.rows), out_dev);
        cols[py::str(r.key)] = back(cat_or_empty(r.cols), out_dev);
        eidx[py::str(r.key)] = back(cat_or_empty(r.eidx), out_dev);
        los[py::str(r.key)] = r.layer_offsets;
    }
    return py::make_tuple(samples, rows, cols, eidx, los);
}

// ---------------------------------------------------------------- python.rs:584-608
Tensor random_walk(const Tensor &row_ptrs, const Tensor &col_indices, const Tensor &start, int64_t walk_length, float p,
                   float q) {
    const c10::Device dev = compute_device({&row_ptrs, &col_indices, &start});
    DeviceGuard guard(dev);
    Tensor ptrs = on_graph(row_ptrs, dev, at::kLong), idx = on_graph(col_indices, dev, at::kLong);
    Tensor st = on(start, dev, at::kLong).reshape({-1});
    if (walk_length < 0) throw py::value_error("walk_length must be >= 0");
    RangeCheck rc(dev);
    rc.add(st, ptrs.numel() - 1);
    rc.verify("random_walk start");
    if (walk_length > 1) check_graph_ids(idx, ptrs.numel() - 1, dev, "random_walk col_indices");
    Tensor walks = at::empty({st.numel(), walk_length + 1}, i64(dev));
    tg_graph g{};
    g.ptrs = ptrs.data_ptr<int64_t>();
    g.indices = idx.data_ptr<int64_t>();
    g.n_major = ptrs.numel() - 1;
    g.n_edges = idx.numel();
    const tg_rng rng = next_rng();
    // p != q: every proposal asks has_edge (random_walk.rs:57-63) -- from the graph's edge set once a large call paid for it
    Tensor edge_set;
    if (p > 0.0f && q > 0.0f && !(p == 1.0f && q == 1.0f) && walk_length > 1 && st.numel() > 0)
        edge_set = EdgeSets::instance().get(ptrs, idx, g, dev, st.numel() * walk_length >= ((int64_t)1 << 20));
    check_rc(tg_random_walk_es(&g, edge_set.defined() ? edge_set.data_ptr<int64_t>() : nullptr,
                               edge_set.defined() ? (int64_t)edge_set.nbytes() : 0, st.numel() ? st.data_ptr<int64_t>() : nullptr,
                               st.numel(), walk_length, p, q, &rng, walks.data_ptr<int64_t>(), stream_of(dev)));
    return back(walks, start.device()); // random_walk.rs:19-23 allocates on start.device()
}

// ---------------------------------------------------------------- python.rs:611-642
std::tuple<Tensor, Tensor> tempo_random_walk(const Tensor &row_ptrs, const Tensor &col_indices,
                                             const Tensor &node_timestamps, const Tensor &edge_timestamps,
                                             const Tensor &start, const Tensor &start_timestamps, int64_t walk_length,
                                             std::pair<int64_t, int64_t> window) {
    const c10::Device dev = compute_device({&row_ptrs, &col_indices, &start});
    DeviceGuard guard(dev);
    Tensor ptrs = on_graph(row_ptrs, dev, at::kLong), idx = on_graph(col_indices, dev, at::kLong);
    Tensor nts = on_graph(node_timestamps, dev, at::kLong), ets = on_graph(edge_timestamps, dev, at::kLong);
    Tensor st = on(start, dev, at::kLong).reshape({-1}), sts = on(start_timestamps, dev, at::kLong).reshape({-1});
    if (walk_length < 0) throw py::value_error("walk_length must be >= 0");
    if (sts.numel() != st.numel()) throw py::value_error("start_timestamps must have one entry per start node");
    if (nts.numel() < ptrs.numel() - 1) throw py::value_error("node_timestamps must have one entry per node");
    if (ets.numel() != idx.numel()) throw py::value_error("edge_timestamps must have one entry per edge");
    RangeCheck rc(dev);
    rc.add(st, ptrs.numel() - 1);
    rc.verify("tempo_random_walk start");
    if (walk_length > 1) check_graph_ids(idx, ptrs.numel() - 1, dev, "tempo_random_walk col_indices");
    Tensor walks = at::full({st.numel(), walk_length}, -1, i64(dev));
    Tensor wts = at::full({st.numel(), walk_length}, -1, i64(dev));
    tg_graph g{};
    g.ptrs = ptrs.data_ptr<int64_t>();
    g.indices = idx.data_ptr<int64_t>();
    g.n_major = ptrs.numel() - 1;
    g.n_edges = idx.numel();
    const tg_rng rng = next_rng();
    check_rc(tg_tempo_random_walk(&g, nts.data_ptr<int64_t>(), ets.numel() ? ets.data_ptr<int64_t>() : nullptr,
                                  st.numel() ? st.data_ptr<int64_t>() : nullptr,
                                  sts.numel() ? sts.data_ptr<int64_t>() : nullptr, st.numel(), walk_length,
                                  window.first, window.second, &rng, walks.data_ptr<int64_t>(),
                                  wts.data_ptr<int64_t>(), stream_of(dev)));
    return {back(walks, start.device()), back(wts, start.device())};
}

// python.rs:645-687
std::tuple<Tensor, Tensor> biased_tempo_random_walk(const Tensor &row_ptrs, const Tensor &col_indices,
                                                    const Tensor &node_timestamps, const Tensor &edge_timestamps,
                                                    const Tensor &start, const Tensor &start_timestamps,
                                                    int64_t walk_length, const std::string &bias_type, bool forward,
                                                    int64_t retry_count) {
    int32_t bias;
    if (bias_type == "uniform") bias = TG_BIAS_UNIFORM;
    else if (bias_type == "linear") bias = TG_BIAS_LINEAR;
    else if (bias_type == "exponential") bias = TG_BIAS_EXPONENTIAL;
    else throw py::value_error("Unknown bias type: " + bias_type); // python.rs:666-671
    const c10::Device dev = compute_device({&row_ptrs, &col_indices, &start});
    DeviceGuard guard(dev);
    Tensor ptrs = on_graph(row_ptrs, dev, at::kLong), idx = on_graph(col_indices, dev, at::kLong);
    Tensor nts = on_graph(node_timestamps, dev, at::kLong), ets = on_graph(edge_timestamps, dev, at::kLong);
    Tensor st = on(start, dev, at::kLong).reshape({-1}), sts = on(start_timestamps, dev, at::kLong).reshape({-1});
    if (walk_length < 1) throw py::value_error("walk_length must be >= 1");
    if (retry_count < 0) throw py::value_error("retry_count must be >= 0");
    if (sts.numel() != st.numel()) throw py::value_error("start_timestamps must have one entry per start node");
    if (nts.numel() < ptrs.numel() - 1) throw py::value_error("node_timestamps must have one entry per node");
    if (ets.numel() != idx.numel()) throw py::value_error("edge_timestamps must have one entry per edge");
    RangeCheck rc(dev);
    rc.add(st, ptrs.numel() - 1);
    rc.verify("biased_tempo_random_walk start");
    if (walk_length > 1) check_graph_ids(idx, ptrs.numel() - 1, dev, "biased_tempo_random_walk col_indices");
    Tensor walks = at::empty({st.numel(), walk_length}, i64(dev));
    Tensor wts = at::empty({st.numel(), walk_length}, i64(dev));
    if (st.numel() == 0) return {back(walks, start.device()), back(wts, start.device())};
    // the linear bias sorts a row's candidates: rows above the LDS capacity need a slab sized by the largest row
    int64_t max_degree = 0;
    if (bias == TG_BIAS_LINEAR && ptrs.numel() > 1)
        max_degree = read_scalar<int64_t>((ptrs.slice(0, 1) - ptrs.slice(0, 0, ptrs.numel() - 1)).max());
    int64_t ws_bytes = 0;
    check_rc(tg_biased_walk_workspace_bytes(st.numel(), max_degree, bias, &ws_bytes));
    Tensor ws = at::empty({ws_bytes / 8 + 1}, i64(dev));
    Tensor status = at::zeros({1}, at::TensorOptions().dtype(at::kInt).device(dev));
    tg_graph g{};
    g.ptrs = ptrs.data_ptr<int64_t>();
    g.indices = idx.numel() ? idx.data_ptr<int64_t>() : nullptr;
    g.n_major = ptrs.numel() - 1;
    g.n_edges = idx.numel();
    const tg_rng rng = next_rng();
    check_rc(tg_biased_tempo_random_walk(&g, nts.data_ptr<int64_t>(), ets.numel() ? ets.data_ptr<int64_t>() : nullptr,
                                         st.data_ptr<int64_t>(), sts.data_ptr<int64_t>(), st.numel(), walk_length, bias,
                                         forward ? 1 : 0, retry_count, max_degree, &rng, walks.data_ptr<int64_t>(),
                                         wts.data_ptr<int64_t>(), status.data_ptr<int32_t>(), ws.data_ptr<int64_t>(),
                                         ws_bytes, stream_of(dev)));
    const int32_t flags = read_scalar<int32_t>(status);
    if (flags & 2)
        throw PanicError("cannot sample empty range: every bias weight so far underflowed to zero (the "
                                 "reference panics here, utils/sampling.rs:49)");
    if (flags & 1) throw std::runtime_error("biased_tempo_random_walk: sort slab too small for a CSR row");
    return {back(walks, start.device()), back(wts, start.device())};
}

} // namespace

void register_more(py::module_ &m); // negative sampling + hgt (python_module_more.cpp)

// The loader's mini-batches are views of a super-batch's flat tensors (tch_geometric/loader.py).  Building the views of one
// mini-batch in Python is a narrow() per tensor, ~2 us each -- more than the GPU spends sampling the mini-batch (~0.5 us in
// a 16 384-batch launch); this object holds the super-batch's tensors once and cuts all views of mini-batch j in ONE call.
struct BatchViews {
    std::vector<Tensor> bases;
    std::vector<int64_t> dims;  // the dimension the mini-batches are laid out along
    std::vector<int64_t> kinds; // 0: a node tensor (sliced by node_ptr), 1: an edge tensor (edge_ptr)
    std::vector<int64_t> node_ptr, edge_ptr;
    BatchViews(std::vector<Tensor> b, std::vector<int64_t> d, std::vector<int64_t> k, std::vector<int64_t> np,
               std::vector<int64_t> ep)
        : bases(std::move(b)), dims(std::move(d)), kinds(std::move(k)), node_ptr(std::move(np)), edge_ptr(std::move(ep)) {
        if (bases.size() != dims.size() || bases.size() != kinds.size() || node_ptr.size() != edge_ptr.size() ||
            node_ptr.empty())
            throw py::value_error("BatchViews: tensors, dims and kinds must have one length; node_ptr and edge_ptr too");
    }
    py::tuple at(int64_t j) const {
        if (j < 0 || j + 1 >= (int64_t)node_ptr.size()) throw py::index_error("mini-batch index out of range");
        const int64_t an = node_ptr[(size_t)j], ln = node_ptr[(size_t)j + 1] - an;
        const int64_t ae = edge_ptr[(size_t)j], le = edge_ptr[(size_t)j + 1] - ae;
        py::tuple out(bases.size());
        for (size_t i = 0; i < bases.size(); ++i)
            out[i] = kinds[i] == 0 ? bases[i].narrow(dims[i], an, ln) : bases[i].narrow(dims[i], ae, le);
        return out;
    }
};

PYBIND11_MODULE(tch_geometric, m) {
    m.doc() = "MI355X-native backend behind tch-geometric's operator surface (reference: src/python.rs)";
    // additive: the reference's RNG cannot be seeded from Python (utils/random.rs:14-17 is not exported)
    py::register_exception<PanicError>(m, "PanicException", PyExc_RuntimeError);
    m.def("seed", [](uint64_t s) {
        RngState &st = rng_state();
        std::lock_guard<std::mutex> lk(st.mu);
        st.seed = s;
        st.call = 0;
    }, py::arg("seed"), "Seed the global (seed, call counter) state; every operator call consumes one call id.");
    // additive: CPU-resident adjacency tensors are uploaded once and kept on the device (host_common.h ResidentGraphs)
    m.def("graph_cache_info", [] {
        ResidentGraphs &g = ResidentGraphs::instance();
        std::lock_guard<std::mutex> lk(g.mu);
        py::dict d;
        d["entries"] = g.entries.size();
        d["bytes"] = g.bytes_locked();
        d["limit_bytes"] = g.limit_bytes();
        d["hits"] = g.hits;
        d["uploads"] = g.uploads;
        {
            EdgeSets &es = EdgeSets::instance();
            std::lock_guard<std::mutex> lk2(es.mu);
            d["edge_sets"] = es.entries.size();
            d["edge_set_bytes"] = es.bytes_locked();
            d["edge_set_limit_bytes"] = EdgeSets::limit_bytes();
            d["edge_set_hits"] = es.hits;
            d["edge_set_builds"] = es.builds;
        }
        return d;
    }, "Device copies kept of CPU-resident adjacency tensors: {entries, bytes, limit_bytes, hits, uploads}, and the edge sets "
       "kept for node2vec walks with p != q: {edge_sets, edge_set_bytes, edge_set_limit_bytes, edge_set_hits, edge_set_builds}.");
    m.def("graph_cache_clear", [] {
        {
            ResidentGraphs &g = ResidentGraphs::instance();
            std::lock_guard<std::mutex> lk(g.mu);
            g.entries.clear();
        }
        EdgeSets &es = EdgeSets::instance();
        std::lock_guard<std::mutex> lk2(es.mu);
        es.entries.clear();
    }, "Drop every device copy of a CPU-resident adjacency tensor and every edge set.");
    py::class_<BatchViews>(m, "BatchViews", "All tensor views of mini-batch j of a loader super-batch in one call.")
        .def(py::init<std::vector<Tensor>, std::vector<int64_t>, std::vector<int64_t>, std::vector<int64_t>,
                      std::vector<int64_t>>(),
             py::arg("tensors"), py::arg("dims"), py::arg("kinds"), py::arg("node_ptr"), py::arg("edge_ptr"))
        .def("at", &BatchViews::at, py::arg("j"));
    m.def("rng_state", [] {
        RngState &st = rng_state();
        std::lock_guard<std::mutex> lk(st.mu);
        return std::make_pair(st.seed, st.call);
    });
    m.def("set_rng_state", [](uint64_t s, uint64_t c) {
        RngState &st = rng_state();
        std::lock_guard<std::mutex> lk(st.mu);
        st.seed = s;
        st.call = c;
    });
    m.def("backend_version", [] { return std::string(tg_version()); });

    m.def("to_csc", [](const Tensor &rc, const py::object &size) { return to_csx(rc, size, true); }, py::arg("row_col"),
          py::arg("size"));
    m.def("to_csr", [](const Tensor &rc, const py::object &size) { return to_csx(rc, size, false); },
          py::arg("row_col"), py::arg("size"));
    m.def("neighbor_sampling_homogenous", &neighbor_sampling_homogenous, py::arg("col_ptrs"), py::arg("row_indices"),
          py::arg("inputs"), py::arg("num_neighbors"), py::arg("sampler") = py::none(), py::arg("filter") = py::none());
    m.def("neighbor_sampling_heterogenous", &neighbor_sampling_heterogenous, py::arg("node_types"),
          py::arg("edge_types"), py::arg("col_ptrs"), py::arg("row_indices"), py::arg("inputs"),
          py::arg("num_neighbors"), py::arg("num_hops"), py::arg("sampler") = py::none(),
          py::arg("filter") = py::none());
    // (tensor-only signatures: the whole body runs without the GIL)
    m.def("random_walk", &random_walk, py::arg("row_ptrs"), py::arg("col_indices"), py::arg("start"),
          py::arg("walk_length"), py::arg("p"), py::arg("q"), py::call_guard<py::gil_scoped_release>());
    m.def("tempo_random_walk", &tempo_random_walk, py::arg("row_ptrs"), py::arg("col_indices"),
          py::arg("node_timestamps"), py::arg("edge_timestamps"), py::arg("start"), py::arg("start_timestamps"),
          py::arg("walk_length"), py::arg("window"), py::call_guard<py::gil_scoped_release>());
    m.def("biased_tempo_random_walk", &biased_tempo_random_walk, py::arg("row_ptrs"), py::arg("col_indices"),
          py::arg("node_timestamps"), py::arg("edge_timestamps"), py::arg("start"), py::arg("start_timestamps"),
          py::arg("walk_length"), py::arg("bias_type"), py::arg("forward"), py::arg("retry_count"),
          py::call_guard<py::gil_scoped_release>());
    register_more(m);
}
