// Third translation unit of the host module: hgt_sampling (python.rs:399-482).
#include <optional>

#include "host_common.h"

using namespace tghost;

namespace {

py::tuple hgt_sampling(const std::vector<std::string> &node_types,
                       const std::vector<std::tuple<std::string, std::string, std::string>> &edge_types,
                       const py::dict &col_ptrs, const py::dict &row_indices, const py::object &row_timestamps,
                       const py::dict &inputs, const py::object &input_timestamps, const py::dict &num_samples,
                       int64_t num_hops, const py::object &timerange) {
    const int T = (int)node_types.size();
    std::map<std::string, int> tix;
    for (int t = 0; t < T; ++t) tix[node_types[(size_t)t]] = t;
    Tensor first;
    for (auto item : col_ptrs) {
        first = item.second.cast<Tensor>();
        break;
    }
    const c10::Device dev = compute_device({&first});
    DeviceGuard guard(dev);
    if (num_hops < 0) throw py::value_error("num_hops must be >= 0");
    const int H = (int)num_hops;

    // relations in `edge_types` order; graphs are keyed by col_ptrs (python.rs:418-435)
    std::vector<std::string> keys;
    std::vector<int32_t> rel_src, rel_dst;
    std::vector<Tensor> ptrs, idx, rts;
    std::vector<tg_graph> graphs;
    py::dict rts_dict = row_timestamps.is_none() ? py::dict() : row_timestamps.cast<py::dict>();
    for (const auto &et : edge_types) {
        const std::string key = rel_key(et);
        if (!col_ptrs.contains(py::str(key))) continue;
        keys.push_back(key);
        rel_src.push_back(tix.at(std::get<0>(et)));
        rel_dst.push_back(tix.at(std::get<2>(et)));
        ptrs.push_back(on_graph(col_ptrs[py::str(key)].cast<Tensor>(), dev, at::kLong));
        idx.push_back(on_graph(row_indices[py::str(key)].cast<Tensor>(), dev, at::kLong));
        rts.push_back(rts_dict.contains(py::str(key)) ? on(rts_dict[py::str(key)].cast<Tensor>(), dev, at::kLong)
                                                      : Tensor());
    }
    const int R = (int)keys.size();
    for (int r = 0; r < R; ++r) {
        tg_graph g{};
        g.ptrs = ptrs[(size_t)r].data_ptr<int64_t>();
        g.indices = idx[(size_t)r].numel() ? idx[(size_t)r].data_ptr<int64_t>() : nullptr;
        g.timestamps = rts[(size_t)r].defined() ? rts[(size_t)r].data_ptr<int64_t>() : nullptr;
        g.n_major = ptrs[(size_t)r].numel() - 1;
        g.n_edges = idx[(size_t)r].numel();
        graphs.push_back(g);
    }
    std::vector<Tensor> in((size_t)T), in_ts((size_t)T);
    std::vector<const int64_t *> in_ptr((size_t)T, nullptr), its_ptr((size_t)T, nullptr);
    std::vector<int64_t> n_in((size_t)T, -1), ns((size_t)T * (size_t)std::max(H, 1), -1);
    const bool has_its = !input_timestamps.is_none();
    py::dict its_dict = has_its ? input_timestamps.cast<py::dict>() : py::dict();
    c10::Device out_dev = dev;
    bool out_dev_set = false;
    for (int t = 0; t < T; ++t) {
        const std::string &name = node_types[(size_t)t];
        if (inputs.contains(py::str(name))) {
            Tensor x = inputs[py::str(name)].cast<Tensor>();
            if (!out_dev_set) {
                out_dev = x.device();
                out_dev_set = true;
            }
            in[(size_t)t] = on(x, dev, at::kLong).reshape({-1});
            n_in[(size_t)t] = in[(size_t)t].numel();
            in_ptr[(size_t)t] = n_in[(size_t)t] ? in[(size_t)t].data_ptr<int64_t>() : nullptr;
            if (has_its) { // :172 `ts.get(node_type).unwrap()`
                if (!its_dict.contains(py::str(name))) throw py::key_error(name);
                in_ts[(size_t)t] = on(its_dict[py::str(name)].cast<Tensor>(), dev, at::kLong).reshape({-1});
                if (in_ts[(size_t)t].numel() != n_in[(size_t)t])
                    throw py::value_error("input_timestamps[" + name + "] must have one entry per input");
                its_ptr[(size_t)t] = n_in[(size_t)t] ? in_ts[(size_t)t].data_ptr<int64_t>() : nullptr;
            }
        }
        if (num_samples.contains(py::str(name))) {
            auto v = num_samples[py::str(name)].cast<std::vector<int64_t>>();
            if ((int64_t)v.size() < num_hops) throw py::index_error("num_samples[" + name + "] is shorter than num_hops");
            for (int l = 0; l < H; ++l) {
                if (v[(size_t)l] < 0) throw py::value_error("num_samples entries must be >= 0");
                ns[(size_t)t * H + l] = v[(size_t)l];
            }
        }
    }
    std::pair<int64_t, int64_t> tr{0, 0};
    const bool has_tr = !timerange.is_none();
    if (has_tr) tr = timerange.cast<std::pair<int64_t, int64_t>>();
    // ---- from here to the result dicts nothing touches Python: the GIL is released, so worker threads (each on its own
    // HIP stream) overlap their calls
    std::optional<NoGil> nogil;
    nogil.emplace();
    {
        RangeCheck rc(dev); // an input of type t indexes the columns of every relation whose dst is t
        for (int r = 0; r < R; ++r)
            if (n_in[(size_t)rel_dst[(size_t)r]] > 0) rc.add(in[(size_t)rel_dst[(size_t)r]], ptrs[(size_t)r].numel() - 1);
        rc.verify("hgt_sampling inputs");
    }
    tg_hgt_problem pb{};
    pb.n_types = T;
    pb.n_rels = R;
    pb.n_hops = H;
    pb.rel_src = rel_src.data();
    pb.rel_dst = rel_dst.data();
    pb.graphs = graphs.data();
    pb.inputs = in_ptr.data();
    pb.input_ts = has_its ? its_ptr.data() : nullptr;
    pb.n_inputs = n_in.data();
    pb.num_samples = ns.data();
    if (has_tr) {
        pb.has_timerange = 1;
        pb.tr_lo = tr.first;
        pb.tr_hi = tr.second;
    }
    // outputs
    std::vector<Tensor> samples, sample_ts, rows, cols, eidx;
    std::vector<int64_t *> sp, tp, rp, cp, ep;
    std::vector<int64_t> cap_nodes((size_t)T, 0);
    for (int t = 0; t < T; ++t) {
        int64_t cap = std::max<int64_t>(n_in[(size_t)t], 0);
        for (int l = 0; l < H; ++l) cap += std::max<int64_t>(ns[(size_t)t * H + l], 0);
        cap_nodes[(size_t)t] = cap;
        samples.push_back(at::empty({cap + 1}, i64(dev)));
        sample_ts.push_back(at::empty({cap + 1}, i64(dev)));
        sp.push_back(samples.back().data_ptr<int64_t>());
        tp.push_back(sample_ts.back().data_ptr<int64_t>());
    }
    for (int r = 0; r < R; ++r) {
        const int64_t cap = 50 * std::max<int64_t>(cap_nodes[(size_t)rel_dst[(size_t)r]], 1) + 1;
        rows.push_back(at::empty({cap}, i64(dev)));
        cols.push_back(at::empty({cap}, i64(dev)));
        eidx.push_back(at::empty({cap}, i64(dev)));
        rp.push_back(rows.back().data_ptr<int64_t>());
        cp.push_back(cols.back().data_ptr<int64_t>());
        ep.push_back(eidx.back().data_ptr<int64_t>());
    }
    Tensor counts = at::zeros({T + R + 1}, i64(dev));
    tg_hgt_out out{};
    out.samples = sp.data();
    out.sample_ts = tp.data();
    out.rows = rp.data();
    out.cols = cp.data();
    out.edge_index = ep.data();
    out.n_samples = counts.data_ptr<int64_t>();
    out.n_edges = counts.data_ptr<int64_t>() + T;
    out.panic = reinterpret_cast<int32_t *>(counts.data_ptr<int64_t>() + T + R);
    int64_t ws_bytes = 0;
    check_rc(tg_hgt_workspace_bytes(&pb, &ws_bytes));
    Tensor ws = at::empty({ws_bytes / 8 + 2}, i64(dev));
    const tg_rng rng = next_rng();
    check_rc(tg_hgt_sample(&pb, &rng, &out, ws.data_ptr<int64_t>(), ws_bytes, stream_of(dev)));
    Tensor c = to_host(counts); // the call's only synchronisation
    nogil.reset();
    if ((c[T + R].item<int64_t>() & 0xffffffff) != 0)
        throw PanicError("hgt_sampling: num_samples has no entry for a node type that owns a budget, or a "
                                 "weight sum was not positive (the reference panics here)");
    py::dict d_samples, d_ts, d_rows, d_cols, d_eidx;
    for (int t = 0; t < T; ++t) {
        const int64_t n = c[t].item<int64_t>();
        // nodes_dict only has the types that were inputs or own a budget; others come back empty
        d_samples[py::str(node_types[(size_t)t])] = back(samples[(size_t)t].narrow(0, 0, n), out_dev);
        d_ts[py::str(node_types[(size_t)t])] = back(sample_ts[(size_t)t].narrow(0, 0, n), out_dev);
    }
    for (int r = 0; r < R; ++r) {
        const int64_t n = c[T + r].item<int64_t>();
        d_rows[py::str(keys[(size_t)r])] = back(rows[(size_t)r].narrow(0, 0, n), out_dev);
        d_cols[py::str(keys[(size_t)r])] = back(cols[(size_t)r].narrow(0, 0, n), out_dev);
        d_eidx[py::str(keys[(size_t)r])] = back(eidx[(size_t)r].narrow(0, 0, n), out_dev);
    }
    return py::make_tuple(d_samples, d_ts, d_rows, d_cols, d_eidx);
}

} // namespace

void register_hgt(py::module_ &m) {
    m.def("hgt_sampling", &hgt_sampling, py::arg("node_types"), py::arg("edge_types"), py::arg("col_ptrs"),
          py::arg("row_indices"), py::arg("row_timestamps"), py::arg("inputs"), py::arg("input_timestamps"),
          py::arg("num_samples"), py::arg("num_hops"), py::arg("timerange") = py::none());
}
