// Second translation unit of the host module: negative sampling (python.rs:690-783) and HGT sampling
// (python.rs:399-482) bindings.
#include "host_common.h"

using namespace tghost;

namespace {

struct NegRun {
    std::vector<Tensor> samples, rows, cols;
    std::vector<int64_t> n_samples, n_edges;
};

// runs tg_neg_sample for relations (rel_src, rel_dst over T node types) and trims the outputs
NegRun run_neg(const c10::Device &dev, int T, const std::vector<int32_t> &rel_src, const std::vector<int32_t> &rel_dst,
               const std::vector<Tensor> &ptrs, const std::vector<Tensor> &indices, const std::vector<int64_t> &node_count,
               const std::vector<Tensor> &inputs, const std::vector<int64_t> &n_inputs, int64_t num_neg,
               int64_t try_count, bool inbound, bool homogeneous) {
    const int R = (int)rel_src.size();
    if (num_neg < 0 || try_count < 0) throw py::value_error("num_neg and try_count must be >= 0");
    std::vector<tg_graph> graphs((size_t)R);
    for (int r = 0; r < R; ++r) {
        graphs[r] = tg_graph{};
        graphs[r].ptrs = ptrs[r].data_ptr<int64_t>();
        graphs[r].indices = indices[r].numel() ? indices[r].data_ptr<int64_t>() : nullptr;
        graphs[r].n_major = ptrs[r].numel() - 1;
        graphs[r].n_edges = indices[r].numel();
    }
    std::vector<const int64_t *> in_ptrs((size_t)T, nullptr);
    int64_t m = 0;
    for (int t = 0; t < T; ++t) {
        if (n_inputs[t] > 0) {
            in_ptrs[t] = inputs[t].data_ptr<int64_t>();
            m += n_inputs[t] * num_neg;
        }
    }
    tg_neg_problem pb{};
    pb.n_types = T;
    pb.n_rels = R;
    pb.homogeneous = homogeneous ? 1 : 0;
    pb.inbound = inbound ? 1 : 0;
    pb.rel_src = rel_src.data();
    pb.rel_dst = rel_dst.data();
    pb.graphs = graphs.data();
    pb.node_count = node_count.data();
    pb.inputs = in_ptrs.data();
    pb.n_inputs = n_inputs.data();
    pb.num_neg = num_neg;
    pb.try_count = try_count;

    NegRun run;
    std::vector<int64_t *> s_ptrs((size_t)T), r_ptrs((size_t)R), c_ptrs((size_t)R);
    for (int t = 0; t < T; ++t) {
        run.samples.push_back(at::empty({std::max<int64_t>(n_inputs[t], 0) + m + 1}, i64(dev)));
        s_ptrs[t] = run.samples[t].data_ptr<int64_t>();
    }
    for (int r = 0; r < R; ++r) {
        const int64_t cap = std::max<int64_t>(n_inputs[rel_src[r]], 0) * num_neg + 1;
        run.rows.push_back(at::empty({cap}, i64(dev)));
        run.cols.push_back(at::empty({cap}, i64(dev)));
        r_ptrs[r] = run.rows[r].data_ptr<int64_t>();
        c_ptrs[r] = run.cols[r].data_ptr<int64_t>();
    }
    Tensor counts = at::zeros({T + R + 1}, i64(dev)); // n_samples | n_edges | panic (int32 in the last word)
    int64_t ws_bytes = 0;
    check_rc(tg_neg_workspace_bytes(&pb, &ws_bytes));
    Tensor ws = at::empty({ws_bytes / 8 + 1}, i64(dev));
    tg_neg_out out{};
    out.samples = s_ptrs.data();
    out.rows = r_ptrs.data();
    out.cols = c_ptrs.data();
    out.n_samples = counts.data_ptr<int64_t>();
    out.n_edges = counts.data_ptr<int64_t>() + T;
    out.panic = reinterpret_cast<int32_t *>(counts.data_ptr<int64_t>() + T + R);
    const tg_rng rng = next_rng();
    check_rc(tg_neg_sample(&pb, &rng, &out, ws.data_ptr<int64_t>(), stream_of(dev)));
    Tensor c = to_host(counts); // the call's only synchronisation
    if ((c[T + R].item<int64_t>() & 0xffffffff) != 0)
        throw PanicError("inbound negative sampling indexed a CSR row out of range (the reference panics here, "
                                 "negative_sampling.rs:113)");
    for (int t = 0; t < T; ++t) run.n_samples.push_back(c[t].item<int64_t>());
    for (int r = 0; r < R; ++r) run.n_edges.push_back(c[T + r].item<int64_t>());
    return run;
}

// python.rs:690-720
py::tuple negative_sample_neighbors_homogenous(const Tensor &row_ptrs, const Tensor &col_indices,
                                               std::pair<int64_t, int64_t> graph_size, const Tensor &inputs,
                                               int64_t num_neg, int64_t try_count) {
    const c10::Device dev = compute_device({&row_ptrs, &col_indices, &inputs});
    DeviceGuard guard(dev);
    Tensor ptrs = on_graph(row_ptrs, dev, at::kLong), idx = on_graph(col_indices, dev, at::kLong);
    Tensor in = on(inputs, dev, at::kLong).reshape({-1});
    if (graph_size.second < 1) throw py::value_error("graph_size[1] must be >= 1 (the reference panics on an empty range)");
    NegRun r;
    {
        NoGil nogil; // nothing in here touches Python
        RangeCheck rc(dev);
        rc.add(in, ptrs.numel() - 1);
        rc.verify("negative_sample_neighbors_homogenous inputs");
        r = run_neg(dev, 1, {0}, {0}, {ptrs}, {idx}, {graph_size.second}, {in}, {in.numel()}, num_neg, try_count, false,
                    true);
    }
    const c10::Device out_dev = inputs.device();
    return py::make_tuple(back(r.samples[0].narrow(0, 0, r.n_samples[0]), out_dev),
                          back(r.rows[0].narrow(0, 0, r.n_edges[0]), out_dev),
                          back(r.cols[0].narrow(0, 0, r.n_edges[0]), out_dev), in.numel());
}

// python.rs:724-783
py::tuple negative_sample_neighbors_heterogenous(const std::vector<std::string> &node_types,
                                                 const std::vector<std::tuple<std::string, std::string, std::string>> &edge_types,
                                                 const py::dict &row_ptrs, const py::dict &col_indices,
                                                 const py::dict &sizes, const py::dict &inputs, int64_t num_neg,
                                                 int64_t try_count, bool inbound) {
    const int T = (int)node_types.size();
    std::map<std::string, int> tix;
    for (int t = 0; t < T; ++t) tix[node_types[(size_t)t]] = t;
    Tensor first;
    for (auto item : row_ptrs) {
        first = item.second.cast<Tensor>();
        break;
    }
    const c10::Device dev = compute_device({&first});
    DeviceGuard guard(dev);
    std::vector<int32_t> rel_src, rel_dst;
    std::vector<Tensor> ptrs, idx;
    std::vector<int64_t> node_count;
    std::vector<std::string> keys;
    for (const auto &et : edge_types) {
        const std::string key = rel_key(et);
        if (!row_ptrs.contains(py::str(key))) throw py::key_error(key); // graphs[rel_type] panics in the reference
        keys.push_back(key);
        rel_src.push_back(tix.at(std::get<0>(et)));
        rel_dst.push_back(tix.at(std::get<2>(et)));
        ptrs.push_back(on_graph(row_ptrs[py::str(key)].cast<Tensor>(), dev, at::kLong));
        idx.push_back(on_graph(col_indices[py::str(key)].cast<Tensor>(), dev, at::kLong));
        auto sz = sizes[py::str(key)].cast<std::pair<int64_t, int64_t>>();
        if (sz.second < 1) throw py::value_error("sizes[" + key + "][1] must be >= 1");
        node_count.push_back(sz.second);
    }
    std::vector<Tensor> in((size_t)T);
    std::vector<int64_t> n_in((size_t)T, -1);
    c10::Device out_dev = dev;
    bool out_dev_set = false;
    for (int t = 0; t < T; ++t) {
        if (!inputs.contains(py::str(node_types[(size_t)t]))) continue;
        Tensor x = inputs[py::str(node_types[(size_t)t])].cast<Tensor>();
        if (!out_dev_set) {
            out_dev = x.device();
            out_dev_set = true;
        }
        in[(size_t)t] = on(x, dev, at::kLong).reshape({-1});
        n_in[(size_t)t] = in[(size_t)t].numel();
    }
    {
        RangeCheck rc(dev); // an input of type t indexes the rows of every relation whose src is t
        for (size_t k = 0; k < keys.size(); ++k)
            if (n_in[(size_t)rel_src[k]] > 0) rc.add(in[(size_t)rel_src[k]], ptrs[k].numel() - 1);
        rc.verify("negative_sample_neighbors_heterogenous inputs");
    }
    NegRun r = run_neg(dev, T, rel_src, rel_dst, ptrs, idx, node_count, in, n_in, num_neg, try_count, inbound, false);
    py::dict samples, rows, cols, counts;
    for (int t = 0; t < T; ++t) {
        samples[py::str(node_types[(size_t)t])] = back(r.samples[(size_t)t].narrow(0, 0, r.n_samples[(size_t)t]), out_dev);
        counts[py::str(node_types[(size_t)t])] = std::max<int64_t>(n_in[(size_t)t], 0); // negative_sampling.rs:96
    }
    for (size_t k = 0; k < keys.size(); ++k) {
        rows[py::str(keys[k])] = back(r.rows[k].narrow(0, 0, r.n_edges[k]), out_dev);
        cols[py::str(keys[k])] = back(r.cols[k].narrow(0, 0, r.n_edges[k]), out_dev);
    }
    return py::make_tuple(samples, rows, cols, counts);
}

} // namespace

void register_hgt(py::module_ &m);    // python_module_hgt.cpp
void register_budget(py::module_ &m); // python_module_budget.cpp

void register_more(py::module_ &m) {
    m.def("negative_sample_neighbors_homogenous", &negative_sample_neighbors_homogenous, py::arg("row_ptrs"),
          py::arg("col_indices"), py::arg("graph_size"), py::arg("inputs"), py::arg("num_neg"), py::arg("try_count"));
    m.def("negative_sample_neighbors_heterogenous", &negative_sample_neighbors_heterogenous, py::arg("node_types"),
          py::arg("edge_types"), py::arg("row_ptrs"), py::arg("col_indices"), py::arg("sizes"), py::arg("inputs"),
          py::arg("num_neg"), py::arg("try_count"), py::arg("inbound"));
    register_hgt(m);
    register_budget(m);
}
