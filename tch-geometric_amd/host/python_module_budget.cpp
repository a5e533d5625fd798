// Fourth translation unit of the host module: budget_sampling (python.rs:486-581; SURVEY.md 8(f) "next" row).
// The per-node work runs in tg_budget_layer; this file drives the (layer, node type) loop of
// budget_sampling.rs:223-257 and appends the selected candidates in the reference's order.
#include "host_common.h"

using namespace tghost;

namespace {

py::tuple budget_sampling(const std::vector<std::string> &node_types,
                          const std::vector<std::tuple<std::string, std::string, std::string>> &edge_types,
                          const py::dict &col_ptrs, const py::dict &row_indices, const py::object &row_timestamps,
                          const py::dict &inputs, const py::object &input_timestamps, const py::dict &num_neighbors,
                          int64_t num_hops, const py::object &window, bool forward, bool relative) {
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
    const tg_rng rng = next_rng();

    struct Rel {
        std::string key;
        int src, dst;
        Tensor ptrs, idx, ts;
        std::vector<Tensor> rows, cols, eidx;
    };
    std::vector<Rel> rels;
    py::dict rts = row_timestamps.is_none() ? py::dict() : row_timestamps.cast<py::dict>();
    for (const auto &et : edge_types) {
        Rel r;
        r.key = rel_key(et);
        if (!col_ptrs.contains(py::str(r.key))) continue; // graphs are keyed by col_ptrs (python.rs:509-523)
        r.src = tix.at(std::get<0>(et));
        r.dst = tix.at(std::get<2>(et));
        r.ptrs = on_graph(col_ptrs[py::str(r.key)].cast<Tensor>(), dev, at::kLong);
        r.idx = on_graph(row_indices[py::str(r.key)].cast<Tensor>(), dev, at::kLong);
        if (rts.contains(py::str(r.key))) r.ts = on(rts[py::str(r.key)].cast<Tensor>(), dev, at::kLong);
        rels.push_back(std::move(r));
    }
    const int R = (int)rels.size();
    std::vector<int64_t> rel_src_host((size_t)std::max(R, 1), 0);
    for (int r = 0; r < R; ++r) rel_src_host[(size_t)r] = rels[(size_t)r].src;
    Tensor rel_src_dev = at::tensor(rel_src_host, at::TensorOptions().dtype(at::kLong)).to(dev);

    py::dict its = input_timestamps.is_none() ? py::dict() : input_timestamps.cast<py::dict>();
    std::vector<std::vector<Tensor>> chunks((size_t)T), ts_chunks((size_t)T), new_chunks((size_t)T), new_ts((size_t)T);
    std::vector<Tensor> frontier((size_t)T), frontier_ts((size_t)T);
    std::vector<int64_t> len((size_t)T, 0), fbegin((size_t)T, 0);
    c10::Device out_dev = dev;
    bool out_dev_set = false;
    for (int t = 0; t < T; ++t) { // :181-197
        const std::string &name = node_types[(size_t)t];
        if (inputs.contains(py::str(name))) {
            Tensor x = inputs[py::str(name)].cast<Tensor>();
            if (!out_dev_set) {
                out_dev = x.device();
                out_dev_set = true;
            }
            frontier[(size_t)t] = on(x, dev, at::kLong).reshape({-1});
        } else {
            frontier[(size_t)t] = at::empty({0}, i64(dev));
        }
        const int64_t n = frontier[(size_t)t].numel();
        if (its.contains(py::str(name))) {
            frontier_ts[(size_t)t] = on(its[py::str(name)].cast<Tensor>(), dev, at::kLong).reshape({-1});
            if (frontier_ts[(size_t)t].numel() != n) throw py::value_error("input_timestamps[" + name + "] must match inputs");
        } else {
            frontier_ts[(size_t)t] = at::full({n}, -1, i64(dev)); // :195 NAN_TIMESTAMP
        }
        chunks[(size_t)t].push_back(frontier[(size_t)t]);
        ts_chunks[(size_t)t].push_back(frontier_ts[(size_t)t]);
        len[(size_t)t] = n;
    }
    {
        RangeCheck rc(dev); // an input of type t indexes the columns of every relation whose dst is t
        for (const Rel &r : rels) rc.add(frontier[(size_t)r.dst], r.ptrs.numel() - 1);
        rc.verify("budget_sampling inputs");
    }
    tg_budget_layer_in in{};
    if (!window.is_none()) { // python.rs:541-548
        auto w = window.cast<std::pair<int64_t, int64_t>>();
        in.filter_on = 1;
        in.forward = forward ? 1 : 0;
        in.relative = relative ? 1 : 0;
        in.win_lo = w.first;
        in.win_hi = w.second;
    }
    // ---- the whole operator in one stream-ordered call (tg_budget_sample): no host bookkeeping, one read-back
    bool fused = T <= 8 && R <= 16 && H <= TG_MAX_HOPS;
    std::vector<int64_t> quota((size_t)std::max(T * H, 1), 0);
    for (int t = 0; t < T && H > 0; ++t) {
        const std::string &name = node_types[(size_t)t];
        if (!num_neighbors.contains(py::str(name)))
            throw PanicError("budget_sampling: num_neighbors has no entry for node type " + name +
                                     " (the reference panics here, budget_sampling.rs:226)");
        auto q = num_neighbors[py::str(name)].cast<std::vector<int64_t>>();
        if ((int64_t)q.size() < H) throw py::index_error("num_neighbors[" + name + "] is shorter than num_hops");
        for (int h = 0; h < H; ++h) {
            quota[(size_t)(t * H + h)] = q[(size_t)h];
            if (q[(size_t)h] < 0 || q[(size_t)h] > 64) fused = false;
        }
    }
    if (fused) {
        std::vector<int32_t> rs((size_t)std::max(R, 1)), rd((size_t)std::max(R, 1));
        std::vector<tg_graph> graphs((size_t)std::max(R, 1));
        for (int r = 0; r < R; ++r) {
            const Rel &rl = rels[(size_t)r];
            rs[(size_t)r] = rl.src;
            rd[(size_t)r] = rl.dst;
            tg_graph g{};
            g.ptrs = rl.ptrs.data_ptr<int64_t>();
            g.indices = rl.idx.numel() ? rl.idx.data_ptr<int64_t>() : nullptr;
            g.timestamps = rl.ts.defined() ? rl.ts.data_ptr<int64_t>() : nullptr;
            g.n_major = rl.ptrs.numel() - 1;
            g.n_edges = rl.idx.numel();
            graphs[(size_t)r] = g;
        }
        std::vector<const int64_t *> in_ptr((size_t)T, nullptr), in_ts_ptr((size_t)T, nullptr);
        std::vector<int64_t> n_in((size_t)T, 0), cap_n((size_t)T, 0), cap_e((size_t)std::max(R, 1), 0);
        for (int t = 0; t < T; ++t) {
            n_in[(size_t)t] = frontier[(size_t)t].numel();
            if (n_in[(size_t)t]) {
                in_ptr[(size_t)t] = frontier[(size_t)t].data_ptr<int64_t>();
                in_ts_ptr[(size_t)t] = frontier_ts[(size_t)t].data_ptr<int64_t>();
            }
        }
        tg_budget_problem pb{};
        pb.n_types = T;
        pb.n_rels = R;
        pb.n_hops = H;
        pb.filter_on = in.filter_on;
        pb.forward = in.forward;
        pb.relative = in.relative;
        pb.win_lo = in.win_lo;
        pb.win_hi = in.win_hi;
        pb.rel_src = rs.data();
        pb.rel_dst = rd.data();
        pb.graphs = graphs.data();
        pb.num_neighbors = quota.data();
        pb.inputs = in_ptr.data();
        pb.input_ts = in_ts_ptr.data();
        pb.n_inputs = n_in.data();
        check_rc(tg_budget_capacity(&pb, cap_n.data(), cap_e.data()));
        int64_t ws_bytes = 0;
        check_rc(tg_budget_workspace_bytes(&pb, &ws_bytes));
        Tensor ws = at::empty({ws_bytes / 8 + 1}, i64(dev));
        std::vector<Tensor> S((size_t)T), TS((size_t)T), RW((size_t)R), CL((size_t)R), EI((size_t)R);
        std::vector<int64_t *> s_ptr((size_t)T), ts_ptr((size_t)T), r_ptr((size_t)std::max(R, 1)), c_ptr((size_t)std::max(R, 1)),
            e_ptr((size_t)std::max(R, 1));
        for (int t = 0; t < T; ++t) {
            S[(size_t)t] = at::empty({std::max<int64_t>(cap_n[(size_t)t], 1)}, i64(dev));
            TS[(size_t)t] = at::empty({std::max<int64_t>(cap_n[(size_t)t], 1)}, i64(dev));
            s_ptr[(size_t)t] = S[(size_t)t].data_ptr<int64_t>();
            ts_ptr[(size_t)t] = TS[(size_t)t].data_ptr<int64_t>();
        }
        for (int r = 0; r < R; ++r) {
            const int64_t c = std::max<int64_t>(cap_e[(size_t)r], 1);
            RW[(size_t)r] = at::empty({c}, i64(dev));
            CL[(size_t)r] = at::empty({c}, i64(dev));
            EI[(size_t)r] = at::empty({c}, i64(dev));
            r_ptr[(size_t)r] = RW[(size_t)r].data_ptr<int64_t>();
            c_ptr[(size_t)r] = CL[(size_t)r].data_ptr<int64_t>();
            e_ptr[(size_t)r] = EI[(size_t)r].data_ptr<int64_t>();
        }
        Tensor counts = at::zeros({T + R + 1}, i64(dev));
        tg_budget_out o{};
        o.samples = s_ptr.data();
        o.sample_ts = ts_ptr.data();
        o.cap_nodes = cap_n.data();
        o.rows = r_ptr.data();
        o.cols = c_ptr.data();
        o.edge_index = e_ptr.data();
        o.cap_edges = cap_e.data();
        o.counts = counts.data_ptr<int64_t>();
        check_rc(tg_budget_sample(&pb, &rng, &o, ws.data_ptr<int64_t>(), ws_bytes, stream_of(dev)));
        Tensor c = to_host(counts); // the call's only synchronisation
        const int64_t *ch = c.data_ptr<int64_t>();
        py::dict d_samples, d_ts, d_rows, d_cols, d_eidx;
        for (int t = 0; t < T; ++t) {
            d_samples[py::str(node_types[(size_t)t])] = back(S[(size_t)t].narrow(0, 0, ch[t]), out_dev);
            d_ts[py::str(node_types[(size_t)t])] = back(TS[(size_t)t].narrow(0, 0, ch[t]), out_dev);
        }
        for (int r = 0; r < R; ++r) {
            const int64_t ne = ch[T + r];
            d_rows[py::str(rels[(size_t)r].key)] = back(RW[(size_t)r].narrow(0, 0, ne), out_dev);
            d_cols[py::str(rels[(size_t)r].key)] = back(CL[(size_t)r].narrow(0, 0, ne), out_dev);
            d_eidx[py::str(rels[(size_t)r].key)] = back(EI[(size_t)r].narrow(0, 0, ne), out_dev);
        }
        return py::make_tuple(d_samples, d_ts, d_rows, d_cols, d_eidx);
    }

    for (int layer = 0; layer < H; ++layer) { // :223
        for (int t = 0; t < T; ++t) {
            new_chunks[(size_t)t].clear();
            new_ts[(size_t)t].clear();
        }
        for (int t = 0; t < T; ++t) { // :225 node_types order
            const std::string &name = node_types[(size_t)t];
            if (!num_neighbors.contains(py::str(name)))
                throw PanicError("budget_sampling: num_neighbors has no entry for node type " + name +
                                         " (the reference panics here, budget_sampling.rs:226)");
            auto quota = num_neighbors[py::str(name)].cast<std::vector<int64_t>>();
            if ((int64_t)quota.size() <= layer) throw py::index_error("num_neighbors[" + name + "] is shorter than num_hops");
            const int64_t k = quota[(size_t)layer], F = frontier[(size_t)t].numel();
            if (F == 0 || k == 0) continue;
            std::vector<tg_graph> gs;
            std::vector<int32_t> ids;
            for (int r = 0; r < R; ++r) {
                const Rel &rl = rels[(size_t)r];
                if (rl.dst != t) continue;
                tg_graph g{};
                g.ptrs = rl.ptrs.data_ptr<int64_t>();
                g.indices = rl.idx.numel() ? rl.idx.data_ptr<int64_t>() : nullptr;
                g.timestamps = rl.ts.defined() ? rl.ts.data_ptr<int64_t>() : nullptr;
                g.n_major = rl.ptrs.numel() - 1;
                g.n_edges = rl.idx.numel();
                gs.push_back(g);
                ids.push_back(r);
            }
            if (gs.empty()) continue;
            Tensor sel_v = at::empty({F * k}, i64(dev)), sel_ts = at::empty({F * k}, i64(dev));
            Tensor sel_rel = at::empty({F * k}, i64(dev)), sel_i = at::empty({F * k}, i64(dev));
            in.graphs = gs.data();
            in.rel_ids = ids.data();
            in.n_rels = (int32_t)gs.size();
            in.node_type = t;
            in.fanout = (int32_t)k;
            in.nodes = frontier[(size_t)t].data_ptr<int64_t>();
            in.nodes_ts = frontier_ts[(size_t)t].data_ptr<int64_t>();
            in.n_front = F;
            in.id_base = fbegin[(size_t)t];
            tg_budget_layer_out out{sel_v.data_ptr<int64_t>(), sel_ts.data_ptr<int64_t>(), sel_rel.data_ptr<int64_t>(),
                                    sel_i.data_ptr<int64_t>()};
            check_rc(tg_budget_layer(&in, &rng, &out, stream_of(dev)));
            // selected candidates in (node, slot) order (:140-151)
            Tensor pick = at::nonzero(sel_rel >= 0).reshape({-1});
            if (pick.numel() == 0) continue;
            Tensor v = sel_v.index_select(0, pick), vt = sel_ts.index_select(0, pick);
            Tensor rel = sel_rel.index_select(0, pick), ci = sel_i.index_select(0, pick);
            Tensor j = at::floor_divide(pick, k) + fbegin[(size_t)t];
            Tensor stype = rel_src_dev.index_select(0, rel);
            for (int s = 0; s < T; ++s) { // appends to a source type keep (node, slot) order
                Tensor ps = at::nonzero(stype == s).reshape({-1});
                const int64_t cnt = ps.numel();
                if (cnt == 0) continue;
                Tensor new_index = at::arange(len[(size_t)s], len[(size_t)s] + cnt, i64(dev)); // :147
                Tensor vs = v.index_select(0, ps), ts_s = vt.index_select(0, ps), rs = rel.index_select(0, ps);
                Tensor js = j.index_select(0, ps), cs = ci.index_select(0, ps);
                chunks[(size_t)s].push_back(vs);
                ts_chunks[(size_t)s].push_back(ts_s);
                new_chunks[(size_t)s].push_back(vs);
                new_ts[(size_t)s].push_back(ts_s);
                len[(size_t)s] += cnt;
                for (int r : ids) { // :150 push_edge(i, j, edge_ptr)
                    if (rels[(size_t)r].src != s) continue;
                    Tensor pr = at::nonzero(rs == r).reshape({-1});
                    if (pr.numel() == 0) continue;
                    rels[(size_t)r].rows.push_back(new_index.index_select(0, pr));
                    rels[(size_t)r].cols.push_back(js.index_select(0, pr));
                    rels[(size_t)r].eidx.push_back(cs.index_select(0, pr));
                }
            }
        }
        for (int t = 0; t < T; ++t) { // :240-243
            fbegin[(size_t)t] += frontier[(size_t)t].numel();
            frontier[(size_t)t] = new_chunks[(size_t)t].empty() ? at::empty({0}, i64(dev)) : at::cat(new_chunks[(size_t)t]);
            frontier_ts[(size_t)t] = new_ts[(size_t)t].empty() ? at::empty({0}, i64(dev)) : at::cat(new_ts[(size_t)t]);
        }
    }
    py::dict d_samples, d_ts, d_rows, d_cols, d_eidx;
    for (int t = 0; t < T; ++t) {
        d_samples[py::str(node_types[(size_t)t])] = back(at::cat(chunks[(size_t)t]), out_dev);
        d_ts[py::str(node_types[(size_t)t])] = back(at::cat(ts_chunks[(size_t)t]), out_dev);
    }
    auto cat_or_empty = [&](const std::vector<Tensor> &v) { return v.empty() ? at::empty({0}, i64(dev)) : at::cat(v); };
    for (Rel &r : rels) {
        d_rows[py::str(r.key)] = back(cat_or_empty(r.rows), out_dev);
        d_cols[py::str(r.key)] = back(cat_or_empty(r.cols), out_dev);
        d_eidx[py::str(r.key)] = back(cat_or_empty(r.eidx), out_dev);
    }
    return py::make_tuple(d_samples, d_ts, d_rows, d_cols, d_eidx);
}

} // namespace

void register_budget(py::module_ &m) {
    m.def("budget_sampling", &budget_sampling, py::arg("node_types"), py::arg("edge_types"), py::arg("col_ptrs"),
          py::arg("row_indices"), py::arg("row_timestamps"), py::arg("inputs"), py::arg("input_timestamps"),
          py::arg("num_neighbors"), py::arg("num_hops"), py::arg("window"), py::arg("forward"), py::arg("relative"));
}
