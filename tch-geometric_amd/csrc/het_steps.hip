// Device-side bookkeeping of neighbor_sampling_heterogenous when its (hop, relation) steps run as flat hops
// (temporal filters, weighted sampler, fan-outs or problem sizes beyond the fused launch of ns_hetero.hip):
// neighbor_sampling.rs:292-352 keeps list lengths, frontier slices and relation edge counts on the host; here they
// live in a device array (`meta`), so a whole call issues its launches without one read-back in between.
//
//   meta = len[T] | fbeg[T] | fend[T] | ne[R] | layer_offsets[R][H][3] | snap[max(4, T + 2R)]
//
//   tg_het_step_begin   relation r, hop h: snapshot (len[src], ne[r], fbeg[dst]); layer_offsets[r][h] =
//                       (len[src], ne[r], len[dst]) (:314-315); the frontier = list[dst][fbeg, fend) copied into a
//                       buffer of the hop's worst-case size, padded with -1 (the flat hops skip negative vertices);
//                       draw ids = fbeg + i (the slot of the vertex in its type's list), filter states alongside.
//   [tg_ns_hop / tg_ns_hop_scan / tg_ns_hop_weighted over the padded frontier]
//   tg_het_step_end     appends the step's samples to list[src] (and their filter states), (row = new index in
//                       src's list, col = frontier slot, edge pointer) to the relation's lists (:333-340) and
//                       advances len[src] and ne[r] (same launch: the append reads the snapshot, not the live lengths).
//   tg_het_hop_end      slices[t] = (end, len[t]) for every node type (:345-348).
//
// ALL relations of a hop in one set of launches (tg_het_hop_begin_all / tg_ns_hop_segments / tg_het_hop_end_all): the
// frontier of a hop is fixed when the hop starts (:345-348 advance the slices only at its end), so the relations'
// frontiers can be concatenated, sampled together, and their appends replayed in relation order afterwards -- the
// list positions the reference reaches one relation at a time are prefix sums of the relations' totals.
#include "tg_device.h"
#include "tg_host.h"

namespace tg {

struct HetMeta {
    int64_t *len, *fbeg, *fend, *ne, *lo, *snap;
};
__host__ __device__ inline HetMeta het_meta(int64_t *m, int T, int R, int H) {
    HetMeta x;
    x.len = m;
    x.fbeg = m + T;
    x.fend = m + 2 * T;
    x.ne = m + 3 * T;
    x.lo = m + 3 * T + R;
    x.snap = x.lo + (int64_t)R * H * 3;
    return x;
}
__host__ __device__ inline int64_t het_snap_words(int T, int R) { return (T + 2 * R > 4) ? T + 2 * R : 4; }

// ---------------------------------------------------------------- all relations of a hop at once
struct HetEntries {
    tg_het_entry e[TG_HET_HOP_MAX_ENTRIES];
    int32_t n;
};
// kernel arguments -> LDS (dynamic indexing of by-value arguments would go through scratch memory)
__device__ __forceinline__ void het_load_entries(const HetEntries &a, tg_het_entry *E) {
#pragma unroll
    for (int j = 0; j < TG_HET_HOP_MAX_ENTRIES; ++j)
        if ((int)threadIdx.x == j && j < a.n) E[j] = a.e[j];
    __syncthreads();
}

// snapshot of the list lengths / edge counts / frontier starts; the concatenated frontier (padded with -1), its draw
// ids (= slot of the vertex in its type's list) and filter states
__global__ void het_hop_begin_all_kernel(const HetEntries a, int64_t *meta, int T, int R, int H, int64_t m_total,
                                         int64_t *frontier, int64_t *fstate, int64_t *ids, int64_t *layout_dev) {
    __shared__ tg_het_entry E[TG_HET_HOP_MAX_ENTRIES];
    __shared__ int64_t seg_at[TG_HET_HOP_MAX_ENTRIES], seg_len[TG_HET_HOP_MAX_ENTRIES], m_real;
    het_load_entries(a, E);
    const HetMeta M = het_meta(meta, T, R, H);
    int64_t *snap_len = M.snap, *snap_ne = M.snap + T, *snap_fb = M.snap + T + R;
    if (threadIdx.x == 0) { // where every segment's frontier starts: padded (host layout) or packed back to back
        int64_t at = 0;
        int n_seg = 0;
        for (int j = 0; j < a.n; ++j) {
            seg_at[j] = seg_len[j] = 0;
            if (E[j].segment < 0) continue;
            const int64_t live = min(max(M.fend[E[j].dst] - M.fbeg[E[j].dst], (int64_t)0), E[j].cap);
            seg_at[j] = layout_dev ? at : E[j].begin;
            seg_len[j] = layout_dev ? live : E[j].cap;
            if (layout_dev && blockIdx.x == 0) layout_dev[n_seg] = at;
            at += live;
            ++n_seg;
        }
        m_real = layout_dev ? at : m_total;
        if (layout_dev && blockIdx.x == 0) layout_dev[n_seg] = at;
    }
    if (blockIdx.x == 0) {
        for (int t = threadIdx.x; t < T; t += blockDim.x) snap_len[t] = M.len[t];
        for (int j = threadIdx.x; j < a.n; j += blockDim.x) {
            snap_ne[E[j].rel] = M.ne[E[j].rel];
            snap_fb[E[j].rel] = M.fbeg[E[j].dst];
        }
    }
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m_real; i += (int64_t)gridDim.x * blockDim.x) {
        int j = -1;
        for (int q = 0; q < a.n; ++q)
            if (E[q].segment >= 0 && i >= seg_at[q] && i < seg_at[q] + seg_len[q]) j = q;
        int64_t v = -1, st = 0, id = 0;
        if (j >= 0) {
            const int64_t fb = M.fbeg[E[j].dst], fe = M.fend[E[j].dst], at = fb + (i - seg_at[j]);
            id = at;
            if (at < fe) {
                v = E[j].list_dst[at];
                if (fstate) st = E[j].state_dst[at];
            }
        }
        frontier[i] = v;
        if (fstate) fstate[i] = st;
        ids[i] = id;
    }
}

// replays the relations' appends in relation order: layer offsets (:314-315), sample lists (:333-334), edge lists
// (:340), list lengths and edge counts; with `last` also the frontier slices of the next hop (:345-348)
__global__ void het_hop_end_all_kernel(const HetEntries a, const int64_t *__restrict__ offsets, const int64_t *__restrict__ nbr,
                                       const int64_t *__restrict__ ep, const int64_t *__restrict__ par,
                                       const int64_t *__restrict__ st_out, int64_t m_total,
                                       const int64_t *__restrict__ layout_dev, int64_t *meta, int T, int R, int H, int hop,
                                       int last, int32_t *status) {
    __shared__ tg_het_entry E[TG_HET_HOP_MAX_ENTRIES];
    __shared__ int64_t len_run[64], base_s[TG_HET_HOP_MAX_ENTRIES], base_e[TG_HET_HOP_MAX_ENTRIES],
        ostart[TG_HET_HOP_MAX_ENTRIES], total[TG_HET_HOP_MAX_ENTRIES], seg_at[TG_HET_HOP_MAX_ENTRIES], m_real;
    __shared__ int bad;
    het_load_entries(a, E);
    const HetMeta M = het_meta(meta, T, R, H);
    const int64_t *snap_len = M.snap, *snap_ne = M.snap + T, *snap_fb = M.snap + T + R;
    if (threadIdx.x == 0) {
        bad = 0;
        m_real = m_total;
        for (int t = 0; t < T; ++t) len_run[t] = snap_len[t];
        int n_seg = 0;
        for (int j = 0; j < a.n; ++j)
            if (E[j].segment >= 0) ++n_seg;
        if (layout_dev) m_real = layout_dev[n_seg];
        n_seg = 0;
        for (int j = 0; j < a.n; ++j) {
            const tg_het_entry &e = E[j];
            const int64_t lo_s = len_run[e.src], lo_e = snap_ne[e.rel], lo_d = len_run[e.dst];
            int64_t o0 = 0, tot = 0;
            seg_at[j] = 0;
            if (e.segment >= 0) { // the segment's slots in the frontier the hop really ran over
                const int64_t b0 = layout_dev ? layout_dev[n_seg] : e.begin;
                const int64_t b1 = layout_dev ? layout_dev[n_seg + 1] : e.begin + e.cap;
                ++n_seg;
                seg_at[j] = b0;
                o0 = offsets[b0];
                tot = offsets[b1] - o0;
            }
            base_s[j] = lo_s;
            base_e[j] = lo_e;
            ostart[j] = o0;
            total[j] = tot;
            if (lo_s + tot > e.cap_list_src || lo_e + tot > e.cap_edges) bad = 1; // cannot happen with worst-case capacities
            if (blockIdx.x == 0) {
                int64_t *lo = M.lo + ((int64_t)e.rel * H + hop) * 3;
                lo[0] = lo_s;
                lo[1] = lo_e;
                lo[2] = lo_d;
                M.ne[e.rel] = lo_e + tot;
            }
            len_run[e.src] = lo_s + tot;
        }
        if (blockIdx.x == 0) {
            if (bad) atomicOr(status, 4);
            for (int t = 0; t < T; ++t) {
                M.len[t] = len_run[t];
                if (last) { // :345-348
                    M.fbeg[t] = M.fend[t];
                    M.fend[t] = len_run[t];
                }
            }
        }
    }
    __syncthreads();
    if (bad || m_total == 0) return;
    const int64_t n_out = offsets[m_real];
    for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < n_out; o += (int64_t)gridDim.x * blockDim.x) {
        int j = 0;
        for (int q = 0; q < a.n; ++q)
            if (E[q].segment >= 0 && o >= ostart[q] && o < ostart[q] + total[q]) j = q;
        const tg_het_entry &e = E[j];
        const int64_t x = o - ostart[j];
        e.list_src[base_s[j] + x] = nbr[o];                       // :333
        if (st_out) e.state_src[base_s[j] + x] = st_out[o];       // :334
        e.rows[base_e[j] + x] = base_s[j] + x;                    // :340 j
        e.cols[base_e[j] + x] = (par[o] - seg_at[j]) + snap_fb[e.rel]; // :340 i
        e.edge_index[base_e[j] + x] = ep[o];
    }
}

__global__ void het_begin_kernel(const int64_t *__restrict__ list_dst, const int64_t *__restrict__ state_dst, int64_t *meta,
                                 int T, int R, int H, int src, int dst, int rel, int hop, int64_t cap_f, int64_t *frontier,
                                 int64_t *fstate, int64_t *ids) {
    const HetMeta M = het_meta(meta, T, R, H);
    const int64_t fb = M.fbeg[dst], fe = M.fend[dst];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        M.snap[0] = M.len[src];
        M.snap[1] = M.ne[rel];
        M.snap[2] = fb;
        int64_t *lo = M.lo + ((int64_t)rel * H + hop) * 3; // neighbor_sampling.rs:314-315
        lo[0] = M.len[src];
        lo[1] = M.ne[rel];
        lo[2] = M.len[dst];
    }
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cap_f; i += (int64_t)gridDim.x * blockDim.x) {
        const bool live = fb + i < fe;
        frontier[i] = live ? list_dst[fb + i] : -1;
        if (fstate) fstate[i] = live ? state_dst[fb + i] : 0;
        ids[i] = fb + i;
    }
}

__global__ void het_end_kernel(const int64_t *__restrict__ offsets, const int64_t *__restrict__ nbr,
                               const int64_t *__restrict__ ep, const int64_t *__restrict__ par,
                               const int64_t *__restrict__ st_out, int64_t cap_f, int64_t *meta, int T, int R, int H, int src,
                               int rel, int64_t *list_src, int64_t *state_src, int64_t cap_list, int64_t *rows, int64_t *cols,
                               int64_t *eidx, int64_t cap_e, int32_t *status) {
    const HetMeta M = het_meta(meta, T, R, H);
    const int64_t total = offsets[cap_f];
    const int64_t base_s = M.snap[0], base_e = M.snap[1], fb = M.snap[2];
    if (base_s + total > cap_list || base_e + total > cap_e) { // cannot happen with worst-case capacities
        if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr(status, 4);
        return;
    }
    // the append works from the snapshot tg_het_step_begin took, so the live lengths can advance in the same launch
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        M.len[src] = base_s + total;
        M.ne[rel] = base_e + total;
    }
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < total; j += (int64_t)gridDim.x * blockDim.x) {
        list_src[base_s + j] = nbr[j];                       // :333
        if (state_src) state_src[base_s + j] = st_out[j];    // :334
        rows[base_e + j] = base_s + j;                       // :340 j
        cols[base_e + j] = par[j] + fb;                      // :340 i
        eidx[base_e + j] = ep[j];
    }
}

__global__ void het_hop_end_kernel(int64_t *meta, int T, int R, int H) {
    const HetMeta M = het_meta(meta, T, R, H);
    const int t = threadIdx.x;
    if (t < T) { // :345-348
        M.fbeg[t] = M.fend[t];
        M.fend[t] = M.len[t];
    }
}

} // namespace tg

extern "C" int tg_het_meta_words(int32_t n_types, int32_t n_rels, int32_t n_hops, int64_t *words) {
    TG_REQUIRE(words && n_types >= 1 && n_rels >= 0 && n_hops >= 0, "tg_het_meta_words: bad arguments");
    *words = 3 * (int64_t)n_types + n_rels + (int64_t)n_rels * n_hops * 3 + tg::het_snap_words(n_types, n_rels);
    return TG_OK;
}

extern "C" int tg_het_step_begin(const int64_t *list_dst, const int64_t *state_dst, int64_t *meta, int32_t n_types,
                                 int32_t n_rels, int32_t n_hops, int32_t src, int32_t dst, int32_t rel, int32_t hop,
                                 int64_t cap_f, int64_t *frontier, int64_t *fstate, int64_t *ids, void *stream) {
    TG_REQUIRE(meta && frontier && ids && cap_f >= 1 && src >= 0 && src < n_types && dst >= 0 && dst < n_types && rel >= 0 &&
                   rel < n_rels && hop >= 0 && hop < n_hops,
               "tg_het_step_begin: bad arguments");
    TG_REQUIRE((fstate == nullptr) == (state_dst == nullptr), "tg_het_step_begin: states come with their list");
    int64_t g = (cap_f + 255) / 256;
    if (g > 1024) g = 1024;
    hipLaunchKernelGGL(tg::het_begin_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, list_dst, state_dst, meta,
                       (int)n_types, (int)n_rels, (int)n_hops, (int)src, (int)dst, (int)rel, (int)hop, cap_f, frontier, fstate,
                       ids);
    TG_LAUNCH_CHECK();
    return TG_OK;
}

extern "C" int tg_het_step_end(const tg_hop_out *out, const int64_t *states_out, int64_t cap_f, int32_t fanout, int64_t *meta,
                               int32_t n_types, int32_t n_rels, int32_t n_hops, int32_t src, int32_t rel, int64_t *list_src,
                               int64_t *state_src, int64_t cap_list, int64_t *rows, int64_t *cols, int64_t *edge_index,
                               int64_t cap_edges, int32_t *status, void *stream) {
    TG_REQUIRE(out && out->offsets && out->neighbors && out->edge_ptrs && out->parents && meta && list_src && rows && cols &&
                   edge_index && status && cap_f >= 1 && fanout >= 1,
               "tg_het_step_end: bad arguments");
    TG_REQUIRE((state_src == nullptr) == (states_out == nullptr), "tg_het_step_end: states come with their list");
    int64_t g = (cap_f * fanout + 255) / 256;
    if (g > 2048) g = 2048;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(tg::het_end_kernel, dim3((unsigned)g), dim3(256), 0, s, out->offsets, out->neighbors, out->edge_ptrs,
                       out->parents, states_out, cap_f, meta, (int)n_types, (int)n_rels, (int)n_hops, (int)src, (int)rel,
                       list_src, state_src, cap_list, rows, cols, edge_index, cap_edges, status);
    TG_LAUNCH_CHECK();
    return TG_OK;
}

extern "C" int tg_het_hop_end(int64_t *meta, int32_t n_types, int32_t n_rels, int32_t n_hops, void *stream) {
    TG_REQUIRE(meta && n_types >= 1 && n_types <= 64, "tg_het_hop_end: bad arguments");
    hipLaunchKernelGGL(tg::het_hop_end_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, meta, (int)n_types, (int)n_rels,
                       (int)n_hops);
    TG_LAUNCH_CHECK();
    return TG_OK;
}

static int het_entries(tg::HetEntries &a, const tg_het_entry *entries, int32_t n_entries, int32_t n_types, int32_t n_rels,
                       int64_t m_total, bool begin) {
    TG_REQUIRE(entries && n_entries >= 1 && n_entries <= TG_HET_HOP_MAX_ENTRIES, "tg_het_hop_*_all: 1 .. %d entries",
               TG_HET_HOP_MAX_ENTRIES);
    a.n = n_entries;
    for (int j = 0; j < n_entries; ++j) {
        const tg_het_entry &e = entries[j];
        TG_REQUIRE(e.rel >= 0 && e.rel < n_rels && e.src >= 0 && e.src < n_types && e.dst >= 0 && e.dst < n_types,
                   "tg_het_hop_*_all: entry %d names an unknown relation or node type", j);
        if (e.segment >= 0) {
            TG_REQUIRE(e.begin >= 0 && e.cap >= 1 && e.begin + e.cap <= m_total, "tg_het_hop_*_all: entry %d leaves the frontier", j);
            TG_REQUIRE(begin ? e.list_dst != nullptr : (e.list_src && e.rows && e.cols && e.edge_index),
                       "tg_het_hop_*_all: entry %d has null lists", j);
        }
        a.e[j] = e;
    }
    return TG_OK;
}

extern "C" int tg_het_hop_begin_all(const tg_het_entry *entries, int32_t n_entries, int64_t *meta, int32_t n_types,
                                    int32_t n_rels, int32_t n_hops, int64_t m_total, int64_t *frontier, int64_t *fstate,
                                    int64_t *ids, int64_t *layout_dev, void *stream) {
    TG_REQUIRE(meta && n_types >= 1 && n_types <= 64 && m_total >= 0 && (m_total == 0 || (frontier && ids)),
               "tg_het_hop_begin_all: bad arguments");
    tg::HetEntries a{};
    int rc = het_entries(a, entries, n_entries, n_types, n_rels, m_total, true);
    if (rc != TG_OK) return rc;
    int64_t g = (m_total + 255) / 256;
    g = g < 1 ? 1 : (g > 1024 ? 1024 : g);
    hipLaunchKernelGGL(tg::het_hop_begin_all_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, a, meta, (int)n_types,
                       (int)n_rels, (int)n_hops, m_total, frontier, fstate, ids, layout_dev);
    TG_LAUNCH_CHECK();
    return TG_OK;
}

extern "C" int tg_het_hop_end_all(const tg_het_entry *entries, int32_t n_entries, const tg_hop_out *out,
                                  const int64_t *states_out, int64_t m_total, int64_t out_cap, const int64_t *layout_dev,
                                  int64_t *meta, int32_t n_types, int32_t n_rels, int32_t n_hops, int32_t hop, int32_t last,
                                  int32_t *status, void *stream) {
    TG_REQUIRE(meta && status && n_types >= 1 && n_types <= 64 && hop >= 0 && hop < n_hops && m_total >= 0 && out_cap >= 0,
               "tg_het_hop_end_all: bad arguments");
    TG_REQUIRE(m_total == 0 || (out && out->offsets && out->neighbors && out->edge_ptrs && out->parents),
               "tg_het_hop_end_all: null hop outputs");
    tg::HetEntries a{};
    int rc = het_entries(a, entries, n_entries, n_types, n_rels, m_total, false);
    if (rc != TG_OK) return rc;
    int64_t g = (out_cap + 255) / 256;
    g = g < 1 ? 1 : (g > 2048 ? 2048 : g);
    hipLaunchKernelGGL(tg::het_hop_end_all_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, a,
                       m_total ? out->offsets : nullptr, m_total ? out->neighbors : nullptr, m_total ? out->edge_ptrs : nullptr,
                       m_total ? out->parents : nullptr, states_out, m_total, layout_dev, meta, (int)n_types, (int)n_rels,
                       (int)n_hops, (int)hop, (int)last, status);
    TG_LAUNCH_CHECK();
    return TG_OK;
}
