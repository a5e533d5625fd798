// Device-side bookkeeping of neighbor_sampling_heterogenous when its (hop, relation) steps run as flat hops
// (temporal filters, weighted sampler, fan-outs or problem sizes beyond the fused launch of ns_hetero.hip):
// neighbor_sampling.rs:292-352 keeps list lengths, frontier slices and relation edge counts on the host; here they
// live in a device array (`meta`), so a whole call issues its launches without one read-back in between.
//
//   meta = len[T] | fbeg[T] | fend[T] | ne[R] | layer_offsets[R][H][3] | snap[4]
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
    *words = 3 * (int64_t)n_types + n_rels + (int64_t)n_rels * n_hops * 3 + 4;
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
