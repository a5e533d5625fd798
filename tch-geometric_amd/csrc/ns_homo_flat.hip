// neighbor_sampling_homogenous under a temporal filter / with the weighted sampler for FEW seed batches: hop by hop over
// the whole device (round 4).
//
// tg_ns_homo_batched gives a batch ONE workgroup (ns_homo_scan.hip): right from a few hundred batches on, wrong below -- 64
// batches of 1 024 seeds on RMAT-24 leave three quarters of the CUs idle while each workgroup inspects 3 * 10^7 edges
// (weighted 54 ms, temporal 5.5 ms; through the flat hops 15.3 / 2.4 ms).  The reference has ONE entry point
// (src/python.rs:210-257), so the C ABI switches by itself: with a workspace (tg_ns_homo_batched_workspace_bytes) a launch of
// at most 256 weighted / filtered batches runs, per hop,
//   frontier   every batch's frontier slice -> one flat array (vertex, draw id = id_base + slot, call id = call + batch,
//              filter state), the batches' REAL frontiers back to back (a one-workgroup scan of their lengths first), the
//              unused tail up to the hop's worst case filled with -1: the flat hops' wavefronts over the tail leave at once;
//   flat hop   tg_ns_hop_scan / tg_ns_hop_weighted_groups: the columns cut into 512-edge groups processed all over the
//              device (the draws are named by (call id, slot): the same as the per-batch kernel's and the oracle's);
//   emit       one workgroup per batch copies its slice of the compact hop output into the batch's slabs in slot order
//              (neighbor_sampling.rs:212-217) and advances the batch's state.
// Everything stays on the device.  Two things the per-batch kernel reports and a flat hop cannot attribute to a batch --
// a column-group bound that was too low, a weighted column whose sum is not positive (the batch's counts[0] = -1) -- raise
// a status word; the per-batch kernel is launched BEHIND the flat path and returns at once unless that word is set, so the
// result is always the per-batch kernel's, bit for bit.
#include <algorithm>

#include "tg_device.h"
#include "tg_host.h"

int tg_ns_homo_filtered_launch_if(const tg_graph *csc, const int64_t *seeds, int64_t n_batches, int64_t n_seeds,
                                  const int64_t *fanout, int32_t n_hops, const tg_ns_config *cfg, const tg_rng *rng,
                                  const tg_ns_out *out, const int32_t *only_if, hipStream_t stream); // ns_homo_scan.hip

namespace tg {

struct FlatState {
    int64_t begin, end, ne;
};

struct FlatParams {
    const int64_t *seeds, *seeds_state;
    int64_t n_seeds, n_batches, cap_nodes, cap_edges, id_base, pitch;
    int32_t n_hops, hop;
    uint64_t call_id;
    int64_t *samples, *rows, *cols, *edge_index, *layer_offsets, *counts, *states;
    FlatState *st;
    int64_t *boff; // [n_batches + 1] start of every batch's slice in the hop's flat frontier
    int64_t *vertices, *ids, *call_ids, *fstates;                              // the hop's flat frontier [n_batches * pitch]
    int64_t *cnt, *offsets, *neighbors, *edge_ptrs, *parents, *states_out;     // the flat hop's outputs
};

__global__ void flat_begin_kernel(const FlatParams p) {
    const int64_t b = blockIdx.x;
    for (int64_t i = threadIdx.x; i < p.n_seeds; i += blockDim.x) {
        p.samples[b * p.cap_nodes + i] = p.seeds[b * p.n_seeds + i]; // :184
        if (p.states) p.states[b * p.cap_nodes + i] = p.seeds_state[b * p.n_seeds + i];
    }
    if (threadIdx.x == 0) {
        p.st[b] = FlatState{0, p.n_seeds, 0};
        if (p.n_hops == 0) {
            p.counts[b * 2 + 0] = p.n_seeds;
            p.counts[b * 2 + 1] = 0;
        }
    }
}

// boff = exclusive prefix of the batches' frontier lengths (n_batches <= 256: one workgroup, one pass)
__global__ void __launch_bounds__(256) flat_offsets_kernel(const FlatParams p) {
    __shared__ int64_t wave_tot[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t len = tid < p.n_batches ? p.st[tid].end - p.st[tid].begin : 0;
    int64_t incl = len;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int64_t u = __shfl_up(incl, off, 64);
        if (lane >= off) incl += u;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    int64_t carry = 0;
    for (int w = 0; w < wave; ++w) carry += wave_tot[w];
    if (tid < p.n_batches) p.boff[tid] = carry + incl - len;
    if (tid == p.n_batches - 1) p.boff[p.n_batches] = carry + incl;
}

__global__ void flat_frontier_kernel(const FlatParams p) {
    const int64_t m = p.n_batches * p.pitch, total = p.boff[p.n_batches];
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < m; j += (int64_t)gridDim.x * blockDim.x) {
        if (j >= total) { // the unused tail: the flat hops skip negative vertices
            p.vertices[j] = -1;
            p.ids[j] = 0;
            p.call_ids[j] = 0;
            p.fstates[j] = 0;
            continue;
        }
        int lo = 0, hi = (int)p.n_batches; // the batch whose slice holds j: largest b with boff[b] <= j
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (p.boff[mid] <= j)
                lo = mid;
            else
                hi = mid;
        }
        const int64_t b = lo;
        const int64_t slot = p.st[b].begin + (j - p.boff[b]);
        p.vertices[j] = p.samples[b * p.cap_nodes + slot];
        p.ids[j] = p.id_base + slot;
        p.call_ids[j] = (int64_t)(p.call_id + (uint64_t)b);
        p.fstates[j] = p.states ? p.states[b * p.cap_nodes + slot] : 0;
    }
}

__global__ void flat_emit_kernel(const FlatParams p) {
    const int64_t b = blockIdx.x;
    const FlatState st = p.st[b];
    const int64_t f0 = p.boff[b]; // the batch's slice of the flat frontier: [f0, boff[b + 1])
    const int64_t base = p.offsets[f0], tot = p.offsets[p.boff[b + 1]] - base;
    int64_t *samples = p.samples + b * p.cap_nodes, *rows = p.rows + b * p.cap_edges;
    int64_t *cols = p.cols + b * p.cap_edges, *eidx = p.edge_index + b * p.cap_edges;
    for (int64_t q = threadIdx.x; q < tot; q += blockDim.x) {
        const int64_t e = st.ne + q;
        samples[p.n_seeds + e] = p.neighbors[base + q];                  // :215
        rows[e] = p.n_seeds + e;                                         // :217
        cols[e] = st.begin + (p.parents[base + q] - f0);
        eidx[e] = p.edge_ptrs[base + q];
        if (p.states) p.states[b * p.cap_nodes + p.n_seeds + e] = p.states_out[base + q];
    }
    __syncthreads(); // every thread has read the state before it moves on
    if (threadIdx.x == 0) {
        int64_t *lo = p.layer_offsets + (b * p.n_hops + p.hop) * 3; // :193
        lo[0] = p.n_seeds + st.ne;
        lo[1] = st.ne;
        lo[2] = p.n_seeds + st.ne;
        const FlatState nx{st.end, p.n_seeds + st.ne + tot, st.ne + tot}; // :221-222
        p.st[b] = nx;
        if (p.hop == p.n_hops - 1) {
            p.counts[b * 2 + 0] = nx.end;
            p.counts[b * 2 + 1] = nx.ne;
        }
    }
}

struct FlatLayout {
    size_t st, status, boff, vertices, ids, call_ids, fstates, cnt, offsets, neighbors, edge_ptrs, parents, states_out, hop_ws, total;
    int64_t m_max, out_max, group_cap, hop_ws_bytes;
};

static int flat_layout(const tg_graph *csc, int64_t n_batches, int64_t n_seeds, const int64_t *fanout, int32_t n_hops,
                       bool weighted, FlatLayout *L) {
    int64_t layer = n_seeds, m_max = n_seeds, out_max = 0;
    int32_t kmax = 1;
    for (int h = 0; h < n_hops; ++h) {
        m_max = std::max(m_max, layer);
        layer *= fanout[h];
        out_max = std::max(out_max, layer);
        kmax = std::max<int32_t>(kmax, (int32_t)fanout[h]);
    }
    L->m_max = n_batches * m_max;
    L->out_max = n_batches * out_max;
    // the frontier's columns cut into 512-edge groups.  Every column at least one and all edges once over is NOT enough for
    // frontiers drawn by degree (hop 2 of [15, 10] on RMAT-24 repeats hubs: ~10 groups per slot), so 16 per slot are provided
    // for (at most 2^26 groups: 64 bytes each in the weighted form); beyond that the status word falls back to the
    // per-batch kernel
    L->group_cap = std::max<int64_t>({1024, csc->n_edges / 512 + 2 * L->m_max + 2,
                                      std::min<int64_t>(16 * L->m_max, (int64_t)1 << 26)});
    int64_t hb = 0;
    const int rc = weighted ? tg_ns_hop_weighted_workspace_bytes(L->m_max, kmax, L->group_cap, &hb)
                            : tg_ns_hop_scan_workspace_bytes(L->m_max, kmax, L->group_cap, &hb);
    if (rc != TG_OK) return rc;
    L->hop_ws_bytes = hb;
    size_t at = 0;
    auto take = [&](size_t bytes) {
        const size_t here = at;
        at += (bytes + 255) & ~(size_t)255;
        return here;
    };
    L->st = take((size_t)n_batches * sizeof(FlatState));
    L->status = take(256);
    L->boff = take((size_t)(n_batches + 1) * 8);
    const size_t m8 = (size_t)L->m_max * 8, o8 = (size_t)L->out_max * 8;
    L->vertices = take(m8);
    L->ids = take(m8);
    L->call_ids = take(m8);
    L->fstates = take(m8);
    L->cnt = take(m8);
    L->offsets = take(m8 + 8);
    L->neighbors = take(o8);
    L->edge_ptrs = take(o8);
    L->parents = take(o8);
    L->states_out = take(o8);
    L->hop_ws = take((size_t)hb);
    L->total = at;
    return TG_OK;
}

} // namespace tg

// measured on RMAT-24, 1 024 seeds, [15, 10] (profiles/r04/flat_scan_*.json; per-batch workgroups -> this path): weighted 8 /
// 64 / 256 batches 47 -> 2.2, 53 -> 15.2, 117 -> 57 ms; temporal filter 5.2 -> 0.55, 5.5 -> 2.3, 10.4 -> 6.9 ms; at 1 024
// batches the per-batch kernel is the faster form again (157 against 220 ms weighted, 19 against 29 ms filtered)
#define TG_NS_FLAT_MAX_BATCHES 256
#define TG_NS_FLAT_MAX_BATCHES_FILTER 256

static bool flat_applies(const tg_graph *csc, int64_t n_batches, int64_t n_seeds, const int64_t *fanout, int32_t n_hops,
                         const tg_ns_config *cfg) {
    if (!cfg || !csc) return false;
    const bool weighted = cfg->sampler == TG_SAMPLER_WEIGHTED, filtered = cfg->filter_mode != TG_FILTER_NONE;
    if (!weighted && !filtered) return false;
    if (cfg->seed_ids || cfg->seed_call_ids) return false;
    if (n_batches < 1 || n_batches > (weighted ? TG_NS_FLAT_MAX_BATCHES : TG_NS_FLAT_MAX_BATCHES_FILTER) || n_seeds < 1 ||
        n_hops < 1)
        return false;
    for (int h = 0; h < n_hops; ++h)
        if (fanout[h] < 1 || fanout[h] > 64) return false; // the per-batch kernel's bound: it is the fall-back
    return true;
}

extern "C" int tg_ns_homo_batched_workspace_bytes(const tg_graph *csc, int64_t n_batches, int64_t n_seeds,
                                                  const int64_t *fanout, int32_t n_hops, const tg_ns_config *cfg,
                                                  int64_t *n_bytes) {
    TG_REQUIRE(csc && n_bytes && n_batches >= 0 && n_seeds >= 0 && n_hops >= 0 && n_hops <= TG_MAX_HOPS &&
                   (fanout || n_hops == 0),
               "tg_ns_homo_batched_workspace_bytes: bad arguments");
    if (flat_applies(csc, n_batches, n_seeds, fanout, n_hops, cfg)) {
        tg::FlatLayout L;
        const int rc = tg::flat_layout(csc, n_batches, n_seeds, fanout, n_hops, cfg->sampler == TG_SAMPLER_WEIGHTED, &L);
        if (rc != TG_OK) return rc;
        *n_bytes = (int64_t)L.total;
        return TG_OK;
    }
    const bool scanning = cfg && (cfg->sampler == TG_SAMPLER_WEIGHTED || cfg->filter_mode != TG_FILTER_NONE);
    if (scanning) { // many batches: one workgroup per batch, no workspace
        *n_bytes = 0;
        return TG_OK;
    }
    return tg_ns_homo_workspace_bytes_for(csc, n_batches, n_seeds, fanout, n_hops, n_bytes);
}

// -> 1 when the launch took the flat path (the caller returns), 0 when it does not apply (the caller goes on), < 0 never;
// errors come back through *rc
int tg_ns_homo_flat_launch(const tg_graph *csc, const int64_t *seeds, int64_t n_batches, int64_t n_seeds,
                           const int64_t *fanout, int32_t n_hops, const tg_ns_config *cfg, const tg_rng *rng,
                           const tg_ns_out *out, void *ws, int64_t ws_bytes, int32_t mode, hipStream_t stream, int *rc_out) {
    using namespace tg;
    *rc_out = TG_OK;
    if (!ws || mode == TG_NS_FORM_FUSED || !flat_applies(csc, n_batches, n_seeds, fanout, n_hops, cfg)) return 0;
    const bool weighted = cfg->sampler == TG_SAMPLER_WEIGHTED, filtered = cfg->filter_mode != TG_FILTER_NONE;
    if ((weighted && !csc->weights) || (filtered && (!csc->timestamps || !cfg->seeds_state || !out->states))) return 0; // the per-batch launch reports it
    FlatLayout L;
    if (flat_layout(csc, n_batches, n_seeds, fanout, n_hops, weighted, &L) != TG_OK || ws_bytes < (int64_t)L.total ||
        ((uintptr_t)ws & 255) != 0)
        return 0;
    auto run = [&]() -> int {
        unsigned char *w = static_cast<unsigned char *>(ws);
        FlatParams p;
        p.seeds = seeds;
        p.seeds_state = cfg->seeds_state;
        p.n_seeds = n_seeds;
        p.n_batches = n_batches;
        p.cap_nodes = out->cap_nodes;
        p.cap_edges = out->cap_edges;
        p.id_base = cfg->id_base;
        p.n_hops = n_hops;
        p.call_id = rng->call_id;
        p.samples = out->samples;
        p.rows = out->rows;
        p.cols = out->cols;
        p.edge_index = out->edge_index;
        p.layer_offsets = out->layer_offsets;
        p.counts = out->counts;
        p.states = filtered ? out->states : nullptr;
        p.st = reinterpret_cast<FlatState *>(w + L.st);
        int32_t *status = reinterpret_cast<int32_t *>(w + L.status);
        p.boff = reinterpret_cast<int64_t *>(w + L.boff);
        p.vertices = reinterpret_cast<int64_t *>(w + L.vertices);
        p.ids = reinterpret_cast<int64_t *>(w + L.ids);
        p.call_ids = reinterpret_cast<int64_t *>(w + L.call_ids);
        p.fstates = reinterpret_cast<int64_t *>(w + L.fstates);
        p.cnt = reinterpret_cast<int64_t *>(w + L.cnt);
        p.offsets = reinterpret_cast<int64_t *>(w + L.offsets);
        p.neighbors = reinterpret_cast<int64_t *>(w + L.neighbors);
        p.edge_ptrs = reinterpret_cast<int64_t *>(w + L.edge_ptrs);
        p.parents = reinterpret_cast<int64_t *>(w + L.parents);
        p.states_out = reinterpret_cast<int64_t *>(w + L.states_out);
        TG_HIP(hipMemsetAsync(status, 0, sizeof(int32_t), stream));
        hipLaunchKernelGGL(flat_begin_kernel, dim3((unsigned)n_batches), dim3(256), 0, stream, p);
        TG_LAUNCH_CHECK();
        int64_t pitch = n_seeds;
        for (int h = 0; h < n_hops; ++h) {
            p.hop = h;
            p.pitch = pitch;
            const int64_t m = n_batches * pitch;
            hipLaunchKernelGGL(flat_offsets_kernel, dim3(1), dim3(256), 0, stream, p);
            hipLaunchKernelGGL(flat_frontier_kernel, dim3((unsigned)std::min<int64_t>((m + 255) / 256, 8192)), dim3(256), 0,
                               stream, p);
            TG_LAUNCH_CHECK();
            tg_hop_in in{};
            in.vertices = p.vertices;
            in.ids = p.ids;
            in.call_ids = p.call_ids;
            in.m = m;
            in.id_base = 0;
            in.fanout = (int32_t)fanout[h];
            in.sampler = weighted ? TG_SAMPLER_UNIFORM : cfg->sampler;
            in.rng_tag = cfg->rng_tag;
            tg_hop_out ho{p.cnt, p.offsets, p.neighbors, p.edge_ptrs, p.parents};
            tg_hop_filter flt{};
            flt.filter_mode = cfg->filter_mode;
            flt.forward = cfg->forward;
            flt.win_lo = cfg->win_lo;
            flt.win_hi = cfg->win_hi;
            flt.states = p.fstates;
            int rc;
            if (weighted)
                rc = tg_ns_hop_weighted_groups(csc, &in, filtered ? &flt : nullptr, rng, &ho, p.states_out, status, w + L.hop_ws,
                                               L.hop_ws_bytes, L.group_cap, stream);
            else
                rc = tg_ns_hop_scan(csc, &in, &flt, rng, &ho, p.states_out, status, w + L.hop_ws, L.hop_ws_bytes, L.group_cap,
                                    stream);
            if (rc != TG_OK) return rc;
            hipLaunchKernelGGL(flat_emit_kernel, dim3((unsigned)n_batches), dim3(512), 0, stream, p);
            TG_LAUNCH_CHECK();
            pitch *= fanout[h];
        }
        // the per-batch kernel behind it: returns at once unless the status word says the flat path could not finish
        return tg_ns_homo_filtered_launch_if(csc, seeds, n_batches, n_seeds, fanout, n_hops, cfg, rng, out, status, stream);
    };
    *rc_out = run();
    return 1;
}
