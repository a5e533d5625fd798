// budget_sampling on gfx950 -- replaces the per-node work of src/algo/budget_sampling.rs:63-153 (reference;
// SURVEY.md 8(f) "next" row).  One WAVEFRONT per frontier node of the node type being expanded:
//   Budget::update (:81-122)  for every relation into the type, lanes take the first min(deg, 50) column
//                             entries, test the temporal filter, and append the admissible ones to the node's
//                             candidate list in LDS (ballot + popcount keep the reference's order);
//   Budget::sample (:137-151) reservoir over the list: everything when it is short, else the ticket chain
//                             (lane s owns slot s; displaced entries are found with ballots);
// and the chosen candidates are written at a fixed stride (node, slot) for the host to append in order.
// Launch-bound small work; no synchronisation.
#include "tg_device.h"
#include "tg_host.h"

namespace tg {

constexpr int BUD_MAX_NB = 50;   // MAX_NEIGHBORS budget_sampling.rs:10
constexpr int BUD_MAX_RELS = 16; // relations into one node type
constexpr int64_t BUD_NAN_TS = -1;
constexpr uint32_t TAG_BUDGET = 10u;

struct BudRel {
    const int64_t *ptrs, *indices, *ts;
    int32_t rel; // index in the caller's relation list
    int32_t _pad;
};
struct BudParams {
    BudRel rels[BUD_MAX_RELS];
    int32_t n_rels, k;
    int32_t filter_on, forward, relative, _pad;
    int64_t win_lo, win_hi; // half open (python.rs:541-548)
    const int64_t *nodes, *nodes_ts;
    int64_t n_front, id_base;
    uint64_t seed, call_id;
    uint32_t tag;
    int64_t *sel_v, *sel_ts, *sel_rel, *sel_i; // [n_front * k]; sel_rel < 0 = empty slot
};

__global__ void budget_layer_kernel(const BudParams p) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
    const int cap = BUD_MAX_NB * p.n_rels;
    int64_t *cv = reinterpret_cast<int64_t *>(smem) + (size_t)wave * 3 * cap; // node | timestamp | rel << 8 | index
    int64_t *ct = cv + cap, *cr = ct + cap;
    const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const CallKey ck = call_key(p.seed, p.call_id, p.tag);
    for (int64_t j = (int64_t)blockIdx.x * n_waves + wave; j < p.n_front; j += (int64_t)gridDim.x * n_waves) {
        const int64_t w = p.nodes[j], w_t = p.nodes_ts[j];
        uint32_t n = 0;
        for (int q = 0; q < p.n_rels; ++q) { // :81 relations into this type, caller's order
            const BudRel R = p.rels[q];
            const int64_t b = R.ptrs[w], len = R.ptrs[w + 1] - b;
            const int cnt = (int)min(len, (int64_t)BUD_MAX_NB); // :100 a prefix of the column, no draws
            int64_t v = 0, v_t = BUD_NAN_TS;
            bool ok = lane < cnt;
            if (ok) {
                v = R.indices[b + lane];
                v_t = R.ts ? R.ts[b + lane] : BUD_NAN_TS; // :103
                if (v_t == BUD_NAN_TS) v_t = w_t;         // :104-106
                if (p.filter_on && !(w_t == BUD_NAN_TS || v_t == BUD_NAN_TS)) { // :20-29
                    const int64_t x = p.forward ? (v_t - w_t) : -(v_t - w_t);
                    ok = p.win_lo <= x && x < p.win_hi;
                }
            }
            const uint64_t mask = __ballot(ok);
            if (ok) {
                const uint32_t at = n + (uint32_t)__popcll(mask & lt_mask);
                cv[at] = v;
                ct[at] = p.filter_on ? (p.relative ? w_t : v_t) : v_t; // :117-119
                cr[at] = ((int64_t)R.rel << 8) | lane;                  // :116 the index inside the column
            }
            n += (uint32_t)__popcll(mask);
        }
        wave_lds_handoff();
        // ---- Budget::sample: reservoir_sampling(rng, 0..n, idx[k])  (:137-138)
        const int k = p.k;
        uint32_t pos = (uint32_t)lane;
        const uint32_t cnt = min(n, (uint32_t)k);
        if (n > (uint32_t)k) { // reservoir by tickets, lane s owns slot s
            uint32_t myK = 0xffffffffu, myV = 0;
            Draw d;
            const uint64_t id = (uint64_t)(p.id_base + j);
            for (int s = 0; s < k; ++s) {
                const uint32_t m = (n - 1u) - (uint32_t)s;
                if ((s & 3) == 0) d = draw(ck, id, (uint32_t)(s >> 2), 0u);
                const uint32_t r = slot_draw_from(d, ck, id, (uint32_t)s, 0u, m), last = m - 1u;
                const uint64_t mr = __ballot(lane < s && myK == r);
                const uint64_t ml = __ballot(lane < s && myK == last);
                const uint32_t vr = __shfl(myV, mr ? 63 - __clzll((long long)mr) : 0, 64);
                const uint32_t vl = __shfl(myV, ml ? 63 - __clzll((long long)ml) : 0, 64);
                const uint32_t tr = mr ? vr : r, tl = ml ? vl : last;
                if (lane == s) {
                    myK = r;
                    myV = tl;
                    pos = (tr < n - (uint32_t)k) ? (uint32_t)k + tr : (uint32_t)s;
                }
            }
        }
        if (lane < k) {
            const int64_t o = j * k + lane;
            if ((uint32_t)lane < cnt) {
                const int64_t packed = cr[pos];
                p.sel_v[o] = cv[pos];
                p.sel_ts[o] = ct[pos];
                p.sel_rel[o] = packed >> 8;
                p.sel_i[o] = packed & 0xff;
            } else {
                p.sel_rel[o] = -1;
            }
        }
        wave_lds_handoff();
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The whole operator without host bookkeeping (tg_budget_sample).  Sample lists, their lengths and the frontier bounds
// live on the device; per (layer, node type) step the host only enqueues
//   step_select   the per-node work above, frontier bounds read from the device state;
//   step_count    one wavefront per 64 (node, slot) cells: how many chosen candidates of every relation into the type,
//                 and of every source node type, the chunk holds;
//   step_scan     one workgroup: exclusive scan of those counts over the chunks, per category;
//   step_scatter  the same wavefronts append their candidates at (list length + chunk base + rank inside the chunk):
//                 a stable multi-way partition, so every list receives its entries in (node, slot) order as the
//                 reference's loop does (budget_sampling.rs:230-236, :140-151);
//   step_advance  list lengths += totals.
constexpr int BUDF_MAX_TYPES = 8, BUDF_MAX_RELS = 16;
constexpr int BUDF_CATS = BUDF_MAX_TYPES + BUDF_MAX_RELS; // categories: [0, T) source types, [T, T + R) relations

struct BudState { // device
    int64_t len[BUDF_MAX_TYPES], fbegin[BUDF_MAX_TYPES], fend[BUDF_MAX_TYPES], ne[BUDF_MAX_RELS];
};
struct BudFused {
    BudRel rels[BUD_MAX_RELS]; // relations INTO the type being expanded
    int32_t rel_src[BUDF_MAX_RELS]; // by global relation index
    int32_t n_rels_in, k, type, n_types, n_rels;
    int32_t filter_on, forward, relative;
    int64_t win_lo, win_hi;
    uint64_t seed, call_id;
    BudState *st;
    int64_t *samples[BUDF_MAX_TYPES], *ts[BUDF_MAX_TYPES];
    int64_t *rows[BUDF_MAX_RELS], *cols[BUDF_MAX_RELS], *eidx[BUDF_MAX_RELS];
    int64_t *sel_v, *sel_ts, *sel_rel, *sel_i; // [max_front * k]
    int64_t *cnt, *base, *totals;              // [n_chunks_cap * BUDF_CATS], same, [BUDF_CATS]
    int64_t chunks_cap;
    int32_t n_active, active[BUDF_CATS]; // categories the current step can produce
};

struct BudInit {
    int64_t n[BUDF_MAX_TYPES];
};
__global__ void budf_init_kernel(BudState *st, int n_types, int n_rels, const BudInit in) {
    const int t = threadIdx.x;
    if (t < n_types) {
        st->len[t] = in.n[t];
        st->fbegin[t] = 0;
        st->fend[t] = in.n[t];
    }
    if (t < n_rels) st->ne[t] = 0;
}

__global__ void budf_select_kernel(const BudFused f) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
    const int cap = BUD_MAX_NB * (f.n_rels_in > 0 ? f.n_rels_in : 1);
    int64_t *cv = reinterpret_cast<int64_t *>(smem) + (size_t)wave * 3 * cap;
    int64_t *ct = cv + cap, *cr = ct + cap;
    const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const CallKey ck = call_key(f.seed, f.call_id, TAG_BUDGET | ((uint32_t)f.type << 8));
    const int64_t begin = f.st->fbegin[f.type], n_front = f.st->fend[f.type] - begin;
    const int64_t *nodes = f.samples[f.type] + begin, *nodes_ts = f.ts[f.type] + begin;
    for (int64_t j = (int64_t)blockIdx.x * n_waves + wave; j < n_front; j += (int64_t)gridDim.x * n_waves) {
        const int64_t w = nodes[j], w_t = nodes_ts[j];
        uint32_t n = 0;
        for (int q = 0; q < f.n_rels_in; ++q) { // :81 relations into this type, caller's order
            const BudRel R = f.rels[q];
            const int64_t b = R.ptrs[w], len = R.ptrs[w + 1] - b;
            const int cnt = (int)min(len, (int64_t)BUD_MAX_NB);
            int64_t v = 0, v_t = BUD_NAN_TS;
            bool ok = lane < cnt;
            if (ok) {
                v = R.indices[b + lane];
                v_t = R.ts ? R.ts[b + lane] : BUD_NAN_TS;
                if (v_t == BUD_NAN_TS) v_t = w_t;
                if (f.filter_on && !(w_t == BUD_NAN_TS || v_t == BUD_NAN_TS)) {
                    const int64_t x = f.forward ? (v_t - w_t) : -(v_t - w_t);
                    ok = f.win_lo <= x && x < f.win_hi;
                }
            }
            const uint64_t mask = __ballot(ok);
            if (ok) {
                const uint32_t at = n + (uint32_t)__popcll(mask & lt_mask);
                cv[at] = v;
                ct[at] = f.filter_on ? (f.relative ? w_t : v_t) : v_t;
                cr[at] = ((int64_t)R.rel << 8) | lane;
            }
            n += (uint32_t)__popcll(mask);
        }
        wave_lds_handoff();
        const int k = f.k;
        uint32_t pos = (uint32_t)lane;
        const uint32_t cnt = min(n, (uint32_t)k);
        if (n > (uint32_t)k) {
            uint32_t myK = 0xffffffffu, myV = 0;
            Draw d;
            const uint64_t id = (uint64_t)(begin + j); // slot of the node in its type's list
            for (int s = 0; s < k; ++s) {
                const uint32_t m = (n - 1u) - (uint32_t)s;
                if ((s & 3) == 0) d = draw(ck, id, (uint32_t)(s >> 2), 0u);
                const uint32_t r = slot_draw_from(d, ck, id, (uint32_t)s, 0u, m), last = m - 1u;
                const uint64_t mr = __ballot(lane < s && myK == r);
                const uint64_t ml = __ballot(lane < s && myK == last);
                const uint32_t vr = __shfl(myV, mr ? 63 - __clzll((long long)mr) : 0, 64);
                const uint32_t vl = __shfl(myV, ml ? 63 - __clzll((long long)ml) : 0, 64);
                const uint32_t tr = mr ? vr : r, tl = ml ? vl : last;
                if (lane == s) {
                    myK = r;
                    myV = tl;
                    pos = (tr < n - (uint32_t)k) ? (uint32_t)k + tr : (uint32_t)s;
                }
            }
        }
        if (lane < k) {
            const int64_t o = j * k + lane;
            if ((uint32_t)lane < cnt) {
                const int64_t packed = cr[pos];
                f.sel_v[o] = cv[pos];
                f.sel_ts[o] = ct[pos];
                f.sel_rel[o] = packed >> 8;
                f.sel_i[o] = packed & 0xff;
            } else {
                f.sel_rel[o] = -1;
            }
        }
        wave_lds_handoff();
    }
}

// rank of this lane among the lanes of its wavefront that hold the same category, and that category's count
__device__ __forceinline__ void budf_rank(int cat, uint64_t lt_mask, int lane, uint32_t &rank, uint32_t &count) {
    rank = 0;
    count = 0;
    uint64_t todo = __ballot(cat >= 0);
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int c = __shfl(cat, leader, 64);
        const uint64_t same = __ballot(cat == c);
        if (cat == c) {
            rank = (uint32_t)__popcll(same & lt_mask);
            count = (uint32_t)__popcll(same);
        }
        todo &= ~same;
    }
}

template <bool SCATTER> __global__ void budf_partition_kernel(const BudFused f) {
    __shared__ int64_t stage_all[4][BUDF_CATS];
    const int lane = threadIdx.x & 63;
    const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const int64_t begin = f.st->fbegin[f.type], n_front = f.st->fend[f.type] - begin;
    const int64_t n_cells = n_front * f.k, n_chunks = (n_cells + 63) >> 6;
    const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t c = wave0; c < n_chunks; c += n_waves) {
        const int64_t q = (c << 6) + lane;
        int rel = -1;
        if (q < n_cells) rel = (int)f.sel_rel[q];
        const int src = rel >= 0 ? f.rel_src[rel] : -1;
        uint32_t r_rank, r_cnt, s_rank, s_cnt;
        budf_rank(rel, lt_mask, lane, r_rank, r_cnt);
        budf_rank(src, lt_mask, lane, s_rank, s_cnt);
        if (!SCATTER) { // the chunk's counts per category, staged in LDS so that every cell is written once
            int64_t *stage = stage_all[threadIdx.x >> 6];
            if (lane < BUDF_CATS) stage[lane] = 0;
            wave_lds_handoff();
            if (rel >= 0 && r_rank == 0) stage[f.n_types + rel] = r_cnt;
            if (src >= 0 && s_rank == 0) stage[src] = s_cnt;
            wave_lds_handoff();
            if (lane < BUDF_CATS) f.cnt[c * BUDF_CATS + lane] = stage[lane];
            wave_lds_handoff();
        } else if (rel >= 0) {
            const int64_t *bs = f.base + c * BUDF_CATS;
            const int64_t i_new = f.st->len[src] + bs[src] + s_rank;             // :147
            f.samples[src][i_new] = f.sel_v[q];
            f.ts[src][i_new] = f.sel_ts[q];
            const int64_t e = f.st->ne[rel] + bs[f.n_types + rel] + r_rank;      // :150 push_edge(i, j, edge_ptr)
            f.rows[rel][e] = i_new;
            f.cols[rel][e] = begin + q / f.k;
            f.eidx[rel][e] = f.sel_i[q];
        }
    }
}

// one workgroup: base[c][cat] = sum of cnt[c'][cat] over c' < c; totals[cat]
__global__ void budf_scan_kernel(const BudFused f) {
    __shared__ int64_t wave_tot[16];
    __shared__ int64_t carry_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n_w = blockDim.x >> 6;
    const int64_t n_front = f.st->fend[f.type] - f.st->fbegin[f.type];
    const int64_t n_chunks = (n_front * f.k + 63) >> 6;
    for (int a = 0; a < f.n_active; ++a) {
        const int cat = f.active[a];
        if (tid == 0) carry_s = 0;
        __syncthreads();
        for (int64_t base = 0; base < n_chunks; base += blockDim.x) {
            const int64_t c = base + tid;
            const int64_t v = (c < n_chunks) ? f.cnt[c * BUDF_CATS + cat] : 0;
            const int64_t incl = wave_inclusive_scan(v);
            if (lane == 63) wave_tot[wave] = incl;
            __syncthreads();
            int64_t before = carry_s;
            for (int w = 0; w < wave; ++w) before += wave_tot[w];
            if (c < n_chunks) f.base[c * BUDF_CATS + cat] = before + incl - v;
            __syncthreads();
            if (tid == 0) {
                int64_t t = carry_s;
                for (int w = 0; w < n_w; ++w) t += wave_tot[w];
                carry_s = t;
            }
            __syncthreads();
        }
        if (tid == 0) f.totals[cat] = carry_s;
        __syncthreads();
    }
}

__global__ void budf_advance_kernel(const BudFused f) {
    const int a = threadIdx.x;
    if (a < f.n_active) {
        const int cat = f.active[a];
        if (cat < f.n_types)
            f.st->len[cat] += f.totals[cat];
        else
            f.st->ne[cat - f.n_types] += f.totals[cat];
    }
}
__global__ void budf_next_layer_kernel(BudState *st, int n_types) { // :240-243
    const int t = threadIdx.x;
    if (t < n_types) {
        st->fbegin[t] = st->fend[t];
        st->fend[t] = st->len[t];
    }
}
__global__ void budf_counts_kernel(const BudState *st, int n_types, int n_rels, int64_t *counts) {
    const int t = threadIdx.x;
    if (t < n_types) counts[t] = st->len[t];
    if (t < n_rels) counts[n_types + t] = st->ne[t];
}
__global__ void budf_copy_inputs_kernel(const int64_t *__restrict__ in, const int64_t *__restrict__ in_ts, int64_t n,
                                        int64_t *samples, int64_t *ts) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        samples[i] = in[i];
        ts[i] = in_ts ? in_ts[i] : BUD_NAN_TS; // :195
    }
}

struct BudfPlan {
    int64_t cap_nodes[BUDF_MAX_TYPES], cap_edges[BUDF_MAX_RELS], max_cells, chunks_cap;
    size_t bytes;
};
static int budf_plan(const tg_budget_problem *pb, BudfPlan &pl) {
    TG_REQUIRE(pb, "tg_budget: null problem");
    TG_REQUIRE(pb->n_types >= 1 && pb->n_types <= BUDF_MAX_TYPES && pb->n_rels >= 0 && pb->n_rels <= BUDF_MAX_RELS,
               "tg_budget: %d node types / %d relations (supported: <= %d / <= %d)", pb->n_types, pb->n_rels, BUDF_MAX_TYPES,
               BUDF_MAX_RELS);
    TG_REQUIRE(pb->n_hops >= 0 && pb->n_hops <= TG_MAX_HOPS && pb->n_inputs && pb->num_neighbors,
               "tg_budget: bad hops or null arrays");
    TG_REQUIRE(pb->n_rels == 0 || (pb->rel_src && pb->rel_dst && pb->graphs), "tg_budget: null relation arrays");
    int64_t front[BUDF_MAX_TYPES], fresh[BUDF_MAX_TYPES];
    pl.max_cells = 1;
    for (int t = 0; t < pb->n_types; ++t) {
        TG_REQUIRE(pb->n_inputs[t] >= 0, "tg_budget: negative input count");
        front[t] = pb->n_inputs[t];
        pl.cap_nodes[t] = front[t];
    }
    for (int r = 0; r < pb->n_rels; ++r) {
        TG_REQUIRE(pb->rel_src[r] >= 0 && pb->rel_src[r] < pb->n_types && pb->rel_dst[r] >= 0 && pb->rel_dst[r] < pb->n_types,
                   "tg_budget: relation %d names a node type outside [0, %d)", r, pb->n_types);
        pl.cap_edges[r] = 0;
    }
    for (int h = 0; h < pb->n_hops; ++h) {
        for (int t = 0; t < pb->n_types; ++t) fresh[t] = 0;
        for (int t = 0; t < pb->n_types; ++t) {
            const int64_t k = pb->num_neighbors[(size_t)t * pb->n_hops + h];
            TG_REQUIRE(k >= 0 && k <= 64, "tg_budget: num_neighbors %lld outside [0, 64]", (long long)k);
            TG_REQUIRE(k == 0 || front[t] <= (INT64_MAX / 16) / k, "tg_budget: capacity overflows int64");
            const int64_t cells = front[t] * k;
            if (cells > pl.max_cells) pl.max_cells = cells;
            bool seen[BUDF_MAX_TYPES] = {false};
            for (int r = 0; r < pb->n_rels; ++r)
                if (pb->rel_dst[r] == t) {
                    pl.cap_edges[r] += cells;           // a node's k picks may all come from one relation
                    if (!seen[pb->rel_src[r]]) fresh[pb->rel_src[r]] += cells;
                    seen[pb->rel_src[r]] = true;
                }
        }
        for (int t = 0; t < pb->n_types; ++t) {
            front[t] = fresh[t];
            pl.cap_nodes[t] += fresh[t];
        }
    }
    pl.chunks_cap = (pl.max_cells + 63) / 64;
    auto a256 = [](size_t b) { return (b + 255) & ~(size_t)255; };
    pl.bytes = a256(sizeof(BudState)) + 4 * a256(8 * (size_t)pl.max_cells) + 2 * a256(8 * (size_t)pl.chunks_cap * BUDF_CATS) +
               a256(8 * BUDF_CATS) + 256;
    return TG_OK;
}

} // namespace tg

extern "C" int tg_budget_capacity(const tg_budget_problem *pb, int64_t *cap_nodes, int64_t *cap_edges) {
    tg::BudfPlan pl;
    const int rc = tg::budf_plan(pb, pl);
    if (rc != TG_OK) return rc;
    TG_REQUIRE(cap_nodes && (cap_edges || pb->n_rels == 0), "tg_budget_capacity: null outputs");
    for (int t = 0; t < pb->n_types; ++t) cap_nodes[t] = pl.cap_nodes[t];
    for (int r = 0; r < pb->n_rels; ++r) cap_edges[r] = pl.cap_edges[r];
    return TG_OK;
}

extern "C" int tg_budget_workspace_bytes(const tg_budget_problem *pb, int64_t *bytes) {
    tg::BudfPlan pl;
    const int rc = tg::budf_plan(pb, pl);
    if (rc != TG_OK) return rc;
    TG_REQUIRE(bytes, "tg_budget_workspace_bytes: null output");
    *bytes = (int64_t)pl.bytes;
    return TG_OK;
}

extern "C" int tg_budget_sample(const tg_budget_problem *pb, const tg_rng *rng, const tg_budget_out *out, void *workspace,
                                int64_t workspace_bytes, void *stream_) {
    using namespace tg;
    BudfPlan pl;
    int rc = budf_plan(pb, pl);
    if (rc != TG_OK) return rc;
    TG_REQUIRE(rng && out && workspace && (size_t)workspace_bytes >= pl.bytes, "tg_budget_sample: null argument or workspace too small");
    TG_REQUIRE(out->samples && out->sample_ts && out->cap_nodes && out->counts, "tg_budget_sample: null output tables");
    hipStream_t stream = (hipStream_t)stream_;
    const int T = pb->n_types, R = pb->n_rels, H = pb->n_hops;
    unsigned char *base = reinterpret_cast<unsigned char *>(workspace);
    size_t off = 0;
    auto take = [&](size_t bytes) {
        unsigned char *q = base + off;
        off += (bytes + 255) & ~(size_t)255;
        return q;
    };
    BudFused f;
    f.st = reinterpret_cast<BudState *>(take(sizeof(BudState)));
    f.sel_v = reinterpret_cast<int64_t *>(take(8 * (size_t)pl.max_cells));
    f.sel_ts = reinterpret_cast<int64_t *>(take(8 * (size_t)pl.max_cells));
    f.sel_rel = reinterpret_cast<int64_t *>(take(8 * (size_t)pl.max_cells));
    f.sel_i = reinterpret_cast<int64_t *>(take(8 * (size_t)pl.max_cells));
    f.cnt = reinterpret_cast<int64_t *>(take(8 * (size_t)pl.chunks_cap * BUDF_CATS));
    f.base = reinterpret_cast<int64_t *>(take(8 * (size_t)pl.chunks_cap * BUDF_CATS));
    f.totals = reinterpret_cast<int64_t *>(take(8 * BUDF_CATS));
    TG_REQUIRE(off <= pl.bytes, "tg_budget_sample: internal workspace plan mismatch");
    f.chunks_cap = pl.chunks_cap;
    f.n_types = T;
    f.n_rels = R;
    f.filter_on = pb->filter_on;
    f.forward = pb->forward;
    f.relative = pb->relative;
    f.win_lo = pb->win_lo;
    f.win_hi = pb->win_hi;
    f.seed = rng->seed;
    f.call_id = rng->call_id;
    for (int t = 0; t < T; ++t) {
        TG_REQUIRE(out->cap_nodes[t] >= pl.cap_nodes[t], "tg_budget_sample: samples slab of type %d too small", t);
        TG_REQUIRE((out->samples[t] && out->sample_ts[t]) || pl.cap_nodes[t] == 0, "tg_budget_sample: null slab of type %d", t);
        f.samples[t] = out->samples[t];
        f.ts[t] = out->sample_ts[t];
    }
    for (int r = 0; r < R; ++r) {
        TG_REQUIRE(pb->graphs[r].ptrs, "tg_budget_sample: relation %d has no CSC", r);
        TG_REQUIRE(out->cap_edges[r] >= pl.cap_edges[r], "tg_budget_sample: edge slabs of relation %d too small", r);
        TG_REQUIRE(pl.cap_edges[r] == 0 || (out->rows[r] && out->cols[r] && out->edge_index[r]),
                   "tg_budget_sample: null edge slabs of relation %d", r);
        f.rel_src[r] = pb->rel_src[r];
        f.rows[r] = out->rows[r];
        f.cols[r] = out->cols[r];
        f.eidx[r] = out->edge_index[r];
    }
    // inputs -> the heads of the sample lists; their counts -> the device state
    BudInit init;
    for (int t = 0; t < BUDF_MAX_TYPES; ++t) init.n[t] = t < T ? pb->n_inputs[t] : 0;
    for (int t = 0; t < T; ++t)
        if (pb->n_inputs[t] > 0) {
            TG_REQUIRE(pb->inputs && pb->inputs[t], "tg_budget_sample: node type %d has inputs but no pointer", t);
            int64_t g = (pb->n_inputs[t] + 255) / 256;
            if (g > 4096) g = 4096;
            hipLaunchKernelGGL(budf_copy_inputs_kernel, dim3((unsigned)g), dim3(256), 0, stream, pb->inputs[t],
                               (pb->input_ts && pb->input_ts[t]) ? pb->input_ts[t] : (const int64_t *)nullptr, pb->n_inputs[t],
                               out->samples[t], out->sample_ts[t]);
        }
    hipLaunchKernelGGL(budf_init_kernel, dim3(1), dim3(64), 0, stream, f.st, T, R, init);
    int64_t front[BUDF_MAX_TYPES], fresh[BUDF_MAX_TYPES]; // worst-case frontier sizes: they only size the grids
    for (int t = 0; t < T; ++t) front[t] = pb->n_inputs[t];
    for (int h = 0; h < H; ++h) {
        for (int t = 0; t < T; ++t) fresh[t] = 0;
        for (int t = 0; t < T; ++t) { // :225 node_types order
            const int64_t k = pb->num_neighbors[(size_t)t * H + h];
            f.n_rels_in = 0;
            bool seen[BUDF_MAX_TYPES] = {false};
            for (int r = 0; r < R; ++r)
                if (pb->rel_dst[r] == t) {
                    f.rels[f.n_rels_in++] = BudRel{pb->graphs[r].ptrs, pb->graphs[r].indices, pb->graphs[r].timestamps, r, 0};
                    if (!seen[pb->rel_src[r]]) fresh[pb->rel_src[r]] += front[t] * k;
                    seen[pb->rel_src[r]] = true;
                }
            if (k == 0 || front[t] == 0 || f.n_rels_in == 0) continue;
            f.n_active = 0;
            for (int s_t = 0; s_t < T; ++s_t)
                if (seen[s_t]) f.active[f.n_active++] = s_t;
            for (int q = 0; q < f.n_rels_in; ++q) f.active[f.n_active++] = T + f.rels[q].rel;
            f.k = (int32_t)k;
            f.type = t;
            const int n_waves = 2;
            const size_t lds = (size_t)n_waves * 3 * BUD_MAX_NB * f.n_rels_in * sizeof(int64_t);
            int64_t blocks = (front[t] + n_waves - 1) / n_waves;
            if (blocks > 8192) blocks = 8192;
            hipLaunchKernelGGL(budf_select_kernel, dim3((unsigned)blocks), dim3(64 * n_waves), lds, stream, f);
            int64_t pblocks = ((front[t] * k + 63) / 64 + 3) / 4;
            if (pblocks > 4096) pblocks = 4096;
            hipLaunchKernelGGL(budf_partition_kernel<false>, dim3((unsigned)pblocks), dim3(256), 0, stream, f);
            hipLaunchKernelGGL(budf_scan_kernel, dim3(1), dim3(1024), 0, stream, f);
            hipLaunchKernelGGL(budf_partition_kernel<true>, dim3((unsigned)pblocks), dim3(256), 0, stream, f);
            hipLaunchKernelGGL(budf_advance_kernel, dim3(1), dim3(64), 0, stream, f);
        }
        hipLaunchKernelGGL(budf_next_layer_kernel, dim3(1), dim3(64), 0, stream, f.st, T);
        for (int t = 0; t < T; ++t) front[t] = fresh[t];
    }
    hipLaunchKernelGGL(budf_counts_kernel, dim3(1), dim3(64), 0, stream, f.st, T, R, out->counts);
    TG_LAUNCH_CHECK();
    return TG_OK;
}

extern "C" int tg_budget_layer(const tg_budget_layer_in *in, const tg_rng *rng, const tg_budget_layer_out *out,
                               void *stream) {
    using namespace tg;
    TG_REQUIRE(in && rng && out, "tg_budget_layer: null argument");
    TG_REQUIRE(in->n_rels >= 0 && in->n_rels <= BUD_MAX_RELS, "tg_budget_layer: %d relations into one node type (max %d)",
               in->n_rels, BUD_MAX_RELS);
    TG_REQUIRE(in->fanout >= 1 && in->fanout <= 64, "tg_budget_layer: num_neighbors %d outside [1, 64]", in->fanout);
    TG_REQUIRE(in->n_front >= 0, "tg_budget_layer: negative frontier");
    if (in->n_front == 0) return TG_OK;
    TG_REQUIRE(in->nodes && in->nodes_ts && out->sel_v && out->sel_ts && out->sel_rel && out->sel_i,
               "tg_budget_layer: null buffers");
    BudParams p;
    p.n_rels = in->n_rels;
    for (int q = 0; q < in->n_rels; ++q) {
        TG_REQUIRE(in->graphs[q].ptrs, "tg_budget_layer: relation %d has no CSC", q);
        p.rels[q] = BudRel{in->graphs[q].ptrs, in->graphs[q].indices, in->graphs[q].timestamps, in->rel_ids[q], 0};
    }
    p.k = in->fanout;
    p.filter_on = in->filter_on;
    p.forward = in->forward;
    p.relative = in->relative;
    p.win_lo = in->win_lo;
    p.win_hi = in->win_hi;
    p.nodes = in->nodes;
    p.nodes_ts = in->nodes_ts;
    p.n_front = in->n_front;
    p.id_base = in->id_base;
    p.seed = rng->seed;
    p.call_id = rng->call_id;
    p.tag = TAG_BUDGET | ((uint32_t)in->node_type << 8);
    p.sel_v = out->sel_v;
    p.sel_ts = out->sel_ts;
    p.sel_rel = out->sel_rel;
    p.sel_i = out->sel_i;
    const int n_waves = 2;
    const size_t lds = (size_t)n_waves * 3 * BUD_MAX_NB * (in->n_rels > 0 ? in->n_rels : 1) * sizeof(int64_t);
    int64_t blocks = (in->n_front + n_waves - 1) / n_waves;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(budget_layer_kernel, dim3((unsigned)blocks), dim3(64 * n_waves), lds, (hipStream_t)stream, p);
    TG_LAUNCH_CHECK();
    return TG_OK;
}
