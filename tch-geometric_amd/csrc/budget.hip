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
                if ((s & 1) == 0) d = draw(ck, id, (uint32_t)(s >> 1), 0u);
                const uint32_t r = bounded32(d.half(s & 1), m), last = m - 1u;
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

} // namespace tg

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
