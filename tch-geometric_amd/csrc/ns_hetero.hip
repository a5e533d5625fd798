// neighbor_sampling_heterogenous in ONE launch (reference: src/algo/neighbor_sampling.rs:233-356; binding
// python.rs:275-395) for the unweighted, unfiltered samplers -- the default of the operator surface.
//
// One workgroup owns one seed batch and walks the reference's loop nest as it stands: hops outside, relations in the
// caller's `edge_types` order inside (:292-294); for relation (src, rel, dst) the frontier is the slice of dst's
// sample list that existed when the hop started (:288-290, :345-348), every frontier vertex draws <= k in-neighbours
// of type src from the relation's CSC, and the new samples are appended to src's list while (row = new index in
// src's list, col = frontier slot in dst's list, edge pointer) go to the relation's edge lists (:333-341).  All
// list lengths live in LDS, so no size ever travels to the host between steps: the host-driven form of this operator
// pays one launch and one read-back per (hop, relation).  Per step the work is that of ns_homo.hip: lane = frontier
// vertex counts -> LDS scan -> reservoir-by-tickets positions staged in OUTPUT order -> coalesced, pipelined emit.
// Draw address: tag = NS_HETERO | relation << 8, id = slot of the frontier vertex in dst's list, call id = batch's.
#include "ns_tickets.h"
#include "tg_device.h"
#include "tg_host.h"

namespace tg {

constexpr int HET_CHUNKS_PER_ROUND = 1024;
constexpr int HET_EMIT = 4;

struct HetRel {
    const int64_t *ptrs;
    const int64_t *indices;
    int64_t *rows, *cols, *eidx; // [n_batches * cap_edges]
    int64_t cap_edges;
    int32_t src, dst;
    int32_t fanout[TG_MAX_HOPS]; // 0 = relation not sampled in that hop
};
struct HetType {
    const int64_t *inputs; // [n_batches * n_inputs] or null
    int64_t *samples;      // [n_batches * cap_nodes]
    int64_t n_inputs, cap_nodes;
};
struct NsHetParams {
    HetRel rel[TG_HET_MAX_RELS];
    HetType type[TG_HET_MAX_TYPES];
    int32_t n_types, n_rels, n_hops, kmax;
    int64_t *layer_offsets; // [n_batches * n_rels * n_hops * 3]
    int64_t *counts;        // [n_batches * (n_types + n_rels)]
    uint64_t seed, call_id;
};

__host__ __device__ inline size_t het_wave_lds_bytes(int kmax) {
    return 64 * sizeof(int64_t) + (size_t)64 * kmax * sizeof(uint32_t) + (((size_t)64 * kmax + 15) & ~(size_t)15);
}
__host__ __device__ inline size_t het_block_lds_bytes(int kmax, int n_waves) {
    return (((size_t)(HET_CHUNKS_PER_ROUND + 1) * sizeof(uint32_t) + 15) & ~(size_t)15) +
           (size_t)n_waves * het_wave_lds_bytes(kmax);
}

template <int KMAX, bool REPLACE>
__global__ void ns_hetero_kernel(const NsHetParams p) {
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ int64_t len[TG_HET_MAX_TYPES], fbegin[TG_HET_MAX_TYPES], fend[TG_HET_MAX_TYPES];
    __shared__ int64_t ne_rel[TG_HET_MAX_RELS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n_waves = blockDim.x >> 6;
    const int64_t b = blockIdx.x;

    uint32_t *chunk_off = reinterpret_cast<uint32_t *>(smem);
    unsigned char *wbase = smem + ((((size_t)(HET_CHUNKS_PER_ROUND + 1) * sizeof(uint32_t)) + 15) & ~(size_t)15) +
                           (size_t)wave * het_wave_lds_bytes(p.kmax);
    int64_t *ebase = reinterpret_cast<int64_t *>(wbase);
    uint32_t *spos = reinterpret_cast<uint32_t *>(wbase + 64 * sizeof(int64_t));
    uint8_t *slane = reinterpret_cast<uint8_t *>(wbase + 64 * sizeof(int64_t) + (size_t)64 * p.kmax * sizeof(uint32_t));

    for (int t = 0; t < p.n_types; ++t) { // :264-278
        const HetType &ty = p.type[t];
        int64_t *s = ty.samples + b * ty.cap_nodes;
        for (int64_t i = tid; i < ty.n_inputs; i += blockDim.x) s[i] = ty.inputs[b * ty.n_inputs + i];
    }
    if (tid < p.n_types) {
        len[tid] = p.type[tid].n_inputs;
        fbegin[tid] = 0; // :288-290
        fend[tid] = p.type[tid].n_inputs;
    }
    if (tid < p.n_rels) ne_rel[tid] = 0;
    __syncthreads();

    for (int h = 0; h < p.n_hops; ++h) {   // :292
        for (int r = 0; r < p.n_rels; ++r) { // :294 in the caller's relation order
            const HetRel &rl = p.rel[r];
            const int k = rl.fanout[h];
            if (k == 0) continue;
            const int s_t = rl.src, d_t = rl.dst;
            const int64_t begin = fbegin[d_t], end = fend[d_t];
            const int64_t n_src0 = len[s_t], ne0 = ne_rel[r];
            if (tid == 0) { // :314-315
                int64_t *lo = p.layer_offsets + ((b * p.n_rels + r) * p.n_hops + h) * 3;
                lo[0] = n_src0;
                lo[1] = ne0;
                lo[2] = len[d_t];
            }
            const int64_t *front = p.type[d_t].samples + b * p.type[d_t].cap_nodes;
            int64_t *out_s = p.type[s_t].samples + b * p.type[s_t].cap_nodes;
            int64_t *rows = rl.rows + b * rl.cap_edges, *cols = rl.cols + b * rl.cap_edges, *eidx = rl.eidx + b * rl.cap_edges;
            const CallKey ck = call_key(p.seed, p.call_id + (uint64_t)b, TAG_NS_HETERO | ((uint32_t)r << 8));
            int64_t ne = 0; // edges of this (hop, relation) so far
            for (int64_t round_begin = begin; round_begin < end; round_begin += (int64_t)HET_CHUNKS_PER_ROUND * 64) {
                const int64_t round_end = min(end, round_begin + (int64_t)HET_CHUNKS_PER_ROUND * 64);
                const int nc = (int)((round_end - round_begin + 63) >> 6);
                for (int c = wave; c < nc; c += n_waves) { // pass A: per-chunk counts
                    const int64_t i = round_begin + (int64_t)c * 64 + lane;
                    uint32_t cnt = 0;
                    if (i < round_end) {
                        const int64_t w = front[i];
                        const int64_t deg = rl.ptrs[w + 1] - rl.ptrs[w];
                        cnt = (deg <= 0) ? 0u : (REPLACE ? (uint32_t)k : (uint32_t)min(deg, (int64_t)k));
                    }
                    const uint32_t tot = wave_sum(cnt);
                    if (lane == 0) chunk_off[c] = tot;
                }
                __syncthreads();
                if (wave == 0) { // scan of the chunk totals
                    uint32_t carry = 0;
                    for (int c0 = 0; c0 < nc; c0 += 64) {
                        const uint32_t v = (c0 + lane < nc) ? chunk_off[c0 + lane] : 0u;
                        const uint32_t incl = wave_inclusive_scan(v);
                        if (c0 + lane < nc) chunk_off[c0 + lane] = carry + incl - v;
                        carry += __shfl(incl, 63, 64);
                    }
                    if (lane == 0) chunk_off[nc] = carry;
                }
                __syncthreads();
                for (int c = wave; c < nc; c += n_waves) { // pass B: sample, stage in output order, emit
                    const int64_t i0 = round_begin + (int64_t)c * 64;
                    const int64_t i = i0 + lane;
                    int64_t e0 = 0, deg = 0;
                    if (i < round_end) {
                        const int64_t w = front[i];
                        e0 = rl.ptrs[w];
                        deg = rl.ptrs[w + 1] - e0;
                    }
                    const uint32_t cnt = (deg <= 0) ? 0u : (REPLACE ? (uint32_t)k : (uint32_t)min(deg, (int64_t)k));
                    const uint64_t did = (uint64_t)i; // slot of the frontier vertex in dst's list (:318)
                    const uint32_t incl = wave_inclusive_scan(cnt);
                    const uint32_t excl = incl - cnt;
                    const uint32_t total = __shfl(incl, 63, 64);
                    ebase[lane] = e0;
                    if (cnt > 0) {
                        const uint32_t n = (uint32_t)deg;
                        if (REPLACE) { // sampling.rs:57-69
                            sample_replace_any(ck, did, n, k, spos, slane, excl, lane);
                        } else if (deg <= k) { // sampling.rs:12-15
                            for (uint32_t s = 0; s < cnt; ++s) {
                                spos[excl + s] = s;
                                slane[excl + s] = (uint8_t)lane;
                            }
                        } else {
                            sample_tickets<KMAX>(ck, did, n, k, spos, slane, excl, lane);
                        }
                    }
                    wave_lds_handoff();
                    if (total > 0) {
                        const int64_t e_chunk = ne + (int64_t)chunk_off[c];
                        struct Batch {
                            int l[HET_EMIT];
                            int64_t ep[HET_EMIT], v[HET_EMIT];
                        };
                        auto issue = [&](Batch &t, uint32_t q0) { // unconditional gathers (see ns_homo.hip emit_chunk)
#pragma unroll
                            for (int u = 0; u < HET_EMIT; ++u) {
                                const uint32_t q = q0 + (uint32_t)(u * 64 + lane);
                                const uint32_t qq = q < total ? q : 0u;
                                t.l[u] = slane[qq];
                                t.ep[u] = ebase[t.l[u]] + (int64_t)spos[qq];
                            }
#pragma unroll
                            for (int u = 0; u < HET_EMIT; ++u) t.v[u] = __builtin_nontemporal_load(&rl.indices[t.ep[u]]);
                        };
                        auto store = [&](const Batch &t, uint32_t q0) {
#pragma unroll
                            for (int u = 0; u < HET_EMIT; ++u) {
                                const uint32_t q = q0 + (uint32_t)(u * 64 + lane);
                                if (q < total) {
                                    const int64_t e = e_chunk + q;
                                    out_s[n_src0 + e] = t.v[u];                                       // :338
                                    __builtin_nontemporal_store(n_src0 + e, &rows[ne0 + e]);          // :335,340 j
                                    __builtin_nontemporal_store(i0 + (int64_t)t.l[u], &cols[ne0 + e]); // :340 i
                                    __builtin_nontemporal_store(t.ep[u], &eidx[ne0 + e]);
                                }
                            }
                        };
                        Batch ba, bb;
                        issue(ba, 0u);
                        for (uint32_t q0 = 0; q0 < total; q0 += 2u * 64u * HET_EMIT) {
                            issue(bb, q0 + 64u * HET_EMIT);
                            store(ba, q0);
                            issue(ba, q0 + 2u * 64u * HET_EMIT);
                            store(bb, q0 + 64u * HET_EMIT);
                        }
                    }
                    wave_lds_handoff();
                }
                __syncthreads();
                ne += chunk_off[nc];
                __syncthreads(); // chunk_off is rewritten by the next round
            }
            if (tid == 0) {
                len[s_t] = n_src0 + ne;
                ne_rel[r] = ne0 + ne;
            }
            __syncthreads(); // the next relation reads len / ne_rel and may read the samples just written
        }
        if (tid < p.n_types) { // :345-348
            fbegin[tid] = fend[tid];
            fend[tid] = len[tid];
        }
        __syncthreads();
    }
    int64_t *cnt_out = p.counts + b * (p.n_types + p.n_rels);
    if (tid < p.n_types) cnt_out[tid] = len[tid];
    if (tid < p.n_rels) cnt_out[p.n_types + tid] = ne_rel[tid];
}

template <int KMAX, bool REPLACE> static int launch_hetero(const NsHetParams &p, int64_t n_batches, hipStream_t stream) {
    int threads = (n_batches < 512) ? 1024 : 512;
    while (threads > 64 && het_block_lds_bytes(p.kmax, threads / 64) > 60 * 1024) threads >>= 1;
    const size_t lds = het_block_lds_bytes(p.kmax, threads / 64);
    hipLaunchKernelGGL((ns_hetero_kernel<KMAX, REPLACE>), dim3((unsigned)n_batches), dim3(threads), lds, stream, p);
    TG_LAUNCH_CHECK();
    return TG_OK;
}

} // namespace tg

static int het_check_problem(const tg_het_problem *pb) {
    TG_REQUIRE(pb, "tg_ns_hetero: null problem");
    TG_REQUIRE(pb->n_types >= 1 && pb->n_types <= TG_HET_MAX_TYPES, "tg_ns_hetero: %d node types (supported: 1..%d)",
               pb->n_types, TG_HET_MAX_TYPES);
    TG_REQUIRE(pb->n_rels >= 0 && pb->n_rels <= TG_HET_MAX_RELS, "tg_ns_hetero: %d relations (supported: 0..%d)",
               pb->n_rels, TG_HET_MAX_RELS);
    TG_REQUIRE(pb->n_hops >= 0 && pb->n_hops <= TG_MAX_HOPS, "tg_ns_hetero: n_hops %d outside [0, %d]", pb->n_hops,
               TG_MAX_HOPS);
    TG_REQUIRE(pb->n_inputs && (pb->n_rels == 0 || (pb->rel_src && pb->rel_dst && pb->graphs && pb->fanout)),
               "tg_ns_hetero: null problem arrays");
    TG_REQUIRE(pb->sampler == TG_SAMPLER_UNIFORM || pb->sampler == TG_SAMPLER_UNIFORM_REPL,
               "tg_ns_hetero: only the unweighted samplers run fused (drive tg_ns_homo_batched per relation otherwise)");
    for (int t = 0; t < pb->n_types; ++t) TG_REQUIRE(pb->n_inputs[t] >= 0, "tg_ns_hetero: negative input count");
    for (int r = 0; r < pb->n_rels; ++r) {
        TG_REQUIRE(pb->rel_src[r] >= 0 && pb->rel_src[r] < pb->n_types && pb->rel_dst[r] >= 0 &&
                       pb->rel_dst[r] < pb->n_types,
                   "tg_ns_hetero: relation %d names a node type outside [0, %d)", r, pb->n_types);
        for (int h = 0; h < pb->n_hops; ++h) {
            const int64_t k = pb->fanout[(size_t)r * pb->n_hops + h];
            TG_REQUIRE(k >= 0 && k <= TG_MAX_FANOUT, "tg_ns_hetero: fanout %lld of relation %d outside [0, %d]",
                       (long long)k, r, TG_MAX_FANOUT);
        }
    }
    return TG_OK;
}

extern "C" int tg_ns_hetero_capacity(const tg_het_problem *pb, int64_t *cap_nodes, int64_t *cap_edges) {
    const int rc = het_check_problem(pb);
    if (rc != TG_OK) return rc;
    TG_REQUIRE(cap_nodes && (cap_edges || pb->n_rels == 0), "tg_ns_hetero_capacity: null outputs");
    int64_t front[TG_HET_MAX_TYPES], fresh[TG_HET_MAX_TYPES];
    for (int t = 0; t < pb->n_types; ++t) {
        front[t] = pb->n_inputs[t];
        cap_nodes[t] = pb->n_inputs[t];
    }
    for (int r = 0; r < pb->n_rels; ++r) cap_edges[r] = 0;
    for (int h = 0; h < pb->n_hops; ++h) {
        for (int t = 0; t < pb->n_types; ++t) fresh[t] = 0;
        for (int r = 0; r < pb->n_rels; ++r) {
            const int64_t k = pb->fanout[(size_t)r * pb->n_hops + h];
            const int64_t f = front[pb->rel_dst[r]];
            TG_REQUIRE(k == 0 || f <= (INT64_MAX / 8) / k, "tg_ns_hetero_capacity: capacity overflows int64");
            cap_edges[r] += f * k;
            fresh[pb->rel_src[r]] += f * k;
        }
        for (int t = 0; t < pb->n_types; ++t) {
            front[t] = fresh[t];
            cap_nodes[t] += fresh[t];
        }
    }
    return TG_OK;
}

extern "C" int tg_ns_hetero_batched(const tg_het_problem *pb, int64_t n_batches, const tg_rng *rng, const tg_het_out *out,
                                    void *stream) {
    int rc = het_check_problem(pb);
    if (rc != TG_OK) return rc;
    TG_REQUIRE(rng && out && n_batches >= 0 && n_batches <= 0x7fffffff, "tg_ns_hetero_batched: bad arguments");
    TG_REQUIRE(out->samples && out->cap_nodes && out->counts && (pb->n_rels == 0 || pb->n_hops == 0 || out->layer_offsets),
               "tg_ns_hetero_batched: null output tables");
    int64_t need_nodes[TG_HET_MAX_TYPES], need_edges[TG_HET_MAX_RELS > 0 ? TG_HET_MAX_RELS : 1];
    rc = tg_ns_hetero_capacity(pb, need_nodes, need_edges);
    if (rc != TG_OK) return rc;
    if (n_batches == 0) return TG_OK;
    tg::NsHetParams p;
    p.n_types = pb->n_types;
    p.n_rels = pb->n_rels;
    p.n_hops = pb->n_hops;
    p.kmax = 1;
    for (int t = 0; t < pb->n_types; ++t) {
        TG_REQUIRE(out->cap_nodes[t] >= need_nodes[t], "tg_ns_hetero_batched: samples slab of type %d too small (%lld < %lld)",
                   t, (long long)out->cap_nodes[t], (long long)need_nodes[t]);
        TG_REQUIRE(out->samples[t] || out->cap_nodes[t] == 0, "tg_ns_hetero_batched: null samples slab of type %d", t);
        TG_REQUIRE(pb->n_inputs[t] == 0 || (pb->inputs && pb->inputs[t]), "tg_ns_hetero_batched: null inputs of type %d", t);
        p.type[t].inputs = pb->n_inputs[t] ? pb->inputs[t] : nullptr;
        p.type[t].samples = out->samples[t];
        p.type[t].n_inputs = pb->n_inputs[t];
        p.type[t].cap_nodes = out->cap_nodes[t];
    }
    for (int r = 0; r < pb->n_rels; ++r) {
        const tg_graph &g = pb->graphs[r];
        TG_REQUIRE(g.ptrs && (g.indices || g.n_edges == 0), "tg_ns_hetero_batched: null graph of relation %d", r);
        TG_REQUIRE(out->cap_edges[r] >= need_edges[r], "tg_ns_hetero_batched: edge slabs of relation %d too small", r);
        TG_REQUIRE(need_edges[r] == 0 || (out->rows[r] && out->cols[r] && out->edge_index[r]),
                   "tg_ns_hetero_batched: null edge slabs of relation %d", r);
        tg::HetRel &rl = p.rel[r];
        rl.ptrs = g.ptrs;
        rl.indices = g.indices;
        rl.rows = out->rows[r];
        rl.cols = out->cols[r];
        rl.eidx = out->edge_index[r];
        rl.cap_edges = out->cap_edges[r];
        rl.src = pb->rel_src[r];
        rl.dst = pb->rel_dst[r];
        for (int h = 0; h < TG_MAX_HOPS; ++h) rl.fanout[h] = 0;
        for (int h = 0; h < pb->n_hops; ++h) {
            rl.fanout[h] = (int32_t)pb->fanout[(size_t)r * pb->n_hops + h];
            if (rl.fanout[h] > p.kmax) p.kmax = rl.fanout[h];
        }
    }
    p.layer_offsets = out->layer_offsets;
    p.counts = out->counts;
    p.seed = rng->seed;
    p.call_id = rng->call_id;
    hipStream_t s = (hipStream_t)stream;
    const bool repl = pb->sampler == TG_SAMPLER_UNIFORM_REPL;
    if (p.kmax <= 16) return repl ? tg::launch_hetero<16, true>(p, n_batches, s) : tg::launch_hetero<16, false>(p, n_batches, s);
    return repl ? tg::launch_hetero<32, true>(p, n_batches, s) : tg::launch_hetero<32, false>(p, n_batches, s);
}
