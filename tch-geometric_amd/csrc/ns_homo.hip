// neighbor_sampling_homogenous on gfx950 -- replaces the hot loop of
// src/algo/neighbor_sampling.rs:188-223 (reference) for the unweighted,
// unfiltered samplers (UnweightedSampler<false|true> + IdentityFilter).
//
// Shape of the work.  The reference emits a forest without node dedup
// (neighbor_sampling.rs:212-217), so a batch's output positions follow from a
// prefix sum of per-vertex sample counts; no hash table is needed.  One
// workgroup owns one seed batch and walks its hops in order (hop h+1's
// frontier is hop h's output), so hops are separated by a workgroup barrier
// only -- no grid sync, no temporaries in HBM, and thousands of batches run
// side by side in one launch to cover HBM latency.
//
// Per hop and 64-vertex chunk of the frontier (one wavefront):
//   pass A  lane = frontier vertex: load ptrs[w], ptrs[w+1] -> count
//           min(deg,k) (or k with replacement); wave sum -> LDS chunk total
//   scan    wave 0 turns chunk totals into exclusive offsets (LDS)
//   pass B  lane = frontier vertex: draw its <=k neighbour positions from
//           counter-addressed Philox ("reservoir by tickets", k bounded draws,
//           register-resident shuffle list), stage them in LDS in OUTPUT order;
//           then the wave walks that staged slice with consecutive lanes:
//           gather indices[edge_ptr] and write samples/rows/cols/edge_index
//           fully coalesced.
// Algorithmic HBM bytes per hop (F frontier slots, S sampled edges):
//   reads 8F (frontier id) + 16F (ptrs pair) + 8S (gather), writes 32S.
#include <stdlib.h>

#include "ns_tickets.h"
#include "tg_device.h"
#include "tg_host.h"

namespace tg {

constexpr int NS_CHUNKS_PER_ROUND = 1024; // 64-vertex chunks whose totals fit the LDS scan array

struct NsHomoParams {
    const int64_t *ptrs;
    const int64_t *indices;
    const uint32_t *indices32;
    const uint32_t *ptrs32;
    const int64_t *seeds;
    int64_t n_seeds;
    int32_t n_hops;
    int32_t kmax; // max fan-out over hops (sizes the LDS staging)
    int32_t fanout[TG_MAX_HOPS];
    int64_t cap_nodes, cap_edges;
    int64_t *samples, *rows, *cols, *edge_index, *layer_offsets, *counts;
    uint64_t seed, call_id;
    uint32_t tag;
    int64_t id_base;
    const int64_t *seed_ids, *seed_call_ids; // remote-frontier mode (n_hops == 1)
    TG_BOUNDS_FIELDS
};

// per wave: edge base [64] i64 | staged positions [64*k] u32 | staged lanes [64*k] u8 | (big fan-outs only) the
// ticket strips [64 * 2k] u32
__host__ __device__ inline size_t ns_wave_lds_bytes(int kmax, bool strips) {
    return 64 * sizeof(int64_t) + (size_t)64 * kmax * sizeof(uint32_t) + (((size_t)64 * kmax + 15) & ~(size_t)15) +
           (strips ? (size_t)64 * 2 * kmax * sizeof(uint32_t) : 0);
}
__host__ __device__ inline size_t ns_block_lds_bytes(int kmax, int n_waves, bool strips) {
    return (((size_t)(NS_CHUNKS_PER_ROUND + 1) * sizeof(uint32_t) + 15) & ~(size_t)15) +
           (size_t)n_waves * ns_wave_lds_bytes(kmax, strips);
}

constexpr int NS_EMIT = 4; // gathers per lane and batch of the emit loop (two batches in flight)

// Emit of one 64-vertex chunk: the staged (lane, position) pairs are walked by consecutive lanes, the neighbour id is
// gathered and the four output columns are written coalesced (neighbor_sampling.rs:211-217).  The gathers are the
// kernel's HBM-latency exposure (each one is its own 128-byte line), so they are issued in batches of NS_EMIT per
// lane with the next batch in flight while the current one is stored, and unconditionally -- lanes past the end
// re-read element 0's address and skip the stores; a branch around a load would make the compiler wait for every
// load in flight (s_waitcnt vmcnt(0)).
template <typename IDX, bool NT>
__device__ __forceinline__ void emit_chunk(const IDX *__restrict__ idx, uint32_t total, int lane, const uint8_t *slane,
                                           const uint32_t *spos, const int64_t *ebase, int64_t e_chunk, int64_t i0,
                                           int64_t n_seeds, int64_t *samples, int64_t *rows, int64_t *cols,
                                           int64_t *eidx) {
    if (total == 0) return;
    struct Batch {
        int l[NS_EMIT];
        int64_t ep[NS_EMIT];
        IDX v[NS_EMIT];
    };
    auto issue = [&](Batch &t, uint32_t q0) {
#pragma unroll
        for (int u = 0; u < NS_EMIT; ++u) {
            const uint32_t q = q0 + (uint32_t)(u * 64 + lane);
            const uint32_t qq = q < total ? q : 0u;
            t.l[u] = slane[qq];
            t.ep[u] = ebase[t.l[u]] + (int64_t)spos[qq];
        }
#pragma unroll
        for (int u = 0; u < NS_EMIT; ++u) t.v[u] = __builtin_nontemporal_load(&idx[t.ep[u]]); // each line is used once
    };
    auto store = [&](const Batch &t, uint32_t q0) {
#pragma unroll
        for (int u = 0; u < NS_EMIT; ++u) {
            const uint32_t q = q0 + (uint32_t)(u * 64 + lane);
            if (q < total) {
                const int64_t e = e_chunk + q;
                samples[n_seeds + e] = (int64_t)t.v[u]; // :215 (re-read as the next hop's frontier: keep it cacheable)
                if (NT) { // write-once outputs: stream them past L2 so gathers keep the cache
                    __builtin_nontemporal_store(n_seeds + e, &rows[e]);          // :217 j
                    __builtin_nontemporal_store(i0 + (int64_t)t.l[u], &cols[e]); // :217 i
                    __builtin_nontemporal_store(t.ep[u], &eidx[e]);              // :217 edge_ptr
                } else {
                    rows[e] = n_seeds + e;
                    cols[e] = i0 + (int64_t)t.l[u];
                    eidx[e] = t.ep[u];
                }
            }
        }
    };
    Batch a, b;
    issue(a, 0u);
    for (uint32_t q0 = 0; q0 < total; q0 += 2u * 64u * NS_EMIT) {
        issue(b, q0 + 64u * NS_EMIT);
        store(a, q0);
        issue(a, q0 + 2u * 64u * NS_EMIT);
        store(b, q0 + 64u * NS_EMIT);
    }
}

template <int KMAX, bool REPLACE, bool NT>
__global__ void ns_homo_uniform_kernel(const NsHomoParams p) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n_waves = blockDim.x >> 6;
    const int64_t b = blockIdx.x;

    uint32_t *chunk_off = reinterpret_cast<uint32_t *>(smem);
    unsigned char *wbase = smem + ((((size_t)(NS_CHUNKS_PER_ROUND + 1) * sizeof(uint32_t)) + 15) & ~(size_t)15) +
                           (size_t)wave * ns_wave_lds_bytes(p.kmax, KMAX == 0);
    int64_t *ebase = reinterpret_cast<int64_t *>(wbase);
    uint32_t *spos = reinterpret_cast<uint32_t *>(wbase + 64 * sizeof(int64_t));
    uint8_t *slane = reinterpret_cast<uint8_t *>(wbase + 64 * sizeof(int64_t) + (size_t)64 * p.kmax * sizeof(uint32_t));
    uint32_t *strip = reinterpret_cast<uint32_t *>(wbase + 64 * sizeof(int64_t) + (size_t)64 * p.kmax * sizeof(uint32_t) +
                                                   (((size_t)64 * p.kmax + 15) & ~(size_t)15));
    (void)strip;

    int64_t *samples = p.samples + b * p.cap_nodes;
    int64_t *rows = p.rows + b * p.cap_edges;
    int64_t *cols = p.cols + b * p.cap_edges;
    int64_t *eidx = p.edge_index + b * p.cap_edges;
    const int64_t n_seeds = p.n_seeds;

    for (int64_t i = tid; i < n_seeds; i += blockDim.x) samples[i] = p.seeds[b * n_seeds + i]; // :184
    const CallKey ck = call_key(p.seed, p.call_id + (uint64_t)b, p.tag);
    __syncthreads();

    int64_t begin = 0, end = n_seeds; // frontier = samples[begin, end)   (:187)
    int64_t ne = 0;                   // edges emitted so far; samples so far = n_seeds + ne
    for (int h = 0; h < p.n_hops; ++h) {
        const int k = p.fanout[h];
        if (tid == 0) { // :193
            int64_t *lo = p.layer_offsets + (b * p.n_hops + h) * 3;
            lo[0] = n_seeds + ne;
            lo[1] = ne;
            lo[2] = n_seeds + ne;
        }
        for (int64_t round_begin = begin; round_begin < end; round_begin += (int64_t)NS_CHUNKS_PER_ROUND * 64) {
            const int64_t round_end = min(end, round_begin + (int64_t)NS_CHUNKS_PER_ROUND * 64);
            const int nc = (int)((round_end - round_begin + 63) >> 6);
            // ---- pass A: per-chunk sample counts
            for (int c = wave; c < nc; c += n_waves) {
                const int64_t i = round_begin + (int64_t)c * 64 + lane;
                uint32_t cnt = 0;
                if (i < round_end) {
                    int64_t w = samples[i];
                    TG_CHECK_VERTEX(p, w);
                    const int64_t deg = p.ptrs32 ? (int64_t)(p.ptrs32[w + 1] - p.ptrs32[w]) : p.ptrs[w + 1] - p.ptrs[w];
                    cnt = (deg <= 0) ? 0u : (REPLACE ? (uint32_t)k : (uint32_t)min(deg, (int64_t)k));
                }
                const uint32_t tot = wave_sum(cnt);
                if (lane == 0) chunk_off[c] = tot;
            }
            __syncthreads();
            // ---- scan of chunk totals (wave 0)
            if (wave == 0) {
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
            // ---- pass B: sample, stage in output order, coalesced emit
            for (int c = wave; c < nc; c += n_waves) {
                const int64_t i0 = round_begin + (int64_t)c * 64;
                const int64_t i = i0 + lane;
                int64_t e0 = 0, deg = 0;
                if (i < round_end) {
                    int64_t w = samples[i];
                    TG_CHECK_VERTEX(p, w);
                    if (p.ptrs32) {
                        e0 = (int64_t)p.ptrs32[w];
                        deg = (int64_t)p.ptrs32[w + 1] - e0;
                    } else {
                        e0 = p.ptrs[w];
                        deg = p.ptrs[w + 1] - e0;
                    }
                }
                const uint32_t cnt = (deg <= 0) ? 0u : (REPLACE ? (uint32_t)k : (uint32_t)min(deg, (int64_t)k));
                uint64_t did = (uint64_t)(p.id_base + i); // address of this vertex's draws
                CallKey ckl = ck;
                if (p.seed_ids && i < round_end) { // remote-frontier mode: the origin sampler's (call id, slot)
                    did = (uint64_t)p.seed_ids[b * n_seeds + i];
                    ckl = call_key(p.seed, (uint64_t)p.seed_call_ids[b * n_seeds + i], p.tag);
                }
                const uint32_t incl = wave_inclusive_scan(cnt);
                const uint32_t excl = incl - cnt;
                const uint32_t total = __shfl(incl, 63, 64);
                ebase[lane] = e0;
                if (cnt > 0) {
                    const uint32_t n = (uint32_t)deg;
                    if (REPLACE) { // sampling.rs:57-69, k draws of U[0,n)
                        sample_replace_any(ckl, did, n, k, spos, slane, excl, lane);
                    } else if (deg <= k) { // sampling.rs:12-15: the reservoir is just filled
                        for (uint32_t s = 0; s < cnt; ++s) {
                            spos[excl + s] = s;
                            slane[excl + s] = (uint8_t)lane;
                        }
                    } else if constexpr (KMAX == 0) {
                        sample_tickets_lds(ckl, did, n, k, spos, slane, excl, lane, strip);
                    } else {
                        sample_tickets<KMAX>(ckl, did, n, k, spos, slane, excl, lane);
                    }
                }
                wave_lds_handoff();
                const int64_t e_chunk = ne + (int64_t)chunk_off[c];
                if (p.indices32)
                    emit_chunk<uint32_t, NT>(p.indices32, total, lane, slane, spos, ebase, e_chunk, i0, n_seeds, samples,
                                             rows, cols, eidx);
                else
                    emit_chunk<int64_t, NT>(p.indices, total, lane, slane, spos, ebase, e_chunk, i0, n_seeds, samples, rows,
                                            cols, eidx);
                wave_lds_handoff();
            }
            __syncthreads();
            ne += chunk_off[nc];
            __syncthreads(); // chunk_off is rewritten by the next round
        }
        begin = end; // :221-222
        end = n_seeds + ne;
    }
    if (tid == 0) {
        p.counts[b * 2 + 0] = n_seeds + ne;
        p.counts[b * 2 + 1] = ne;
    }
}

static int env_int(const char *name, int dflt) {
    const char *v = getenv(name);
    return v ? atoi(v) : dflt;
}

template <int KMAX, bool REPLACE, bool NT>
static int launch_uniform_nt(const NsHomoParams &p, int64_t n_batches, hipStream_t stream) {
    // few batches: wide workgroups for latency; many batches: narrow ones for occupancy
    int threads = (n_batches < 512) ? 1024 : 512;
    static const int forced = env_int("TG_NS_THREADS", 0); // tuning knob, read once
    if (forced >= 64 && forced <= 1024 && forced % 64 == 0) threads = forced;
    while (threads > 64 && ns_block_lds_bytes(p.kmax, threads / 64, KMAX == 0) > 64 * 1024) // LDS limit
        threads = ((threads >> 1) + 63) & ~63; // stays a whole number of wavefronts (n_waves = blockDim.x >> 6)
    if (threads % 64 != 0) return fail(TG_ERR_INVALID, "tg_ns_homo_batched: %d threads is not a multiple of 64", threads);
    const size_t lds = ns_block_lds_bytes(p.kmax, threads / 64, KMAX == 0);
    if (lds > 160 * 1024)
        return fail(TG_ERR_UNSUPPORTED, "tg_ns_homo_batched: fan-out %d needs %zu B of LDS per wavefront", p.kmax, lds);
    if (lds > 64 * 1024) // one wavefront per workgroup with a large fan-out: opt in to the full 160 KB of a CU
        TG_HIP(hipFuncSetAttribute((const void *)ns_homo_uniform_kernel<KMAX, REPLACE, NT>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((ns_homo_uniform_kernel<KMAX, REPLACE, NT>), dim3((unsigned)n_batches), dim3(threads), lds,
                       stream, p);
    TG_LAUNCH_CHECK();
    return TG_OK;
}
template <int KMAX, bool REPLACE>
static int launch_uniform(const NsHomoParams &p, int64_t n_batches, hipStream_t stream) {
    static const int nt = env_int("TG_NS_NT", 1); // read once
    return nt ? launch_uniform_nt<KMAX, REPLACE, true>(p, n_batches, stream)
              : launch_uniform_nt<KMAX, REPLACE, false>(p, n_batches, stream);
}

} // namespace tg

extern "C" int tg_ns_homo_capacity(int64_t n_seeds, const int64_t *fanout, int32_t n_hops, int64_t *cap_nodes,
                                   int64_t *cap_edges) {
    TG_REQUIRE(n_seeds >= 0 && n_hops >= 0 && (fanout || n_hops == 0), "tg_ns_homo_capacity: bad arguments");
    int64_t layer = n_seeds, edges = 0;
    for (int h = 0; h < n_hops; ++h) {
        TG_REQUIRE(fanout[h] >= 1, "tg_ns_homo_capacity: fanout[%d] = %lld must be >= 1", h, (long long)fanout[h]);
        TG_REQUIRE(layer == 0 || fanout[h] <= INT64_MAX / 4 / (layer > 0 ? layer : 1),
                   "tg_ns_homo_capacity: capacity overflows int64");
        layer *= fanout[h];
        edges += layer;
    }
    if (cap_nodes) *cap_nodes = n_seeds + edges;
    if (cap_edges) *cap_edges = edges;
    return TG_OK;
}

int tg_ns_homo_filtered_launch(const tg_graph *csc, const int64_t *seeds, int64_t n_batches, int64_t n_seeds,
                               const int64_t *fanout, int32_t n_hops, const tg_ns_config *cfg, const tg_rng *rng,
                               const tg_ns_out *out, hipStream_t stream); // ns_homo_scan.hip

int tg_ns_homo_flat_launch(const tg_graph *csc, const int64_t *seeds, int64_t n_batches, int64_t n_seeds,
                           const int64_t *fanout, int32_t n_hops, const tg_ns_config *cfg, const tg_rng *rng,
                           const tg_ns_out *out, void *ws, int64_t ws_bytes, int32_t mode, hipStream_t stream,
                           int *rc_out); // ns_homo_flat.hip
int tg_ns_homo_windowed_applicable(const tg_graph *csc, int64_t n_batches, int64_t n_seeds, const int64_t *fanout,
                                   int32_t n_hops, const tg_ns_config *cfg, const tg_ns_out *out,
                                   int32_t mode); // ns_homo_win.hip
int tg_ns_homo_windowed_launch(const tg_graph *csc, const int64_t *seeds, int64_t n_batches, int64_t n_seeds,
                               const int64_t *fanout, int32_t n_hops, const tg_ns_config *cfg, const tg_rng *rng,
                               const tg_ns_out *out, void *ws, int64_t ws_bytes, int32_t mode, hipStream_t stream);

static int ns_homo_batched_impl(const tg_graph *csc, const int64_t *seeds, int64_t n_batches, int64_t n_seeds,
                                const int64_t *fanout, int32_t n_hops, const tg_ns_config *cfg, const tg_rng *rng,
                                const tg_ns_out *out, void *ws, int64_t ws_bytes, int32_t mode, void *stream) {
    TG_REQUIRE(csc && csc->ptrs && (csc->indices || csc->n_edges == 0), "tg_ns_homo_batched: null graph");
    TG_REQUIRE(rng && out, "tg_ns_homo_batched: null rng/out");
    TG_REQUIRE(n_batches >= 0 && n_seeds >= 0, "tg_ns_homo_batched: negative sizes");
    TG_REQUIRE(n_batches <= 0x7fffffff, "tg_ns_homo_batched: too many batches for one launch");
    TG_REQUIRE(n_hops >= 0 && n_hops <= TG_MAX_HOPS, "tg_ns_homo_batched: n_hops %d outside [0, %d]", n_hops,
               TG_MAX_HOPS);
    TG_REQUIRE(seeds || n_seeds == 0, "tg_ns_homo_batched: null seeds");
    TG_REQUIRE(out->samples && out->counts && (out->layer_offsets || n_hops == 0),
               "tg_ns_homo_batched: null output buffers");
    int64_t need_nodes = 0, need_edges = 0;
    int rc = tg_ns_homo_capacity(n_seeds, fanout, n_hops, &need_nodes, &need_edges);
    if (rc != TG_OK) return rc;
    TG_REQUIRE(out->cap_nodes >= need_nodes && out->cap_edges >= need_edges,
               "tg_ns_homo_batched: output slabs too small (need %lld nodes / %lld edges per batch)",
               (long long)need_nodes, (long long)need_edges);
    TG_REQUIRE(need_edges == 0 || (out->rows && out->cols && out->edge_index),
               "tg_ns_homo_batched: null edge output buffers");
    if (n_batches == 0) return TG_OK;

    const int sampler = cfg ? cfg->sampler : TG_SAMPLER_UNIFORM;
    const int filter = cfg ? cfg->filter_mode : TG_FILTER_NONE;
    TG_REQUIRE(sampler >= TG_SAMPLER_UNIFORM && sampler <= TG_SAMPLER_WEIGHTED, "tg_ns_homo_batched: bad sampler %d",
               sampler);
    TG_REQUIRE(filter >= TG_FILTER_NONE && filter <= TG_FILTER_DYNAMIC, "tg_ns_homo_batched: bad filter %d", filter);
    if (sampler == TG_SAMPLER_WEIGHTED || filter != TG_FILTER_NONE) {
        int rc_flat = TG_OK; // few batches + a workspace: hop by hop over the whole device (ns_homo_flat.hip)
        if (tg_ns_homo_flat_launch(csc, seeds, n_batches, n_seeds, fanout, n_hops, cfg, rng, out, ws, ws_bytes, mode,
                                   (hipStream_t)stream, &rc_flat))
            return rc_flat;
        return tg_ns_homo_filtered_launch(csc, seeds, n_batches, n_seeds, fanout, n_hops, cfg, rng, out,
                                          (hipStream_t)stream);
    }

    if (ws && tg_ns_homo_windowed_applicable(csc, n_batches, n_seeds, fanout, n_hops, cfg, out, mode))
        return tg_ns_homo_windowed_launch(csc, seeds, n_batches, n_seeds, fanout, n_hops, cfg, rng, out, ws, ws_bytes,
                                          mode, (hipStream_t)stream);

    tg::NsHomoParams p;
    p.ptrs = csc->ptrs;
    p.indices = csc->indices;
    p.indices32 = csc->indices32;
    p.ptrs32 = csc->ptrs32;
    p.seeds = seeds;
    p.n_seeds = n_seeds;
    p.n_hops = n_hops;
    p.kmax = 1;
    for (int h = 0; h < TG_MAX_HOPS; ++h) p.fanout[h] = 0;
    for (int h = 0; h < n_hops; ++h) {
        TG_REQUIRE(fanout[h] <= 255, "tg_ns_homo_batched: fanout[%d] = %lld exceeds 255", h, (long long)fanout[h]);
        p.fanout[h] = (int32_t)fanout[h];
        if (p.fanout[h] > p.kmax) p.kmax = p.fanout[h];
    }
    p.cap_nodes = out->cap_nodes;
    p.cap_edges = out->cap_edges;
    p.samples = out->samples;
    p.rows = out->rows;
    p.cols = out->cols;
    p.edge_index = out->edge_index;
    p.layer_offsets = out->layer_offsets;
    p.counts = out->counts;
    p.seed = rng->seed;
    p.call_id = rng->call_id;
    p.tag = (cfg && cfg->rng_tag) ? cfg->rng_tag : TG_TAG_NS_HOMO;
    p.id_base = cfg ? cfg->id_base : 0;
    TG_BOUNDS_INIT(p, csc);
    p.seed_ids = cfg ? cfg->seed_ids : nullptr;
    p.seed_call_ids = cfg ? cfg->seed_call_ids : nullptr;
    TG_REQUIRE((p.seed_ids == nullptr) == (p.seed_call_ids == nullptr),
               "tg_ns_homo_batched: seed_ids and seed_call_ids come together");
    TG_REQUIRE(!p.seed_ids || n_hops == 1, "tg_ns_homo_batched: remote-frontier mode needs n_hops == 1");
    hipStream_t s = (hipStream_t)stream;
    const bool repl = sampler == TG_SAMPLER_UNIFORM_REPL;
    if (p.kmax <= 16)
        return repl ? tg::launch_uniform<16, true>(p, n_batches, s) : tg::launch_uniform<16, false>(p, n_batches, s);
    if (p.kmax <= TG_MAX_FANOUT)
        return repl ? tg::launch_uniform<32, true>(p, n_batches, s) : tg::launch_uniform<32, false>(p, n_batches, s);
    // above the register-resident sampler: ticket strips in LDS (any fan-out the LDS can hold)
    return repl ? tg::launch_uniform<0, true>(p, n_batches, s) : tg::launch_uniform<0, false>(p, n_batches, s);
}

extern "C" int tg_ns_homo_batched(const tg_graph *csc, const int64_t *seeds, int64_t n_batches, int64_t n_seeds,
                                  const int64_t *fanout, int32_t n_hops, const tg_ns_config *cfg, const tg_rng *rng,
                                  const tg_ns_out *out, void *stream) {
    return ns_homo_batched_impl(csc, seeds, n_batches, n_seeds, fanout, n_hops, cfg, rng, out, nullptr, 0, 0, stream);
}

extern "C" int tg_ns_homo_batched_ws(const tg_graph *csc, const int64_t *seeds, int64_t n_batches, int64_t n_seeds,
                                     const int64_t *fanout, int32_t n_hops, const tg_ns_config *cfg, const tg_rng *rng,
                                     const tg_ns_out *out, void *workspace, int64_t workspace_bytes, int32_t mode,
                                     void *stream) {
    TG_REQUIRE(workspace || workspace_bytes == 0, "tg_ns_homo_batched_ws: null workspace");
    TG_REQUIRE(mode >= TG_NS_FORM_AUTO && mode <= TG_NS_FORM_WINDOWED_WIDE, "tg_ns_homo_batched_ws: bad mode %d", mode);
    return ns_homo_batched_impl(csc, seeds, n_batches, n_seeds, fanout, n_hops, cfg, rng, out, workspace,
                                workspace_bytes, mode, stream);
}
