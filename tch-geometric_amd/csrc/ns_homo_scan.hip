// neighbor_sampling_homogenous with a TemporalFilter and/or the WeightedSampler
// (reference: src/algo/neighbor_sampling.rs:36-77 filter, :131-158 weighted
// sampler, src/utils/sampling.rs:6-55).  Every candidate edge has to be looked at
// (its timestamp / weight), so the work shape differs from ns_homo.hip:
//
//   P1  one WAVEFRONT per frontier vertex streams its column once, coalesced:
//       lanes test the filter, ballot + popcount give each admissible edge its
//       rank in the reference's candidate order, and the reservoir is resolved
//       from one addressed Philox draw per candidate beyond the first k (the
//       per-item form of the reference loop; for weights this IS the reference
//       algorithm, with its left-to-right running sum kept exactly).  The <=k
//       chosen edge pointers are parked at a fixed stride in the batch's `rows`
//       slab (free until the hop's rows are written, -1 padded).
//   P2  as in ns_homo.hip: lane = vertex, counts -> LDS scan -> output-ordered
//       LDS staging -> coalesced gather + write of samples / cols / edge_index
//       (+ the filter state of the new sample).
//   P3  rows[e] = n_seeds + e for the hop (the parked pointers are consumed).
// HBM-bound on 8 B (timestamp) [+ 8 B (weight)] per inspected edge.
#include "tg_device.h"
#include "tg_host.h"

namespace tg {

constexpr int NSS_CHUNKS_PER_ROUND = 1024;
constexpr int NSS_PREFETCH = 8; // 64-edge chunks whose loads are in flight together per wavefront
constexpr int NSS_GROUPS = 1024; // running-count slots per wavefront (group = 128 << shift edges)
constexpr int NSS_PAIRS = 4;     // 16-byte loads per lane and round of the count pass (round = 512 edges)
constexpr int NSS_FETCH = 8;     // slots whose groups are re-read together in the fetch pass
typedef long long ts_pair __attribute__((ext_vector_type(2)));

struct NsScanParams {
    const int64_t *ptrs;
    const int64_t *indices;
    const double *weights;
    const int64_t *timestamps;
    const int64_t *seeds;
    const int64_t *seeds_state;
    int64_t n_seeds;
    int32_t n_hops, kmax;
    int32_t fanout[TG_MAX_HOPS];
    int32_t sampler, filter_mode, forward;
    int64_t win_lo, win_hi;
    int64_t cap_nodes, cap_edges;
    int64_t *samples, *rows, *cols, *edge_index, *layer_offsets, *counts, *states;
    uint64_t seed, call_id;
    uint32_t tag;
    int64_t id_base;
    const int32_t *only_if; // non-null: run only when *only_if & 3 != 0 (the fall-back behind the flat path, ns_homo_flat.hip)
};

__host__ __device__ inline size_t nss_wave_lds_bytes(int kmax) {
    // slot_ptr[k] i64 | tgt[k] i64 | slot_rank[k] u32 (padded) | eslot[64*k] u8 | elane[64*k] u8 (padded) | gpref
    return (size_t)kmax * 8 * 2 + (((size_t)kmax * 4 + 15) & ~(size_t)15) + 2 * (((size_t)64 * kmax + 15) & ~(size_t)15) +
           (size_t)NSS_GROUPS * 4 + 64 * 8; // + 64 doubles for the weighted sampler's serial prefix
}
__host__ __device__ inline size_t nss_block_lds_bytes(int kmax, int n_waves) {
    return (((size_t)(NSS_CHUNKS_PER_ROUND + 1) * 4 + 15) & ~(size_t)15) + (size_t)n_waves * nss_wave_lds_bytes(kmax);
}

// neighbor_sampling.rs:55-67
__device__ __forceinline__ bool filter_pass(const NsScanParams &p, int64_t state, int64_t e) {
    if (p.filter_mode == TG_FILTER_NONE) return true;
    const int64_t t = p.timestamps[e];
    int64_t x;
    if (p.filter_mode == TG_FILTER_STATIC)
        x = t;
    else
        x = p.forward ? (t - state) : -(t - state);
    return p.win_lo <= x && x <= p.win_hi;
}

// The same predicate without control flow, for the streaming loops: x = t (static), t - state (forward) or
// state - t (backward) is (t ^ m) - m + off with m = 0 / -1; no filter = the full i64 window.  (Two's-complement
// wrap-around is what the reference's release build does on overflow.)
struct FilterEval {
    int64_t m, off, lo, hi;
    __device__ __forceinline__ FilterEval(const NsScanParams &p, int64_t state) {
        const bool none = p.filter_mode == TG_FILTER_NONE, stat = p.filter_mode == TG_FILTER_STATIC;
        const bool neg = !none && !stat && !p.forward;
        m = neg ? -1 : 0;
        off = (none || stat) ? 0 : (neg ? state : -state);
        lo = none ? INT64_MIN : p.win_lo;
        hi = none ? INT64_MAX : p.win_hi;
    }
    __device__ __forceinline__ bool operator()(int64_t t) const {
        const int64_t x = (int64_t)(((uint64_t)(t ^ m) - (uint64_t)m) + (uint64_t)off);
        return (int)(lo <= x) & (int)(x <= hi);
    }
};

template <bool WEIGHTED>
__global__ void ns_homo_scan_kernel(const NsScanParams p) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n_waves = blockDim.x >> 6;
    const int64_t b = blockIdx.x;
    const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    if (p.only_if && (*p.only_if & 3) == 0) return; // uniform: the flat path finished, nothing to redo

    uint32_t *chunk_off = reinterpret_cast<uint32_t *>(smem);
    unsigned char *wbase = smem + ((((size_t)(NSS_CHUNKS_PER_ROUND + 1) * 4) + 15) & ~(size_t)15) +
                           (size_t)wave * nss_wave_lds_bytes(p.kmax);
    int64_t *slot_ptr = reinterpret_cast<int64_t *>(wbase);
    int64_t *tgt = slot_ptr + p.kmax;
    uint32_t *slot_rank = reinterpret_cast<uint32_t *>(tgt + p.kmax);
    uint8_t *eslot = reinterpret_cast<uint8_t *>(wbase + (size_t)p.kmax * 16 + (((size_t)p.kmax * 4 + 15) & ~(size_t)15));
    uint8_t *elane = eslot + (((size_t)64 * p.kmax + 15) & ~(size_t)15);
    uint32_t *gpref = reinterpret_cast<uint32_t *>(reinterpret_cast<unsigned char *>(elane) +
                                                   (((size_t)64 * p.kmax + 15) & ~(size_t)15));
    double *pbuf = reinterpret_cast<double *>(gpref + NSS_GROUPS);
    (void)pbuf;

    int64_t *samples = p.samples + b * p.cap_nodes;
    int64_t *rows = p.rows + b * p.cap_edges;
    int64_t *cols = p.cols + b * p.cap_edges;
    int64_t *eidx = p.edge_index + b * p.cap_edges;
    const bool has_state = p.filter_mode != TG_FILTER_NONE;
    int64_t *states = has_state ? p.states + b * p.cap_nodes : nullptr;
    const int64_t n_seeds = p.n_seeds;

    for (int64_t i = tid; i < n_seeds; i += blockDim.x) {
        samples[i] = p.seeds[b * n_seeds + i];
        if (has_state) states[i] = p.seeds_state[b * n_seeds + i];
    }
    const CallKey ck = call_key(p.seed, p.call_id + (uint64_t)b, p.tag);
    __shared__ int panic_flag; // the reference panics (empty float range) -- reported through counts
    __shared__ unsigned long long next_vertex;
    if (tid == 0) panic_flag = 0;
    __syncthreads();

    int64_t begin = 0, end = n_seeds, ne = 0;
    for (int h = 0; h < p.n_hops; ++h) {
        const int k = p.fanout[h];
        if (tid == 0) {
            int64_t *lo = p.layer_offsets + (b * p.n_hops + h) * 3;
            lo[0] = n_seeds + ne;
            lo[1] = ne;
            lo[2] = n_seeds + ne;
        }
        int64_t *park = rows + ne; // (end-begin)*k entries, free until P3
        const int64_t hop_ne0 = ne;

        // ================= P1: one wavefront per frontier vertex
        // (the next vertex's id, state and column bounds are loaded while the current column streams)
        // Vertices are handed out dynamically (one LDS counter per workgroup): column lengths differ by orders of
        // magnitude, a static split would leave three wavefronts waiting for the one that met a hub.
        int64_t w_nx = 0, st_nx = 0, e0_nx = 0, e1_nx = 0;
        if (tid == 0) next_vertex = (unsigned long long)(begin + n_waves);
        __syncthreads();
        if (begin + wave < end) {
            w_nx = samples[begin + wave];
            st_nx = has_state ? states[begin + wave] : 0;
            e0_nx = p.ptrs[w_nx];
            e1_nx = p.ptrs[w_nx + 1];
        }
        int64_t i_take = begin + wave;
        while (i_take < end) {
            const int64_t i = i_take;
            const int64_t st = st_nx;
            const int64_t e0 = e0_nx, e1 = e1_nx;
            unsigned long long taken = 0;
            if (lane == 0) taken = atomicAdd(&next_vertex, 1ull);
            i_take = (int64_t)__shfl(taken, 0, 64);
            const int64_t i_nx = min(i_take, end - 1); // clamped: the loads below stay unconditional
            w_nx = samples[i_nx];
            st_nx = has_state ? states[i_nx] : 0;
            const FilterEval fpass(p, st);
            uint32_t n = 0; // admissible candidates seen so far
            if (lane < k) slot_rank[lane] = 0;
            wave_lds_handoff();
            if constexpr (!WEIGHTED) {
                // ---- uniform samplers under a filter: count, draw ranks, fetch only the groups that hold them.
                // pass 1 streams the column once (8 chunks in flight per lane) and keeps the running count of
                // admissible edges at the end of every group of 512 << shift edges in LDS;
                // (16-byte loads: two timestamps per lane, pairs aligned to 16 B, so groups are cut relative to the
                // aligned start a0 <= e0; two rounds of 4 loads per lane are in flight -- the next round is issued
                // before the current one is counted)
                const int64_t a0 = e0 - (int64_t)((((uintptr_t)p.timestamps >> 3) ^ (uintptr_t)e0) & 1);
                const int64_t deg = e1 - a0;
                int shift = 0;
                while (((deg + 127) >> (7 + shift)) > NSS_GROUPS) ++shift;
                const int64_t n_sub = (deg + 127) >> 7; // 128-edge sub-blocks: one 16-byte load per lane
                const int64_t n_rounds = (n_sub + NSS_PAIRS - 1) / NSS_PAIRS;
                const int64_t spg_mask = ((int64_t)1 << shift) - 1;
                int g = 0;
                ts_pair bufA[NSS_PAIRS], bufB[NSS_PAIRS];
                auto issue = [&](ts_pair *buf, int64_t r) {
#pragma unroll
                    for (int u = 0; u < NSS_PAIRS; ++u) {
                        // unconditional (out-of-column lanes re-read the column's first pair and are masked when
                        // counted): a branch per load would make the compiler wait for ALL loads in flight
                        const int64_t e = a0 + (((r * NSS_PAIRS + u) << 6) + lane) * 2;
                        buf[u] = __builtin_nontemporal_load(reinterpret_cast<const ts_pair *>(p.timestamps + ((e < e1) ? e : a0)));
                    }
                };
                auto count = [&](const ts_pair *buf, int64_t r) {
#pragma unroll
                    for (int u = 0; u < NSS_PAIRS; ++u) {
                        const int64_t sb = r * NSS_PAIRS + u;
                        const int64_t e = a0 + ((sb << 6) + lane) * 2;
                        n += (uint32_t)__popcll(__ballot((int)(e >= e0) & (int)(e < e1) & (int)fpass(buf[u].x)));
                        n += (uint32_t)__popcll(__ballot((int)(e + 1 < e1) & (int)fpass(buf[u].y)));
                        if (sb < n_sub && (((sb + 1) & spg_mask) == 0 || sb + 1 == n_sub)) {
                            if (lane == 0) gpref[g] = n;
                            ++g;
                        }
                    }
                };
                // rounds past the column's end are issued too (they re-read its first pair and count nothing): the
                // loop body stays free of branches around loads, which keeps the s_waitcnt counts exact
                issue(bufA, 0);
                for (int64_t r = 0; r < n_rounds; r += 2) {
                    issue(bufB, r + 1);
                    count(bufA, r);
                    issue(bufA, r + 2);
                    count(bufB, r + 1);
                }
                e0_nx = p.ptrs[w_nx]; // next vertex's column bounds: in flight during the fetch pass
                e1_nx = p.ptrs[w_nx + 1];
                const int n_groups = g;
                wave_lds_handoff();
                // the ranks to fetch: every candidate when there are few, k draws of U[0,n) with replacement
                // (sampling.rs:57-69), else the reservoir's ticket chain (lane s owns slot s)
                const uint32_t cnt_sel = (p.sampler == TG_SAMPLER_UNIFORM_REPL) ? (n > 0 ? (uint32_t)k : 0u)
                                                                               : min(n, (uint32_t)k);
                uint32_t myrank = (uint32_t)lane;
                const uint64_t did = (uint64_t)(p.id_base + i);
                if (p.sampler == TG_SAMPLER_UNIFORM_REPL) {
                    if (n > 0 && lane < k) {
                        myrank = slot_draw(ck, did, (uint32_t)lane, D1_REPLACE, n);
                    }
                } else if (n > (uint32_t)k) {
                    uint32_t myK = 0xffffffffu, myV = 0;
                    Draw d;
                    for (int s = 0; s < k; ++s) {
                        const uint32_t m = (n - 1u) - (uint32_t)s;
                        if ((s & 3) == 0) d = draw(ck, did, (uint32_t)(s >> 2), 0u);
                        const uint32_t r = slot_draw_from(d, ck, did, (uint32_t)s, 0u, m), last = m - 1u;
                        const uint64_t mr = __ballot(lane < s && myK == r);
                        const uint64_t ml = __ballot(lane < s && myK == last);
                        const uint32_t vr = __shfl(myV, mr ? 63 - __clzll((long long)mr) : 0, 64);
                        const uint32_t vl = __shfl(myV, ml ? 63 - __clzll((long long)ml) : 0, 64);
                        const uint32_t tr = mr ? vr : r, tl = ml ? vl : last;
                        if (lane == s) {
                            myK = r;
                            myV = tl;
                            myrank = (tr < n - (uint32_t)k) ? (uint32_t)k + tr : (uint32_t)s;
                        }
                    }
                }
                // pass 2: lane s finds the group that holds slot s's rank (first group whose running count exceeds
                // it); the groups of NSS_FETCH slots are then re-read together, one 16-byte load per lane and slot
                int mygroup = 0;
                if ((uint32_t)lane < cnt_sel) {
                    int lo_g = 0, hi_g = n_groups - 1;
                    while (lo_g < hi_g) {
                        const int mid = (lo_g + hi_g) >> 1;
                        if (gpref[mid] > myrank)
                            hi_g = mid;
                        else
                            lo_g = mid + 1;
                    }
                    mygroup = lo_g;
                }
                if (shift == 0) {
                    for (uint32_t s0 = 0; s0 < cnt_sel; s0 += NSS_FETCH) {
                        ts_pair v[NSS_FETCH];
                        int gq[NSS_FETCH];
#pragma unroll
                        for (int u = 0; u < NSS_FETCH; ++u) {
                            gq[u] = __shfl(mygroup, (int)min(s0 + (uint32_t)u, 63u), 64);
                            const int64_t e = a0 + ((int64_t)gq[u] << 7) + 2 * lane;
                            v[u] = *reinterpret_cast<const ts_pair *>(p.timestamps + ((e < e1) ? e : a0));
                        }
#pragma unroll
                        for (int u = 0; u < NSS_FETCH; ++u) {
                            const uint32_t sl = s0 + (uint32_t)u;
                            const uint32_t r = __shfl(myrank, (int)min(sl, 63u), 64);
                            const uint32_t seen = gq[u] > 0 ? gpref[gq[u] - 1] : 0u;
                            const int64_t e = a0 + ((int64_t)gq[u] << 7) + 2 * lane;
                            const bool okx = (int)(e >= e0) & (int)(e < e1) & (int)fpass(v[u].x);
                            const bool oky = (int)(e + 1 < e1) & (int)fpass(v[u].y);
                            const uint64_t mx = __ballot(okx), my = __ballot(oky);
                            const uint32_t before = seen + (uint32_t)__popcll(mx & lt_mask) + (uint32_t)__popcll(my & lt_mask);
                            if (sl < cnt_sel) {
                                if (okx && before == r) slot_ptr[sl] = e;
                                if (oky && before + (uint32_t)okx == r) slot_ptr[sl] = e + 1;
                            }
                        }
                    }
                } else { // columns above 128 * NSS_GROUPS edges: groups of 128 << shift edges, slot by slot
                    const int64_t gsize = (int64_t)128 << shift;
                    for (uint32_t sl = 0; sl < cnt_sel; ++sl) {
                        const uint32_t r = __shfl(myrank, (int)sl, 64);
                        const int lo_g = __shfl(mygroup, (int)sl, 64);
                        uint32_t seen = lo_g > 0 ? gpref[lo_g - 1] : 0u;
                        const int64_t gb = a0 + (int64_t)lo_g * gsize, ge = min(gb + gsize, e1);
                        bool found = false;
                        for (int64_t cb = gb; cb < ge && !found; cb += 64 * NSS_PREFETCH) {
                            int64_t tsv[NSS_PREFETCH];
#pragma unroll
                            for (int u = 0; u < NSS_PREFETCH; ++u) {
                                const int64_t e = cb + u * 64 + lane;
                                tsv[u] = p.timestamps[(e >= e0 && e < ge) ? e : e0];
                            }
#pragma unroll
                            for (int u = 0; u < NSS_PREFETCH; ++u) {
                                if (found) break;
                                const int64_t e = cb + u * 64 + lane;
                                const bool ok = (int)(e >= e0) & (int)(e < ge) & (int)fpass(tsv[u]);
                                const uint64_t mask = __ballot(ok);
                                const uint32_t c = (uint32_t)__popcll(mask);
                                if (r - seen < c) {
                                    if (ok && seen + (uint32_t)__popcll(mask & lt_mask) == r) slot_ptr[sl] = e;
                                    found = true;
                                }
                                seen += c;
                            }
                        }
                    }
                }
            } else {
                e0_nx = p.ptrs[w_nx];
                e1_nx = p.ptrs[w_nx + 1];
                // The column is streamed NSS_PREFETCH chunks at a time: all loads of a group are issued before any
                // of them is consumed, so a wavefront keeps several HBM requests in flight instead of one.
                double w_sum = 0.0;
                constexpr bool weighted = WEIGHTED;
                for (int64_t gbase = e0; gbase < e1; gbase += 64 * NSS_PREFETCH) {
                    int64_t tsv[NSS_PREFETCH];
                    double wvv[NSS_PREFETCH];
#pragma unroll
                    for (int u = 0; u < NSS_PREFETCH; ++u) {
                        const int64_t e = gbase + u * 64 + lane;
                        tsv[u] = (has_state && e < e1) ? __builtin_nontemporal_load(&p.timestamps[e]) : 0;
                        wvv[u] = (weighted && e < e1) ? __builtin_nontemporal_load(&p.weights[e]) : 0.0;
                    }
#pragma unroll
                    for (int u = 0; u < NSS_PREFETCH; ++u) {
                        const int64_t base = gbase + u * 64;
                        if (base >= e1) break;
                        const int64_t e = base + lane;
                        const bool ok = e < e1 && fpass(tsv[u]);
                        const uint64_t mask = __ballot(ok);
                        const uint32_t rank = n + (uint32_t)__popcll(mask & lt_mask);
                        uint32_t hit_slot = 0xffffffffu;
                        if (weighted) { // sampling.rs:28-55
                            const double wv = ok ? wvv[u] : 0.0;
                            double tot;
                            const double pref = wave_blocked_prefix_f64(wv, w_sum, &tot);
                            w_sum = tot;
                            if (ok && rank >= (uint32_t)k) {
                                if (!(0.0 < pref)) {
                                    panic_flag = 1;
                                } else {
                                    const Draw d = draw(ck, (uint64_t)(p.id_base + i), rank, D1_WEIGHTED);
                                    const double j = u64_to_f64_01(d.a()) * pref + 0.0;
                                    if (j < wv) hit_slot = (uint32_t)bounded64(d.b(), (uint64_t)k);
                                }
                            }
                        } else if (ok && rank >= (uint32_t)k) { // sampling.rs:17-24, one addressed draw per item
                            const Draw d = draw(ck, (uint64_t)(p.id_base + i), rank, D1_LITERAL);
                            const uint64_t j = bounded64(d.a(), (uint64_t)rank);
                            if (j < (uint64_t)k) hit_slot = (uint32_t)j;
                        }
                        const bool fill = ok && rank < (uint32_t)k;
                        if (__ballot(fill || hit_slot != 0xffffffffu) != 0ull) { // most chunks of a long column: no hit
                            if (fill) slot_ptr[rank] = e; // sampling.rs:12-15 / :37-45
                            if (hit_slot != 0xffffffffu) atomicMax(&slot_rank[hit_slot], rank);
                            wave_lds_handoff();
                            if (hit_slot != 0xffffffffu && slot_rank[hit_slot] == rank) slot_ptr[hit_slot] = e; // last hit
                            wave_lds_handoff();
                        }
                        n += (uint32_t)__popcll(mask);
                    }
                }
            }
            wave_lds_handoff();
            const uint32_t cnt = (p.sampler == TG_SAMPLER_UNIFORM_REPL) ? (n > 0 ? (uint32_t)k : 0u)
                                                                       : min(n, (uint32_t)k);
            if (lane < k) park[(i - begin) * k + lane] = ((uint32_t)lane < cnt) ? slot_ptr[lane] : -1;
            wave_lds_handoff();
        }
        __syncthreads();

        // ================= P2: compaction + emit
        for (int64_t round_begin = begin; round_begin < end; round_begin += (int64_t)NSS_CHUNKS_PER_ROUND * 64) {
            const int64_t round_end = min(end, round_begin + (int64_t)NSS_CHUNKS_PER_ROUND * 64);
            const int nc = (int)((round_end - round_begin + 63) >> 6);
            for (int c = wave; c < nc; c += n_waves) {
                const int64_t i = round_begin + (int64_t)c * 64 + lane;
                uint32_t cnt = 0;
                if (i < round_end)
                    for (int s = 0; s < k; ++s) cnt += (park[(i - begin) * k + s] >= 0);
                const uint32_t tot = wave_sum(cnt);
                if (lane == 0) chunk_off[c] = tot;
            }
            __syncthreads();
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
            for (int c = wave; c < nc; c += n_waves) {
                const int64_t i0 = round_begin + (int64_t)c * 64;
                const int64_t i = i0 + lane;
                uint32_t cnt = 0;
                if (i < round_end)
                    for (int s = 0; s < k; ++s) cnt += (park[(i - begin) * k + s] >= 0);
                const uint32_t incl = wave_inclusive_scan(cnt);
                const uint32_t excl = incl - cnt;
                const uint32_t total = __shfl(incl, 63, 64);
                for (uint32_t s = 0; s < cnt; ++s) { // only (lane, slot) is staged; the pointer is re-read from `park`
                    eslot[excl + s] = (uint8_t)s;
                    elane[excl + s] = (uint8_t)lane;
                }
                wave_lds_handoff();
                const int64_t e_chunk = ne + (int64_t)chunk_off[c];
#pragma unroll 2
                for (uint32_t q = lane; q < total; q += 64) {
                    const int l = elane[q];
                    const int64_t ep = park[(i0 + l - begin) * k + eslot[q]];
                    const int64_t e = e_chunk + q;
                    samples[n_seeds + e] = p.indices[ep];
                    cols[e] = i0 + l;
                    eidx[e] = ep;
                    if (has_state) // neighbor_sampling.rs:69-76
                        states[n_seeds + e] = (p.filter_mode == TG_FILTER_DYNAMIC) ? p.timestamps[ep] : states[i0 + l];
                }
                wave_lds_handoff();
            }
            __syncthreads();
            ne += chunk_off[nc];
            __syncthreads();
        }
        // ================= P3: rows of this hop (its parked pointers are all consumed)
        for (int64_t e = hop_ne0 + tid; e < ne; e += blockDim.x) rows[e] = n_seeds + e;
        __syncthreads();
        begin = end;
        end = n_seeds + ne;
    }
    if (tid == 0) {
        p.counts[b * 2 + 0] = panic_flag ? -1 : n_seeds + ne;
        p.counts[b * 2 + 1] = ne;
    }
}

} // namespace tg

int tg_ns_homo_filtered_launch_if(const tg_graph *csc, const int64_t *seeds, int64_t n_batches, int64_t n_seeds,
                                  const int64_t *fanout, int32_t n_hops, const tg_ns_config *cfg, const tg_rng *rng,
                                  const tg_ns_out *out, const int32_t *only_if, hipStream_t stream);
int tg_ns_homo_filtered_launch(const tg_graph *csc, const int64_t *seeds, int64_t n_batches, int64_t n_seeds,
                               const int64_t *fanout, int32_t n_hops, const tg_ns_config *cfg, const tg_rng *rng,
                               const tg_ns_out *out, hipStream_t stream) {
    return tg_ns_homo_filtered_launch_if(csc, seeds, n_batches, n_seeds, fanout, n_hops, cfg, rng, out, nullptr, stream);
}
int tg_ns_homo_filtered_launch_if(const tg_graph *csc, const int64_t *seeds, int64_t n_batches, int64_t n_seeds,
                                  const int64_t *fanout, int32_t n_hops, const tg_ns_config *cfg, const tg_rng *rng,
                                  const tg_ns_out *out, const int32_t *only_if, hipStream_t stream) {
    tg::NsScanParams p;
    p.only_if = only_if;
    p.ptrs = csc->ptrs;
    p.indices = csc->indices;
    p.weights = csc->weights;
    p.timestamps = csc->timestamps;
    p.seeds = seeds;
    p.seeds_state = cfg->seeds_state;
    p.n_seeds = n_seeds;
    p.n_hops = n_hops;
    p.kmax = 1;
    for (int h = 0; h < TG_MAX_HOPS; ++h) p.fanout[h] = 0;
    for (int h = 0; h < n_hops; ++h) {
        TG_REQUIRE(fanout[h] <= 64, "tg_ns_homo_batched: fanout[%d] = %lld exceeds 64 (filtered / weighted path)", h,
                   (long long)fanout[h]);
        p.fanout[h] = (int32_t)fanout[h];
        if (p.fanout[h] > p.kmax) p.kmax = p.fanout[h];
    }
    p.sampler = cfg->sampler;
    p.filter_mode = cfg->filter_mode;
    p.forward = cfg->forward;
    p.win_lo = cfg->win_lo;
    p.win_hi = cfg->win_hi;
    TG_REQUIRE(!cfg->seed_ids && !cfg->seed_call_ids,
               "tg_ns_homo_batched: remote-frontier mode is not available with filters / weights yet");
    TG_REQUIRE(p.sampler != TG_SAMPLER_WEIGHTED || csc->weights, "tg_ns_homo_batched: weighted sampler without weights");
    if (p.filter_mode != TG_FILTER_NONE) {
        TG_REQUIRE(csc->timestamps, "tg_ns_homo_batched: temporal filter without edge timestamps");
        TG_REQUIRE(cfg->seeds_state || n_seeds == 0, "tg_ns_homo_batched: temporal filter without seeds_state");
        TG_REQUIRE(out->states, "tg_ns_homo_batched: temporal filter needs the `states` workspace");
    }
    p.cap_nodes = out->cap_nodes;
    p.cap_edges = out->cap_edges;
    p.samples = out->samples;
    p.rows = out->rows;
    p.cols = out->cols;
    p.edge_index = out->edge_index;
    p.layer_offsets = out->layer_offsets;
    p.counts = out->counts;
    p.states = out->states;
    p.seed = rng->seed;
    p.call_id = rng->call_id;
    p.tag = cfg->rng_tag ? cfg->rng_tag : TG_TAG_NS_HOMO;
    p.id_base = cfg->id_base;
    int threads = (n_batches < 256) ? 1024 : 256;
    while (threads > 64 && tg::nss_block_lds_bytes(p.kmax, threads / 64) > 64 * 1024) threads >>= 1;
    const size_t lds = tg::nss_block_lds_bytes(p.kmax, threads / 64);
    if (p.sampler == TG_SAMPLER_WEIGHTED)
        hipLaunchKernelGGL(tg::ns_homo_scan_kernel<true>, dim3((unsigned)n_batches), dim3(threads), lds, stream, p);
    else
        hipLaunchKernelGGL(tg::ns_homo_scan_kernel<false>, dim3((unsigned)n_batches), dim3(threads), lds, stream, p);
    TG_LAUNCH_CHECK();
    return TG_OK;
}
