// hgt_sampling on gfx950 -- replaces src/algo/hgt_sampling.rs:13-278 (reference).
//
// The reference is sequential and order-sensitive: budgets are HashMaps whose f64 scores accumulate in
// visiting order, samples are drawn by a weighted reservoir over the budget in iteration order, and local
// ids follow insertion order.  This file restates each step as an order-preserving parallel primitive on
// device-resident state, launched from ONE host function without any synchronisation (sizes the host
// cannot know are handled with capacity upper bounds + device-side counters):
//
//  state per node type  nodes/ts lists, `to_local` hash map (node -> slot), budget = entry arrays
//                       (key, score f64, ts, alive) in INSERTION order + hash map key -> entry
//  update_budget        per relation into the sampled type: contributions (sample j, neighbour i < 50,
//                       hgt_sampling.rs:72 takes a PREFIX of the column) are generated in the reference's
//                       order; new keys get entries in first-contribution order (min-position hash map +
//                       prefix sum); every entry's contributions are collected in a bucket (count, offsets,
//                       scatter) and put in contribution order inside one lane / one wavefront as they are
//                       summed -- same f64 rounding as the reference's `+=` chain.  Relations that feed
//                       different budgets run side by side (blockIdx.y)
//  sample_from          one workgroup per node type, one launch per layer: live entries compacted in entry
//                       order, the reference's weighted reservoir (sampling.rs:28-55) with philox-mode's
//                       blocked running sum, one addressed Philox draw per candidate, then the append
//  edges                wavefront per destination node: <= 50 column positions (reservoir by tickets when the
//                       column is longer), kept when the source is a sampled node; compaction by prefix sum;
//                       the relations side by side (blockIdx.y)
//
// Canonical order: node types in `node_types` order, relations in `edge_types` order (the reference's
// HashMap order is not reproducible).  HBM traffic is tiny next to neighbor sampling; this path is bound by the
// LENGTH of its chain of launches (48 for 3 node types / 5 relations / 2 layers), which is why steps share launches.
#include <cstring>

#include <rocprim/device/device_scan.hpp>

#include <algorithm>
#include <vector>

#include "tg_device.h"
#include "tg_host.h"
#include "tg_scan.h"
#include "tg_map.h"
#define HGT_MULTI_TYPES_STAMPS 8

namespace tg {

constexpr int HGT_MAX_NB = 50; // MAX_NEIGHBORS hgt_sampling.rs:10
constexpr int64_t HGT_NAN_TS = -1;

// device counters of one node type
struct HgtTypeCtr {
    int64_t n_nodes;   // length of the nodes list
    int64_t lay_begin; // nodes[lay_begin, lay_end) = what the next update_budget processes
    int64_t lay_end;
    int64_t n_budget; // budget entries ever created
    int64_t present;  // budget_dict has an entry for the type
};

struct HgtType {
    int64_t *nodes, *ts;
    int64_t *tl_keys, *tl_vals, tl_mask;
    int64_t *bkey, *bts, *balive;
    double *bscore;
    int64_t *bm_keys, *bm_vals, bm_mask;
    HgtTypeCtr *ctr;
};

// ---------------------------------------------------------------- small scans and fills
// total of a flag array after its exclusive scan
// exclusive scan of flag[0 .. n) into rank[0 .. n] (rank[n] = the total, also written to *total) in one launch of one
// workgroup (tg_scan.h): the scans of a call are short and the call is bound by its number of launches
__device__ __forceinline__ void hgt_scan1_body(const int64_t *__restrict__ flag, int64_t n, int64_t *rank, int64_t *total) {
    block_scan_exclusive_plus1(n, [&](int64_t i) { return flag[i]; }, rank);
    __syncthreads();
    if (threadIdx.x == 0) total[0] = rank[n];
}
__device__ __forceinline__ void fill2_i64_body(int64_t *a, int64_t na, int64_t va, int64_t *b, int64_t nb, int64_t vb) {
    const int64_t n = na > nb ? na : nb;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        if (i < na) a[i] = va;
        if (i < nb) b[i] = vb;
    }
}
__global__ void scan_total_kernel(const int64_t *__restrict__ in, const int64_t *__restrict__ excl, int64_t n,
                                  int64_t *total) {
    if (threadIdx.x == 0 && blockIdx.x == 0) total[0] = n > 0 ? excl[n - 1] + in[n - 1] : 0;
}

// ---------------------------------------------------------------- inputs (hgt_sampling.rs:167-180)
__global__ void hgt_init_inputs_kernel(HgtType ty, const int64_t *__restrict__ inputs,
                                       const int64_t *__restrict__ input_ts, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t v = inputs[i];
        ty.nodes[i] = v;
        ty.ts[i] = input_ts ? input_ts[i] : HGT_NAN_TS;
        const int64_t s = map_slot_insert(ty.tl_keys, ty.tl_mask, v);
        atomicMax(reinterpret_cast<long long *>(&ty.tl_vals[s]), (long long)i); // insert overwrites: last slot wins
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        ty.ctr->n_nodes = n;
        ty.ctr->lay_begin = 0;
        ty.ctr->lay_end = n;
    }
}

// ---------------------------------------------------------------- update_budget (hgt_sampling.rs:27-102)
// per sample of the layer: number of contributions = min(deg, 50), and their exclusive prefix in the same launch (one
// workgroup: a layer is a few thousand samples); also marks the source budget present
__device__ __forceinline__ void hgt_contrib_count_scan_body(HgtType dst, HgtTypeCtr *src_ctr, const int64_t *__restrict__ ptrs,
                                                            int64_t *ccnt, int64_t *coff, int64_t cap, int64_t *total) {
    const int64_t b = dst.ctr->lay_begin, e = dst.ctr->lay_end;
    block_scan_exclusive_plus1(
        cap,
        [&](int64_t j) {
            int64_t c = 0;
            if (b + j < e) {
                const int64_t w = dst.nodes[b + j];
                c = min(ptrs[w + 1] - ptrs[w], (int64_t)HGT_MAX_NB);
            }
            ccnt[j] = c;
            return c;
        },
        coff);
    __syncthreads();
    if (threadIdx.x == 0) {
        total[0] = coff[cap];
        if (e > b) src_ctr->present = 1; // :38-40, :55
    }
}
// one lane per (sample j, neighbour i): key or -1, 1/count, timestamp  (:58-100)
__device__ __forceinline__ void hgt_contrib_gen_body(HgtType dst, HgtType src, const int64_t *__restrict__ ptrs,
                                       const int64_t *__restrict__ indices, const int64_t *__restrict__ edge_ts,
                                       int has_timerange, int64_t tr_lo, int64_t tr_hi, const int64_t *ccnt,
                                       const int64_t *coff, int64_t cap, int64_t *ckey, double *cinv, int64_t *cts) {
    const int64_t b = dst.ctr->lay_begin, e = dst.ctr->lay_end;
    const int64_t total = cap * HGT_MAX_NB;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (int64_t)gridDim.x * blockDim.x) {
        const int64_t j = q / HGT_MAX_NB, i = q - j * HGT_MAX_NB;
        if (b + j >= e) continue;
        const int64_t cnt = ccnt[j];
        if (i >= cnt) continue;
        const int64_t p = coff[j] + i;
        const int64_t w = dst.nodes[b + j];
        const int64_t ep = ptrs[w] + i; // the first min(deg,50) neighbours, :72
        const int64_t v = indices[ep];
        int64_t key = v;
        int64_t v_ts = edge_ts ? edge_ts[ep] : HGT_NAN_TS; // :82
        if (v_ts == HGT_NAN_TS) v_ts = dst.ts[b + j];      // :83-85
        if (map_slot_find(src.tl_keys, src.tl_mask, v) >= 0) key = -1; // :80 already sampled
        if (has_timerange && v_ts != HGT_NAN_TS && !(tr_lo <= v_ts && v_ts < tr_hi)) key = -1; // :88-92
        ckey[p] = key;
        cinv[p] = 1.0 / (double)cnt; // :73
        cts[p] = v_ts;
    }
}
// existing entry -> its index; new key -> remember the smallest contribution position
__device__ __forceinline__ void hgt_contrib_slots_body(HgtType src, const int64_t *mc, const int64_t *__restrict__ ckey,
                                         int64_t *cslot, int64_t *tmp_keys, int64_t *tmp_vals, int64_t tmp_mask) {
    const int64_t n = *mc;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (int64_t)gridDim.x * blockDim.x) {
        const int64_t v = ckey[p];
        if (v < 0) {
            cslot[p] = -2;
            continue;
        }
        const int64_t s = map_slot_find(src.bm_keys, src.bm_mask, v);
        if (s >= 0) {
            cslot[p] = src.bm_vals[s];
        } else {
            cslot[p] = -1;
            const int64_t t = map_slot_insert(tmp_keys, tmp_mask, v);
            atomicMin(reinterpret_cast<long long *>(&tmp_vals[t]), (long long)p);
        }
    }
}
// flags "p is the first contribution of a new key", kept per 64-position chunk: cmask[c] = the chunk's flags as a bit
// mask, flag[c] = their number.  A scan over the few hundred chunk counts (one launch of one workgroup) then ranks any
// position: rank(q) = prefix[q / 64] + popcount(cmask[q / 64] below bit q % 64).
__device__ __forceinline__ void hgt_first_flags_body(const int64_t *mc, const int64_t *__restrict__ ckey,
                                       const int64_t *__restrict__ cslot, const int64_t *tmp_keys,
                                       const int64_t *tmp_vals, int64_t tmp_mask, int64_t cap, int64_t *flag,
                                       uint64_t *cmask) {
    const int64_t n = *mc;
    const int lane = threadIdx.x & 63;
    const int64_t n_chunks = (cap + 63) >> 6;
    for (int64_t c = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; c < n_chunks;
         c += ((int64_t)gridDim.x * blockDim.x) >> 6) {
        const int64_t p = (c << 6) + lane;
        bool f = false;
        if (p < n && cslot[p] == -1) {
            const int64_t t = map_slot_find(tmp_keys, tmp_mask, ckey[p]);
            f = (tmp_vals[t] == p);
        }
        const uint64_t m = __ballot(f);
        if (lane == 0) {
            cmask[c] = m;
            flag[c] = __popcll(m);
        }
    }
}
__device__ __forceinline__ int64_t hgt_chunk_rank(const int64_t *__restrict__ chunk_prefix, const uint64_t *__restrict__ cmask,
                                                  int64_t q) {
    const uint64_t below = (q & 63) ? (cmask[q >> 6] & (~0ull >> (64 - (q & 63)))) : 0ull;
    return chunk_prefix[q >> 6] + __popcll(below);
}
// new keys get entries n_budget + rank(first contribution); the entry order is the reference's insertion order.  The
// same lane counts the contribution in its entry's bucket (below).
__device__ __forceinline__ void hgt_new_slots_body(HgtType src, const int64_t *mc, const int64_t *__restrict__ ckey,
                                                   int64_t *cslot, const int64_t *tmp_keys, const int64_t *tmp_vals,
                                                   int64_t tmp_mask, const int64_t *__restrict__ rank,
                                                   const uint64_t *__restrict__ cmask, uint32_t *bcnt) {
    const int64_t n = *mc, nb = src.ctr->n_budget;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (int64_t)gridDim.x * blockDim.x) {
        int64_t entry = cslot[p];
        if (entry == -1) {
            const int64_t v = ckey[p];
            const int64_t t = map_slot_find(tmp_keys, tmp_mask, v);
            const int64_t first = tmp_vals[t];
            entry = nb + hgt_chunk_rank(rank, cmask, first);
            cslot[p] = entry;
            if (first == p) { // :95 entry(v).or_default()
                src.bkey[entry] = v;
                src.bscore[entry] = 0.0;
                src.bts[entry] = 0;
                src.balive[entry] = 1;
                const int64_t s = map_slot_insert(src.bm_keys, src.bm_mask, v);
                src.bm_vals[s] = entry;
            }
        }
        if (entry >= 0) atomicAdd(&bcnt[entry], 1u);
    }
}

// ---- the contributions of every entry, in contribution order, WITHOUT a device-wide sort
// (a stable rocPRIM sort by entry was 9 launches and ~60 us per round: block sort + 8 merge passes; its onesweep radix
// form 5 kernels + 7 memsets and ~100 us.)  Buckets instead: new_slots counted every entry's contributions; (1) each
// entry gets the start of its bucket -- prefix inside its 64-entry chunk here, the chunk totals scanned by one
// workgroup; (2) the positions drop into their buckets in ARBITRARY order (atomic cursor); (3) accumulate puts each
// bucket into ascending position order -- positions are unique, so that IS the contribution order -- as it sums.
__device__ __forceinline__ void hgt_bucket_offsets_body(const HgtType &src, const uint32_t *__restrict__ bcnt, int64_t cap,
                                                        uint32_t *bwithin, int64_t *tflag, const int64_t *n_new,
                                                        uint32_t *long_list, int64_t *n_long) {
    // the budget grows by the entries hgt_new_slots_body placed (nothing else in this launch reads the counter)
    if (blockIdx.x == 0 && threadIdx.x == 0) src.ctr->n_budget += *n_new;
    const int lane = threadIdx.x & 63;
    const int64_t n_chunks = (cap + 63) >> 6;
    for (int64_t c = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; c < n_chunks;
         c += ((int64_t)gridDim.x * blockDim.x) >> 6) {
        const int64_t e = (c << 6) + lane;
        const uint32_t v = e < cap ? bcnt[e] : 0u;
        const uint32_t incl = wave_inclusive_scan(v);
        if (e < cap) bwithin[e] = incl - v;
        if (lane == 63) tflag[c] = (int64_t)incl;
        // entries with more than 8 contributions go on a list (any order): accumulate gives each a wavefront of its own
        const uint64_t lm = __ballot(v > 8u);
        if (lm) {
            uint32_t at = 0;
            if (lane == 0) at = (uint32_t)atomicAdd(reinterpret_cast<unsigned long long *>(n_long), (unsigned long long)__popcll(lm));
            at = (uint32_t)__builtin_amdgcn_readfirstlane((int)at);
            if (v > 8u) long_list[at + __popcll(lm & (lane ? (~0ull >> (64 - lane)) : 0ull))] = (uint32_t)e;
        }
    }
}
__device__ __forceinline__ void hgt_bucket_scatter_body(const int64_t *mc, const int64_t *__restrict__ cslot,
                                                        const int64_t *__restrict__ trank,
                                                        const uint32_t *__restrict__ bwithin, uint32_t *bcur,
                                                        uint32_t *bucket) {
    const int64_t n = *mc;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (int64_t)gridDim.x * blockDim.x) {
        const int64_t e = cslot[p];
        if (e < 0) continue;
        bucket[(uint32_t)trank[e >> 6] + bwithin[e] + atomicAdd(&bcur[e], 1u)] = (uint32_t)p;
    }
}
__device__ __forceinline__ void hgt_ce(uint32_t &x, uint32_t &y) {
    const uint32_t lo = min(x, y), hi = max(x, y);
    x = lo;
    y = hi;
}
// Wavefront-wide stable LSD radix sort of n positions (8-bit digits; equal digits keep their order through ballot matching)
// between the arrays a and b; returns the one that holds the result.  `hist` = 256 words of this wavefront's LDS.
template <bool IN_GLOBAL>
__device__ __forceinline__ uint32_t *hgt_wave_radix_sort(uint32_t *a, uint32_t *b, uint32_t n, int pbits, uint32_t *hist) {
    const int lane = threadIdx.x & 63;
    const uint64_t lt_mask = lane ? (~0ull >> (64 - lane)) : 0ull;
    for (int shift = 0; shift < pbits; shift += 8) {
        for (int i = lane; i < 256; i += 64) hist[i] = 0;
        wave_lds_handoff();
        for (uint32_t t0 = 0; t0 < n; t0 += 256) { // four tiles' loads in flight
            uint32_t v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = t0 + u * 64 + lane < n ? a[t0 + u * 64 + lane] : 0u;
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (t0 + u * 64 + lane < n) atomicAdd(&hist[(v[u] >> shift) & 255u], 1u);
        }
        wave_lds_handoff();
        { // exclusive scan of the 256 digit counts, four per lane
            const uint32_t x0 = hist[4 * lane], x1 = hist[4 * lane + 1], x2 = hist[4 * lane + 2], x3 = hist[4 * lane + 3];
            const uint32_t sm = x0 + x1 + x2 + x3;
            const uint32_t base = wave_inclusive_scan(sm) - sm;
            wave_lds_handoff();
            hist[4 * lane] = base;
            hist[4 * lane + 1] = base + x0;
            hist[4 * lane + 2] = base + x0 + x1;
            hist[4 * lane + 3] = base + x0 + x1 + x2;
        }
        wave_lds_handoff();
        uint32_t nxt = lane < n ? a[lane] : 0u;
        for (uint32_t t0 = 0; t0 < n; t0 += 64) { // tiles in order; the next tile's load overlaps this tile's ranking
            const bool valid = t0 + lane < n;
            const uint32_t v = nxt;
            if (t0 + 64 < n) nxt = t0 + 64 + lane < n ? a[t0 + 64 + lane] : 0u;
            const uint32_t d = (v >> shift) & 255u;
            uint64_t same = __ballot(valid);
#pragma unroll
            for (int bit = 0; bit < 8; ++bit) {
                const bool one = (d >> bit) & 1u;
                const uint64_t bal = __ballot(one);
                same &= one ? bal : ~bal;
            }
            const uint32_t before = (uint32_t)__popcll(same & lt_mask), group = (uint32_t)__popcll(same);
            const uint32_t at = hist[d];
            wave_lds_handoff();
            if (valid) {
                b[at + before] = v;
                if (before + 1 == group) hist[d] = at + group;
            }
            wave_lds_handoff();
        }
        if (IN_GLOBAL) { // the next pass reads what this one wrote
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        uint32_t *t = a;
        a = b;
        b = t;
    }
    return a;
}
// score += 1/deg in contribution order (:96), timestamp = the last contribution's (:97).  One lane per entry while its
// bucket holds <= 8 positions (a 19-comparator network in registers; most entries hold 1-3); a longer bucket takes the
// whole wavefront: up to 64 positions are ranked by counting in registers, more go through the radix sort above -- in
// the wavefront's LDS up to HGT_RUN_LDS positions, between the bucket and its twin in global memory beyond -- linear in
// the bucket's length, so a hub that every sample points at costs its length, not its square.  The f64 sum itself stays
// left to right.
constexpr int HGT_ACC_THREADS = 256;
constexpr uint32_t HGT_RUN_LDS = 1024;
constexpr int HGT_ACC_LONG_BLOCKS = 128; // workgroups at the end of accumulate's grid that take the long buckets
__device__ __forceinline__ void hgt_accumulate_body(const HgtType &src, const uint32_t *__restrict__ bcnt,
                                                    const uint32_t *__restrict__ bwithin, const int64_t *__restrict__ trank,
                                                    uint32_t *bucket, uint32_t *bucket2, int64_t cap, int pbits,
                                                    const double *__restrict__ cinv, const int64_t *__restrict__ cts,
                                                    const uint32_t *__restrict__ long_list, const int64_t *n_long) {
    __shared__ uint32_t hist_s[HGT_ACC_THREADS / 64][256];
    __shared__ uint32_t perm_s[HGT_ACC_THREADS / 64][64];
    __shared__ double sum_s[HGT_ACC_THREADS / 64][64];
    __shared__ uint32_t run_s[HGT_ACC_THREADS / 64][2][HGT_RUN_LDS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t *hist = hist_s[wave], *perm = perm_s[wave];
    double *sbuf = sum_s[wave];
    const int64_t nbud = src.ctr->n_budget;
    const int64_t n_chunks = (min(cap, nbud) + 63) >> 6;
    const int64_t short_blocks = (int64_t)gridDim.x - HGT_ACC_LONG_BLOCKS;
    if ((int64_t)blockIdx.x >= short_blocks) { // the long buckets, one wavefront each
        const int64_t n_l = *n_long, stride = (int64_t)HGT_ACC_LONG_BLOCKS * (HGT_ACC_THREADS / 64);
        for (int64_t i = ((int64_t)blockIdx.x - short_blocks) * (HGT_ACC_THREADS / 64) + wave; i < n_l; i += stride) {
            const int64_t e2 = long_list[i];
            const uint32_t L2 = bcnt[e2];
            const uint32_t st2 = (uint32_t)trank[e2 >> 6] + bwithin[e2];
            double score = src.bscore[e2];
            uint32_t last = 0;
            if (L2 <= 64) {
                const uint32_t p = (uint32_t)lane < L2 ? bucket[st2 + lane] : 0xffffffffu;
                uint32_t r = 0;
                for (uint32_t j = 0; j < L2; ++j) r += (uint32_t)__builtin_amdgcn_readlane((int)p, (int)j) < p ? 1u : 0u;
                if ((uint32_t)lane < L2) perm[r] = p;
                wave_lds_handoff();
                const uint32_t ps = (uint32_t)lane < L2 ? perm[lane] : 0u;
                const double cv = (uint32_t)lane < L2 ? cinv[ps] : 0.0;
                double total;
                (void)wave_serial_prefix_f64(cv, score, &total, sbuf); // x + 0.0 = x: the lanes past the bucket add nothing
                score = total;
                last = (uint32_t)__builtin_amdgcn_readlane((int)ps, (int)(L2 - 1));
            } else {
                const uint32_t *sorted;
                if (L2 <= HGT_RUN_LDS) {
                    uint32_t *ra = run_s[wave][0], *rb = run_s[wave][1];
                    for (uint32_t t0 = 0; t0 < L2; t0 += 256) {
                        uint32_t v[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) v[u] = t0 + u * 64 + lane < L2 ? bucket[st2 + t0 + u * 64 + lane] : 0u;
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            if (t0 + u * 64 + lane < L2) ra[t0 + u * 64 + lane] = v[u];
                    }
                    wave_lds_handoff();
                    sorted = hgt_wave_radix_sort<false>(ra, rb, L2, pbits, hist);
                } else {
                    sorted = hgt_wave_radix_sort<true>(bucket + st2, bucket2 + st2, L2, pbits, hist);
                }
                for (uint32_t t0 = 0; t0 < L2; t0 += 256) { // four tiles' gathers in flight, then their sums in order
                    uint32_t ps[4];
                    double cv[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const bool valid = t0 + u * 64 + lane < L2;
                        ps[u] = valid ? sorted[t0 + u * 64 + lane] : 0u;
                        cv[u] = valid ? cinv[ps[u]] : 0.0;
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (t0 + u * 64 >= L2) break; // uniform
                        double total;
                        (void)wave_serial_prefix_f64(cv[u], score, &total, sbuf);
                        score = total;
                        if (t0 + u * 64 + 64 >= L2) last = (uint32_t)__builtin_amdgcn_readlane((int)ps[u], (int)(L2 - 1 - t0 - u * 64));
                    }
                }
            }
            if (lane == 0) {
                src.bscore[e2] = score;
                src.bts[e2] = cts[last];
            }
        }
        return;
    }
    for (int64_t c = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; c < n_chunks; c += (short_blocks * blockDim.x) >> 6) {
        const int64_t e = (c << 6) + lane;
        const uint32_t L = (e < nbud && e < cap) ? bcnt[e] : 0u;
        const uint32_t start = L ? (uint32_t)trank[c] + bwithin[e] : 0u;
        if (L > 0 && L <= 8) {
            uint32_t a[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) a[u] = (uint32_t)u < L ? bucket[start + u] : 0xffffffffu;
            hgt_ce(a[0], a[1]), hgt_ce(a[2], a[3]), hgt_ce(a[4], a[5]), hgt_ce(a[6], a[7]);
            hgt_ce(a[0], a[2]), hgt_ce(a[1], a[3]), hgt_ce(a[4], a[6]), hgt_ce(a[5], a[7]);
            hgt_ce(a[1], a[2]), hgt_ce(a[5], a[6]), hgt_ce(a[0], a[4]), hgt_ce(a[3], a[7]);
            hgt_ce(a[1], a[5]), hgt_ce(a[2], a[6]);
            hgt_ce(a[1], a[4]), hgt_ce(a[3], a[6]);
            hgt_ce(a[2], a[4]), hgt_ce(a[3], a[5]);
            hgt_ce(a[3], a[4]);
            double cv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) cv[u] = (uint32_t)u < L ? cinv[a[u]] : 0.0; // the gathers are issued together
            double score = src.bscore[e];
            uint32_t last = a[0];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if ((uint32_t)u < L) {
                    score = score + cv[u];
                    last = a[u];
                }
            src.bscore[e] = score;
            src.bts[e] = cts[last];
        }
    }
}

// ---------------------------------------------------------------- update_budget steps side by side (gridDim.y = steps)
// A call is a chain of launches of a few microseconds each: bound by the LENGTH of the chain, not by work.  Steps of
// update_budget that feed DIFFERENT source types' budgets are independent (a step reads its destination type's layer and
// writes its source type's budget), so up to HGT_MAX_PAR of them run as ONE launch per phase, blockIdx.y = step, each on
// scratch of its own; steps that feed the same budget stay in their canonical order (later rounds).  cfg4's eight steps
// take three rounds.  (The same over six HIP streams cost more in fork / join events than it won: DESIGN.md 4.6.)
constexpr int HGT_MAX_PAR = 4;
struct HgtStep {
    HgtType dst, src;
    HgtTypeCtr *src_ctr;
    const int64_t *ptrs, *indices, *edge_ts;
    int64_t pad; // the source budget's capacity
    int64_t *ccnt, *coff, *ckey, *cts, *cslot, *tmp_keys, *tmp_vals, *flag, *rank, *scal, *tflag, *trank;
    uint32_t *bcnt, *bwithin, *bcur, *bucket, *bucket2, *bucket3;
    double *cinv;
    uint64_t *cmask;
};
struct HgtSteps {
    HgtStep s[HGT_MAX_PAR];
    int pbits; // bits of a contribution position
};
__global__ void __launch_bounds__(SCAN1_THREADS) hgt_count_scan_steps_kernel(const HgtSteps S, int64_t cap) {
    const HgtStep &a = S.s[blockIdx.y];
    hgt_contrib_count_scan_body(a.dst, a.src_ctr, a.ptrs, a.ccnt, a.coff, cap, a.scal + 0);
}
// contributions; also the empty min-position map of the step's new keys and the zeroed bucket counters
__global__ void hgt_gen_steps_kernel(const HgtSteps S, int has_timerange, int64_t tr_lo, int64_t tr_hi, int64_t cap,
                                     int64_t tmp_cap) {
    const HgtStep &a = S.s[blockIdx.y];
    fill2_i64_body(a.tmp_keys, tmp_cap, MAP_EMPTY, a.tmp_vals, tmp_cap, (int64_t)INT64_MAX);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.pad; i += (int64_t)gridDim.x * blockDim.x) {
        a.bcnt[i] = 0;
        a.bcur[i] = 0;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) a.scal[3] = 0; // long buckets listed
    hgt_contrib_gen_body(a.dst, a.src, a.ptrs, a.indices, a.edge_ts, has_timerange, tr_lo, tr_hi, a.ccnt, a.coff, cap, a.ckey,
                         a.cinv, a.cts);
}
__global__ void hgt_slots_steps_kernel(const HgtSteps S, int64_t tmp_mask) {
    const HgtStep &a = S.s[blockIdx.y];
    hgt_contrib_slots_body(a.src, a.scal + 0, a.ckey, a.cslot, a.tmp_keys, a.tmp_vals, tmp_mask);
}
__global__ void hgt_first_flags_steps_kernel(const HgtSteps S, int64_t tmp_mask, int64_t cap) {
    const HgtStep &a = S.s[blockIdx.y];
    hgt_first_flags_body(a.scal + 0, a.ckey, a.cslot, a.tmp_keys, a.tmp_vals, tmp_mask, cap, a.flag, a.cmask);
}
__global__ void __launch_bounds__(SCAN1_THREADS) hgt_scan1_steps_kernel(const HgtSteps S, int64_t n) {
    const HgtStep &a = S.s[blockIdx.y];
    hgt_scan1_body(a.flag, n, a.rank, a.scal + 1);
}
__global__ void hgt_new_slots_steps_kernel(const HgtSteps S, int64_t tmp_mask) {
    const HgtStep &a = S.s[blockIdx.y];
    hgt_new_slots_body(a.src, a.scal + 0, a.ckey, a.cslot, a.tmp_keys, a.tmp_vals, tmp_mask, a.rank, a.cmask, a.bcnt);
}
__global__ void hgt_bucket_offsets_steps_kernel(const HgtSteps S) {
    const HgtStep &a = S.s[blockIdx.y];
    hgt_bucket_offsets_body(a.src, a.bcnt, a.pad, a.bwithin, a.tflag, a.scal + 1, a.bucket2, a.scal + 3);
}
__global__ void __launch_bounds__(SCAN1_THREADS) hgt_bucket_scan_steps_kernel(const HgtSteps S) {
    const HgtStep &a = S.s[blockIdx.y];
    hgt_scan1_body(a.tflag, (a.pad + 63) >> 6, a.trank, a.scal + 2);
}
__global__ void hgt_bucket_scatter_steps_kernel(const HgtSteps S) {
    const HgtStep &a = S.s[blockIdx.y];
    hgt_bucket_scatter_body(a.scal + 0, a.cslot, a.trank, a.bwithin, a.bcur, a.bucket);
}
__global__ void __launch_bounds__(HGT_ACC_THREADS) hgt_accumulate_steps_kernel(const HgtSteps S) {
    const HgtStep &a = S.s[blockIdx.y];
    hgt_accumulate_body(a.src, a.bcnt, a.bwithin, a.trank, a.bucket, a.bucket3, a.pad, S.pbits, a.cinv, a.cts, a.bucket2,
                        a.scal + 3);
}

// ---------------------------------------------------------------- sample_from (hgt_sampling.rs:104-135)
// The node types of a layer sample from their own budgets independently (:201-221): ONE launch per layer, one workgroup
// per type, which (1) compacts the type's live budget entries in entry order -- every wavefront counts the live entries
// of its run of chunks, then places them behind the earlier wavefronts' -- (2) runs the weighted reservoir below and (3) moves
// the chosen entries to the node list.  The phases hand their arrays over through global memory inside the workgroup.
// -DTG_HGT_STAMPS (an experiment build, not the product): thread 0 of every sample_from workgroup stamps the 100 MHz
// wall clock at its phase boundaries; tg_hgt_sample prints the last layer's phase times of every type to stderr.
#ifdef TG_HGT_STAMPS
__device__ long long hgt_stamps[HGT_MULTI_TYPES_STAMPS][16];
#define HGT_STAMP(i)                                                                                                   \
    do {                                                                                                               \
        if (threadIdx.x == 0) hgt_stamps[blockIdx.x][i] = (long long)wall_clock64();                                   \
    } while (0)
#else
#define HGT_STAMP(i)                                                                                                   \
    do {                                                                                                               \
    } while (0)
#endif
__device__ __forceinline__ void hgt_wg_handoff() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}
__device__ __forceinline__ void hgt_live_list_body(const HgtType &ty, int64_t *__restrict__ live, double *__restrict__ wlive,
                                                   int64_t *n_live) {
    __shared__ int64_t wave_live[SCAN1_THREADS / 64];
    const int64_t n = ty.ctr->n_budget;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n_waves = blockDim.x >> 6;
    const uint64_t lt_mask = lane ? (~0ull >> (64 - lane)) : 0ull;
    const int64_t n_chunks = (n + 63) >> 6;
    const int64_t per = (n_chunks + n_waves - 1) / n_waves; // every wavefront takes a contiguous run of 64-entry chunks
    const int64_t c_lo = min(n_chunks, (int64_t)wave * per), c_hi = min(n_chunks, c_lo + per);
    int64_t mine = 0;
    for (int64_t c0 = c_lo; c0 < c_hi; c0 += 16) { // its live entries, sixteen chunks' loads in flight
        int64_t alive[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int64_t i = ((c0 + u) << 6) + lane;
            alive[u] = (c0 + u < c_hi && i < n) ? ty.balive[i] : 0;
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) mine += alive[u] != 0;
    }
    mine = wave_sum(mine);
    if (lane == 0) wave_live[wave] = mine;
    __syncthreads();
    int64_t at = 0, all = 0;
    for (int w = 0; w < n_waves; ++w) {
        if (w < wave) at += wave_live[w];
        all += wave_live[w];
    }
    if (tid == 0) *n_live = all;
    HGT_STAMP(1);
    HGT_STAMP(2);
    for (int64_t c0 = c_lo; c0 < c_hi; c0 += 8) { // second sweep: every live entry's place and weight, in entry order
        int64_t alive[8];
        double sc[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int64_t i = ((c0 + u) << 6) + lane;
            const bool in = c0 + u < c_hi && i < n;
            alive[u] = in ? ty.balive[i] : 0;
            sc[u] = in ? ty.bscore[i] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const uint64_t m = __ballot(alive[u] != 0);
            if (alive[u] != 0) {
                const int64_t r = at + __popcll(m & lt_mask);
                live[r] = ((c0 + u) << 6) + lane;
                wlive[r] = sc[u] * sc[u]; // the reservoir's weight (:104-110)
            }
            at += __popcll(m);
        }
    }
    hgt_wg_handoff();
}
constexpr int64_t HGT_LDS_SLOTS = 8192; // samples per layer whose slot tables fit 64 KB of LDS
// The reference's weighted reservoir over the live entries, weights score^2 (:104-135), by ONE WORKGROUP.  With
// philox-mode's blocked running sum (tg_device.h wave_blocked_prefix_f64) the chunks of 64 entries only meet in the
// left-to-right sum of their totals, so: (A) every wavefront forms the totals of its chunks; (B) one lane adds them up left
// to right (the defined order; ~12 ns per chunk); (C) every wavefront draws for its chunks -- a candidate m >= k that is
// accepted raises slot_rank[its slot] to m with an atomic max: the LAST accepted candidate of a slot wins, as in the
// reference's sequential loop, whatever the order the chunks are visited in; (D) slot s holds entry slot_rank[s], or s
// itself if nothing hit it.  One wavefront used to walk the whole list: 276 us per layer for 50-80 K entries.
__device__ __forceinline__ void hgt_reservoir_body(const HgtType &ty, const int64_t *n_live_ptr,
                                                   const int64_t *__restrict__ live, int64_t k, uint64_t seed,
                                                   uint64_t call_id, uint64_t draw_id, int64_t *chosen, int64_t *n_chosen,
                                                   int *panic, unsigned char *smem, double *carries, uint32_t *slots_global,
                                                   const double *__restrict__ wlive) {
    // slot table: LDS up to 8192 samples per layer, the caller's global scratch beyond
    uint32_t *slot_rank = (k > HGT_LDS_SLOTS) ? slots_global : reinterpret_cast<uint32_t *>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n_waves = blockDim.x >> 6;
    const int64_t n = *n_live_ptr;
    if (k < 0) { // num_samples has no entry for this type: the reference panics once the budget exists (:202)
        if (tid == 0) {
            if (ty.ctr->present) *panic = 1;
            *n_chosen = 0;
        }
        return;
    }
    if (k == 0) {
        if (tid == 0) *n_chosen = 0;
        return;
    }
    const CallKey ck = call_key(seed, call_id, TAG_HGT);
    const bool gslots = k > HGT_LDS_SLOTS;
    auto block_handoff = [&]() { // slot-table entries / chunk totals pass between the wavefronts of this workgroup
        if (gslots) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        if (gslots) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    };
    const int64_t nc = (n + 63) >> 6;
    for (int64_t s = tid; s < k; s += blockDim.x) slot_rank[s] = 0;
    constexpr int U = 4; // chunks in flight per wavefront
    HGT_STAMP(3);
    for (int64_t c0 = (int64_t)wave * U; c0 < nc; c0 += (int64_t)n_waves * U) { // (A) weights and chunk totals
        double wv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t m = (c0 + u) * 64 + lane;
            wv[u] = m < n ? wlive[m] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (c0 + u >= nc) break; // uniform
            double tot;
            (void)wave_blocked_prefix_f64(wv[u], 0.0, &tot);
            if (lane == 0) carries[c0 + u] = tot;
        }
    }
    hgt_wg_handoff();
    HGT_STAMP(4);
    { // (B) carries[c] = sum of the totals before chunk c, added left to right: tiles of 1024 totals through LDS, one lane
      // runs the dependent adds (the LDS reads pipeline ahead of them)
        __shared__ double tile_s[1024];
        __shared__ double run_s;
        if (tid == 0) run_s = 0.0;
        for (int64_t c0 = 0; c0 < nc; c0 += 1024) {
            const int64_t c = c0 + tid;
            if (tid < 1024 && c < nc) tile_s[tid] = carries[c];
            __syncthreads();
            if (tid == 0) {
                double run = run_s;
                const int cnt_t = (int)min((int64_t)1024, nc - c0);
                for (int i = 0; i < cnt_t; ++i) {
                    const double t = tile_s[i];
                    tile_s[i] = run;
                    run = run + t;
                }
                run_s = run;
            }
            __syncthreads();
            if (tid < 1024 && c < nc) carries[c] = tile_s[tid];
            __syncthreads();
        }
    }
    hgt_wg_handoff();
    HGT_STAMP(5);
    for (int64_t c0 = (int64_t)wave * U; c0 < nc; c0 += (int64_t)n_waves * U) { // (C) draws
        double wv[U], cr[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t m = (c0 + u) * 64 + lane;
            wv[u] = m < n ? wlive[m] : 0.0;
            cr[u] = c0 + u < nc ? carries[c0 + u] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (c0 + u >= nc) break; // uniform
            const int64_t m = (c0 + u) * 64 + lane;
            double tot;
            const double pref = wave_blocked_prefix_f64(wv[u], cr[u], &tot); // blocked running sum, sampling.rs:40,48
            if (m < n && m >= k) {
                if (!(0.0 < pref)) {
                    *panic = 1;
                } else {
                    const Draw d = draw(ck, draw_id, (uint32_t)m, D1_WEIGHTED);
                    const double j = u64_to_f64_01(d.a()) * pref + 0.0;
                    if (j < wv[u]) atomicMax(&slot_rank[bounded64(d.b(), (uint64_t)k)], (uint32_t)m);
                }
            }
        }
    }
    block_handoff();
    HGT_STAMP(6);
    const int64_t cnt = min(n, k); // (D)  (m >= k >= 1, so a rank of 0 means "never hit": the slot keeps entry s, :37-45)
    for (int64_t s = tid; s < cnt; s += blockDim.x) {
        const uint32_t r = gslots ? __atomic_load_n(&slot_rank[s], __ATOMIC_RELAXED) : slot_rank[s];
        chosen[s] = r ? (int64_t)r : s;
    }
    if (tid == 0) *n_chosen = cnt;
}
// :213-221 move the samples to the node list, give them local ids, erase them from the budget
__device__ __forceinline__ void hgt_append_body(const HgtType &ty, const int64_t *__restrict__ live,
                                                const int64_t *__restrict__ chosen, const int64_t *n_chosen) {
    const int64_t cnt = *n_chosen, base = ty.ctr->n_nodes;
    for (int64_t s = threadIdx.x; s < cnt; s += blockDim.x) {
        const int64_t entry = live[chosen[s]];
        const int64_t v = ty.bkey[entry];
        ty.nodes[base + s] = v;
        ty.ts[base + s] = ty.bts[entry];
        const int64_t h = map_slot_insert(ty.tl_keys, ty.tl_mask, v);
        ty.tl_vals[h] = base + s;
        ty.balive[entry] = 0;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        ty.ctr->lay_begin = base;
        ty.ctr->lay_end = base + cnt;
        ty.ctr->n_nodes = base + cnt;
    }
}
constexpr int HGT_MULTI_TYPES = 8; // node types per launch (more types: more launches)
struct HgtSampleArgs {
    HgtType ty[HGT_MULTI_TYPES];
    int64_t *n_live[HGT_MULTI_TYPES], *live[HGT_MULTI_TYPES], *chosen[HGT_MULTI_TYPES], *n_chosen[HGT_MULTI_TYPES];
    int64_t k[HGT_MULTI_TYPES];
    uint32_t *slots[HGT_MULTI_TYPES];
    double *carries[HGT_MULTI_TYPES], *wlive[HGT_MULTI_TYPES];
};
__global__ void __launch_bounds__(SCAN1_THREADS)
    hgt_sample_layer_kernel(const HgtSampleArgs a, uint64_t seed, uint64_t call_id, int64_t layer, int first_type, int n_types,
                            int *panic) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int t = blockIdx.x;
    HGT_STAMP(0);
    hgt_live_list_body(a.ty[t], a.live[t], a.wlive[t], a.n_live[t]);
    hgt_reservoir_body(a.ty[t], a.n_live[t], a.live[t], a.k[t], seed, call_id, (uint64_t)(layer * n_types + first_type + t),
                       a.chosen[t], a.n_chosen[t], panic, smem, a.carries[t], a.slots[t], a.wlive[t]);
    hgt_wg_handoff();
    HGT_STAMP(7);
    hgt_append_body(a.ty[t], a.live[t], a.chosen[t], a.n_chosen[t]);
    HGT_STAMP(8);
}
// empty type tables and zeroed counters, all types of a call in one launch
struct HgtInitArgs {
    HgtType ty[HGT_MULTI_TYPES];
    int64_t tl_cap[HGT_MULTI_TYPES], bm_cap[HGT_MULTI_TYPES];
};
__global__ void hgt_init_types_kernel(const HgtInitArgs a, int64_t *scalars, int n_scalars) {
    const HgtType &y = a.ty[blockIdx.y];
    fill2_i64_body(y.tl_keys, a.tl_cap[blockIdx.y], MAP_EMPTY, y.tl_vals, a.tl_cap[blockIdx.y], (int64_t)-1);
    fill2_i64_body(y.bm_keys, a.bm_cap[blockIdx.y], MAP_EMPTY, nullptr, 0, 0);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        HgtTypeCtr z;
        z.n_nodes = z.lay_begin = z.lay_end = z.n_budget = z.present = 0;
        *y.ctr = z;
    }
    if (blockIdx.x == 0 && blockIdx.y == 0 && (int)threadIdx.x < n_scalars) scalars[threadIdx.x] = 0;
}

// ---------------------------------------------------------------- edges (hgt_sampling.rs:244-268)
// one WAVEFRONT per destination node; candidates at a fixed stride of 50, -1 where dropped.  Lane s owns
// reservoir slot s: the ticket chain (k = 50 sequential bounded draws over a shrinking urn) is resolved with
// ballots over the lanes' displaced entries, then all slots gather and look up `to_local` in parallel.
__device__ __forceinline__ void hgt_edge_candidates_body(const HgtType &dst, const HgtType &src,
                                                        const int64_t *__restrict__ ptrs,
                                                        const int64_t *__restrict__ indices, int64_t cap_nodes,
                                                        uint64_t seed, uint64_t call_id, uint32_t tag, int64_t *cand_j,
                                                        int64_t *cand_ep, int64_t *kept) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
    const int64_t n_nodes = dst.ctr->n_nodes;
    const CallKey ck = call_key(seed, call_id, tag);
    for (int64_t i = (int64_t)blockIdx.x * n_waves + wave; i < cap_nodes; i += (int64_t)gridDim.x * n_waves) {
        int64_t *cj = cand_j + i * HGT_MAX_NB, *ce = cand_ep + i * HGT_MAX_NB;
        if (i >= n_nodes) {
            if (lane < HGT_MAX_NB) cj[lane] = -1;
            if (lane == 0) kept[i] = 0;
            continue;
        }
        const int64_t w = dst.nodes[i];
        const int64_t b = ptrs[w], len = ptrs[w + 1] - b;
        const int k = (int)min(len, (int64_t)HGT_MAX_NB); // :258
        int64_t pos = lane;
        if (len > HGT_MAX_NB) { // reservoir by tickets, k = 50
            const uint32_t n = (uint32_t)len;
            uint32_t myK = 0xffffffffu, myV = 0, mypos = 0;
            Draw d;
            for (int s = 0; s < HGT_MAX_NB; ++s) {
                const uint32_t m = (n - 1u) - (uint32_t)s;
                if ((s & 3) == 0) d = draw(ck, (uint64_t)i, (uint32_t)(s >> 2), 0u);
                const uint32_t r = slot_draw_from(d, ck, (uint64_t)i, (uint32_t)s, 0u, m), last = m - 1u;
                const uint64_t mr = __ballot(lane < s && myK == r);     // the latest displaced entry wins
                const uint64_t ml = __ballot(lane < s && myK == last);
                const uint32_t vr = __shfl(myV, mr ? 63 - __clzll((long long)mr) : 0, 64);
                const uint32_t vl = __shfl(myV, ml ? 63 - __clzll((long long)ml) : 0, 64);
                const uint32_t tr = mr ? vr : r, tl = ml ? vl : last;
                if (lane == s) {
                    myK = r;
                    myV = tl;
                    mypos = (tr < n - (uint32_t)HGT_MAX_NB) ? (uint32_t)HGT_MAX_NB + tr : (uint32_t)s;
                }
            }
            pos = mypos;
        }
        int64_t j = -1, ep = -1;
        if (lane < k) {
            ep = b + pos;
            const int64_t v = indices[ep];
            const int64_t h = map_slot_find(src.tl_keys, src.tl_mask, v); // :263
            if (h >= 0) j = src.tl_vals[h];
        }
        if (lane < HGT_MAX_NB) {
            cj[lane] = j;
            ce[lane] = ep;
        }
        const uint64_t m = __ballot(j >= 0);
        if (lane == 0) kept[i] = __popcll(m); // edges of this destination node; their scan places them (:264 order)
    }
}
// one wavefront per destination node: its kept candidates, in slot order, from off[i] on
__device__ __forceinline__ void hgt_edge_emit_body(const int64_t *__restrict__ cand_j, const int64_t *__restrict__ cand_ep,
                                                   const int64_t *__restrict__ off, int64_t n_nodes_cap, int64_t *rows,
                                                   int64_t *cols, int64_t *eidx) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
    const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    for (int64_t i = (int64_t)blockIdx.x * n_waves + wave; i < n_nodes_cap; i += (int64_t)gridDim.x * n_waves) {
        const int64_t q = i * HGT_MAX_NB + lane;
        const int64_t j = (lane < HGT_MAX_NB) ? cand_j[q] : -1;
        const uint64_t m = __ballot(j >= 0);
        if (j >= 0) {
            const int64_t e = off[i] + __popcll(m & lt_mask);
            rows[e] = j;           // :264 j
            cols[e] = i;           //      i
            eidx[e] = cand_ep[q];  //      edge_ptr
        }
    }
}
// the relations of a call side by side (blockIdx.y = relation, HGT_EDGE_PAR per launch, scratch of its own each)
constexpr int HGT_EDGE_PAR = 8;
struct HgtEdgeRel {
    HgtType dst, src;
    const int64_t *ptrs, *indices;
    int64_t cap_n;
    uint32_t tag;
    int64_t *cand_j, *cand_ep, *kept, *off, *rows, *cols, *eidx, *n_edges;
};
struct HgtEdgeRels {
    HgtEdgeRel r[HGT_EDGE_PAR];
};
__global__ void hgt_edge_candidates_rels_kernel(const HgtEdgeRels E, uint64_t seed, uint64_t call_id) {
    const HgtEdgeRel &a = E.r[blockIdx.y];
    hgt_edge_candidates_body(a.dst, a.src, a.ptrs, a.indices, a.cap_n, seed, call_id, a.tag, a.cand_j, a.cand_ep, a.kept);
}
__global__ void __launch_bounds__(SCAN1_THREADS) hgt_edge_scan_rels_kernel(const HgtEdgeRels E) {
    const HgtEdgeRel &a = E.r[blockIdx.y];
    hgt_scan1_body(a.kept, a.cap_n, a.off, a.n_edges);
}
__global__ void hgt_edge_emit_rels_kernel(const HgtEdgeRels E) {
    const HgtEdgeRel &a = E.r[blockIdx.y];
    hgt_edge_emit_body(a.cand_j, a.cand_ep, a.off, a.cap_n, a.rows, a.cols, a.eidx);
}
__global__ void hgt_finish_kernel(const HgtTypeCtr *ctr, int n_types, int64_t *n_samples, const int *panic, int *panic_out) {
    for (int t = threadIdx.x; t < n_types; t += blockDim.x) n_samples[t] = ctr[t].n_nodes;
    if (threadIdx.x == 0) *panic_out = *panic;
}

// ---------------------------------------------------------------- workspace layout
struct HgtPlan {
    int T, R, H;
    std::vector<int64_t> cap_nodes, cap_budget, tl_cap, bm_cap;
    int64_t max_layer, mc_cap, tmp_cap, max_budget, max_k, max_nodes, edge_cap, scan_cap;
    int edge_lanes; // relations whose edges are rebuilt side by side
    size_t scan_temp_bytes;
    size_t total_bytes;
};
static inline size_t align16(size_t x) { return (x + 15) & ~(size_t)15; }

struct HgtStepScratch {
    int64_t *ccnt, *coff, *ckey, *cts, *cslot, *tmp_keys, *tmp_vals, *flag, *rank, *scal, *tflag, *trank;
    uint32_t *bcnt, *bwithin, *bcur, *bucket, *bucket2, *bucket3;
    double *cinv;
    uint64_t *cmask;
};
struct HgtEdgeScratch {
    int64_t *cand_j, *cand_ep, *kept, *off;
};
// every array of a call; hgt_carve hands them out of the workspace (base = NULL: only adds up the bytes)
struct HgtBuffers {
    HgtTypeCtr *ctr;
    int64_t *scal; // 8 words behind the counters; the panic flag is the last
    int *panic;
    std::vector<HgtType> ty;
    HgtStepScratch sc[HGT_MAX_PAR];
    std::vector<int64_t *> live, chosen;
    std::vector<uint32_t *> slots;
    std::vector<double *> carries, wlive;
    int64_t *n_live, *n_chosen;
    std::vector<HgtEdgeScratch> ed;
    void *scan_temp;
};
static size_t hgt_carve(const HgtPlan &pl, unsigned char *base, HgtBuffers &B) {
    size_t off = 0;
    auto take = [&](size_t bytes) {
        unsigned char *p = base ? base + off : nullptr;
        off += align16(bytes);
        return p;
    };
    auto i64 = [&](size_t n) { return reinterpret_cast<int64_t *>(take(8 * n)); };
    const int T = pl.T;
    unsigned char *ctr_block = take(sizeof(HgtTypeCtr) * T + 64);
    B.ctr = reinterpret_cast<HgtTypeCtr *>(ctr_block);
    B.scal = reinterpret_cast<int64_t *>(ctr_block ? ctr_block + sizeof(HgtTypeCtr) * T : nullptr);
    B.panic = reinterpret_cast<int *>(B.scal ? B.scal + 7 : nullptr);
    B.ty.assign((size_t)T, HgtType{});
    for (int t = 0; t < T; ++t) {
        HgtType &y = B.ty[(size_t)t];
        y.tl_keys = i64((size_t)pl.tl_cap[t]);
        y.tl_vals = i64((size_t)pl.tl_cap[t]);
        y.tl_mask = pl.tl_cap[t] - 1;
        y.bkey = i64((size_t)pl.cap_budget[t]);
        y.bts = i64((size_t)pl.cap_budget[t]);
        y.balive = i64((size_t)pl.cap_budget[t]);
        y.bscore = reinterpret_cast<double *>(i64((size_t)pl.cap_budget[t]));
        y.bm_keys = i64((size_t)pl.bm_cap[t]);
        y.bm_vals = i64((size_t)pl.bm_cap[t]);
        y.bm_mask = pl.bm_cap[t] - 1;
        y.ctr = B.ctr ? B.ctr + t : nullptr;
    }
    const size_t step_chunks = (size_t)(pl.mc_cap / 64 + 3);
    for (int y = 0; y < HGT_MAX_PAR; ++y) {
        HgtStepScratch &s = B.sc[y];
        s.ccnt = i64((size_t)pl.max_layer);
        s.coff = i64((size_t)pl.max_layer + 1);
        s.ckey = i64((size_t)pl.mc_cap);
        s.cinv = reinterpret_cast<double *>(i64((size_t)pl.mc_cap));
        s.cts = i64((size_t)pl.mc_cap);
        s.cslot = i64((size_t)pl.mc_cap);
        s.tmp_keys = i64((size_t)pl.tmp_cap);
        s.tmp_vals = i64((size_t)pl.tmp_cap);
        s.flag = i64(step_chunks);
        s.rank = i64(step_chunks);
        s.cmask = reinterpret_cast<uint64_t *>(i64(step_chunks));
        s.scal = i64(8); // [0] contributions [1] new entries [2] kept contributions [3] long buckets
        const size_t budget_chunks = (size_t)(pl.max_budget / 64 + 3);
        s.tflag = i64(budget_chunks);
        s.trank = i64(budget_chunks);
        s.bcnt = reinterpret_cast<uint32_t *>(take(4 * (size_t)pl.max_budget));
        s.bwithin = reinterpret_cast<uint32_t *>(take(4 * (size_t)pl.max_budget));
        s.bcur = reinterpret_cast<uint32_t *>(take(4 * (size_t)pl.max_budget));
        s.bucket = reinterpret_cast<uint32_t *>(take(4 * (size_t)pl.mc_cap));
        s.bucket2 = reinterpret_cast<uint32_t *>(take(4 * (size_t)pl.mc_cap)); // the list of long buckets
        s.bucket3 = reinterpret_cast<uint32_t *>(take(4 * (size_t)pl.mc_cap)); // the radix sort's second array
    }
    B.live.assign((size_t)T, nullptr), B.chosen.assign((size_t)T, nullptr), B.slots.assign((size_t)T, nullptr);
    B.carries.assign((size_t)T, nullptr);
    B.wlive.assign((size_t)T, nullptr);
    for (int t = 0; t < T; ++t) {
        const size_t chunks = (size_t)(pl.cap_budget[t] / 64 + 3);
        B.live[(size_t)t] = i64((size_t)pl.cap_budget[t]);
        B.chosen[(size_t)t] = i64((size_t)pl.max_k);
        B.slots[(size_t)t] = reinterpret_cast<uint32_t *>(i64((size_t)pl.max_k));
        B.carries[(size_t)t] = reinterpret_cast<double *>(i64(chunks));
        B.wlive[(size_t)t] = reinterpret_cast<double *>(i64((size_t)pl.cap_budget[t]));
    }
    B.n_live = i64(2 * (size_t)T);
    B.n_chosen = B.n_live ? B.n_live + T : nullptr;
    B.ed.assign((size_t)pl.edge_lanes, HgtEdgeScratch{});
    for (int l = 0; l < pl.edge_lanes; ++l) {
        B.ed[(size_t)l].cand_j = i64((size_t)pl.edge_cap);
        B.ed[(size_t)l].cand_ep = i64((size_t)pl.edge_cap);
        B.ed[(size_t)l].kept = i64((size_t)pl.max_nodes + 1);
        B.ed[(size_t)l].off = i64((size_t)pl.max_nodes + 2);
    }
    B.scan_temp = take(pl.scan_temp_bytes);
    return off + 256;
}

static int hgt_make_plan(const tg_hgt_problem *pb, HgtPlan &pl) {
    pl.T = pb->n_types;
    pl.R = pb->n_rels;
    pl.H = pb->n_hops;
    pl.cap_nodes.assign(pl.T, 0);
    pl.cap_budget.assign(pl.T, 0);
    pl.max_layer = 1;
    pl.max_k = 1;
    std::vector<int64_t> upd(pl.T, 0); // nodes of the type that an update_budget ever processes
    for (int t = 0; t < pl.T; ++t) {
        const int64_t n_in = pb->n_inputs[t] > 0 ? pb->n_inputs[t] : 0;
        pl.cap_nodes[t] = n_in;
        upd[t] = n_in;
        if (n_in > pl.max_layer) pl.max_layer = n_in;
        for (int l = 0; l < pl.H; ++l) {
            const int64_t k = pb->num_samples[(size_t)t * pl.H + l];
            if (k > 0) {
                pl.cap_nodes[t] += k;
                if (l < pl.H - 1) upd[t] += k;
                if (k > pl.max_layer) pl.max_layer = k;
                if (k > pl.max_k) pl.max_k = k;
            }
        }
    }
    for (int r = 0; r < pl.R; ++r) pl.cap_budget[pb->rel_src[r]] += upd[pb->rel_dst[r]] * HGT_MAX_NB;
    pl.max_budget = 1;
    pl.max_nodes = 1;
    pl.tl_cap.assign(pl.T, 0);
    pl.bm_cap.assign(pl.T, 0);
    for (int t = 0; t < pl.T; ++t) {
        if (pl.cap_budget[t] < 1) pl.cap_budget[t] = 1;
        // a budget can never hold more live entries than were inserted; samples per layer are capped by both
        if (pl.cap_budget[t] > pl.max_budget) pl.max_budget = pl.cap_budget[t];
        if (pl.cap_nodes[t] > pl.max_nodes) pl.max_nodes = pl.cap_nodes[t];
        pl.tl_cap[t] = pow2_at_least(2 * pl.cap_nodes[t] + 2);
        pl.bm_cap[t] = pow2_at_least(2 * pl.cap_budget[t] + 2);
    }
    pl.mc_cap = pl.max_layer * HGT_MAX_NB;
    TG_REQUIRE(pl.mc_cap < ((int64_t)1 << 31), "tg_hgt: layers of %lld samples are not supported", (long long)pl.max_layer);
    pl.tmp_cap = pow2_at_least(2 * pl.mc_cap + 2);
    pl.edge_cap = pl.max_nodes * HGT_MAX_NB;
    pl.edge_lanes = std::max(1, std::min(pl.R, HGT_EDGE_PAR));
    pl.scan_cap = std::max(std::max(pl.mc_cap, pl.max_budget) / 64 + 3, pl.max_nodes + 2);
    hipError_t e;
    size_t sc = 0;
    e = rocprim::exclusive_scan(nullptr, sc, (int64_t *)nullptr, (int64_t *)nullptr, (int64_t)0, (size_t)pl.scan_cap,
                                rocprim::plus<int64_t>(), (hipStream_t)0, false);
    if (e != hipSuccess) return tg::fail(TG_ERR_HIP, "rocprim::exclusive_scan size query failed: %s", hipGetErrorString(e));
    pl.scan_temp_bytes = sc;
    HgtBuffers B;
    pl.total_bytes = hgt_carve(pl, nullptr, B);
    return TG_OK;
}

} // namespace tg

extern "C" int tg_hgt_workspace_bytes(const tg_hgt_problem *pb, int64_t *bytes) {
    TG_REQUIRE(pb && bytes, "tg_hgt_workspace_bytes: null argument");
    TG_REQUIRE(pb->n_types >= 1 && pb->n_types <= 1024 && pb->n_rels >= 0 && pb->n_hops >= 0,
               "tg_hgt_workspace_bytes: bad sizes");
    tg::HgtPlan pl;
    int rc = tg::hgt_make_plan(pb, pl);
    if (rc != TG_OK) return rc;
    *bytes = (int64_t)pl.total_bytes;
    return TG_OK;
}

extern "C" int tg_hgt_sample(const tg_hgt_problem *pb, const tg_rng *rng, const tg_hgt_out *out, void *workspace,
                             int64_t workspace_bytes, void *stream_) {
    using namespace tg;
    TG_REQUIRE(pb && rng && out && workspace, "tg_hgt_sample: null argument");
    TG_REQUIRE(pb->n_types >= 1 && pb->n_types <= 1024 && pb->n_rels >= 0 && pb->n_hops >= 0, "tg_hgt_sample: bad sizes");
    hipStream_t stream = (hipStream_t)stream_;
    HgtPlan pl;
    int rc = hgt_make_plan(pb, pl);
    if (rc != TG_OK) return rc;
    TG_REQUIRE((size_t)workspace_bytes >= pl.total_bytes, "tg_hgt_sample: workspace too small (%lld < %lld)",
               (long long)workspace_bytes, (long long)pl.total_bytes);
    const int T = pl.T, R = pl.R, H = pl.H;
    HgtBuffers B;
    (void)hgt_carve(pl, reinterpret_cast<unsigned char *>(workspace), B);
    std::vector<HgtType> &ty = B.ty;
    HgtTypeCtr *ctr = B.ctr;
    int *panic = B.panic;
    for (int t = 0; t < T; ++t) {
        ty[(size_t)t].nodes = out->samples[t];
        ty[(size_t)t].ts = out->sample_ts[t];
    }

    // ---- empty maps, zero counters: the types side by side
    for (int t0 = 0; t0 < T; t0 += HGT_MULTI_TYPES) {
        const int n = std::min(HGT_MULTI_TYPES, T - t0);
        HgtInitArgs ia;
        std::memset(&ia, 0, sizeof(ia));
        int64_t widest = 1;
        for (int i = 0; i < n; ++i) {
            ia.ty[i] = ty[(size_t)(t0 + i)];
            ia.tl_cap[i] = pl.tl_cap[t0 + i];
            ia.bm_cap[i] = pl.bm_cap[t0 + i];
            widest = std::max(widest, std::max(ia.tl_cap[i], ia.bm_cap[i]));
        }
        hipLaunchKernelGGL(hgt_init_types_kernel, dim3(grid_1d(widest), (unsigned)n), dim3(256), 0, stream, ia, B.scal, 8);
    }
    TG_LAUNCH_CHECK();

    // exclusive scan of flag[0..n) into rank[0..n), total into *total, for the sizes one workgroup is too slow for
    auto library_scan = [&](const int64_t *flag, int64_t *rank, int64_t n, int64_t *total) -> int {
        TG_REQUIRE(n <= pl.scan_cap, "tg_hgt_sample: scan of %lld elements exceeds the plan", (long long)n);
        size_t stb = pl.scan_temp_bytes;
        TG_HIP(rocprim::exclusive_scan(B.scan_temp, stb, flag, rank, (int64_t)0, (size_t)n, rocprim::plus<int64_t>(), stream,
                                       false));
        hipLaunchKernelGGL(scan_total_kernel, dim3(1), dim3(64), 0, stream, flag, rank, n, total);
        return TG_OK;
    };
    constexpr int64_t ONE_WORKGROUP_SCAN = 16384; // elements up to which one workgroup beats the library's launches

    // ---- update_budget (:27-102) for the layers of the node types `which`, in order (:47 relations in canonical order).
    // A step = (node type nt, relation r into nt); steps are dealt to ROUNDS: a step goes to the round after the last
    // step that fed the same source type's budget (their order must hold), else to the first round with a free place;
    // the steps of a round run as one launch per phase (blockIdx.y = step).
    auto update_budgets = [&](const std::vector<int> &which) -> int {
        std::vector<std::vector<std::pair<int, int>>> rounds;
        std::vector<int> last_round((size_t)T, -1);
        for (int nt : which)
            for (int r = 0; r < R; ++r) {
                if (pb->rel_dst[r] != nt) continue;
                size_t rd = (size_t)(last_round[(size_t)pb->rel_src[r]] + 1);
                while (rd < rounds.size() && rounds[rd].size() >= (size_t)HGT_MAX_PAR) ++rd;
                if (rd >= rounds.size()) rounds.resize(rd + 1);
                rounds[rd].push_back({nt, r});
                last_round[(size_t)pb->rel_src[r]] = (int)rd;
            }
        const int64_t n_chunks = (pl.mc_cap + 63) / 64;
        for (const auto &round : rounds) {
            const unsigned Y = (unsigned)round.size();
            if (Y == 0) continue;
            HgtSteps S;
            std::memset(&S, 0, sizeof(S));
            int64_t widest_budget = 1;
            for (unsigned y = 0; y < Y; ++y) {
                const int nt = round[y].first, r = round[y].second, st = pb->rel_src[r];
                const tg_graph &g = pb->graphs[r];
                const HgtStepScratch &sc = B.sc[y];
                HgtStep &a = S.s[y];
                a.dst = ty[(size_t)nt];
                a.src = ty[(size_t)st];
                a.src_ctr = ctr + st;
                a.ptrs = g.ptrs;
                a.indices = g.indices;
                a.edge_ts = g.timestamps;
                a.pad = pl.cap_budget[st];
                widest_budget = std::max(widest_budget, a.pad);
                a.ccnt = sc.ccnt, a.coff = sc.coff, a.ckey = sc.ckey, a.cts = sc.cts, a.cslot = sc.cslot;
                a.tmp_keys = sc.tmp_keys, a.tmp_vals = sc.tmp_vals, a.flag = sc.flag, a.rank = sc.rank;
                a.scal = sc.scal, a.cinv = sc.cinv, a.cmask = sc.cmask, a.tflag = sc.tflag, a.trank = sc.trank;
                a.bcnt = sc.bcnt, a.bwithin = sc.bwithin, a.bcur = sc.bcur, a.bucket = sc.bucket, a.bucket2 = sc.bucket2;
                a.bucket3 = sc.bucket3;
            }
            S.pbits = 1;
            while (((int64_t)1 << S.pbits) < pl.mc_cap) ++S.pbits;
            auto g2 = [&](int64_t n) { return dim3(grid_1d(n), Y); };
            hipLaunchKernelGGL(hgt_count_scan_steps_kernel, dim3(1, Y), dim3(SCAN1_THREADS), 0, stream, S, pl.max_layer);
            hipLaunchKernelGGL(hgt_gen_steps_kernel, g2(pl.mc_cap), dim3(256), 0, stream, S, pb->has_timerange, pb->tr_lo,
                               pb->tr_hi, pl.max_layer, pl.tmp_cap);
            hipLaunchKernelGGL(hgt_slots_steps_kernel, g2(pl.mc_cap), dim3(256), 0, stream, S, pl.tmp_cap - 1);
            hipLaunchKernelGGL(hgt_first_flags_steps_kernel, g2(pl.mc_cap), dim3(256), 0, stream, S, pl.tmp_cap - 1, pl.mc_cap);
            if (n_chunks <= ONE_WORKGROUP_SCAN) { // over the chunks' counts: one workgroup per step
                hipLaunchKernelGGL(hgt_scan1_steps_kernel, dim3(1, Y), dim3(SCAN1_THREADS), 0, stream, S, n_chunks);
            } else {
                for (unsigned y = 0; y < Y; ++y)
                    if (int rcs = library_scan(S.s[y].flag, S.s[y].rank, n_chunks, S.s[y].scal + 1)) return rcs;
            }
            hipLaunchKernelGGL(hgt_new_slots_steps_kernel, g2(pl.mc_cap), dim3(256), 0, stream, S, pl.tmp_cap - 1);
            hipLaunchKernelGGL(hgt_bucket_offsets_steps_kernel, g2(widest_budget), dim3(256), 0, stream, S);
            if ((widest_budget + 63) / 64 <= ONE_WORKGROUP_SCAN) { // over the budget chunks' contribution counts
                hipLaunchKernelGGL(hgt_bucket_scan_steps_kernel, dim3(1, Y), dim3(SCAN1_THREADS), 0, stream, S);
            } else {
                for (unsigned y = 0; y < Y; ++y)
                    if (int rcs = library_scan(S.s[y].tflag, S.s[y].trank, (S.s[y].pad + 63) / 64, S.s[y].scal + 2)) return rcs;
            }
            hipLaunchKernelGGL(hgt_bucket_scatter_steps_kernel, g2(pl.mc_cap), dim3(256), 0, stream, S);
            hipLaunchKernelGGL(hgt_accumulate_steps_kernel, dim3(grid_1d(widest_budget) + HGT_ACC_LONG_BLOCKS, Y),
                               dim3(HGT_ACC_THREADS), 0, stream, S);
            TG_LAUNCH_CHECK();
        }
        return TG_OK;
    };

    // ---- :167-196 inputs, then the first budgets
    for (int t = 0; t < T; ++t) {
        const int64_t n_in = pb->n_inputs[t];
        if (n_in > 0) {
            TG_REQUIRE(pb->inputs[t], "tg_hgt_sample: node type %d has n_inputs > 0 but no pointer", t);
            hipLaunchKernelGGL(hgt_init_inputs_kernel, dim3(grid_1d(n_in)), dim3(256), 0, stream, ty[(size_t)t], pb->inputs[t],
                               pb->input_ts ? pb->input_ts[t] : (const int64_t *)nullptr, n_in);
        }
    }
    TG_LAUNCH_CHECK();
    {
        std::vector<int> which;
        for (int t = 0; t < T; ++t)
            if (pb->n_inputs[t] >= 0) which.push_back(t);
        rc = update_budgets(which);
        if (rc != TG_OK) return rc;
    }
    // ---- :198-242 layers
    for (int layer = 0; layer < H; ++layer) {
        for (int t0 = 0; t0 < T; t0 += HGT_MULTI_TYPES) { // :201 every type that owns a budget samples from it
            const int n = std::min(HGT_MULTI_TYPES, T - t0);
            HgtSampleArgs sa;
            std::memset(&sa, 0, sizeof(sa));
            size_t lds = 8;
            for (int i = 0; i < n; ++i) {
                const size_t t = (size_t)(t0 + i);
                sa.ty[i] = ty[t];
                sa.n_live[i] = B.n_live + t, sa.live[i] = B.live[t], sa.chosen[i] = B.chosen[t], sa.n_chosen[i] = B.n_chosen + t;
                sa.k[i] = pb->num_samples[t * (size_t)H + (size_t)layer];
                sa.slots[i] = B.slots[t], sa.carries[i] = B.carries[t], sa.wlive[i] = B.wlive[t];
                if (sa.k[i] > 0 && sa.k[i] <= HGT_LDS_SLOTS) lds = std::max(lds, (size_t)sa.k[i] * 4);
            }
            hipLaunchKernelGGL(hgt_sample_layer_kernel, dim3((unsigned)n), dim3(SCAN1_THREADS), lds, stream, sa, rng->seed,
                               rng->call_id, (int64_t)layer, t0, T, panic);
        }
        TG_LAUNCH_CHECK();
        if (layer < H - 1) { // :227 (types without samples return at :38-40)
            std::vector<int> which;
            for (int t = 0; t < T; ++t) which.push_back(t);
            rc = update_budgets(which);
            if (rc != TG_OK) return rc;
        }
    }
    // ---- :244-268 edges among the sampled nodes, the relations side by side
    for (int r0 = 0; r0 < R; r0 += pl.edge_lanes) {
        const int n = std::min(pl.edge_lanes, R - r0);
        HgtEdgeRels E;
        std::memset(&E, 0, sizeof(E));
        int64_t widest = 1;
        for (int i = 0; i < n; ++i) {
            const int r = r0 + i, st = pb->rel_src[r], dt = pb->rel_dst[r];
            HgtEdgeRel &a = E.r[i];
            a.dst = ty[(size_t)dt], a.src = ty[(size_t)st];
            a.ptrs = pb->graphs[r].ptrs, a.indices = pb->graphs[r].indices;
            a.cap_n = pl.cap_nodes[dt] > 0 ? pl.cap_nodes[dt] : 1;
            a.tag = TAG_HGT | ((uint32_t)(r + 1) << 8);
            a.cand_j = B.ed[(size_t)i].cand_j, a.cand_ep = B.ed[(size_t)i].cand_ep;
            a.kept = B.ed[(size_t)i].kept, a.off = B.ed[(size_t)i].off;
            a.rows = out->rows[r], a.cols = out->cols[r], a.eidx = out->edge_index[r], a.n_edges = out->n_edges + r;
            widest = std::max(widest, a.cap_n);
        }
        const dim3 grid(grid_1d(widest * 64), (unsigned)n);
        hipLaunchKernelGGL(hgt_edge_candidates_rels_kernel, grid, dim3(256), 0, stream, E, rng->seed, rng->call_id);
        if (widest <= ONE_WORKGROUP_SCAN) { // over the nodes' kept-edge counts
            hipLaunchKernelGGL(hgt_edge_scan_rels_kernel, dim3(1, (unsigned)n), dim3(SCAN1_THREADS), 0, stream, E);
        } else {
            for (int i = 0; i < n; ++i)
                if (int rcs = library_scan(E.r[i].kept, E.r[i].off, E.r[i].cap_n, E.r[i].n_edges)) return rcs;
        }
        hipLaunchKernelGGL(hgt_edge_emit_rels_kernel, grid, dim3(256), 0, stream, E);
        TG_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(hgt_finish_kernel, dim3(1), dim3(256), 0, stream, ctr, T, out->n_samples, panic, out->panic);
    TG_LAUNCH_CHECK();
#ifdef TG_HGT_STAMPS
    {
        long long st[HGT_MULTI_TYPES_STAMPS][16];
        TG_HIP(hipStreamSynchronize(stream));
        TG_HIP(hipMemcpyFromSymbol(st, HIP_SYMBOL(hgt_stamps), sizeof(st)));
        static const char *names[8] = {"flags", "scan", "list", "A", "B", "C", "D", "append"};
        for (int t = 0; t < T && t < HGT_MULTI_TYPES_STAMPS; ++t) {
            fprintf(stderr, "hgt stamps type %d:", t);
            for (int i = 0; i < 8; ++i) fprintf(stderr, " %s %.1f us", names[i], (double)(st[t][i + 1] - st[t][i]) * 0.01);
            fprintf(stderr, "\n");
        }
    }
#endif
    return TG_OK;
}
