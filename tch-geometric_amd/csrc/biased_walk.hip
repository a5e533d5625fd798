// biased_tempo_random_walk (reference: src/algo/random_walk.rs:160-288; binding python.rs:645-687).
//
// One WAVEFRONT per walker, persistent over walkers.  A step streams the current vertex's CSR row (64 edges per
// pass, coalesced), keeps the time-respecting candidates (random_walk.rs:228-251), weighs them by BiasType::apply
// (:165-182) and runs the one-slot weighted reservoir (utils/sampling.rs:28-55) over them in candidate order:
//
//   uniform      weights are 1, the running sum is min(c + 1, 2^24) in f32 -> one pass, no serial work;
//   exponential  softmax of integer time differences: pass 1 finds the maximum, pass 2 histograms the integer
//                exponents (<= 103, beyond that exp() is 0 in f32) in LDS, the denominator is summed from the
//                histogram in a fixed order, pass 3 selects;
//   linear       the weights are the descending-time argsort PERMUTATION (random_walk.rs:171, not the ranks): pass 1
//                writes (time, position) keys, the wavefront sorts them -- bitonically in LDS up to 1024 candidates;
//                beyond, a stable LSD radix sort (8 bits per pass, only over the key bytes that differ inside the
//                row) between two global slabs, O(n) per pass where the bitonic network was O(n log^2 n) -- pass 2
//                selects.
//
// The reservoir's running f32 sum (sampling.rs:48) is philox-mode's BLOCKED sum for the linear and exponential biases:
// Kogge-Stone inside every chunk of 64 raw row positions, the chunks' totals carried left to right (the CPU checker
// restates exactly that; its ref-mode keeps the reference's literal sum).  Draws are addressed per candidate, so
// everything else is order-free.  A walker whose candidate set is empty restarts from its start vertex, up to
// retry_count times (:217, :273-276); timestamps written by abandoned attempts stay, as in the reference (:223-225
// resets the vertices only).
#include "row_stream.h"
#include "tg_device.h"
#include "tg_exp_table.h"
#include "tg_host.h"

namespace tg {

constexpr uint32_t TAG_RW_BIASED = 11u;
constexpr int BW_WAVES = 4;
constexpr int BW_SORT_LDS = 1024; // keys per wavefront sorted in LDS
constexpr int BW_HIST = 256;      // >= TG_EXP_NEG_BITS_N; also the radix sort's digit counters
constexpr int BW_P = 4;           // 64-edge chunks per load round of the row streamer (two rounds in flight); 1 is slower here: 116 / 483 / 452 ms

struct BwWaveLds {
    uint32_t hist[BW_HIST];
    uint64_t keys[BW_SORT_LDS];
};

__device__ __forceinline__ void wave_mem_handoff(bool global) {
    if (global) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    } else {
        wave_lds_handoff();
    }
}

// ascending bitonic sort of n_pad (power of two) u64 keys by one wavefront
__device__ void wave_bitonic_sort(uint64_t *buf, uint32_t n_pad, bool global) {
    const uint32_t lane = (uint32_t)lane_id();
    for (uint32_t k = 2; k <= n_pad; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t i = lane; i < n_pad; i += 64) {
                const uint32_t x = i ^ j;
                if (x > i) {
                    const uint64_t a = buf[i], b = buf[x];
                    const bool asc = (i & k) == 0;
                    if ((a > b) == asc) {
                        buf[i] = b;
                        buf[x] = a;
                    }
                }
            }
            wave_mem_handoff(global);
        }
    }
}

// Stable LSD radix sort of n u64 keys by their HIGH words (ascending), one wavefront, between two global buffers; only
// the bytes of the high word that differ somewhere in the row (`vary`) get a pass.  Entries start in candidate order, so
// equal times stay in candidate order: the argsort's tie rule.  Returns the buffer that holds the result.
__device__ uint64_t *wave_radix_sort(uint64_t *src, uint64_t *dst, uint32_t n, uint32_t vary, uint32_t *hist) {
    const int lane = lane_id();
    const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    for (int byte = 0; byte < 4; ++byte) {
        if (((vary >> (8 * byte)) & 0xffu) == 0u) continue;
        const int shift = 32 + 8 * byte;
        for (int i = lane; i < 256; i += 64) hist[i] = 0u;
        wave_lds_handoff();
        for (uint32_t c = lane; c < n; c += 64) atomicAdd(&hist[(uint32_t)(src[c] >> shift) & 0xffu], 1u);
        wave_lds_handoff();
        { // exclusive prefix of the 256 counters: four per lane
            const uint32_t h0 = hist[4 * lane], h1 = hist[4 * lane + 1], h2 = hist[4 * lane + 2], h3 = hist[4 * lane + 3];
            const uint32_t s = h0 + h1 + h2 + h3;
            const uint32_t base = wave_inclusive_scan(s) - s;
            wave_lds_handoff();
            hist[4 * lane] = base;
            hist[4 * lane + 1] = base + h0;
            hist[4 * lane + 2] = base + h0 + h1;
            hist[4 * lane + 3] = base + h0 + h1 + h2;
        }
        wave_lds_handoff();
        for (uint32_t c0 = 0; c0 < n; c0 += 64) { // stable scatter, 64 keys at a time in order
            const uint32_t c = c0 + (uint32_t)lane;
            const bool valid = c < n;
            const uint64_t key = valid ? src[c] : 0ull;
            const uint32_t d = (uint32_t)(key >> shift) & 0xffu;
            uint64_t same = __ballot(valid); // lanes holding this lane's digit
#pragma unroll
            for (int bit = 0; bit < 8; ++bit) {
                const uint64_t bm = __ballot(valid && ((d >> bit) & 1u));
                same &= ((d >> bit) & 1u) ? bm : ~bm;
            }
            const uint32_t rank = (uint32_t)__popcll(same & lt_mask);
            uint32_t pos = 0;
            if (valid) pos = hist[d] + rank;
            wave_lds_handoff();
            if (valid) {
                dst[pos] = key;
                if (rank == 0) hist[d] += (uint32_t)__popcll(same);
            }
            wave_lds_handoff();
        }
        wave_mem_handoff(true);
        uint64_t *t = src;
        src = dst;
        dst = t;
    }
    return src;
}

// philox-mode's BLOCKED running weight sum in f32 (tg_device.h wave_blocked_prefix_f64 is the f64 twin; the CPU checker
// restates it as orc_blocked_prefix_f32): Kogge-Stone inside the chunk of 64 raw row positions (non-candidates add 0),
// then the carry of the earlier chunks.  Six dependent adds per chunk instead of 64: the literal left-to-right sum made a
// step's time the length of the row's f32 chain (exponential bias: 521 ms per 1 M walkers x 20 steps).
__device__ __forceinline__ float wave_blocked_prefix_f32(float v, float carry, float *total) {
    const int lane = lane_id();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const float u = __shfl_up(v, d, 64);
        if (lane >= d) v = v + u;
    }
    *total = carry + __shfl(v, 63, 64);
    return carry + v;
}

struct Cand {
    bool ok;
    int32_t time32;
};

// candidate test of one streamed edge (random_walk.rs:239-256)
__device__ __forceinline__ Cand classify(bool valid, int64_t ts, int64_t cur_ts) {
    Cand c;
    c.ok = valid && ((ts == -1 || cur_ts == -1) || (cur_ts <= ts));
    c.time32 = (int32_t)(uint32_t)(uint64_t)(ts == -1 ? cur_ts : ts);
    return c;
}

__device__ __forceinline__ float exp_delta(int32_t time32, uint32_t t32, int forward) {
    return (float)(int32_t)(forward ? t32 - (uint32_t)time32 : (uint32_t)time32 - t32);
}

__global__ __launch_bounds__(64 * BW_WAVES) void biased_walk_kernel(
    const int64_t *__restrict__ ptrs, const int64_t *__restrict__ indices, const int64_t *__restrict__ node_ts,
    const int64_t *__restrict__ edge_ts, const int64_t *__restrict__ start, const int64_t *__restrict__ start_ts, int64_t n,
    int64_t L, int32_t bias, int32_t forward, int64_t R, uint64_t seed, uint64_t call_id, int64_t *walks,
    int64_t *walks_ts, uint64_t *scratch, int64_t slab, int32_t *status) {
    __shared__ BwWaveLds lds_all[BW_WAVES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    BwWaveLds &lds = lds_all[wave];
    const int64_t gwave = (int64_t)blockIdx.x * BW_WAVES + wave, n_gwaves = (int64_t)gridDim.x * BW_WAVES;
    const CallKey ck = call_key(seed, call_id, TAG_RW_BIASED);
    const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    uint64_t *slab_keys = scratch ? scratch + (size_t)gwave * 2 * (size_t)slab : nullptr; // two buffers of `slab` keys

    for (int64_t i = gwave; i < n; i += n_gwaves) {
        for (int64_t c = lane; c < L; c += 64) { // :199-208 (position c is always written by lane c & 63)
            walks[i * L + c] = -1;
            walks_ts[i * L + c] = -1;
        }
        bool fatal = false;
        for (int64_t attempt = 0; attempt < R && !fatal; ++attempt) { // :217
            int64_t cur = start[i], cur_ts = start_ts[i];
            for (int64_t c = lane; c < L; c += 64) { // :221-225
                walks[i * L + c] = (c == 0) ? cur : -1;
                if (c == 0) walks_ts[i * L] = cur_ts;
            }
            bool restart = false;
            for (int64_t l = 0; l < L - 1; ++l) {
                const uint64_t step_id = ((uint64_t)i * (uint64_t)R + (uint64_t)attempt) * (uint64_t)L + (uint64_t)l;
                const int64_t b = ptrs[cur], e = ptrs[cur + 1];
                const int32_t bias_now = (cur_ts == -1) ? 0 : bias; // :258-262
                const uint32_t t32 = (uint32_t)(uint64_t)cur_ts;
                uint32_t n_c = 0;
                int64_t best_rank = -1, best_v = -1, best_t = -1; // this lane's last reservoir hit
                int64_t first_v = -1, first_t = -1;               // candidate 0 (held by one lane)
                bool has_first = false, panic = false;

                if (bias_now == 0) {
                    // ---- uniform: one pass; w_sum after candidate c is min(c + 1, 2^24) exactly in f32
                    stream_row_ts<BW_P>(indices, edge_ts, node_ts, b, e, lane, [&](int64_t v, bool valid, int64_t ts) { // v: edge position
                        const Cand cd = classify(valid, ts, cur_ts);
                        const uint64_t mask = __ballot(cd.ok);
                        if (cd.ok) {
                            const uint32_t c = n_c + (uint32_t)__popcll(mask & lt_mask);
                            if (c == 0) {
                                first_v = v;
                                first_t = ts;
                                has_first = true;
                            } else {
                                const float w_sum = (float)(c < 16777216u ? c + 1u : 16777216u);
                                const float j = u32_to_f32_01(draw(ck, step_id, c, D1_WEIGHTED).w[0]) * w_sum + 0.0f;
                                if (j < 1.0f) {
                                    best_rank = c;
                                    best_v = v;
                                    best_t = ts;
                                }
                            }
                        }
                        n_c += (uint32_t)__popcll(mask);
                    });
                } else {
                    // ---- pass 1: count, first candidate, maximum exponent / sort keys
                    const int64_t deg = e - b;
                    const bool keys_global = (bias_now == 1) && deg > BW_SORT_LDS;
                    uint64_t *keys = keys_global ? slab_keys : lds.keys;
                    if (keys_global && (slab_keys == nullptr || deg > slab)) {
                        if (lane == 0) atomicOr(status, 1);
                        fatal = true;
                        break;
                    }
                    float mx = -__builtin_inff();
                    uint32_t key_or = 0u, key_and = ~0u; // which bits of the time keys differ inside the row
                    stream_row_ts<BW_P>(indices, edge_ts, node_ts, b, e, lane, [&](int64_t v, bool valid, int64_t ts) { // v: edge position
                        const Cand cd = classify(valid, ts, cur_ts);
                        const uint64_t mask = __ballot(cd.ok);
                        if (cd.ok) {
                            const uint32_t c = n_c + (uint32_t)__popcll(mask & lt_mask);
                            if (c == 0) {
                                first_v = v;
                                first_t = ts;
                                has_first = true;
                            }
                            if (bias_now == 1) { // descending time, ties by ascending position
                                const uint32_t hi = ~((uint32_t)cd.time32 ^ 0x80000000u);
                                keys[c] = ((uint64_t)hi << 32) | (uint64_t)c;
                                key_or |= hi;
                                key_and &= hi;
                            } else
                                mx = fmaxf(mx, exp_delta(cd.time32, t32, forward));
                        }
                        n_c += (uint32_t)__popcll(mask);
                    });
                    if (n_c >= 2) {
                        float den;
                        if (bias_now == 1) {
                            if (keys_global) {
#pragma unroll
                                for (int off = 32; off > 0; off >>= 1) {
                                    key_or |= __shfl_xor(key_or, off, 64);
                                    key_and &= __shfl_xor(key_and, off, 64);
                                }
                                wave_mem_handoff(true);
                                keys = wave_radix_sort(keys, keys + slab, n_c, key_or ^ key_and, lds.hist);
                            } else {
                                uint32_t n_pad = 2;
                                while (n_pad < n_c) n_pad <<= 1;
                                for (uint32_t c = n_c + lane; c < n_pad; c += 64) keys[c] = ~0ull;
                                wave_mem_handoff(false);
                                wave_bitonic_sort(keys, n_pad, false);
                            }
                            den = (float)((int64_t)n_c * ((int64_t)n_c - 1) / 2);
                        } else {
#pragma unroll
                            for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
                            for (int m = lane; m < BW_HIST; m += 64) lds.hist[m] = 0u;
                            wave_lds_handoff();
                            // ---- pass 2 (exponential): histogram of the integer exponents
                            stream_row_ts<BW_P>(indices, edge_ts, node_ts, b, e, lane, [&](int64_t, bool valid, int64_t ts) {
                                const Cand cd = classify(valid, ts, cur_ts);
                                if (cd.ok) {
                                    const float m = mx - exp_delta(cd.time32, t32, forward);
                                    if (m < (float)TG_EXP_NEG_BITS_N) atomicAdd(&lds.hist[(int)m], 1u);
                                }
                            });
                            wave_lds_handoff();
                            double acc = 0.0;
                            if (lane == 0) {
                                for (int m = 0; m < TG_EXP_NEG_BITS_N; ++m)
                                    acc = acc + (double)lds.hist[m] * (double)__uint_as_float(tg_exp_neg_bits[m]);
                            }
                            den = __shfl((float)acc, 0, 64);
                            wave_lds_handoff();
                        }
                        // ---- selection pass: weighted reservoir with one slot (sampling.rs:47-53)
                        float carry = 0.0f;
                        n_c = 0;
                        stream_row_ts<BW_P>(indices, edge_ts, node_ts, b, e, lane, [&](int64_t v, bool valid, int64_t ts) { // v: edge position
                            const Cand cd = classify(valid, ts, cur_ts);
                            const uint64_t mask = __ballot(cd.ok);
                            const uint32_t c = n_c + (uint32_t)__popcll(mask & lt_mask);
                            float w = 0.0f;
                            if (cd.ok) {
                                if (bias_now == 1) {
                                    w = (float)(uint32_t)(keys[c] & 0xffffffffull) / den;
                                } else {
                                    const float m = mx - exp_delta(cd.time32, t32, forward);
                                    const float ex = (m < (float)TG_EXP_NEG_BITS_N) ? __uint_as_float(tg_exp_neg_bits[(int)m]) : 0.0f;
                                    w = ex / den;
                                }
                            }
                            float total;
                            const float w_sum = wave_blocked_prefix_f32(w, carry, &total);
                            carry = total;
                            if (cd.ok && c >= 1) {
                                if (!(0.0f < w_sum)) {
                                    panic = true;
                                } else {
                                    const float j = u32_to_f32_01(draw(ck, step_id, c, D1_WEIGHTED).w[0]) * w_sum + 0.0f;
                                    if (j < w) {
                                        best_rank = c;
                                        best_v = v;
                                        best_t = ts;
                                    }
                                }
                            }
                            n_c += (uint32_t)__popcll(mask);
                        });
                    }
                }
                if (__ballot(panic) != 0ull) {
                    if (lane == 0) atomicOr(status, 2);
                    fatal = true;
                    break;
                }
                if (n_c == 0) { // :273-276
                    restart = true;
                    break;
                }
                int64_t mxr = best_rank;
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) mxr = max(mxr, __shfl_xor(mxr, off, 64));
                const uint64_t owner = (mxr >= 1) ? __ballot(best_rank == mxr) : __ballot(has_first);
                const int src = __ffsll((long long)owner) - 1;
                const int64_t next = indices[__shfl((mxr >= 1) ? best_v : first_v, src, 64)]; // the one id the step needs
                const int64_t next_t = __shfl((mxr >= 1) ? best_t : first_t, src, 64);
                cur = next; // :278-284
                if (next_t != -1) cur_ts = next_t;
                if (lane == (int)((l + 1) & 63)) {
                    walks[i * L + l + 1] = cur;
                    walks_ts[i * L + l + 1] = next_t;
                }
            }
            if (!restart) break; // :286
        }
    }
}

static inline int64_t pow2_ceil(int64_t x) {
    int64_t p = 1;
    while (p < x) p <<= 1;
    return p;
}
static inline unsigned bw_grid(int64_t n, bool with_slab) {
    int64_t g = (n + BW_WAVES - 1) / BW_WAVES;
    const int64_t cap = with_slab ? 1024 : 2048;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (unsigned)g;
}
static inline bool bw_needs_slab(int32_t bias, int64_t max_degree) { return bias == TG_BIAS_LINEAR && max_degree > BW_SORT_LDS; }

} // namespace tg

extern "C" int tg_biased_walk_workspace_bytes(int64_t n, int64_t max_degree, int32_t bias, int64_t *bytes) {
    TG_REQUIRE(bytes && n >= 0 && max_degree >= 0, "tg_biased_walk_workspace_bytes: bad arguments");
    TG_REQUIRE(max_degree < (1ll << 31), "tg_biased_walk_workspace_bytes: max_degree %lld does not fit the sort keys",
               (long long)max_degree);
    *bytes = 0;
    if (tg::bw_needs_slab(bias, max_degree))
        *bytes = (int64_t)tg::bw_grid(n, true) * tg::BW_WAVES * 2 * max_degree * (int64_t)sizeof(uint64_t);
    return TG_OK;
}

extern "C" int tg_biased_tempo_random_walk(const tg_graph *csr, const int64_t *node_ts, const int64_t *edge_ts,
                                           const int64_t *start, const int64_t *start_ts, int64_t n, int64_t walk_length,
                                           int32_t bias, int32_t forward, int64_t retry_count, int64_t max_degree,
                                           const tg_rng *rng, int64_t *walks, int64_t *walks_ts, int32_t *status,
                                           void *workspace, int64_t workspace_bytes, void *stream) {
    TG_REQUIRE(csr && csr->ptrs && (csr->indices || csr->n_edges == 0), "tg_biased_tempo_random_walk: null graph");
    TG_REQUIRE(rng && n >= 0 && walk_length >= 0 && retry_count >= 0 && max_degree >= 0,
               "tg_biased_tempo_random_walk: bad arguments");
    TG_REQUIRE(bias == TG_BIAS_UNIFORM || bias == TG_BIAS_LINEAR || bias == TG_BIAS_EXPONENTIAL,
               "tg_biased_tempo_random_walk: unknown bias %d", bias);
    if (n == 0 || walk_length == 0) return TG_OK;
    TG_REQUIRE(node_ts && (edge_ts || csr->n_edges == 0) && start && start_ts && walks && walks_ts && status,
               "tg_biased_tempo_random_walk: null buffers");
    int64_t need = 0;
    const int rc = tg_biased_walk_workspace_bytes(n, max_degree, bias, &need);
    if (rc != TG_OK) return rc;
    TG_REQUIRE(need == 0 || (workspace && workspace_bytes >= need),
               "tg_biased_tempo_random_walk: workspace of %lld bytes needed, %lld given", (long long)need,
               (long long)workspace_bytes);
    const bool slab = need > 0;
    hipLaunchKernelGGL(tg::biased_walk_kernel, dim3(tg::bw_grid(n, slab)), dim3(64 * tg::BW_WAVES), 0, (hipStream_t)stream,
                       csr->ptrs, csr->indices, node_ts, edge_ts, start, start_ts, n, walk_length, bias, forward,
                       retry_count, rng->seed, rng->call_id, walks, walks_ts, slab ? (uint64_t *)workspace : nullptr,
                       slab ? max_degree : 0, status);
    TG_LAUNCH_CHECK();
    return TG_OK;
}
