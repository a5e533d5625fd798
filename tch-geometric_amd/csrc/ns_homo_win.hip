// neighbor_sampling_homogenous, window-ordered form of the launch-wide gather (gfx950).
//
// Same outputs as ns_homo.hip (the fused per-batch kernel), for launches of MANY seed batches.  The fused kernel
// issues `indices[edge_ptr]` in batch order: on RMAT-24 a 4 096-batch launch makes 120.8 M 4-byte gathers into 8.3 M
// distinct 128-byte lines (~14.6 touches per line) in an order no cache holds, so ~12 of its 13 GB of reads are
// re-fetches (a random 4-byte gather costs a whole line request whether HBM or the Infinity Cache serves it: 55 G
// requests/s; only an L2 hit is cheaper, 258 G/s -- tools/probe_gather_sizes.py).  Here the gathers of a hop are
// brought into ADDRESS order instead, hop by hop over the whole device:
//
//   K1  win_emit_kernel    (one workgroup per batch): per frontier vertex the column bounds and sample count, LDS scan
//                          -> every vertex's output offset, draws; writes rows / cols / edge_index -- which need no
//                          gather -- as coalesced streams and one ITEM per frontier vertex (column start, degree,
//                          batch, slot, output offset: 16 bytes, 24 for graphs / launches beyond 32-bit offsets).
//   P1-3 win_hist / win_colscan / win_basescan / win_scatter: one counting-sort pass of the hop's items by WINDOW of
//                          the column start (window = 2^shift edge pointers = a few hundred KB of `indices`), laid
//                          out XCD-major (windows x, x+8, x+16 ... form queue x); no global atomics.
//   K4  win_gather_kernel  (persistent; the blocks of one XCD sweep that XCD's queue in order, a block reserving the
//                          next slice with one atomic): re-derives the item's positions (counter-addressed Philox:
//                          the same draws as K1 and the fused kernel), gathers `indices[e0 + pos]` -- now L2 hits,
//                          every line of a window is fetched by ONE XCD about once -- and writes
//                          samples[n_seeds + e ...] in the reference's slot order.  (Scattered 8-byte-granular runs
//                          are the expensive kind of write, so only the one array that needs the gather goes here.)
//
// neighbor_sampling.rs:188-223 is unchanged in meaning: the output position of every edge is fixed by the per-batch
// prefix sums of K1, so the ORDER in which items are gathered cannot change a single output word (tests compare this
// path with the fused kernel and the oracle bit for bit).
#include <stdlib.h>

#include <algorithm>
#include <mutex>

#include "ns_tickets.h"
#include "stage_bits.h"
#include "tg_device.h"
#include "tg_host.h"

namespace tg {

constexpr int WIN_PART_BLOCKS = 512;  // blocks of the partition kernels = rows of the histogram matrix
constexpr int WIN_PART_THREADS = 512;
constexpr int WIN_TILE = 4096;        // items per partition tile
constexpr int WIN_MAX_BUCKETS = 8192; // windows per hop (LDS: 32 KB of counters)
constexpr int WIN_MAX_ROWS = 1024;    // rows of the histogram matrix in the folded form (= persistent emit workgroups)
constexpr int WIN_MAX_PARTS = 16;     // staged form: parts of a launch in flight on two streams

struct WinState { // per batch, lives in the workspace
    int64_t begin, end, ne, fbase;
};

struct WinQueues { // items of XCD-group x in the sorted array: [head, end); one 256-byte line per group
    struct alignas(256) Q {
        unsigned long long head, end;
    } q[8];
};

struct WinParams {
    const int64_t *ptrs;
    const int64_t *indices;
    const uint32_t *indices32;
    const uint32_t *ptrs32;
    const int64_t *seeds;
    int64_t n_seeds, n_batches;
    int32_t n_hops, hop, k, kmax;
    int64_t cap_nodes, cap_edges;
    int64_t *samples, *rows, *cols, *edge_index, *layer_offsets, *counts;
    uint64_t seed, call_id;
    uint32_t tag;
    int64_t id_base;
    // workspace
    WinState *state;
    CallKey *call_keys;          // [n_batches] Philox key of batch b's draws
    unsigned long long *n_items; // [TG_MAX_HOPS] items written by K1 of each hop
    void *items_in, *items_sorted;
    uint32_t *hist; // [WIN_PART_BLOCKS][n_buckets]
    uint32_t *base; // [n_buckets + 1]
    WinQueues *queues;
    int32_t n_buckets, shift; // bucket count (multiple of 8), window = e0 >> shift
    int32_t slot_bits;        // narrow items: slot in the low bits, batch above
    // folded histogram (win_tuning().fold_hist): the emit kernels are persistent -- workgroup r walks batches r, r + n_rows,
    // ... and keeps the window histogram of their items in LDS (row r of `hist`); batch b's items lie at b * item_pitch
    int32_t n_rows;
    int64_t item_pitch;
    // staged form (ns_homo_stage.inl)
    uint32_t *vtab;  // [n_windows] first vertex whose column starts in window i or later
    uint32_t *stage; // [max_items][stage_words] {column start, degree, neighbours}
    int32_t n_windows, idx_bits, next_idx_bits;
    int32_t n_wbuckets; // window buckets (multiple of 64); the staged form sorts by n_buckets = n_wbuckets / 8 COARSE buckets first
    uint32_t store_align; // 7: the emit passes' store instructions start on 64-byte boundaries; 1: on 16-byte ones (round 3)
    void *items_fine;   // staged form: the items after the second sort level (what the gather kernel reads)
    uint16_t *item_keys;             // [max_items] staged form: an item's COARSE sort key, at the item's own index (written by whoever
                                     // counts the item: the first kernel or the histogram pass; read by the level-1 scatter)
    uint32_t *fine_rowtot;           // [n_buckets + 1] items per coarse bucket as the second level counted them
    uint32_t *fine_tot, *fine_start; // [(n_buckets + 1) * 128] items per absolute fine key / start of each key's run
    uint32_t *fine_tile_off;         // [tiles][256] offset of a tile's items inside each key's run
    int64_t next_pitch;
    int64_t b0; // staged form in parts: the kernels of a part walk batches [b0, b0 + n_batches)
    int32_t fine_sub_bits; // staged form: log2 of the sub-ranges a window's vertices are ordered into by the second sort level
    TG_BOUNDS_FIELDS
};

// One frontier vertex with something to sample.  Narrow form: launches whose edge pointers and per-batch offsets fit 32
// bits and whose (batch, slot) pair packs into one word.
struct WinItemN {
    uint32_t e0, deg, bs, e;
    __device__ __forceinline__ static WinItemN make(uint64_t e0, uint32_t deg, uint32_t b, uint32_t slot, uint32_t e,
                                                    int slot_bits) {
        return WinItemN{(uint32_t)e0, deg, (b << slot_bits) | slot, e};
    }
    __device__ __forceinline__ uint64_t col() const { return e0; }
    __device__ __forceinline__ uint32_t batch(int slot_bits) const { return bs >> slot_bits; }
    __device__ __forceinline__ uint32_t slot(int slot_bits) const { return bs & ((1u << slot_bits) - 1u); }
};
struct WinItemW {
    uint64_t e0;
    uint32_t deg, b, s, e;
    __device__ __forceinline__ static WinItemW make(uint64_t e0, uint32_t deg, uint32_t b, uint32_t slot, uint32_t e,
                                                    int) {
        return WinItemW{e0, deg, b, slot, e};
    }
    __device__ __forceinline__ uint64_t col() const { return e0; }
    __device__ __forceinline__ uint32_t batch(int) const { return b; }
    __device__ __forceinline__ uint32_t slot(int) const { return s; }
};
static_assert(sizeof(WinItemN) == 16 && sizeof(WinItemW) == 24, "item layouts");

__device__ __forceinline__ uint32_t win_bucket(uint64_t e0, int shift, int n_buckets) {
    const uint32_t w = (uint32_t)(e0 >> shift);
    return (w & 7u) * (uint32_t)(n_buckets >> 3) + (w >> 3); // XCD-major
}

// ---------------------------------------------------------------- init: seeds -> samples, per-batch state and keys
__global__ void win_init_kernel(const WinParams p, int64_t n_batches) {
    const int64_t b = blockIdx.x;
    int64_t *samples = p.samples + b * p.cap_nodes;
    for (int64_t i = threadIdx.x; i < p.n_seeds; i += blockDim.x) samples[i] = p.seeds[b * p.n_seeds + i]; // :184
    if (threadIdx.x == 0) {
        p.state[b] = WinState{0, p.n_seeds, 0, 0};
        p.call_keys[b] = call_key(p.seed, p.call_id + (uint64_t)b, p.tag);
        if (p.n_hops == 0) {
            p.counts[b * 2 + 0] = p.n_seeds;
            p.counts[b * 2 + 1] = 0;
        }
        if (b == 0)
            for (int h = 0; h < TG_MAX_HOPS; ++h) p.n_items[h] = 0ull;
    }
}

// K1 works through a batch's frontier in rounds of WIN_ROUND_CHUNKS 64-vertex chunks whose column bounds stay in LDS
// between the counting pass and the emitting pass (one `ptrs` look-up per frontier vertex and hop).
constexpr int WIN_ROUND_CHUNKS = 32;
constexpr int WIN_EMIT_MAX_THREADS = 512; // launch bound of the STAGED form's first and side kernels: without one the compiler
// budgets registers for 1 024-thread workgroups (128 VGPRs) and the ticket chain spills (50-200 B per lane).  The push form's
// emit kernels keep the 128-VGPR budget and their spills on purpose: bound to 512 threads they take 150-176 VGPRs, two
// wavefronts per SIMD instead of four, and the three output streams need the wavefronts (first hops 3.7 -> 4.7 ms, measured)
constexpr int WIN_STAGE_FIRST_RC = 16; // staged form: rounds of the first kernel (hop 0: usually <= 1 024 seeds) ...
constexpr int WIN_STAGE_SIDE_RC = 8;   // ... and of the side pass, which shares the CUs with the gather kernel

// K1's LDS: chunk offsets | fbase | column starts [slots] i64 | degrees [slots] u32 | per wave: staged positions
// [64*k] u32, staged lanes [64*k] u8
__host__ __device__ inline size_t win_emit_wave_lds_bytes(int kmax) { // ... + the chunk's drawing lanes [64] u8
    return (size_t)64 * kmax * sizeof(uint32_t) + (((size_t)64 * kmax + 15) & ~(size_t)15) + 64;
}

// rc = chunks per round (WIN_ROUND_CHUNKS by default; the staged form's first and side kernels take fewer: less LDS, more
// resident workgroups)
__host__ __device__ inline size_t win_emit_head_bytes(int rc = WIN_ROUND_CHUNKS) {
    return (((size_t)(rc + 1) * sizeof(uint32_t) + 15) & ~(size_t)15) + 16 +
           (size_t)rc * 64 * (sizeof(int64_t) + sizeof(uint32_t));
}
__host__ __device__ inline size_t win_emit_lds_bytes(int kmax, int n_waves, int rc = WIN_ROUND_CHUNKS) {
    return win_emit_head_bytes(rc) + (size_t)n_waves * win_emit_wave_lds_bytes(kmax);
}

// ---------------------------------------------------------------- K1: counts, offsets, draws, the three streams, items
// DIRECT: the hop's gathers are issued here, in batch order, and `samples` is written with the other streams -- no
// items, no sort, no K4.  That is the right form for hop 0: the seeds are arbitrary vertices, so a launch touches a
// line of `indices` only ~1.4 times there (RMAT-24, 4 096 batches: 11.3 M gathers over 8.3 M lines) and ordering them
// buys nothing; deeper frontiers are drawn by degree, repeat the hubs and touch every line ~13 times.
// One hop of one batch by one workgroup; returns the batch's state after the hop.  `hop` / `k` are arguments (not
// p.hop / p.k) so that one kernel can run consecutive hops of its batch back to back.
__device__ __forceinline__ void win_next_item(const WinParams &p, int64_t b, uint32_t rel, uint32_t v, uint32_t *lhist); // staged form

// NEXT (DIRECT only; staged form): every new sample is also handed to the next hop as an 8-byte item.
// ITEMS = false (staged form's side pass): only the three streams -- no items, no item-range atomics.
template <typename Item, int KMAX, bool REPLACE, bool DIRECT, bool FOLD = false, bool NEXT = false, bool ITEMS = true,
          int RC = WIN_ROUND_CHUNKS>
__device__ __forceinline__ WinState win_emit_hop(const WinParams &p, unsigned char *smem, const int64_t b, const int hop,
                                                 const int k, const WinState st, const CallKey ck,
                                                 uint32_t *lhist = nullptr) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n_waves = blockDim.x >> 6;
    uint32_t *chunk_off = reinterpret_cast<uint32_t *>(smem);
    constexpr int64_t ROUND_SLOTS = (int64_t)RC * 64;
    const size_t off_bytes = (((size_t)(RC + 1) * sizeof(uint32_t)) + 15) & ~(size_t)15;
    int64_t *shared_fbase = reinterpret_cast<int64_t *>(smem + off_bytes);
    int64_t *col0 = reinterpret_cast<int64_t *>(smem + off_bytes + 16);
    uint32_t *cdeg = reinterpret_cast<uint32_t *>(smem + off_bytes + 16 + (size_t)ROUND_SLOTS * sizeof(int64_t));
    unsigned char *wbase = smem + win_emit_head_bytes(RC) + (size_t)wave * win_emit_wave_lds_bytes(p.kmax);
    uint32_t *spos = reinterpret_cast<uint32_t *>(wbase);
    uint8_t *slane = wbase + (size_t)64 * p.kmax * sizeof(uint32_t);
    Item *items = static_cast<Item *>(p.items_in);

    int64_t *samples = p.samples + b * p.cap_nodes;
    int64_t *rows = p.rows + b * p.cap_edges;
    int64_t *cols = p.cols + b * p.cap_edges;
    int64_t *eidx = p.edge_index + b * p.cap_edges;
    const int64_t n_seeds = p.n_seeds;
    const int64_t begin = st.begin, end = st.end;
    int64_t ne = st.ne;

    if (tid == 0) {
        int64_t *lo = p.layer_offsets + (b * p.n_hops + hop) * 3; // :193
        lo[0] = n_seeds + ne;
        lo[1] = ne;
        lo[2] = n_seeds + ne;
        // this batch's range of the hop's flat item array (order between batches does not matter)
        if (!FOLD) *shared_fbase = (DIRECT || !ITEMS) ? 0 : (int64_t)atomicAdd(&p.n_items[hop], (unsigned long long)(end - begin));
    }
    if (!FOLD) __syncthreads();
    const int64_t fbase = FOLD ? b * p.item_pitch : *shared_fbase;

    for (int64_t round_begin = begin; round_begin < end; round_begin += ROUND_SLOTS) {
        const int64_t round_end = min(end, round_begin + ROUND_SLOTS);
        const int nc = (int)((round_end - round_begin + 63) >> 6);
        for (int c = wave; c < nc; c += n_waves) { // pass A: column bounds -> LDS, per-chunk sample counts
            const int64_t i = round_begin + (int64_t)c * 64 + lane;
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
            col0[c * 64 + lane] = e0;
            cdeg[c * 64 + lane] = deg > 0 ? (uint32_t)deg : 0u;
            const uint32_t cnt = (deg <= 0) ? 0u : (REPLACE ? (uint32_t)k : (uint32_t)min(deg, (int64_t)k));
            const uint32_t tot = wave_sum(cnt);
            if (lane == 0) chunk_off[c] = tot;
        }
        __syncthreads();
        if (wave == 0) { // scan of chunk totals (nc <= 64)
            const uint32_t v = lane < nc ? chunk_off[lane] : 0u;
            const uint32_t incl = wave_inclusive_scan(v);
            if (lane < nc) chunk_off[lane] = incl - v;
            if (lane == 63) chunk_off[nc] = incl;
        }
        __syncthreads();
        for (int c = wave; c < nc; c += n_waves) { // pass B: draws, items, the three streams
            const int64_t i0 = round_begin + (int64_t)c * 64;
            const int64_t i = i0 + lane;
            const int64_t e0 = col0[c * 64 + lane];
            const uint32_t n = cdeg[c * 64 + lane];
            const uint32_t cnt = (n == 0) ? 0u : (REPLACE ? (uint32_t)k : min(n, (uint32_t)k));
            const uint64_t did = (uint64_t)(p.id_base + i);
            const uint32_t incl = wave_inclusive_scan(cnt);
            const uint32_t excl = incl - cnt;
            const uint32_t total = __shfl(incl, 63, 64);
            const int64_t e_chunk = ne + (int64_t)chunk_off[c];
            if (!DIRECT && ITEMS && i < round_end) {
                items[fbase + (i - begin)] =
                    Item::make((uint64_t)e0, n, (uint32_t)b, (uint32_t)i, (uint32_t)(e_chunk + excl), p.slot_bits);
                if (FOLD && n) atomicAdd(&lhist[win_bucket((uint64_t)e0, p.shift, p.n_buckets)], 1u);
            }
            // Which lanes need Philox at all?  A frontier of arbitrary vertices (hop 0) has few of them per chunk (columns
            // longer than the fan-out), yet a wavefront pays the slot's ceil(k/2) Philox blocks for all 64 lanes.  With at
            // most half the lanes drawing, the (drawing slot, Philox block) pairs are dealt out over ALL lanes instead --
            // every lane computes one block and its two bounded draws per pass, the results meet in LDS (the staging
            // area, not yet in use), and the drawing lanes run only the cheap ticket chain.  Same draws (they are named
            // by (call, slot, block)), same positions.
            const bool draws = REPLACE ? (cnt > 0) : (n > (uint32_t)k);
            const uint64_t dmask = __ballot(draws);
            const int n_draw = __popcll(dmask);
            const bool spread = n_draw > 0 && n_draw <= 32;
            if (spread) {
                uint8_t *dl = slane + (((size_t)64 * p.kmax + 15) & ~(size_t)15); // drawing lanes, in lane order
                const int my_rank = __popcll(dmask & ((lane == 0) ? 0ull : (~0ull >> (64 - lane))));
                if (draws) dl[my_rank] = (uint8_t)lane;
                wave_lds_handoff();
                const int nb = (k + 3) >> 2; // Philox blocks per drawing slot: four 32-bit words each
                for (int t = lane; t < n_draw * nb; t += 64) {
                    const int rank = t / nb, blk = t - rank * nb;
                    const int src = dl[rank];
                    const uint32_t ns = cdeg[c * 64 + src];
                    const uint64_t sid = (uint64_t)(p.id_base + i0 + src);
                    const Draw d = draw(ck, sid, (uint32_t)blk, REPLACE ? D1_REPLACE : 0u);
#pragma unroll
                    for (int h = 0; h < 4; ++h) {
                        const int sl = 4 * blk + h;
                        if (sl < k)
                            spos[rank * k + sl] = slot_draw_from(d, ck, sid, (uint32_t)sl, REPLACE ? D1_REPLACE : 0u,
                                                                 REPLACE ? ns : (ns - 1u) - (uint32_t)sl);
                    }
                }
                wave_lds_handoff();
                uint32_t r[KMAX];
                if (draws) {
#pragma unroll
                    for (int sl = 0; sl < KMAX; ++sl)
                        if (sl < k) r[sl] = spos[my_rank * k + sl];
                }
                wave_lds_handoff(); // every lane holds its draws: the staging area may now be overwritten
                if (cnt > 0) {
                    if (REPLACE) {
#pragma unroll
                        for (int sl = 0; sl < KMAX; ++sl)
                            if (sl < k) {
                                spos[excl + sl] = r[sl];
                                slane[excl + sl] = (uint8_t)lane;
                            }
                    } else if (!draws) { // sampling.rs:12-15
                        for (uint32_t sl = 0; sl < cnt; ++sl) {
                            spos[excl + sl] = sl;
                            slane[excl + sl] = (uint8_t)lane;
                        }
                    } else {
                        sample_tickets_given<KMAX>(r, n, k, spos, slane, excl, lane);
                    }
                }
            } else if (cnt > 0) {
                if (REPLACE) { // sampling.rs:57-69
                    sample_replace_any(ck, did, n, k, spos, slane, excl, lane);
                } else if (n <= (uint32_t)k) { // sampling.rs:12-15
                    for (uint32_t s = 0; s < cnt; ++s) {
                        spos[excl + s] = s;
                        slane[excl + s] = (uint8_t)lane;
                    }
                } else {
                    sample_tickets<KMAX>(ck, did, n, k, spos, slane, excl, lane);
                }
            }
            wave_lds_handoff();
            if constexpr (DIRECT) { // gathers four per lane before the first store; unconditional loads (ns_homo.hip's emit)
                for (uint32_t q0 = 0; q0 < total; q0 += 256) {
                    int l4[4];
                    int64_t ep[4], v[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const uint32_t q = q0 + (uint32_t)(u * 64 + lane);
                        const uint32_t qq = q < total ? q : 0u;
                        l4[u] = slane[qq];
                        ep[u] = col0[c * 64 + l4[u]] + (int64_t)spos[qq];
                    }
                    if (p.indices32) {
#pragma unroll
                        for (int u = 0; u < 4; ++u) v[u] = (int64_t)__builtin_nontemporal_load(&p.indices32[ep[u]]);
                    } else {
#pragma unroll
                        for (int u = 0; u < 4; ++u) v[u] = __builtin_nontemporal_load(&p.indices[ep[u]]);
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const uint32_t q = q0 + (uint32_t)(u * 64 + lane);
                        if (q < total) {
                            const int64_t e = e_chunk + q;
                            samples[n_seeds + e] = v[u]; // :215 (the next hop's frontier)
                            __builtin_nontemporal_store(n_seeds + e, &rows[e]);
                            __builtin_nontemporal_store(i0 + (int64_t)l4[u], &cols[e]);
                            __builtin_nontemporal_store(ep[u], &eidx[e]);
                            if (NEXT) win_next_item(p, b, (uint32_t)(n_seeds + e - end), (uint32_t)v[u], lhist);
                        }
                    }
                }
            } else
            { // three write-once streams (:217).  The kernel is bound by HBM writes (without the draws it takes the same
                // time; with one stream instead of three, half), so the stores are shaped for the memory system:
                // streaming (plain ones: +12 %), one stream after the other instead of interleaved (-7 %), 16 bytes per
                // lane (-8 %; together -9 %; 32 bytes per lane and plain 16-byte stores were slower).  A `torch.fill_` of
                // the same bytes runs at 6.7 TB/s, this kernel at 4.4: tools/probe_write_bw.py.
                typedef long long i64x2 __attribute__((ext_vector_type(2)));
                const int64_t ea = e_chunk;
                // an element at an address that is not 16-byte aligned is stored alone (rows / cols / edge_index share the
                // parity: equal pitch, 16-byte aligned bases); taken from the ADDRESS -- with an odd cap_edges the slabs of
                // odd batches start on an odd element
                // ... up to the next 64-byte boundary (WinParams.store_align = 7; round 3: 16-byte, = 1): from there every store
                // instruction covers whole aligned chunks -- a chunk that two non-temporal store instructions share goes out as
                // two partial writes
                const uint32_t am = p.store_align;
                const uint32_t head = (uint32_t)((am + 1u - (uint32_t)(((uintptr_t)(rows + ea) >> 3) & am)) & am);
                if ((uint32_t)lane < head && (uint32_t)lane < total) {
                    __builtin_nontemporal_store(n_seeds + ea + (int64_t)lane, &rows[ea + lane]);
                    __builtin_nontemporal_store(i0 + (int64_t)slane[lane], &cols[ea + lane]);
                    __builtin_nontemporal_store(col0[c * 64 + slane[lane]] + (int64_t)spos[lane], &eidx[ea + lane]);
                }
                for (uint32_t q = head + 2u * lane; q < total; q += 128) {
                    const int64_t e = ea + q;
                    if (q + 1 < total) {
                        i64x2 r = {n_seeds + e, n_seeds + e + 1};
                        __builtin_nontemporal_store(r, reinterpret_cast<i64x2 *>(&rows[e]));
                    } else
                        __builtin_nontemporal_store(n_seeds + e, &rows[e]);
                }
                for (uint32_t q = head + 2u * lane; q < total; q += 128) {
                    const int64_t e = ea + q;
                    if (q + 1 < total) {
                        i64x2 cc = {i0 + (int64_t)slane[q], i0 + (int64_t)slane[q + 1]};
                        __builtin_nontemporal_store(cc, reinterpret_cast<i64x2 *>(&cols[e]));
                    } else
                        __builtin_nontemporal_store(i0 + (int64_t)slane[q], &cols[e]);
                }
                for (uint32_t q = head + 2u * lane; q < total; q += 128) {
                    const int64_t e = ea + q;
                    const int l0 = slane[q];
                    if (q + 1 < total) {
                        const int l1 = slane[q + 1];
                        i64x2 x = {col0[c * 64 + l0] + (int64_t)spos[q], col0[c * 64 + l1] + (int64_t)spos[q + 1]};
                        __builtin_nontemporal_store(x, reinterpret_cast<i64x2 *>(&eidx[e]));
                    } else
                        __builtin_nontemporal_store(col0[c * 64 + l0] + (int64_t)spos[q], &eidx[e]);
                }
            }
            wave_lds_handoff();
        }
        __syncthreads();
        ne += chunk_off[nc];
        __syncthreads();
    }
    return WinState{end, n_seeds + ne, ne, FOLD ? begin : fbase}; // :221-222  (FOLD: .fbase = where this hop's frontier began)
}

__device__ __forceinline__ void win_store_state(const WinParams &p, int64_t b, int hop, const WinState &st) {
    if (threadIdx.x == 0) {
        p.state[b] = st;
        if (hop == p.n_hops - 1) {
            p.counts[b * 2 + 0] = st.end;
            p.counts[b * 2 + 1] = st.ne;
        }
    }
}

template <typename Item, int KMAX, bool REPLACE, bool DIRECT>
__global__ void win_emit_kernel(const WinParams p) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int64_t b = blockIdx.x;
    const WinState st = win_emit_hop<Item, KMAX, REPLACE, DIRECT>(p, smem, b, p.hop, p.k, p.state[b], p.call_keys[b]);
    win_store_state(p, b, p.hop, st);
}

// The first hops of a batch in ONE kernel: seeds -> samples (win_init's work), hop 0 DIRECT (its gathers issued here, in
// batch order: bound by the rate of random line requests, ~55 G/s) and, when the launch has a second hop, that hop's
// emit pass (bound by its three write streams) -- workgroups in different phases run side by side, so the random reads
// of one batch's hop 0 hide under the streaming writes of another's hop 1, which two kernels one after the other cannot do.
// The workgroup reads back the samples it wrote itself (workgroup barrier between the hops).
template <typename Item, int KMAX, bool REPLACE>
__global__ void win_first_hops_kernel(const WinParams p, const int k0, const int k1) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int64_t b = blockIdx.x;
    int64_t *samples = p.samples + b * p.cap_nodes;
    for (int64_t i = threadIdx.x; i < p.n_seeds; i += blockDim.x) samples[i] = p.seeds[b * p.n_seeds + i]; // :184
    const CallKey ck = call_key(p.seed, p.call_id + (uint64_t)b, p.tag);
    if (threadIdx.x == 0) p.call_keys[b] = ck;
    __syncthreads();
    WinState st = win_emit_hop<Item, KMAX, REPLACE, true>(p, smem, b, 0, k0, WinState{0, p.n_seeds, 0, 0}, ck);
    if (p.n_hops > 1) {
        __syncthreads(); // hop 0's samples (plain stores of this workgroup) are the frontier read next
        st = win_emit_hop<Item, KMAX, REPLACE, false>(p, smem, b, 1, k1, st, ck);
        win_store_state(p, b, 1, st);
    } else
        win_store_state(p, b, 0, st);
}


// Folded-histogram forms (persistent): workgroup r walks batches r, r + gridDim.x, ...; the window histogram of the
// items it writes accumulates in LDS and leaves as row r of `hist` -- the separate histogram pass over the items (and the
// atomics that dealt out item ranges) are gone.  win_scatter_fold_kernel walks the same batches per row.
template <typename Item, int KMAX, bool REPLACE>
__global__ void win_emit_fold_kernel(const WinParams p) {
    extern __shared__ __align__(16) unsigned char smem[];
    uint32_t *lhist = reinterpret_cast<uint32_t *>(smem + win_emit_lds_bytes(p.kmax, blockDim.x >> 6));
    for (int i = threadIdx.x; i < p.n_buckets; i += blockDim.x) lhist[i] = 0;
    __syncthreads();
    for (int64_t b = blockIdx.x; b < p.n_batches; b += gridDim.x) {
        const WinState st =
            win_emit_hop<Item, KMAX, REPLACE, false, true>(p, smem, b, p.hop, p.k, p.state[b], p.call_keys[b], lhist);
        win_store_state(p, b, p.hop, st);
    }
    __syncthreads();
    uint32_t *row = p.hist + (size_t)blockIdx.x * p.n_buckets;
    for (int i = threadIdx.x; i < p.n_buckets; i += blockDim.x) row[i] = lhist[i];
}

template <typename Item, int KMAX, bool REPLACE>
__global__ void win_first_hops_fold_kernel(const WinParams p, const int k0, const int k1) {
    extern __shared__ __align__(16) unsigned char smem[];
    uint32_t *lhist = reinterpret_cast<uint32_t *>(smem + win_emit_lds_bytes(p.kmax, blockDim.x >> 6));
    for (int i = threadIdx.x; i < p.n_buckets; i += blockDim.x) lhist[i] = 0;
    for (int64_t b = blockIdx.x; b < p.n_batches; b += gridDim.x) {
        int64_t *samples = p.samples + b * p.cap_nodes;
        for (int64_t i = threadIdx.x; i < p.n_seeds; i += blockDim.x) samples[i] = p.seeds[b * p.n_seeds + i]; // :184
        const CallKey ck = call_key(p.seed, p.call_id + (uint64_t)b, p.tag);
        if (threadIdx.x == 0) p.call_keys[b] = ck;
        __syncthreads();
        WinState st = win_emit_hop<Item, KMAX, REPLACE, true>(p, smem, b, 0, k0, WinState{0, p.n_seeds, 0, 0}, ck);
        __syncthreads();
        st = win_emit_hop<Item, KMAX, REPLACE, false, true>(p, smem, b, 1, k1, st, ck, lhist);
        win_store_state(p, b, 1, st);
        __syncthreads(); // the LDS staging of this batch is done before the next batch reuses it
    }
    __syncthreads();
    uint32_t *row = p.hist + (size_t)blockIdx.x * p.n_buckets;
    for (int i = threadIdx.x; i < p.n_buckets; i += blockDim.x) row[i] = lhist[i];
}

template <typename Item>
__global__ void __launch_bounds__(WIN_PART_THREADS) win_scatter_fold_kernel(const WinParams p) {
    __shared__ uint32_t cur[WIN_MAX_BUCKETS];
    const int nb = p.n_buckets;
    const Item *items = static_cast<const Item *>(p.items_in);
    Item *sorted = static_cast<Item *>(p.items_sorted);
    const uint32_t *row = p.hist + (size_t)blockIdx.x * nb;
    for (int i = threadIdx.x; i < nb; i += blockDim.x) cur[i] = p.base[i] + row[i];
    __syncthreads();
    constexpr int U = 4; // items per thread and round, all loads in flight together
    for (int64_t b = blockIdx.x; b < p.n_batches; b += gridDim.x) {
        const WinState st = p.state[b];
        const int64_t n = st.begin - st.fbase; // the frontier the emit pass just walked
        const Item *src = items + b * p.item_pitch;
        for (int64_t t0 = 0; t0 < n; t0 += (int64_t)U * WIN_PART_THREADS) {
            Item it[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t j = t0 + (int64_t)u * WIN_PART_THREADS + threadIdx.x;
                it[u].deg = 0;
                if (j < n) it[u] = src[j];
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (it[u].deg) sorted[atomicAdd(&cur[win_bucket(it[u].col(), p.shift, nb)], 1u)] = it[u];
        }
    }
}

// ---------------------------------------------------------------- P1: per-block bucket histogram of the hop's items
template <typename Item>
__global__ void __launch_bounds__(WIN_PART_THREADS) win_hist_kernel(const WinParams p) {
    __shared__ uint32_t h[WIN_MAX_BUCKETS];
    const int nb = p.n_buckets;
    const Item *items = static_cast<const Item *>(p.items_in);
    for (int i = threadIdx.x; i < nb; i += blockDim.x) h[i] = 0;
    __syncthreads();
    const uint64_t n = p.n_items[p.hop];
    for (uint64_t t0 = (uint64_t)blockIdx.x * WIN_TILE; t0 < n; t0 += (uint64_t)gridDim.x * WIN_TILE) {
        for (uint64_t j = t0 + threadIdx.x; j < min(n, t0 + WIN_TILE); j += blockDim.x) {
            const Item it = items[j];
            if (it.deg) atomicAdd(&h[win_bucket(it.col(), p.shift, nb)], 1u);
        }
    }
    __syncthreads();
    uint32_t *row = p.hist + (size_t)blockIdx.x * nb;
    for (int i = threadIdx.x; i < nb; i += blockDim.x) row[i] = h[i];
}

// ---------------------------------------------------------------- P2a: per bucket, exclusive running sum over the blocks
// block = 64 buckets x 16 row groups: a thread sums its 32 rows, the 16 partial sums of a bucket are scanned through LDS,
// then the thread rewrites its rows as running offsets (depth 32 instead of 512 dependent steps)
constexpr int WIN_SCAN_GROUPS = 16;
__global__ void __launch_bounds__(64 * WIN_SCAN_GROUPS) win_colscan_kernel(const WinParams p, int n_rows) {
    __shared__ uint32_t part[WIN_SCAN_GROUPS][64];
    const int bl = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int bkt = blockIdx.x * 64 + bl;
    const int rows_per = (n_rows + WIN_SCAN_GROUPS - 1) / WIN_SCAN_GROUPS;
    const int r0 = g * rows_per, r1 = min(n_rows, r0 + rows_per);
    uint32_t sum = 0;
    if (bkt < p.n_buckets)
        for (int r = r0; r < r1; ++r) sum += p.hist[(size_t)r * p.n_buckets + bkt];
    part[g][bl] = sum;
    __syncthreads();
    uint32_t run = 0;
    for (int gg = 0; gg < g; ++gg) run += part[gg][bl];
    if (bkt < p.n_buckets) {
        for (int r = r0; r < r1; ++r) {
            uint32_t *cell = p.hist + (size_t)r * p.n_buckets + bkt;
            const uint32_t v = *cell;
            *cell = run;
            run += v;
        }
        if (g == WIN_SCAN_GROUPS - 1) p.base[bkt] = run; // column total (scanned in place by P2b)
    }
}

// ---------------------------------------------------------------- P2b: bucket bases + the per-XCD queues (one block)
__global__ void win_basescan_kernel(const WinParams p) {
    __shared__ uint32_t wave_tot[16];
    __shared__ uint32_t carry_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int c0 = 0; c0 < p.n_buckets; c0 += blockDim.x) {
        const int i = c0 + tid;
        const uint32_t v = i < p.n_buckets ? p.base[i] : 0u;
        const uint32_t incl = wave_inclusive_scan(v);
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        uint32_t off = carry_s;
        for (int w = 0; w < wave; ++w) off += wave_tot[w];
        if (i < p.n_buckets) p.base[i] = off + incl - v;
        __syncthreads();
        if (tid == 0) {
            uint32_t t = carry_s;
            for (int w = 0; w < nw; ++w) t += wave_tot[w];
            carry_s = t;
        }
        __syncthreads();
    }
    if (tid == 0) p.base[p.n_buckets] = carry_s;
    __syncthreads();
    if (tid < 8) { // queue x = buckets [x*nb/8, (x+1)*nb/8)
        const int per = p.n_buckets >> 3;
        p.queues->q[tid].head = p.base[tid * per];
        p.queues->q[tid].end = p.base[(tid + 1) * per];
    }
}

// ---------------------------------------------------------------- P3: scatter the items into window order
template <typename Item>
__global__ void __launch_bounds__(WIN_PART_THREADS) win_scatter_kernel(const WinParams p) {
    __shared__ uint32_t cur[WIN_MAX_BUCKETS];
    const int nb = p.n_buckets;
    const Item *items = static_cast<const Item *>(p.items_in);
    Item *sorted = static_cast<Item *>(p.items_sorted);
    const uint32_t *row = p.hist + (size_t)blockIdx.x * nb;
    for (int i = threadIdx.x; i < nb; i += blockDim.x) cur[i] = p.base[i] + row[i];
    __syncthreads();
    const uint64_t n = p.n_items[p.hop];
    constexpr int U = WIN_TILE / WIN_PART_THREADS; // items per thread and tile, all loads in flight together
    for (uint64_t t0 = (uint64_t)blockIdx.x * WIN_TILE; t0 < n; t0 += (uint64_t)gridDim.x * WIN_TILE) {
        Item it[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint64_t j = t0 + (uint64_t)u * WIN_PART_THREADS + threadIdx.x;
            it[u].deg = 0;
            if (j < n) it[u] = items[j];
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (it[u].deg) sorted[atomicAdd(&cur[win_bucket(it[u].col(), p.shift, nb)], 1u)] = it[u];
    }
}

// ---------------------------------------------------------------- K4: window-ordered gather
#ifndef TG_WIN_EMIT
#define TG_WIN_EMIT 5
#endif
constexpr int WIN_EMIT = TG_WIN_EMIT;

// per wave: column start [64] i64 | samples base [64] i64 | staged positions [64*k] u32 | lanes u8
__host__ __device__ inline size_t win_gather_wave_lds_bytes(int kmax) {
    return 2 * 64 * sizeof(int64_t) + (size_t)64 * kmax * sizeof(uint32_t) + (((size_t)64 * kmax + 15) & ~(size_t)15);
}

// The staged (lane, position) pairs of one round are walked by consecutive lanes: gather the neighbour id and write it
// (neighbor_sampling.rs:211-215; edge_index was written by K1).  WIN_EMIT gathers per lane and batch, two batches in flight;
// unconditional loads (lanes past the end re-read element 0's address and skip the stores).
template <typename IDX>
__device__ __forceinline__ void win_gather_chunk(const IDX *__restrict__ idx, uint32_t total, int lane,
                                                 const uint8_t *slane, const uint32_t *spos, const int64_t *ebase,
                                                 const int64_t *obase, int64_t *samples) {
    if (total == 0) return;
    struct Batch {
        int64_t o[WIN_EMIT];
        IDX v[WIN_EMIT];
    };
    auto issue = [&](Batch &t, uint32_t q0) {
        int64_t ep[WIN_EMIT];
#pragma unroll
        for (int u = 0; u < WIN_EMIT; ++u) {
            const uint32_t q = q0 + (uint32_t)(u * 64 + lane);
            const uint32_t qq = q < total ? q : 0u;
            const int l = slane[qq];
            ep[u] = ebase[l] + (int64_t)spos[qq];
            t.o[u] = obase[l] + (int64_t)qq; // the base already holds "minus the item's first q"
        }
#pragma unroll
        for (int u = 0; u < WIN_EMIT; ++u) t.v[u] = idx[ep[u]];
    };
    auto store = [&](const Batch &t, uint32_t q0) {
#pragma unroll
        for (int u = 0; u < WIN_EMIT; ++u) {
            const uint32_t q = q0 + (uint32_t)(u * 64 + lane);
            // :215  plain stores: the partial lines of neighbouring runs merge in L2 (streaming them was slower)
            if (q < total) samples[t.o[u]] = (int64_t)t.v[u];
        }
    };
    Batch a, b;
    issue(a, 0u);
    for (uint32_t q0 = 0; q0 < total; q0 += 2u * 64u * WIN_EMIT) {
        issue(b, q0 + 64u * WIN_EMIT);
        store(a, q0);
        issue(a, q0 + 2u * 64u * WIN_EMIT);
        store(b, q0 + 64u * WIN_EMIT);
    }
}

// Work split: blocks with equal blockIdx % 8 are observed to share an XCD (dispatch is round-robin; a speed matter
// only), so group x = blockIdx % 8 sweeps queue x -- the windows w with w % 8 == x -- IN ORDER: a block takes the
// queue's next slice of 64 items per wave with ONE agent-scope atomic (the following slice is reserved while the
// current one is processed, so the atomic's latency is hidden), i.e. a few thousand atomics per queue, each queue head
// on a line of its own.  At any time the group works inside a front of (blocks x 2 slices) items = a few windows,
// which its XCD's 4 MB L2 holds: every line of `indices` is fetched from HBM about once per hop.  Every slice of
// every queue goes to exactly one block whatever the placement.
template <typename Item, int KMAX, bool REPLACE>
__global__ void win_gather_kernel(const WinParams p) {
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ unsigned long long slice_lo[2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned char *wbase = smem + (size_t)wave * win_gather_wave_lds_bytes(p.kmax);
    int64_t *ebase = reinterpret_cast<int64_t *>(wbase);
    int64_t *obase = ebase + 64;
    uint32_t *spos = reinterpret_cast<uint32_t *>(wbase + 128 * sizeof(int64_t));
    uint8_t *slane = reinterpret_cast<uint8_t *>(wbase + 128 * sizeof(int64_t) + (size_t)64 * p.kmax * sizeof(uint32_t));
    const Item *items = static_cast<const Item *>(p.items_sorted);
    const int k = p.k;
    WinQueues::Q *Q = &p.queues->q[blockIdx.x & 7];
    const unsigned long long qend = Q->end;
    const unsigned long long slice = blockDim.x; // 64 items per wave

    // (reserving two slices ahead to prefetch the next slice's items, and half-filled wavefronts with twice the
    // blocks, were both measured: each widens the front or thins the arithmetic and neither was faster)
    if (tid == 0) slice_lo[0] = atomicAdd(&Q->head, slice);
    __syncthreads();
    for (int buf = 0;; buf ^= 1) {
        const unsigned long long lo = slice_lo[buf];
        if (lo >= qend) break; // uniform: every wave reads the same word
        unsigned long long nxt = 0;
        if (tid == 0) nxt = atomicAdd(&Q->head, slice); // used at the end of this slice
        const unsigned long long j = lo + (unsigned long long)wave * 64 + lane;
        Item it;
        it.deg = 0;
        if (j < qend) it = items[j];
        const uint32_t n = it.deg;
        const uint32_t cnt = (n == 0) ? 0u : (REPLACE ? (uint32_t)k : min(n, (uint32_t)k));
        const uint32_t incl = wave_inclusive_scan(cnt);
        const uint32_t excl = incl - cnt;
        const uint32_t total = __shfl(incl, 63, 64);
        if (cnt > 0) {
            const uint32_t b = it.batch(p.slot_bits);
            const CallKey ck = p.call_keys[b];
            const uint64_t did = (uint64_t)(p.id_base + (int64_t)it.slot(p.slot_bits));
            ebase[lane] = (int64_t)it.col();
            obase[lane] = (int64_t)b * p.cap_nodes + p.n_seeds + (int64_t)it.e - (int64_t)excl;
            if (REPLACE) { // sampling.rs:57-69, k draws of U[0,n)
                sample_replace_any(ck, did, n, k, spos, slane, excl, lane);
            } else if (n <= (uint32_t)k) { // sampling.rs:12-15: the reservoir is just filled
                for (uint32_t s = 0; s < cnt; ++s) {
                    spos[excl + s] = s;
                    slane[excl + s] = (uint8_t)lane;
                }
            } else {
                sample_tickets<KMAX>(ck, did, n, k, spos, slane, excl, lane);
            }
        }
        wave_lds_handoff();
        if (p.indices32)
            win_gather_chunk<uint32_t>(p.indices32, total, lane, slane, spos, ebase, obase, p.samples);
        else
            win_gather_chunk<int64_t>(p.indices, total, lane, slane, spos, ebase, obase, p.samples);
        if (tid == 0) slice_lo[buf ^ 1] = nxt;
        __syncthreads(); // the next slice's start is published; also fences this wave's LDS staging
    }
}

#include "ns_homo_stage.inl"

// lhist (first kernel): the workgroup's coarse histogram in LDS, its vertex table right behind it
__device__ __forceinline__ void win_next_item(const WinParams &p, int64_t b, uint32_t rel, uint32_t v, uint32_t *lhist) {
    static_cast<WinItem8 *>(p.items_in)[b * p.next_pitch + rel] = WinItem8{v, ((uint32_t)b << p.next_idx_bits) | rel};
    if (lhist) { // the one search of the vertex table this item costs: the key is kept for the level-1 scatter
        const uint32_t c = win_stage_key(lhist + p.n_buckets, p.n_windows, p.n_wbuckets, v, 4).coarse;
        atomicAdd(&lhist[c], 1u);
        p.item_keys[b * p.next_pitch + rel] = (uint16_t)c;
    }
}

static int win_env_int(const char *name, int dflt) {
    const char *v = getenv(name);
    return v ? atoi(v) : dflt;
}

// Tuning of the window-ordered launch: defaults <- environment (read once) <- tg_ns_win_tuning_set (any time; tests use
// it to force many small windows on a small graph).  Outputs never depend on any of these.
struct WinTuning {
    int64_t window_bytes;
    int32_t gather_blocks, gather_threads, emit_threads, direct_hop0, fuse_first_hops, fold_hist, emit_blocks;
    int32_t staged, stage_round_chunks, stage_gather_threads, stage_gather_blocks, stage_emit_threads, stage_parts, stage_part_min_batches, stage_sort_blocks;
    int32_t stage_fine, stage_concurrent, stage_split, stage_split_round_chunks, store_align64, stage_fine_sub_bits, stage_fine_blocks;
};
static WinTuning &win_tuning() {
    static WinTuning t = {
        (int64_t)win_env_int("TG_WIN_KIB", 512) * 1024,
        win_env_int("TG_WIN_GATHER_BLOCKS", 256),
        win_env_int("TG_WIN_GATHER_THREADS", 512),
        win_env_int("TG_WIN_EMIT_THREADS", 256),
        win_env_int("TG_WIN_DIRECT_HOP0", 1),
        win_env_int("TG_WIN_FUSE_FIRST_HOPS", 1),
        win_env_int("TG_WIN_FOLD_HIST", 1),
        win_env_int("TG_WIN_EMIT_BLOCKS", 768),
        win_env_int("TG_WIN_STAGED", 2),
        win_env_int("TG_WIN_STAGE_ROUND_CHUNKS", 2),
        win_env_int("TG_WIN_STAGE_GATHER_THREADS", 512),
        win_env_int("TG_WIN_STAGE_GATHER_BLOCKS", 512),
        win_env_int("TG_WIN_STAGE_EMIT_THREADS", 256),
        win_env_int("TG_WIN_STAGE_PARTS", 1),
        win_env_int("TG_WIN_STAGE_PART_MIN_BATCHES", 1024),
        win_env_int("TG_WIN_STAGE_SORT_BLOCKS", 768),
        win_env_int("TG_WIN_STAGE_FINE", 1),
        win_env_int("TG_WIN_STAGE_CONCURRENT", 0),
        win_env_int("TG_WIN_STAGE_SPLIT", 0),
        win_env_int("TG_WIN_STAGE_SPLIT_ROUND_CHUNKS", 4),
        win_env_int("TG_WIN_STORE_ALIGN64", 1),
        win_env_int("TG_WIN_STAGE_FINE_SUB_BITS", 7),
        win_env_int("TG_WIN_STAGE_FINE_BLOCKS", 2048),
    };
    return t;
}

// Optional per-stage timing of the last launch (tg_ns_win_stage_timing): HIP events between the kernels, on the
// launch's stream.  Off by default; when on, a launch records (stages + 1) events and nothing else changes.
constexpr int WIN_MAX_STAGES = 8 * TG_MAX_HOPS + 2;
struct WinStageClock {
    bool enabled = false;
    int n = 0;
    hipEvent_t ev[WIN_MAX_STAGES + 1] = {};
    char name[WIN_MAX_STAGES][24] = {};
    void begin(hipStream_t s) {
        n = 0;
        if (!enabled) return;
        if (!ev[0])
            for (auto &e : ev) (void)hipEventCreate(&e);
        (void)hipEventRecord(ev[0], s);
    }
    void mark(const char *what, int hop, hipStream_t s) {
        if (!enabled || n >= WIN_MAX_STAGES) return;
        snprintf(name[n], sizeof(name[n]), "%s.h%d", what, hop);
        ++n;
        (void)hipEventRecord(ev[n], s);
    }
};
static WinStageClock &win_clock() {
    static WinStageClock c;
    return c;
}

struct WinLayout {
    size_t state, call_keys, n_items, queues, hist, base, items_in, items_sorted, vtab, fine_tot, fine_start, fine_tile_off, item_keys, stage, total, total_push;
    int64_t max_items;
    int stage_words; // 16 / 32: words per stage slot of the staged form; 0: fan-outs beyond it (push form only)
};

// bit widths of a stage slot's pairs for this graph (ns_homo_stage.inl): vertex ids, positions inside a column
static StageBits win_stage_bits(const tg_graph *csc) {
    StageBits sb;
    if (!csc) {
        sb.bv = sb.bp = 32;
        return sb;
    }
    sb.bv = stage_bits_of((uint64_t)std::max<int64_t>(csc->n_major, 1) - 1);
    const int64_t longest = csc->max_degree > 0 ? csc->max_degree : csc->n_edges;
    sb.bp = stage_bits_of((uint64_t)std::max<int64_t>(longest, 1) - 1);
    return sb;
}
// words per stage slot for the largest fan-out of an ORDERED hop (hop 0 is direct): 16 (one 64-byte chunk), 32 (two), or 0
// when the pairs do not fit two chunks (the launch then takes the push form).  csc == NULL: 32-bit fields assumed.
static int win_stage_words(const tg_graph *csc, const int64_t *fanout, int32_t n_hops) {
    int64_t k = 0;
    for (int h = 1; h < n_hops; ++h) k = std::max(k, fanout[h]);
    if (n_hops < 2 || k > 32) return 0;
    if (!csc) return 32; // sizing without a graph: the most a launch can ask for
    const int bits = stage_slot_bits((int)k, win_stage_bits(csc));
    return (bits <= 512 && k <= 16) ? 16 : (bits <= 1024 ? 32 : 0);
}

static WinLayout win_layout(const tg_graph *csc, int64_t n_batches, int64_t n_seeds, const int64_t *fanout, int32_t n_hops) {
    WinLayout L;
    int64_t layer = n_seeds, widest = n_seeds;
    for (int h = 0; h + 1 < n_hops; ++h) {
        layer *= fanout[h];
        if (layer > widest) widest = layer;
    }
    L.max_items = n_batches * widest;
    size_t at = 0;
    auto take = [&](size_t bytes) {
        const size_t here = at;
        at += (bytes + 255) & ~(size_t)255;
        return here;
    };
    L.state = take((size_t)n_batches * sizeof(WinState));
    L.call_keys = take((size_t)n_batches * sizeof(CallKey));
    L.n_items = take(TG_MAX_HOPS * sizeof(unsigned long long));
    L.queues = take(WIN_MAX_PARTS * sizeof(WinQueues));
    L.hist = take((size_t)std::max(WIN_MAX_ROWS, WIN_MAX_PARTS * WIN_PART_BLOCKS) * WIN_MAX_BUCKETS * sizeof(uint32_t));
    L.base = take((size_t)WIN_MAX_PARTS * (WIN_MAX_BUCKETS + 8) * sizeof(uint32_t));
    L.items_in = take((size_t)L.max_items * sizeof(WinItemW)); // sized for the wide form
    L.items_sorted = take((size_t)L.max_items * sizeof(WinItemW));
    L.vtab = take((size_t)(WIN_MAX_BUCKETS + 8) * sizeof(uint32_t));
    L.total_push = at; // what the push form needs; the staged form's tables and slots come after it
    L.fine_tot = take((size_t)WIN_MAX_PARTS * (WIN_MAX_BUCKETS / 8 + 8) * WIN_FINE_PER_COARSE_MAX * sizeof(uint32_t));
    L.fine_start = take((size_t)WIN_MAX_PARTS * (WIN_MAX_BUCKETS / 8 + 8) * WIN_FINE_PER_COARSE_MAX * sizeof(uint32_t));
    L.fine_tile_off = take(((size_t)L.max_items / WIN_FINE_TILE + WIN_MAX_PARTS + 1) * WIN_FINE_BINS_MAX * sizeof(uint32_t));
    L.item_keys = take((size_t)L.max_items * sizeof(uint16_t));
    L.stage_words = win_stage_words(csc, fanout, n_hops);
    L.stage = take((size_t)L.max_items * L.stage_words * sizeof(uint32_t));
    L.total = at;
    return L;
}

template <typename Item, int KMAX, bool REPLACE>
static int win_run(WinParams p, int64_t n_batches, const int64_t *fanout, int32_t n_hops, hipStream_t stream) {
    const WinTuning t = win_tuning();
    WinStageClock &clk = win_clock();
    const int gather_blocks = (t.gather_blocks + 7) & ~7; // 8 groups of equal size
    // folded histogram: needs a second hop (hop 0 has no items) and the fused first hops
    const bool fold = t.fold_hist && t.direct_hop0 && t.fuse_first_hops && n_hops > 1;
    const size_t hist_lds = fold ? (size_t)p.n_buckets * sizeof(uint32_t) : 0;
    auto emit_threads_for = [&]() {
        int threads = t.emit_threads;
        while (threads > 64 && win_emit_lds_bytes(p.kmax, threads / 64) + hist_lds > 64 * 1024)
            threads = ((threads >> 1) + 63) & ~63;
        return threads;
    };
    p.n_batches = n_batches;
    p.n_rows = (int32_t)std::min<int64_t>(std::min(std::max(t.emit_blocks, 1), WIN_MAX_ROWS), n_batches);
    auto pitch_of = [&](int h) { // worst-case frontier of hop h per batch
        int64_t f = p.n_seeds;
        for (int j = 0; j < h; ++j) f *= fanout[j];
        return f;
    };
    clk.begin(stream);
    int h0 = 0;
    if (t.direct_hop0 && t.fuse_first_hops) { // seeds, hop 0 (direct) and hop 1's emit pass by one kernel
        const int threads = emit_threads_for();
        if (fold) {
            p.item_pitch = pitch_of(1);
            hipLaunchKernelGGL((win_first_hops_fold_kernel<Item, KMAX, REPLACE>), dim3((unsigned)p.n_rows), dim3(threads),
                               win_emit_lds_bytes(p.kmax, threads / 64) + hist_lds, stream, p, (int)fanout[0], (int)fanout[1]);
        } else {
            TG_HIP(hipMemsetAsync(p.n_items, 0, TG_MAX_HOPS * sizeof(unsigned long long), stream));
            hipLaunchKernelGGL((win_first_hops_kernel<Item, KMAX, REPLACE>), dim3((unsigned)n_batches), dim3(threads),
                               win_emit_lds_bytes(p.kmax, threads / 64), stream, p, (int)fanout[0],
                               n_hops > 1 ? (int)fanout[1] : 0);
        }
        TG_LAUNCH_CHECK();
        clk.mark("first_hops", 0, stream);
        h0 = 1;
    } else {
        hipLaunchKernelGGL(win_init_kernel, dim3((unsigned)n_batches), dim3(256), 0, stream, p, n_batches);
        TG_LAUNCH_CHECK();
        clk.mark("init", 0, stream);
    }
    for (int h = h0; h < n_hops; ++h) {
        p.hop = h;
        p.k = (int32_t)fanout[h];
        p.item_pitch = pitch_of(h);
        const bool emitted = (h == 1 && h0 == 1); // the fused kernel already ran hop 1's emit pass
        if (!emitted) {
            const int threads = emit_threads_for();
            if (h == 0 && t.direct_hop0) { // arbitrary seeds: nothing to gain from ordering their gathers
                hipLaunchKernelGGL((win_emit_kernel<Item, KMAX, REPLACE, true>), dim3((unsigned)n_batches), dim3(threads),
                                   win_emit_lds_bytes(p.kmax, threads / 64), stream, p);
                TG_LAUNCH_CHECK();
                clk.mark("emit_direct", h, stream);
                continue;
            }
            if (fold)
                hipLaunchKernelGGL((win_emit_fold_kernel<Item, KMAX, REPLACE>), dim3((unsigned)p.n_rows), dim3(threads),
                                   win_emit_lds_bytes(p.kmax, threads / 64) + hist_lds, stream, p);
            else
                hipLaunchKernelGGL((win_emit_kernel<Item, KMAX, REPLACE, false>), dim3((unsigned)n_batches),
                                   dim3(threads), win_emit_lds_bytes(p.kmax, threads / 64), stream, p);
            TG_LAUNCH_CHECK();
            clk.mark("emit", h, stream);
        }
        if (!fold) {
            hipLaunchKernelGGL(win_hist_kernel<Item>, dim3(WIN_PART_BLOCKS), dim3(WIN_PART_THREADS), 0, stream, p);
            TG_LAUNCH_CHECK();
            clk.mark("hist", h, stream);
        }
        hipLaunchKernelGGL(win_colscan_kernel, dim3((p.n_buckets + 63) / 64), dim3(64 * WIN_SCAN_GROUPS), 0, stream, p,
                           fold ? p.n_rows : WIN_PART_BLOCKS);
        TG_LAUNCH_CHECK();
        hipLaunchKernelGGL(win_basescan_kernel, dim3(1), dim3(1024), 0, stream, p);
        TG_LAUNCH_CHECK();
        clk.mark("scans", h, stream);
        if (fold)
            hipLaunchKernelGGL(win_scatter_fold_kernel<Item>, dim3((unsigned)p.n_rows), dim3(WIN_PART_THREADS), 0, stream, p);
        else
            hipLaunchKernelGGL(win_scatter_kernel<Item>, dim3(WIN_PART_BLOCKS), dim3(WIN_PART_THREADS), 0, stream, p);
        TG_LAUNCH_CHECK();
        clk.mark("scatter", h, stream);
        int gthreads = t.gather_threads;
        while (gthreads > 64 && (size_t)(gthreads / 64) * win_gather_wave_lds_bytes(p.kmax) > 64 * 1024)
            gthreads = ((gthreads >> 1) + 63) & ~63;
        hipLaunchKernelGGL((win_gather_kernel<Item, KMAX, REPLACE>), dim3(gather_blocks), dim3(gthreads),
                           (size_t)(gthreads / 64) * win_gather_wave_lds_bytes(p.kmax), stream, p);
        TG_LAUNCH_CHECK();
        clk.mark("gather", h, stream);
    }
    return TG_OK;
}

// ---------------------------------------------------------------- staged form: host side
// tg_ns_win_tuning.staged: 0 = never, 1 = whenever it applies, 2 (default) = where it measures faster than the push form:
// launches of >= 2 048 batches (every launch that takes a window-ordered form at all) whose stage slots are ONE chunk
// (RMAT-24, [15, 10], final build of round 4: 1.24 against 1.29 ms at 2 048 batches, 2.07 / 2.10 at 4 096, 3.61 / 4.09 at
// 8 192, 5.12 / 5.91 at 12 288, 6.7 / 7.9-8.2 at 16 384; profiles/r04/sweep_launch_size_final.jsonl)
constexpr int64_t WIN_STAGED_AUTO_MIN_BATCHES = 2048;
static bool win_staged_wanted(const WinTuning &t, int64_t n_batches, int stage_words) {
    if (t.staged == 2) return n_batches >= WIN_STAGED_AUTO_MIN_BATCHES && stage_words == 16;
    return t.staged != 0;
}
static bool win_staged_applicable(const WinParams &p, const WinTuning &t, const tg_graph *csc, int64_t n_batches,
                                  const int64_t *fanout, int32_t n_hops, int stage_words) {
    if (!win_staged_wanted(t, n_batches, stage_words) || !t.direct_hop0 || n_hops < 2 || stage_words == 0) return false;
    if (csc->n_major >= ((int64_t)1 << 32) || csc->n_edges >= ((int64_t)1 << 32)) return false; // 32-bit items / slots
    int64_t pitch = p.n_seeds;
    for (int h = 1; h < n_hops; ++h) {
        pitch *= fanout[h - 1];
        int bits = 1;
        while (((int64_t)1 << bits) < pitch) ++bits;
        if (bits >= 32 || n_batches > ((int64_t)1 << (32 - bits))) return false; // batch | index in one word
    }
    return true;
}

// The side stream and events of the staged form (one set per process).  Part p's emit pass runs on the side stream while
// part p + 1's first hop, sort and gather run on the caller's: since round 4 the emit pass is a pure stream (no draws), so
// it overlaps with the gather, which hangs on the vector ALUs.  `mu` is held over the whole enqueue sequence of a launch
// in parts: two host threads launching at once would otherwise re-record an event between the other's record and wait.
struct WinSide {
    hipStream_t stream = nullptr;
    hipEvent_t gathered[WIN_MAX_PARTS] = {}, done = nullptr;
    std::mutex mu;
    int ensure() {
        if (stream) return TG_OK;
        // the emit pass gets the higher dispatch priority (TG_WIN_SIDE_PRIO=0: plain): beside it run the NEXT part's first
        // hop and sort, whose thousands of short workgroups would otherwise take every free slot first
        int least = 0, greatest = 0;
        const int prio = win_env_int("TG_WIN_SIDE_PRIO", 1); // 0 plain, 1 highest, 2 lowest
        if (prio && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && greatest < least)
            TG_HIP(hipStreamCreateWithPriority(&stream, hipStreamNonBlocking, prio == 2 ? least : greatest));
        else
            TG_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        for (auto &e : gathered) TG_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        TG_HIP(hipEventCreateWithFlags(&done, hipEventDisableTiming));
        return TG_OK;
    }
};
static WinSide &win_side() {
    static WinSide s;
    return s;
}

// KFIRST: unroll bound of hop 0's direct kernel (any fan-out up to TG_MAX_FANOUT); KMAX: that of the ordered hops' gather /
// emit / side kernels (<= 16 with one-chunk slots; 12 where the fan-outs allow: ~20 VGPRs less in the gather kernel)
template <int W, int KFIRST, int KMAX, bool REPLACE>
static int win_run_staged(WinParams p, const tg_graph *csc, int64_t n_batches, const int64_t *fanout, int32_t n_hops,
                          hipStream_t stream) {
    const WinTuning t = win_tuning();
    WinStageClock &clk = win_clock();
    const StageBits sb = win_stage_bits(csc);
    p.n_windows = (int32_t)((csc->n_edges >> p.shift) + 1);
    auto pitch_of = [&](int h) {
        int64_t f = p.n_seeds;
        for (int j = 0; j < h; ++j) f *= fanout[j];
        return f;
    };
    auto bits_of = [&](int64_t pitch) {
        int bits = 1;
        while (((int64_t)1 << bits) < pitch) ++bits;
        return bits;
    };
    // two sort levels (ns_homo_stage.inl): window buckets in groups of 8 = coarse buckets, 8 queues of coarse buckets
    p.n_wbuckets = (p.n_buckets + 63) & ~63;
    p.n_buckets = p.n_wbuckets >> 3;
    const size_t tables = win_stage_tables_bytes(p.n_buckets, p.n_windows); // counters + vertex table in LDS
    // parts: only worth it when every part still revisits the lines of a window often (>= 1 024 batches each)
    int parts = std::min(std::max(t.stage_parts, 1), WIN_MAX_PARTS);
    while (parts > 1 && n_batches / parts < std::max(t.stage_part_min_batches, 1)) --parts;
    if (clk.enabled) parts = 1; // stage timing: one stream, one part, so that the events bracket single kernels
    // split: rows / cols / edge_index leave on the side stream beside the sort and the gather (one part: the emit pass that
    // follows is short, and it must wait for the side pass, which reads the state it updates)
    const bool split = t.stage_split && !clk.enabled;
    if (split) parts = 1;
    WinSide &side = win_side();
    std::unique_lock<std::mutex> side_lock(side.mu, std::defer_lock);
    if (parts > 1 || split) {
        side_lock.lock();
        const int rc = side.ensure();
        if (rc != TG_OK) return rc;
    }
    uint32_t *const hist0 = p.hist, *const base0 = p.base;
    WinQueues *const queues0 = p.queues;
    void *const sorted0 = p.items_sorted, *const in0 = p.items_in;
    uint32_t *const fine_tot0 = p.fine_tot, *const fine_start0 = p.fine_start, *const fine_tile_off0 = p.fine_tile_off;
    // two hops: the whole chain of a part -- first hop, sort, gather -- runs beside the previous part's emit pass; deeper
    // launches run their earlier hops whole and only the LAST hop in parts
    const bool first_in_parts = parts > 1 && n_hops == 2;

    int fthreads = std::min(t.emit_threads, WIN_EMIT_MAX_THREADS);
    while (fthreads > 64 && win_emit_lds_bytes(p.kmax, fthreads / 64, WIN_STAGE_FIRST_RC) + tables > 64 * 1024)
        fthreads = ((fthreads >> 1) + 63) & ~63;
    auto rows_of = [&](int64_t nb) { // sort workgroups = rows of the histogram = persistent workgroups of the first kernel
        return (int32_t)std::min<int64_t>(std::min(std::max(t.stage_sort_blocks, 1), WIN_MAX_ROWS), nb);
    };
    auto part_tables = [&](int part) {
        p.hist = hist0 + (size_t)part * WIN_MAX_ROWS * (WIN_MAX_BUCKETS / 8);
        p.base = base0 + (size_t)part * (WIN_MAX_BUCKETS + 8);
        p.queues = queues0 + part;
    };
    auto launch_first = [&](int part, int64_t b0, int64_t nb, hipStream_t ps) { // E0: seeds, hop 0 direct, the items of hop 1 + their histogram
        part_tables(part);
        p.b0 = b0;
        p.n_batches = nb;
        p.n_rows = rows_of(nb);
        p.next_pitch = pitch_of(1);
        p.next_idx_bits = bits_of(p.next_pitch);
        hipLaunchKernelGGL((win_stage_first_kernel<KFIRST, REPLACE>), dim3((unsigned)p.n_rows), dim3(fthreads),
                           win_emit_lds_bytes(p.kmax, fthreads / 64, WIN_STAGE_FIRST_RC) + tables, ps, p, (int)fanout[0]);
    };
    // stage_concurrent (two hops, several parts): the parts' WHOLE chains alternate between the caller's stream and the side
    // stream and run beside each other -- what two launches in flight give a caller, inside one launch
    const bool concurrent = first_in_parts && t.stage_concurrent;

    clk.begin(stream);
    hipLaunchKernelGGL(win_vtab_kernel, dim3((p.n_windows + 256) / 256), dim3(256), 0, stream, p, csc->n_major);
    TG_LAUNCH_CHECK();
    if (!first_in_parts) {
        launch_first(0, 0, n_batches, stream);
        TG_LAUNCH_CHECK();
        clk.mark("first", 0, stream);
    }
    if (concurrent) { // fork: the side stream starts behind the vertex table
        TG_HIP(hipEventRecord(side.gathered[0], stream));
        TG_HIP(hipStreamWaitEvent(side.stream, side.gathered[0], 0));
    }
    for (int h = 1; h < n_hops; ++h) {
        const bool next = h + 1 < n_hops;
        const int hparts = next ? 1 : parts; // a hop with a successor keeps one stream: its items feed the next hop's sort
        for (int part = 0; part < hparts; ++part) {
            const int64_t b0 = n_batches * part / hparts, nb = n_batches * (part + 1) / hparts - b0;
            const bool counted = h == 1 && (first_in_parts || hparts == 1); // the first kernel left this part's histogram
            hipStream_t ps = (concurrent && (part & 1)) ? side.stream : stream;
            if (first_in_parts) {
                launch_first(part, b0, nb, ps);
                TG_LAUNCH_CHECK();
            }
            p.hop = h;
            p.k = (int32_t)fanout[h];
            p.item_pitch = pitch_of(h);
            p.idx_bits = bits_of(p.item_pitch);
            p.next_pitch = next ? pitch_of(h + 1) : 0;
            p.next_idx_bits = next ? bits_of(p.next_pitch) : 0;
            int ethreads = t.stage_emit_threads;
            int rc = std::min(std::max(split ? t.stage_split_round_chunks : t.stage_round_chunks, 1), WIN_STAGE_ROUND_CHUNKS_MAX);
            while (win_stage_emit_lds_bytes(W, p.k, ethreads / 64, rc, split) > 64 * 1024) {
                if (rc > 1)
                    rc >>= 1;
                else if (ethreads > 64)
                    ethreads = ((ethreads >> 1) + 63) & ~63;
                else
                    return tg::fail(TG_ERR_INVALID, "tg_ns_homo_batched_ws: the staged emit kernel does not fit the LDS");
            }
            const size_t elds = win_stage_emit_lds_bytes(W, p.k, ethreads / 64, rc, split);
            int gthreads = t.stage_gather_threads;
            const size_t per_wave = (size_t)(64 * (W + 1) + 64) * sizeof(uint32_t);
            while (gthreads > 64 && (size_t)(gthreads / 64) * per_wave > 128 * 1024) gthreads = ((gthreads >> 1) + 63) & ~63;
            if ((size_t)(gthreads / 64) * per_wave > 64 * 1024) { // gfx950: 160 KB of LDS per CU, a workgroup may take it all when asked
                static bool raised = false; // per instantiation of this function template = per gather kernel
                if (!raised) {
                    TG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&win_stage_gather_kernel<W, KMAX, REPLACE>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
                    raised = true;
                }
            }
            const int gblocks = (std::max(t.stage_gather_blocks, 8) + 7) & ~7;
            part_tables(part);
            p.b0 = b0;
            p.n_batches = nb;
            p.n_rows = rows_of(nb);
            p.items_in = in0;
            p.items_sorted = static_cast<WinItem8 *>(sorted0) + p.b0 * p.item_pitch; // the part's own range of the two flat arrays
            p.items_fine = static_cast<WinItem8 *>(in0) + p.b0 * p.item_pitch; // (the strided items are dead after level 1)
            if (!counted) {
                hipLaunchKernelGGL(win_hist8_kernel, dim3((unsigned)p.n_rows), dim3(WIN_PART_THREADS), tables, ps, p);
                TG_LAUNCH_CHECK();
                clk.mark("hist", h, stream);
            }
            hipLaunchKernelGGL(win_colscan_kernel, dim3((p.n_buckets + 63) / 64), dim3(64 * WIN_SCAN_GROUPS), 0, ps, p,
                               p.n_rows);
            TG_LAUNCH_CHECK();
            hipLaunchKernelGGL(win_basescan_kernel, dim3(1), dim3(1024), 0, ps, p);
            TG_LAUNCH_CHECK();
            clk.mark("scans", h, stream);
            {
                static const int tiled = win_env_int("TG_WIN_SCATTER_TILED", 1);
                if (tiled && p.n_buckets <= WIN_PART_THREADS && win_scatter8_tiled_lds(p.n_buckets, p.n_windows) <= 64 * 1024)
                    hipLaunchKernelGGL(win_scatter8_tiled_kernel, dim3((unsigned)p.n_rows), dim3(WIN_PART_THREADS),
                                       win_scatter8_tiled_lds(p.n_buckets, p.n_windows), ps, p);
                else
                    hipLaunchKernelGGL(win_scatter8_kernel, dim3((unsigned)p.n_rows), dim3(WIN_PART_THREADS), tables, ps, p);
            }
            TG_LAUNCH_CHECK();
            clk.mark("scatter", h, stream);
            if (t.stage_fine) {
                const int64_t worst = p.n_batches * p.item_pitch; // the item count itself lives on the device
                const size_t vt = (size_t)(p.n_windows + 1 + p.n_buckets + 1) * sizeof(uint32_t); // vertex table + coarse starts
                p.fine_sub_bits = std::min(std::max(t.stage_fine_sub_bits, 4), WIN_FINE_SUB_BITS_MAX);
                const int per_coarse = 8 << p.fine_sub_bits;
                const size_t keys = (size_t)(p.n_buckets + 1) * per_coarse;
                p.fine_tot = fine_tot0 + (size_t)part * (WIN_MAX_BUCKETS / 8 + 8) * WIN_FINE_PER_COARSE_MAX;
                p.fine_start = fine_start0 + (size_t)part * (WIN_MAX_BUCKETS / 8 + 8) * WIN_FINE_PER_COARSE_MAX;
                p.fine_tile_off = fine_tile_off0 + ((size_t)(p.b0 * p.item_pitch) / WIN_FINE_TILE + (size_t)part) * WIN_FINE_BINS_MAX;
                p.fine_rowtot = p.fine_start + (size_t)(WIN_MAX_BUCKETS / 8 + 1) * WIN_FINE_PER_COARSE_MAX; // the slack rows behind the starts
                const unsigned tiles = (unsigned)std::max<int64_t>(1, std::min<int64_t>((worst + WIN_FINE_TILE - 1) / WIN_FINE_TILE, t.stage_fine_blocks));
                TG_HIP(hipMemsetAsync(p.fine_tot, 0, keys * sizeof(uint32_t), ps));
                hipLaunchKernelGGL(win_sort_fine_kernel<false>, dim3(tiles), dim3(WIN_FINE_THREADS), vt, ps, p);
                TG_LAUNCH_CHECK();
                hipLaunchKernelGGL(win_fine_rowsum_kernel, dim3((p.n_buckets + 16) / 16), dim3(1024), 0, ps, p);
                TG_LAUNCH_CHECK();
                hipLaunchKernelGGL(win_fine_starts_kernel, dim3((p.n_buckets + 16) / 16), dim3(1024), 0, ps, p);
                TG_LAUNCH_CHECK();
                hipLaunchKernelGGL(win_sort_fine_kernel<true>, dim3(tiles), dim3(WIN_FINE_THREADS), vt, ps, p);
                TG_LAUNCH_CHECK();
                clk.mark("fine", h, stream);
            } else
                p.items_fine = p.items_sorted; // the gather kernel reads the coarse-sorted items
            if (split) TG_HIP(hipEventRecord(side.gathered[h & (WIN_MAX_PARTS - 1)], ps)); // fork point: before the gather
            hipLaunchKernelGGL((win_stage_gather_kernel<W, KMAX, REPLACE>), dim3(gblocks), dim3(gthreads),
                               (size_t)(gthreads / 64) * per_wave, ps, p, sb);
            TG_LAUNCH_CHECK();
            if (split) { // fork: the side pass starts beside the GATHER (beside the sort it slowed the sort's passes 2.5x: they
                         // hang on the latency of their few HBM accesses, which the side pass's streams stretch)
                int sthreads = std::min(t.emit_threads, WIN_EMIT_MAX_THREADS);
                while (sthreads > 64 && win_emit_lds_bytes(p.kmax, sthreads / 64, WIN_STAGE_SIDE_RC) > 64 * 1024) sthreads = ((sthreads >> 1) + 63) & ~63;
                TG_HIP(hipStreamWaitEvent(side.stream, side.gathered[h & (WIN_MAX_PARTS - 1)], 0));
                hipLaunchKernelGGL((win_stage_side_kernel<KMAX, REPLACE>), dim3((unsigned)p.n_batches), dim3(sthreads),
                                   win_emit_lds_bytes(p.kmax, sthreads / 64, WIN_STAGE_SIDE_RC), side.stream, p);
                TG_LAUNCH_CHECK();
                TG_HIP(hipEventRecord(side.done, side.stream));
            }
            clk.mark("gather", h, stream);
            // the emit pass of the last hop has nothing after it to wait for: it goes to the side stream, behind this
            // part's gather, and the next part's chain starts beside it
            hipStream_t es = ps;
            if (split) TG_HIP(hipStreamWaitEvent(ps, side.done, 0)); // join: the emit pass updates the state the side pass reads
            if (hparts > 1 && !concurrent) {
                TG_HIP(hipEventRecord(side.gathered[part], stream));
                TG_HIP(hipStreamWaitEvent(side.stream, side.gathered[part], 0));
                es = side.stream;
            }
            if (split) {
                if (next)
                    hipLaunchKernelGGL((win_stage_emit_kernel<W, KMAX, true, true>), dim3((unsigned)p.n_batches),
                                       dim3(ethreads), elds, es, p, sb, rc);
                else
                    hipLaunchKernelGGL((win_stage_emit_kernel<W, KMAX, false, true>), dim3((unsigned)p.n_batches),
                                       dim3(ethreads), elds, es, p, sb, rc);
            } else if (next)
                hipLaunchKernelGGL((win_stage_emit_kernel<W, KMAX, true, false>), dim3((unsigned)p.n_batches), dim3(ethreads),
                                   elds, es, p, sb, rc);
            else
                hipLaunchKernelGGL((win_stage_emit_kernel<W, KMAX, false, false>), dim3((unsigned)p.n_batches), dim3(ethreads),
                                   elds, es, p, sb, rc);
            TG_LAUNCH_CHECK();
            clk.mark("emit", h, stream);
        }
        if (hparts > 1) { // join: the caller's stream continues after the last emit pass
            TG_HIP(hipEventRecord(side.done, side.stream));
            TG_HIP(hipStreamWaitEvent(stream, side.done, 0));
        }
    }
    return TG_OK;
}

template <bool REPLACE>
static int win_dispatch_staged(const WinParams &p, const tg_graph *csc, int stage_words, int64_t n_batches,
                               const int64_t *fanout, int32_t n_hops, hipStream_t stream) {
    int64_t kord = 0; // the largest fan-out of an ordered hop
    for (int h = 1; h < n_hops; ++h) kord = std::max(kord, fanout[h]);
    if (stage_words == 16 && kord <= 12)
        return fanout[0] <= 16 ? win_run_staged<16, 16, 12, REPLACE>(p, csc, n_batches, fanout, n_hops, stream)
                               : win_run_staged<16, 32, 12, REPLACE>(p, csc, n_batches, fanout, n_hops, stream);
    if (stage_words == 16)
        return fanout[0] <= 16 ? win_run_staged<16, 16, 16, REPLACE>(p, csc, n_batches, fanout, n_hops, stream)
                               : win_run_staged<16, 32, 16, REPLACE>(p, csc, n_batches, fanout, n_hops, stream);
    return fanout[0] <= 16 ? win_run_staged<32, 16, 32, REPLACE>(p, csc, n_batches, fanout, n_hops, stream)
                           : win_run_staged<32, 32, 32, REPLACE>(p, csc, n_batches, fanout, n_hops, stream);
}

// window size: a few hundred KB of the gathered array, at most WIN_MAX_BUCKETS windows
static void win_window_geometry(const tg_graph *csc, int32_t *shift_out, int32_t *n_buckets_out) {
    const int elem = csc->indices32 ? 4 : 8;
    int shift = 0;
    while (((int64_t)elem << shift) < win_tuning().window_bytes) ++shift;
    while (((csc->n_edges >> shift) + 1) > WIN_MAX_BUCKETS - 8) ++shift;
    int32_t nb = (int32_t)((((csc->n_edges >> shift) + 1) + 7) & ~(int64_t)7);
    if (nb < 8) nb = 8;
    *shift_out = shift;
    *n_buckets_out = nb;
}

template <typename Item>
static int win_dispatch(const WinParams &p, bool repl, int64_t n_batches, const int64_t *fanout, int32_t n_hops,
                        hipStream_t stream) {
    if (p.kmax <= 16)
        return repl ? win_run<Item, 16, true>(p, n_batches, fanout, n_hops, stream)
                    : win_run<Item, 16, false>(p, n_batches, fanout, n_hops, stream);
    return repl ? win_run<Item, 32, true>(p, n_batches, fanout, n_hops, stream)
                : win_run<Item, 32, false>(p, n_batches, fanout, n_hops, stream);
}

} // namespace tg

extern "C" int tg_ns_homo_workspace_bytes_for(const tg_graph *csc, int64_t n_batches, int64_t n_seeds, const int64_t *fanout,
                                              int32_t n_hops, int64_t *n_bytes) {
    TG_REQUIRE(n_batches >= 0 && n_seeds >= 0 && n_hops >= 0 && n_hops <= TG_MAX_HOPS && (fanout || n_hops == 0) &&
                   n_bytes,
               "tg_ns_homo_workspace_bytes: bad arguments");
    for (int h = 0; h < n_hops; ++h)
        TG_REQUIRE(fanout[h] >= 1 && fanout[h] <= 255, "tg_ns_homo_workspace_bytes: fanout[%d] outside [1, 255]", h);
    // the staged pipeline's stage slots (2.9 GB for the 16 384-batch bench launch with one-chunk slots) only when that
    // pipeline would be taken (tg_ns_win_tuning.staged) at the time of the query; a launch whose workspace lacks them takes
    // the push form
    const tg::WinLayout L = tg::win_layout(csc, n_batches, n_seeds, fanout, n_hops);
    *n_bytes = (int64_t)(tg::win_staged_wanted(tg::win_tuning(), n_batches, L.stage_words) ? L.total : L.total_push);
    return TG_OK;
}
extern "C" int tg_ns_homo_workspace_bytes(int64_t n_batches, int64_t n_seeds, const int64_t *fanout, int32_t n_hops,
                                          int64_t *n_bytes) {
    return tg_ns_homo_workspace_bytes_for(nullptr, n_batches, n_seeds, fanout, n_hops, n_bytes);
}

// Is the window-ordered form applicable / worth it for this launch?  (Same outputs either way.)
int tg_ns_homo_windowed_applicable(const tg_graph *csc, int64_t n_batches, int64_t n_seeds, const int64_t *fanout,
                                   int32_t n_hops, const tg_ns_config *cfg, const tg_ns_out *out, int32_t mode) {
    if (mode == TG_NS_FORM_FUSED) return 0;
    const int sampler = cfg ? cfg->sampler : TG_SAMPLER_UNIFORM;
    const int filter = cfg ? cfg->filter_mode : TG_FILTER_NONE;
    if (sampler == TG_SAMPLER_WEIGHTED || filter != TG_FILTER_NONE) return 0;
    if (cfg && (cfg->seed_ids || cfg->seed_call_ids)) return 0;
    if (n_hops < 1 || n_batches < 1 || n_seeds < 1) return 0;
    int kmax = 1;
    for (int h = 0; h < n_hops; ++h) kmax = fanout[h] > kmax ? (int)fanout[h] : kmax;
    if (kmax > TG_MAX_FANOUT) return 0;
    const tg::WinLayout L = tg::win_layout(csc, n_batches, n_seeds, fanout, n_hops);
    if (L.max_items >= ((int64_t)1 << 32) || out->cap_nodes >= ((int64_t)1 << 32) || n_batches >= ((int64_t)1 << 32))
        return 0;
    if (mode == TG_NS_FORM_WINDOWED || mode == TG_NS_FORM_WINDOWED_WIDE) return 1;
    // worth it when the launch's gathers revisit lines: many batches against a graph larger than the L2s (RMAT-24,
    // 1 024 seeds per batch: 512 batches 29.7 vs 33.9 G edges/s fused, 1 024 batches 37.6 vs 37.1, 2 048 batches 43.8 vs 41.5)
    return n_batches * n_seeds >= ((int64_t)1 << 21) && csc->n_edges >= ((int64_t)1 << 24);
}

int tg_ns_homo_windowed_launch(const tg_graph *csc, const int64_t *seeds, int64_t n_batches, int64_t n_seeds,
                               const int64_t *fanout, int32_t n_hops, const tg_ns_config *cfg, const tg_rng *rng,
                               const tg_ns_out *out, void *ws, int64_t ws_bytes, int32_t mode, hipStream_t stream) {
    using namespace tg;
    const WinLayout L = win_layout(csc, n_batches, n_seeds, fanout, n_hops);
    TG_REQUIRE(ws && ws_bytes >= (int64_t)L.total_push, "tg_ns_homo_batched_ws: workspace too small (%lld < %lld bytes)",
               (long long)ws_bytes, (long long)L.total_push);
    TG_REQUIRE(((uintptr_t)ws & 255) == 0, "tg_ns_homo_batched_ws: workspace must be 256-byte aligned");
    WinParams p;
    p.store_align = win_tuning().store_align64 ? 7u : 1u;

    p.ptrs = csc->ptrs;
    p.indices = csc->indices;
    p.indices32 = csc->indices32;
    p.ptrs32 = csc->ptrs32;
    p.seeds = seeds;
    p.n_seeds = n_seeds;
    p.n_hops = n_hops;
    p.hop = 0;
    p.k = 0;
    p.kmax = 1;
    for (int h = 0; h < n_hops; ++h) p.kmax = fanout[h] > p.kmax ? (int32_t)fanout[h] : p.kmax;
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
    unsigned char *w = static_cast<unsigned char *>(ws);
    p.state = reinterpret_cast<WinState *>(w + L.state);
    p.call_keys = reinterpret_cast<CallKey *>(w + L.call_keys);
    p.n_items = reinterpret_cast<unsigned long long *>(w + L.n_items);
    p.queues = reinterpret_cast<WinQueues *>(w + L.queues);
    p.hist = reinterpret_cast<uint32_t *>(w + L.hist);
    p.base = reinterpret_cast<uint32_t *>(w + L.base);
    p.items_in = w + L.items_in;
    p.items_sorted = w + L.items_sorted;
    win_window_geometry(csc, &p.shift, &p.n_buckets);
    // narrow items: 32-bit edge pointers and offsets, (batch, slot) in one word
    int slot_bits = 1;
    while (((int64_t)1 << slot_bits) < out->cap_nodes) ++slot_bits;
    const bool narrow = csc->n_edges < ((int64_t)1 << 32) && out->cap_edges < ((int64_t)1 << 32) && slot_bits < 32 &&
                        n_batches <= ((int64_t)1 << (32 - slot_bits));
    const bool force_wide = mode == TG_NS_FORM_WINDOWED_WIDE;
    p.slot_bits = slot_bits;
    const bool repl = (cfg ? cfg->sampler : TG_SAMPLER_UNIFORM) == TG_SAMPLER_UNIFORM_REPL;
    p.vtab = reinterpret_cast<uint32_t *>(w + L.vtab);
    p.fine_tot = reinterpret_cast<uint32_t *>(w + L.fine_tot);
    p.fine_start = reinterpret_cast<uint32_t *>(w + L.fine_start);
    p.fine_tile_off = reinterpret_cast<uint32_t *>(w + L.fine_tile_off);
    p.item_keys = reinterpret_cast<uint16_t *>(w + L.item_keys);
    p.stage = reinterpret_cast<uint32_t *>(w + L.stage);
    if (narrow && !force_wide && ws_bytes >= (int64_t)L.total &&
        win_staged_applicable(p, win_tuning(), csc, n_batches, fanout, n_hops, L.stage_words))
        return repl ? win_dispatch_staged<true>(p, csc, L.stage_words, n_batches, fanout, n_hops, stream)
                    : win_dispatch_staged<false>(p, csc, L.stage_words, n_batches, fanout, n_hops, stream);
    if (narrow && !force_wide) return win_dispatch<WinItemN>(p, repl, n_batches, fanout, n_hops, stream);
    return win_dispatch<WinItemW>(p, repl, n_batches, fanout, n_hops, stream);
}

extern "C" int tg_ns_win_tuning_get(tg_ns_win_tuning *t) {
    TG_REQUIRE(t, "tg_ns_win_tuning_get: null");
    const tg::WinTuning &w = tg::win_tuning();
    t->window_bytes = w.window_bytes;
    t->gather_blocks = w.gather_blocks;
    t->gather_threads = w.gather_threads;
    t->emit_threads = w.emit_threads;
    t->direct_hop0 = w.direct_hop0;
    t->fuse_first_hops = w.fuse_first_hops;
    t->fold_hist = w.fold_hist;
    t->emit_blocks = w.emit_blocks;
    t->staged = w.staged;
    t->stage_round_chunks = w.stage_round_chunks;
    t->stage_gather_threads = w.stage_gather_threads;
    t->stage_gather_blocks = w.stage_gather_blocks;
    t->stage_emit_threads = w.stage_emit_threads;
    t->stage_parts = w.stage_parts;
    t->stage_part_min_batches = w.stage_part_min_batches;
    t->stage_sort_blocks = w.stage_sort_blocks;
    t->stage_fine = w.stage_fine;
    t->stage_concurrent = w.stage_concurrent;
    t->stage_split = w.stage_split;
    t->stage_split_round_chunks = w.stage_split_round_chunks;
    t->store_align64 = w.store_align64;
    t->stage_fine_sub_bits = w.stage_fine_sub_bits;
    t->stage_fine_blocks = w.stage_fine_blocks;
    return TG_OK;
}

extern "C" int tg_ns_win_tuning_set(const tg_ns_win_tuning *t) {
    TG_REQUIRE(t, "tg_ns_win_tuning_set: null");
    TG_REQUIRE(t->window_bytes == 0 || (t->window_bytes >= 64 && t->window_bytes <= ((int64_t)1 << 32)),
               "tg_ns_win_tuning_set: window_bytes outside [64, 2^32]");
    TG_REQUIRE(t->gather_threads == 0 || (t->gather_threads >= 64 && t->gather_threads <= 1024 && t->gather_threads % 64 == 0),
               "tg_ns_win_tuning_set: gather_threads must be a multiple of 64 in [64, 1024]");
    TG_REQUIRE(t->emit_threads == 0 || (t->emit_threads >= 64 && t->emit_threads <= 1024 && t->emit_threads % 64 == 0),
               "tg_ns_win_tuning_set: emit_threads must be a multiple of 64 in [64, 1024]");
    TG_REQUIRE(t->gather_blocks >= 0 && t->gather_blocks <= 65536, "tg_ns_win_tuning_set: gather_blocks outside [0, 65536]");
    tg::WinTuning &w = tg::win_tuning();
    if (t->window_bytes) w.window_bytes = t->window_bytes;
    if (t->gather_blocks) w.gather_blocks = t->gather_blocks;
    if (t->gather_threads) w.gather_threads = t->gather_threads;
    if (t->emit_threads) w.emit_threads = t->emit_threads;
    if (t->direct_hop0 >= 0) w.direct_hop0 = t->direct_hop0 != 0;
    if (t->fuse_first_hops >= 0) w.fuse_first_hops = t->fuse_first_hops != 0;
    if (t->fold_hist >= 0) w.fold_hist = t->fold_hist != 0;
    if (t->emit_blocks > 0) w.emit_blocks = t->emit_blocks;
    if (t->staged >= 0) w.staged = t->staged > 2 ? 1 : t->staged;
    if (t->stage_round_chunks > 0) w.stage_round_chunks = t->stage_round_chunks;
    if (t->stage_gather_threads >= 64 && t->stage_gather_threads <= 1024) w.stage_gather_threads = t->stage_gather_threads & ~63;
    if (t->stage_gather_blocks > 0) w.stage_gather_blocks = t->stage_gather_blocks;
    if (t->stage_emit_threads >= 64 && t->stage_emit_threads <= 1024) w.stage_emit_threads = t->stage_emit_threads & ~63;
    if (t->stage_parts > 0) w.stage_parts = t->stage_parts;
    if (t->stage_part_min_batches > 0) w.stage_part_min_batches = t->stage_part_min_batches;
    if (t->stage_sort_blocks > 0) w.stage_sort_blocks = t->stage_sort_blocks;
    if (t->stage_fine >= 0) w.stage_fine = t->stage_fine != 0;
    if (t->stage_concurrent >= 0) w.stage_concurrent = t->stage_concurrent != 0;
    if (t->stage_split >= 0) w.stage_split = t->stage_split != 0;
    if (t->stage_split_round_chunks > 0) w.stage_split_round_chunks = t->stage_split_round_chunks;
    if (t->store_align64 >= 0) w.store_align64 = t->store_align64 != 0;
    if (t->stage_fine_sub_bits > 0) w.stage_fine_sub_bits = std::min(std::max(t->stage_fine_sub_bits, 4), tg::WIN_FINE_SUB_BITS_MAX);
    if (t->stage_fine_blocks > 0) w.stage_fine_blocks = std::min((t->stage_fine_blocks + 7) & ~7, 8192);
    return TG_OK;
}

extern "C" int tg_ns_win_stage_timing(int32_t enable) {
    tg::win_clock().enabled = enable != 0;
    return TG_OK;
}

extern "C" int tg_ns_win_stage_times(float *ms, char *names, int32_t cap, int32_t *n) {
    TG_REQUIRE(n && (cap == 0 || (ms && names)), "tg_ns_win_stage_times: null arguments");
    tg::WinStageClock &c = tg::win_clock();
    *n = c.n;
    if (c.n == 0) return TG_OK;
    TG_HIP(hipEventSynchronize(c.ev[c.n]));
    for (int i = 0; i < c.n && i < cap; ++i) {
        TG_HIP(hipEventElapsedTime(&ms[i], c.ev[i], c.ev[i + 1]));
        snprintf(names + (size_t)i * 24, 24, "%s", c.name[i]);
    }
    return TG_OK;
}

extern "C" int tg_ns_homo_batched_form(const tg_graph *csc, int64_t n_batches, int64_t n_seeds, const int64_t *fanout,
                                       int32_t n_hops, const tg_ns_config *cfg, const tg_ns_out *out,
                                       int64_t workspace_bytes, int32_t mode, int32_t *form, int32_t *n_windows) {
    TG_REQUIRE(csc && out && form && (fanout || n_hops == 0), "tg_ns_homo_batched_form: null arguments");
    TG_REQUIRE(n_hops >= 0 && n_hops <= TG_MAX_HOPS, "tg_ns_homo_batched_form: n_hops %d outside [0, %d]", n_hops, TG_MAX_HOPS);
    *form = TG_NS_FORM_FUSED;
    if (n_windows) *n_windows = 0;
    const int sampler = cfg ? cfg->sampler : TG_SAMPLER_UNIFORM;
    const int filter = cfg ? cfg->filter_mode : TG_FILTER_NONE;
    if (sampler == TG_SAMPLER_WEIGHTED || filter != TG_FILTER_NONE) return TG_OK; // the scanning kernels: neither form
    if (workspace_bytes <= 0 || !tg_ns_homo_windowed_applicable(csc, n_batches, n_seeds, fanout, n_hops, cfg, out, mode))
        return TG_OK;
    if ((int64_t)tg::win_layout(csc, n_batches, n_seeds, fanout, n_hops).total_push > workspace_bytes) return TG_OK; // the launch would refuse
    int32_t shift = 0, nb = 0;
    tg::win_window_geometry(csc, &shift, &nb);
    int slot_bits = 1;
    while (((int64_t)1 << slot_bits) < out->cap_nodes) ++slot_bits;
    const bool narrow = csc->n_edges < ((int64_t)1 << 32) && out->cap_edges < ((int64_t)1 << 32) && slot_bits < 32 &&
                        n_batches <= ((int64_t)1 << (32 - slot_bits));
    *form = (narrow && mode != TG_NS_FORM_WINDOWED_WIDE) ? TG_NS_FORM_WINDOWED : TG_NS_FORM_WINDOWED_WIDE;
    if (n_windows) *n_windows = nb;
    return TG_OK;
}

extern "C" int tg_ns_homo_batched_pipeline(const tg_graph *csc, int64_t n_batches, int64_t n_seeds, const int64_t *fanout,
                                           int32_t n_hops, const tg_ns_config *cfg, const tg_ns_out *out,
                                           int64_t workspace_bytes, int32_t mode, int32_t *staged) {
    TG_REQUIRE(staged, "tg_ns_homo_batched_pipeline: null");
    *staged = 0;
    int32_t form = 0, n_win = 0;
    const int rc = tg_ns_homo_batched_form(csc, n_batches, n_seeds, fanout, n_hops, cfg, out, workspace_bytes, mode, &form, &n_win);
    if (rc != TG_OK || form != TG_NS_FORM_WINDOWED) return rc;
    const tg::WinLayout L = tg::win_layout(csc, n_batches, n_seeds, fanout, n_hops);
    tg::WinParams p{};
    p.n_seeds = n_seeds;
    *staged = workspace_bytes >= (int64_t)L.total &&
              tg::win_staged_applicable(p, tg::win_tuning(), csc, n_batches, fanout, n_hops, L.stage_words);
    return TG_OK;
}
