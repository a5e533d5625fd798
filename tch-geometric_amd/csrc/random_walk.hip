// Random walks on gfx950 -- replace src/algo/random_walk.rs:10-158 (reference).
//
// rw_node2vec_kernel: one LANE per walker (1 M walkers x 80 dependent steps is
//   latency-bound; parallelism comes from walkers).  Per step: ptrs pair,
//   one random neighbour gather, optional has_edge binary search (only when
//   p,q make the three acceptance probabilities differ).  A walker's row is
//   staged 16 steps at a time in LDS and flushed by the whole wave so that each
//   walker's 128 contiguous bytes leave as one full line instead of 16 scattered
//   8-byte stores.  Algorithmic bytes per executed step: 16 + 8 read, 8 written.
//
// rw_tempo_kernel: one WAVEFRONT per walker.  Every step must inspect the whole
//   row (timestamps), so the wave streams indices/edge_ts coalesced, ranks the
//   admissible neighbours with ballot + popcount, and resolves the one-slot
//   reservoir (sampling.rs:17-22 with k = 1: candidate m >= 1 replaces with
//   probability 1/m, candidate 1 always does) from one addressed draw per
//   candidate.  HBM-bound on 16 B per inspected edge.
#include "row_stream.h"
#include "tg_device.h"
#include "tg_host.h"
#include "tg_map.h"

namespace tg {

constexpr int RW_STAGE = 16; // steps staged per walker between flushes

// CSR accessors: the optional u32 shadows (tg_graph.ptrs32 / indices32) hold the same values in half the bytes --
// twice the entries per gathered line, and the whole offset table of RMAT-24 (67 MB) stays in the Infinity Cache
struct CsrView {
    const int64_t *ptrs, *indices;
    const uint32_t *ptrs32, *indices32;
    const uint64_t *edge_set; // optional hash set of the edges (tg_edge_set_build) and its slot mask
    uint64_t edge_mask;
    __device__ __forceinline__ int64_t ptr(int64_t i) const { return ptrs32 ? (int64_t)ptrs32[i] : ptrs[i]; }
    __device__ __forceinline__ int64_t idx(int64_t e) const { return indices32 ? (int64_t)indices32[e] : indices[e]; }
};

// ---- the edge set: has_edge as a hash probe --------------------------------------------------------------------------
// graph.rs:80-83 answers has_edge(x, y) by a binary search of row x: log2(deg) DEPENDENT random line requests, ~13 on
// RMAT-24, and node2vec with p != q asks once per proposal -- the walk then sits on the chip's random-request ceiling.
// The set holds every edge once as the key x << 32 | y in an open-addressing table (linear probing, load <= 1/2, 8-byte
// slots: a probe sequence usually stays inside one 128-byte line), so the same question is ONE line request, with the
// same answer (a multi-edge is one key; ids must be < 2^32 - 1).
constexpr uint64_t EDGE_SET_EMPTY = ~0ull;
__device__ __forceinline__ uint64_t edge_key_hash(uint64_t k) {
    k ^= k >> 33;
    k *= 0xff51afd7ed558ccdull;
    k ^= k >> 29;
    return k;
}
__device__ __forceinline__ bool edge_set_has(const uint64_t *__restrict__ slots, uint64_t mask, int64_t x, int64_t y) {
    const uint64_t key = ((uint64_t)x << 32) | (uint64_t)y;
    for (uint64_t s = edge_key_hash(key) & mask;; s = (s + 1) & mask) {
        const uint64_t v = slots[s];
        if (v == key) return true;
        if (v == EDGE_SET_EMPTY) return false;
    }
}
__global__ void edge_set_clear_kernel(uint64_t *slots, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        slots[i] = EDGE_SET_EMPTY;
}
// the wavefronts walk the edge array in segments of 2 048 edges (coalesced); a segment's rows lie between the rows of its
// first and last edge, found once per segment, so an edge's row is a short search in cached offsets
constexpr int64_t EDGE_SET_SEGMENT = 2048;
__device__ __forceinline__ int64_t row_of_edge(const int64_t *__restrict__ ptrs, int64_t lo, int64_t hi, int64_t e) {
    while (lo < hi) { // last row r in [lo, hi] with ptrs[r] <= e
        const int64_t mid = lo + ((hi - lo + 1) >> 1);
        if (ptrs[mid] <= e)
            lo = mid;
        else
            hi = mid - 1;
    }
    return lo;
}
__global__ void edge_set_insert_kernel(const int64_t *__restrict__ ptrs, const int64_t *__restrict__ indices, int64_t n_major,
                                       int64_t n_edges, uint64_t *slots, uint64_t mask) {
    const int lane = threadIdx.x & 63;
    const int64_t n_seg = (n_edges + EDGE_SET_SEGMENT - 1) / EDGE_SET_SEGMENT;
    for (int64_t sgm = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; sgm < n_seg;
         sgm += ((int64_t)gridDim.x * blockDim.x) >> 6) {
        const int64_t e0 = sgm * EDGE_SET_SEGMENT, e1 = min(n_edges, e0 + EDGE_SET_SEGMENT);
        const int64_t r0 = row_of_edge(ptrs, 0, n_major - 1, e0), r1 = row_of_edge(ptrs, r0, n_major - 1, e1 - 1);
        for (int64_t e = e0 + lane; e < e1; e += 64) {
            const int64_t r = row_of_edge(ptrs, r0, r1, e);
            const uint64_t key = ((uint64_t)r << 32) | (uint64_t)indices[e];
            for (uint64_t s = edge_key_hash(key) & mask;; s = (s + 1) & mask) {
                const uint64_t old = atomicCAS(reinterpret_cast<unsigned long long *>(&slots[s]),
                                               (unsigned long long)EDGE_SET_EMPTY, (unsigned long long)key);
                if (old == EDGE_SET_EMPTY || old == key) break;
            }
        }
    }
}

__device__ __forceinline__ bool has_edge(const CsrView &g, int64_t x, int64_t y) { // graph.rs:80-83
    if (g.edge_set) return edge_set_has(g.edge_set, g.edge_mask, x, y);
    int64_t lo = g.ptr(x), hi = g.ptr(x + 1);
    while (lo < hi) {
        const int64_t mid = lo + ((hi - lo) >> 1);
        const int64_t v = g.idx(mid);
        if (v == y) return true;
        if (v < y)
            lo = mid + 1;
        else
            hi = mid;
    }
    return false;
}

// StageT = uint32_t for the has_edge variants when the vertex ids fit (0xffffffff stands for -1): half the LDS per
// workgroup, twice the resident wavefronts -- those variants are bound by the latency of their dependent loads (p != q:
// 11.8 -> 9.1 ms; p = q = 1 is not: 1.16 ms with int64 staging, 1.27 with u32, so it keeps int64)
template <typename StageT>
__global__ void rw_node2vec_kernel(const CsrView g, const int64_t *__restrict__ start, int64_t n, int64_t walk_length, float prob0,
                                   float prob1, float prob2, uint64_t seed, uint64_t call_id, int64_t *walks) {
    __shared__ StageT stage_all[4][64 * (RW_STAGE + 1)]; // [wave][walker * 17 + step]: odd pitch spreads LDS banks
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    StageT *stage = stage_all[wave];
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t wave_first = t - lane;
    const bool live = t < n;
    const CallKey ck = call_key(seed, call_id, TAG_RW);
    const int64_t L = walk_length + 1;
    const bool always_accept = prob0 >= 1.0f && prob1 >= 1.0f && prob2 >= 1.0f; // r < 1 always holds

    int64_t prev = -1, cur = live ? start[t] : -1;
    bool dead = !live;
    // column 0 is the start node; columns 1..walk_length are steps 0..walk_length-1
    for (int64_t c0 = 0; c0 < L; c0 += RW_STAGE) {
        const int ncols = (int)min((int64_t)RW_STAGE, L - c0);
        for (int j = 0; j < ncols; ++j) {
            const int64_t col = c0 + j;
            int64_t val = -1;
            if (col == 0) {
                val = cur;
            } else if (!dead) {
                const int64_t l = col - 1;
                const int64_t b = g.ptr(cur), e = g.ptr(cur + 1);
                if (e <= b) { // random_walk.rs:45-47
                    dead = true;
                } else {
                    const uint64_t deg = (uint64_t)(e - b);
                    int64_t next;
                    for (uint32_t attempt = 0;; ++attempt) { // :52-66
                        const Draw d = draw(ck, (uint64_t)t, (uint32_t)l, attempt);
                        next = g.idx(b + (int64_t)bounded64(d.a(), deg));
                        if (always_accept) break;
                        const float r = u32_to_f32_01(d.w[2]);
                        if (next == prev) {
                            if (r < prob0) break;
                        } else if (prev >= 0 && has_edge(g, next, prev)) {
                            if (r < prob1) break;
                        } else if (r < prob2) {
                            break;
                        }
                    }
                    prev = cur;
                    cur = next;
                    val = cur;
                }
            }
            stage[lane * (RW_STAGE + 1) + j] = (StageT)val; // -1 -> all ones
        }
        wave_lds_handoff();
        // flush: walker w of this wave owns ncols contiguous int64 at walks[(wave_first+w)*L + c0 ..]
        const int total = 64 * ncols;
        for (int q = lane; q < total; q += 64) {
            const int w = q / ncols, j = q - w * ncols;
            const int64_t tw = wave_first + w;
            if (tw < n) {
                const StageT v = stage[w * (RW_STAGE + 1) + j];
                walks[tw * L + c0 + j] = v == (StageT)-1 ? (int64_t)-1 : (int64_t)v;
            }
        }
        wave_lds_handoff();
    }
}

// one wavefront per walker; LDS keeps the walk so far (node, ts) for restarts
__global__ void rw_tempo_kernel(const int64_t *__restrict__ ptrs, const int64_t *__restrict__ indices,
                                const int64_t *__restrict__ node_ts, const int64_t *__restrict__ edge_ts,
                                const int64_t *__restrict__ start, const int64_t *__restrict__ start_ts, int64_t n,
                                int64_t L, int64_t win0, int64_t win1, uint64_t seed, uint64_t call_id,
                                int64_t *walks, int64_t *walks_ts) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
    int64_t *hist = reinterpret_cast<int64_t *>(smem) + (size_t)wave * 2 * L; // [L] nodes, [L] timestamps
    const int64_t i = (int64_t)blockIdx.x * n_waves + wave;
    if (i >= n) return;
    const CallKey ck = call_key(seed, call_id, TAG_RW_TEMPO);
    const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));

    int64_t cur = start[i];
    const int64_t it = start_ts[i];
    const int64_t wlo = it + win0, whi = it + win1; // half open, random_walk.rs:111
    if (lane == 0 && L > 0) {
        hist[0] = cur;
        hist[L] = it;
    }
    for (int64_t l = 0; l < L - 1; ++l) { // :117
        const uint64_t step_id = (uint64_t)i * (uint64_t)L + (uint64_t)l;
        const int64_t b = ptrs[cur], e = ptrs[cur + 1];
        // one-slot reservoir over the candidates in row order (sampling.rs:12-24 with k = 1), philox-mode: ONE draw per chunk
        // of 64 raw row positions (the CPU checker's orc_reservoir_one_chunked states the law).  Candidates of rank >= 1 are
        // eligible; a chunk with m of them, after `seen` earlier ones, takes the slot with probability m / (seen + m) and
        // gives it to one of its m.  The draws of 64 consecutive chunks are computed TOGETHER, lane l the block of chunk
        // 64 g + l, when the row first needs one of group g: a Philox block per 4 096 row positions and wavefront instead
        // of one per 64 (a block per chunk on the scalar unit costs what the per-candidate blocks cost on the vector unit:
        // both issue once per chunk -- measured 101 ms against 75).
        uint32_t n_pass = 0, seen = 0;
        int64_t best_v = -1, best_t = -1; // the slot's candidate (edge position, time), held by one lane
        bool have_best = false;
        int64_t first_v = -1, first_t = -1; // candidate of rank 0 (held by one lane)
        bool has_first = false;
        uint32_t group = 0xffffffffu; // the group of 64 chunks whose draws the lanes hold
        Draw gd;
        gd.w[0] = gd.w[1] = gd.w[2] = gd.w[3] = 0u;
        auto visit = [&](int64_t v, bool valid, int64_t ts) { // v: edge position
            const bool ok = valid && ((ts == -1 || it == -1) || (wlo <= ts && ts < whi)); // :129-138
            const uint64_t mask = __ballot(ok);
            const uint32_t rank = n_pass + (uint32_t)__popcll(mask & lt_mask);
            if (ok && rank == 0) {
                first_v = v;
                first_t = ts;
                has_first = true;
            }
            const bool eligible = ok && rank >= 1;
            const uint64_t emask = __ballot(eligible);
            const uint32_t m = (uint32_t)__popcll(emask);
            if (m > 0) { // uniform
                const uint32_t chunk = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((v - b) >> 6));
                if ((chunk >> 6) != group) { // uniform
                    group = chunk >> 6;
                    gd = draw(ck, step_id, (group << 6) + (uint32_t)lane, D1_CHUNK);
                }
                const int from = (int)(chunk & 63u);
                const uint64_t da = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)gd.w[0], from) |
                                    ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)gd.w[1], from) << 32);
                const uint64_t db = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)gd.w[2], from) |
                                    ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)gd.w[3], from) << 32);
                if (seen == 0 || bounded64(da, (uint64_t)(seen + m)) < (uint64_t)m) {
                    const uint32_t r = (uint32_t)bounded64(db, (uint64_t)m);
                    have_best = eligible && (uint32_t)__popcll(emask & lt_mask) == r;
                    if (have_best) {
                        best_v = v;
                        best_t = ts;
                    }
                }
                seen += m;
            }
            n_pass += (uint32_t)__popcll(mask);
        };
        // loads in flight per lane: few -- most rows are short and every load of a round is issued whether the row reaches
        // it or not (RMAT-24, 1 M walkers x 20 steps: 97.6 / 85.1 / 74.5 / 73.8 ms with 8 / 4 / 2 / 1 chunks per round)
        if (e - b <= 128)
            stream_row_ts<1>(indices, edge_ts, node_ts, b, e, lane, visit);
        else
            stream_row_ts<2>(indices, edge_ts, node_ts, b, e, lane, visit);
        int64_t next, next_t;
        if (n_pass == 0) { // :144-148 restart from an earlier position of this walk
            const Draw d = draw(ck, step_id, 0u, D1_RESTART);
            const int64_t rr = (int64_t)bounded64(d.a(), (uint64_t)(l + 1));
            wave_lds_handoff();
            next = hist[rr];
            next_t = hist[L + rr];
        } else {
            const uint64_t owner = (seen > 0) ? __ballot(have_best) : __ballot(has_first);
            const int src = __ffsll((long long)owner) - 1;
            next = indices[__shfl((seen > 0) ? best_v : first_v, src, 64)]; // the one neighbour id the step needs
            next_t = __shfl((seen > 0) ? best_t : first_t, src, 64);
        }
        cur = next; // :150-153
        if (lane == 0) {
            hist[l + 1] = next;
            hist[L + l + 1] = next_t;
        }
    }
    wave_lds_handoff();
    for (int64_t c = lane; c < L; c += 64) {
        walks[i * L + c] = hist[c];
        walks_ts[i * L + c] = hist[L + c];
    }
}

} // namespace tg

static int64_t edge_set_slots(int64_t n_edges) {
    int64_t cap = 64;
    while (cap < 2 * n_edges) cap <<= 1;
    return cap;
}
extern "C" int tg_edge_set_bytes(const tg_graph *csr, int64_t *bytes) {
    TG_REQUIRE(csr && bytes && csr->n_edges >= 0 && csr->n_major >= 0, "tg_edge_set_bytes: bad arguments");
    TG_REQUIRE(csr->n_major < (int64_t)0xffffffff, "tg_edge_set_bytes: ids of %lld vertices do not fit the 32-bit halves of a key",
               (long long)csr->n_major);
    *bytes = 8 * edge_set_slots(csr->n_edges);
    return TG_OK;
}
extern "C" int tg_edge_set_build(const tg_graph *csr, void *edge_set, int64_t bytes, void *stream_) {
    TG_REQUIRE(csr && csr->ptrs && (csr->indices || csr->n_edges == 0) && edge_set, "tg_edge_set_build: null argument");
    TG_REQUIRE(csr->n_major < (int64_t)0xffffffff, "tg_edge_set_build: ids of %lld vertices do not fit a key", (long long)csr->n_major);
    const int64_t cap = edge_set_slots(csr->n_edges);
    TG_REQUIRE(bytes >= 8 * cap, "tg_edge_set_build: %lld bytes given, %lld needed", (long long)bytes, (long long)(8 * cap));
    hipStream_t stream = (hipStream_t)stream_;
    uint64_t *slots = reinterpret_cast<uint64_t *>(edge_set);
    hipLaunchKernelGGL(tg::edge_set_clear_kernel, dim3(tg::grid_1d(cap)), dim3(256), 0, stream, slots, cap);
    if (csr->n_edges > 0 && csr->n_major > 0) {
        const int64_t n_seg = (csr->n_edges + tg::EDGE_SET_SEGMENT - 1) / tg::EDGE_SET_SEGMENT;
        hipLaunchKernelGGL(tg::edge_set_insert_kernel, dim3(tg::grid_1d(n_seg * 64)), dim3(256), 0, stream, csr->ptrs, csr->indices,
                           csr->n_major, csr->n_edges, slots, (uint64_t)(cap - 1));
    }
    TG_LAUNCH_CHECK();
    return TG_OK;
}

extern "C" int tg_random_walk_es(const tg_graph *csr, const void *edge_set, int64_t edge_set_bytes, const int64_t *start, int64_t n,
                                 int64_t walk_length, float p, float q, const tg_rng *rng, int64_t *walks, void *stream) {
    TG_REQUIRE(csr && csr->ptrs && (csr->indices || csr->n_edges == 0), "tg_random_walk: null graph");
    TG_REQUIRE(rng && n >= 0 && walk_length >= 0, "tg_random_walk: bad arguments");
    TG_REQUIRE(p > 0.0f && q > 0.0f, "tg_random_walk: p and q must be positive (random_walk.rs:29-30)");
    uint64_t edge_mask = 0;
    if (edge_set) {
        const int64_t cap = edge_set_slots(csr->n_edges);
        TG_REQUIRE(edge_set_bytes == 8 * cap && csr->n_major < (int64_t)0xffffffff,
                   "tg_random_walk_es: the edge set (%lld bytes) was not built for this graph (%lld bytes)",
                   (long long)edge_set_bytes, (long long)(8 * cap));
        edge_mask = (uint64_t)(cap - 1);
    }
    if (n == 0) return TG_OK;
    TG_REQUIRE(start && walks, "tg_random_walk: null buffers");
    // random_walk.rs:29-36, all in f32
    const float inv_p = 1.0f / p, inv_q = 1.0f / q;
    float max_prob = inv_p;
    if (1.0f >= max_prob) max_prob = 1.0f;
    if (inv_q >= max_prob) max_prob = inv_q;
    const float prob0 = 1.0f / p / max_prob, prob1 = 1.0f / max_prob, prob2 = 1.0f / q / max_prob;
    const unsigned blocks = (unsigned)((n + 255) / 256);
    const tg::CsrView view{csr->ptrs, csr->indices, csr->ptrs32, csr->indices32, reinterpret_cast<const uint64_t *>(edge_set),
                           edge_mask};
    const bool always_accept = prob0 >= 1.0f && prob1 >= 1.0f && prob2 >= 1.0f;
    if (!always_accept && csr->n_major < (int64_t)0xffffffff)
        hipLaunchKernelGGL(tg::rw_node2vec_kernel<uint32_t>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, view, start, n,
                           walk_length, prob0, prob1, prob2, rng->seed, rng->call_id, walks);
    else
        hipLaunchKernelGGL(tg::rw_node2vec_kernel<int64_t>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, view, start, n,
                           walk_length, prob0, prob1, prob2, rng->seed, rng->call_id, walks);
    TG_LAUNCH_CHECK();
    return TG_OK;
}
extern "C" int tg_random_walk(const tg_graph *csr, const int64_t *start, int64_t n, int64_t walk_length, float p, float q,
                              const tg_rng *rng, int64_t *walks, void *stream) {
    return tg_random_walk_es(csr, nullptr, 0, start, n, walk_length, p, q, rng, walks, stream);
}

extern "C" int tg_tempo_random_walk(const tg_graph *csr, const int64_t *node_ts, const int64_t *edge_ts,
                                    const int64_t *start, const int64_t *start_ts, int64_t n, int64_t walk_length,
                                    int64_t win0, int64_t win1, const tg_rng *rng, int64_t *walks, int64_t *walks_ts,
                                    void *stream) {
    TG_REQUIRE(csr && csr->ptrs && (csr->indices || csr->n_edges == 0), "tg_tempo_random_walk: null graph");
    TG_REQUIRE(rng && n >= 0 && walk_length >= 0, "tg_tempo_random_walk: bad arguments");
    if (n == 0 || walk_length == 0) return TG_OK;
    TG_REQUIRE(node_ts && (edge_ts || csr->n_edges == 0) && start && start_ts && walks && walks_ts,
               "tg_tempo_random_walk: null buffers");
    int n_waves = 4;
    while (n_waves > 1 && (size_t)n_waves * 2 * walk_length * sizeof(int64_t) > 48 * 1024) n_waves >>= 1;
    const size_t lds = (size_t)n_waves * 2 * walk_length * sizeof(int64_t);
    if (lds > 64 * 1024)
        return tg::fail(TG_ERR_UNSUPPORTED, "tg_tempo_random_walk: walk_length %lld exceeds the LDS walk buffer",
                        (long long)walk_length);
    const unsigned blocks = (unsigned)((n + n_waves - 1) / n_waves);
    hipLaunchKernelGGL(tg::rw_tempo_kernel, dim3(blocks), dim3(64 * n_waves), lds, (hipStream_t)stream, csr->ptrs,
                       csr->indices, node_ts, edge_ts, start, start_ts, n, walk_length, win0, win1, rng->seed,
                       rng->call_id, walks, walks_ts);
    TG_LAUNCH_CHECK();
    return TG_OK;
}
