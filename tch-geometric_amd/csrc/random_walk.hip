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

namespace tg {

constexpr int RW_STAGE = 16; // steps staged per walker between flushes

// CSR accessors: the optional u32 shadows (tg_graph.ptrs32 / indices32) hold the same values in half the bytes --
// twice the entries per gathered line, and the whole offset table of RMAT-24 (67 MB) stays in the Infinity Cache
struct CsrView {
    const int64_t *ptrs, *indices;
    const uint32_t *ptrs32, *indices32;
    __device__ __forceinline__ int64_t ptr(int64_t i) const { return ptrs32 ? (int64_t)ptrs32[i] : ptrs[i]; }
    __device__ __forceinline__ int64_t idx(int64_t e) const { return indices32 ? (int64_t)indices32[e] : indices[e]; }
};

__device__ __forceinline__ bool has_edge(const CsrView &g, int64_t x, int64_t y) { // graph.rs:80-83
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

__global__ void rw_node2vec_kernel(const CsrView g, const int64_t *__restrict__ start, int64_t n, int64_t walk_length, float prob0,
                                   float prob1, float prob2, uint64_t seed, uint64_t call_id, int64_t *walks) {
    __shared__ int64_t stage_all[4][64 * (RW_STAGE + 1)]; // [wave][walker * 17 + step]: odd pitch spreads LDS banks
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int64_t *stage = stage_all[wave];
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
            stage[lane * (RW_STAGE + 1) + j] = val;
        }
        wave_lds_handoff();
        // flush: walker w of this wave owns ncols contiguous int64 at walks[(wave_first+w)*L + c0 ..]
        const int total = 64 * ncols;
        for (int q = lane; q < total; q += 64) {
            const int w = q / ncols, j = q - w * ncols;
            const int64_t tw = wave_first + w;
            if (tw < n) walks[tw * L + c0 + j] = stage[w * (RW_STAGE + 1) + j];
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
        uint32_t n_pass = 0;
        int64_t best_rank = -1, best_v = -1, best_t = -1; // this lane's last reservoir hit
        int64_t first_v = -1, first_t = -1;               // candidate of rank 0 (held by one lane)
        bool has_first = false;
        stream_row<4>(indices, edge_ts, node_ts, b, e, lane, [&](int64_t, bool valid, int64_t v, int64_t ts) {
            const bool ok = valid && ((ts == -1 || it == -1) || (wlo <= ts && ts < whi)); // :129-138
            const uint64_t mask = __ballot(ok);
            if (ok) {
                const uint32_t rank = n_pass + (uint32_t)__popcll(mask & lt_mask);
                if (rank == 0) {
                    first_v = v;
                    first_t = ts;
                    has_first = true;
                } else { // sampling.rs:19-21 with one slot: j drawn from 0..rank, replaces when j == 0
                    const Draw d = draw(ck, step_id, rank, D1_LITERAL);
                    if (bounded64(d.a(), (uint64_t)rank) == 0) {
                        best_rank = rank;
                        best_v = v;
                        best_t = ts;
                    }
                }
            }
            n_pass += (uint32_t)__popcll(mask);
        });
        int64_t next, next_t;
        if (n_pass == 0) { // :144-148 restart from an earlier position of this walk
            const Draw d = draw(ck, step_id, 0u, D1_RESTART);
            const int64_t rr = (int64_t)bounded64(d.a(), (uint64_t)(l + 1));
            wave_lds_handoff();
            next = hist[rr];
            next_t = hist[L + rr];
        } else {
            int64_t mx = best_rank;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) mx = max(mx, __shfl_xor(mx, off, 64));
            const uint64_t owner = (mx >= 1) ? __ballot(best_rank == mx) : __ballot(has_first);
            const int src = __ffsll((long long)owner) - 1;
            next = __shfl((mx >= 1) ? best_v : first_v, src, 64);
            next_t = __shfl((mx >= 1) ? best_t : first_t, src, 64);
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

extern "C" int tg_random_walk(const tg_graph *csr, const int64_t *start, int64_t n, int64_t walk_length, float p,
                              float q, const tg_rng *rng, int64_t *walks, void *stream) {
    TG_REQUIRE(csr && csr->ptrs && (csr->indices || csr->n_edges == 0), "tg_random_walk: null graph");
    TG_REQUIRE(rng && n >= 0 && walk_length >= 0, "tg_random_walk: bad arguments");
    TG_REQUIRE(p > 0.0f && q > 0.0f, "tg_random_walk: p and q must be positive (random_walk.rs:29-30)");
    if (n == 0) return TG_OK;
    TG_REQUIRE(start && walks, "tg_random_walk: null buffers");
    // random_walk.rs:29-36, all in f32
    const float inv_p = 1.0f / p, inv_q = 1.0f / q;
    float max_prob = inv_p;
    if (1.0f >= max_prob) max_prob = 1.0f;
    if (inv_q >= max_prob) max_prob = inv_q;
    const float prob0 = 1.0f / p / max_prob, prob1 = 1.0f / max_prob, prob2 = 1.0f / q / max_prob;
    const unsigned blocks = (unsigned)((n + 255) / 256);
    const tg::CsrView view{csr->ptrs, csr->indices, csr->ptrs32, csr->indices32};
    hipLaunchKernelGGL(tg::rw_node2vec_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, view, start, n,
                       walk_length, prob0, prob1, prob2, rng->seed, rng->call_id, walks);
    TG_LAUNCH_CHECK();
    return TG_OK;
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
