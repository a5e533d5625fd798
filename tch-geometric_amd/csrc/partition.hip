// Device side of neighbor_sampling_homogenous over a RANGE-PARTITIONED CSC (SURVEY.md 8(e) mode 2, BASELINE cfg5).
//
// The origin rank keeps the ordinary per-batch output slabs of tg_ns_homo_batched (samples | rows | cols |
// edge_index | layer_offsets | counts) plus a per-batch state.  All sizes stay on the device: every kernel reads the
// number of requests it works on from device memory, so with world == 1 the whole call runs without one host
// synchronisation, and with world > 1 the host reads back only what `all_to_all_single` needs as split sizes.  Per hop:
//
//   tg_part_requests   origin: every frontier slot of every batch becomes a 16-byte request (vertex, batch, slot),
//                      grouped by the rank that owns the vertex's column: per-batch requests in frontier order ->
//                      counting sort by owner (per-block histograms, column scan, scatter: no global atomics);
//                      req_pos remembers where each frontier slot's request went.  send_counts[world] on the device.
//   [sizes + requests all-to-all]
//   tg_part_count      owner: sample count of every received request (min(deg, k), or k with replacement), their
//                      exclusive prefix = where each request's samples start in the COMPACT reply, and the number of
//                      reply entries per requesting rank (the split sizes of the reply exchange).
//   tg_part_sample     owner: draws with the REQUESTER's address (tag NS_HOMO, id = slot, call id = the requester's
//                      first call id + batch) -- the draws the replicated-graph sampler uses, so results are equal
//                      bit for bit -- gathers the neighbour ids and writes (neighbour, global edge pointer) pairs,
//                      coalesced (consecutive requests have consecutive replies).
//   [counts + replies all-to-all: counts [m] u32 in request order, pairs compact]
//   tg_part_emit       origin: prefix of the returned counts -> each request's reply offset; one workgroup per batch
//                      copies its frontier's replies in slot order (the reference's output order,
//                      neighbor_sampling.rs:195-218) into the slabs (counts -> LDS scan -> LDS staging -> coalesced
//                      writes) and advances the batch state.
#include <stdlib.h>

#include <algorithm>

#include "ns_tickets.h"
#include "stage_bits.h"
#include "tg_device.h"
#include "tg_host.h"

namespace tg {

constexpr int PART_MAX_WORLD = 64;
constexpr int PART_BLOCKS = 256; // blocks of the bucketing kernels = rows of the owner histogram
constexpr int PART_THREADS = 512;
constexpr int PART_TILE = 4096;
constexpr int PART_CHUNKS_PER_ROUND = 1024;
constexpr int PSCAN_BLOCKS_HOST = 1024; // = PSCAN_BLOCKS (block sums of the device-length prefix sum)

struct PartRequest { // 16 bytes on the wire
    int64_t vertex;
    uint32_t batch; // the requester's batch: call id = its first call id + batch
    uint32_t slot;  // slot of the vertex in that batch's sample list = the draw id
};
static_assert(sizeof(PartRequest) == 16, "request layout");

struct PartState {
    int64_t begin, end, ne, fbase;
};

// workspace (int64 words unless noted); host side: part_layout()
struct PartLayout {
    size_t state, n_req, send_counts, hist, base, req_in, rstate_in, req_pos, poff, scan_tmp, total;
    size_t scan_tmp_bytes;
};

static PartLayout part_layout(int64_t n_batches, int64_t request_cap, int world) {
    PartLayout L;
    size_t at = 0;
    auto take = [&](size_t bytes) {
        const size_t here = at;
        at += (bytes + 255) & ~(size_t)255;
        return here;
    };
    L.state = take((size_t)n_batches * sizeof(PartState));
    L.n_req = take(sizeof(unsigned long long) * 2);
    L.send_counts = take(sizeof(int64_t) * (size_t)(world + 1));
    L.hist = take((size_t)PART_BLOCKS * PART_MAX_WORLD * sizeof(uint32_t));
    L.base = take((size_t)(PART_MAX_WORLD + 1) * sizeof(int64_t));
    L.req_in = take((size_t)request_cap * sizeof(PartRequest));
    L.rstate_in = take((size_t)request_cap * sizeof(int64_t));
    L.req_pos = take((size_t)request_cap * sizeof(uint32_t));
    L.poff = take((size_t)(request_cap + 1) * sizeof(int64_t));
    L.scan_tmp_bytes = (size_t)(PSCAN_BLOCKS_HOST + 1) * sizeof(int64_t);
    L.scan_tmp = take(L.scan_tmp_bytes);
    L.total = at;
    return L;
}

// ---------------------------------------------------------------- begin: seeds -> slabs, frontier = the seeds
__global__ void part_init_kernel(const int64_t *__restrict__ seeds, const int64_t *__restrict__ seeds_state,
                                 int64_t n_batches, int64_t n_seeds, int64_t *samples, int64_t *states, int64_t cap_nodes,
                                 PartState *state, int64_t *counts, int32_t n_hops) {
    const int64_t b = blockIdx.x;
    for (int64_t i = threadIdx.x; i < n_seeds; i += blockDim.x) {
        samples[b * cap_nodes + i] = seeds[b * n_seeds + i];
        if (seeds_state) states[b * cap_nodes + i] = seeds_state[b * n_seeds + i];
    }
    if (threadIdx.x == 0) {
        state[b] = PartState{0, n_seeds, 0, 0};
        if (n_hops == 0) {
            counts[b * 2 + 0] = n_seeds;
            counts[b * 2 + 1] = 0;
        }
    }
}

// ---------------------------------------------------------------- requests, step 1: frontier order, per batch
__global__ void part_requests_kernel(const int64_t *__restrict__ samples, const int64_t *__restrict__ states,
                                     int64_t cap_nodes, PartState *state, unsigned long long *n_req, PartRequest *req_in,
                                     int64_t *rstate_in) {
    __shared__ int64_t fbase_s;
    const int64_t b = blockIdx.x;
    const PartState st = state[b];
    if (threadIdx.x == 0) {
        fbase_s = (int64_t)atomicAdd(n_req, (unsigned long long)(st.end - st.begin)); // batch order does not matter
        state[b].fbase = fbase_s;
    }
    __syncthreads();
    const int64_t fbase = fbase_s;
    for (int64_t i = st.begin + threadIdx.x; i < st.end; i += blockDim.x) {
        req_in[fbase + (i - st.begin)] = PartRequest{samples[b * cap_nodes + i], (uint32_t)b, (uint32_t)i};
        if (rstate_in) rstate_in[fbase + (i - st.begin)] = states[b * cap_nodes + i]; // the frontier vertex's filter state
    }
}

__device__ __forceinline__ int part_owner(int64_t v, int64_t shard_size, int world) {
    const int64_t o = v / shard_size;
    return (int)(o < (int64_t)world - 1 ? o : (int64_t)world - 1);
}

// ---------------------------------------------------------------- requests, step 2: counting sort by owner
__global__ void __launch_bounds__(PART_THREADS) part_hist_kernel(const PartRequest *__restrict__ req_in,
                                                                  const unsigned long long *n_req, int64_t shard_size,
                                                                  int world, uint32_t *hist) {
    __shared__ uint32_t h[PART_MAX_WORLD];
    if (threadIdx.x < PART_MAX_WORLD) h[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t n = *n_req;
    for (uint64_t t0 = (uint64_t)blockIdx.x * PART_TILE; t0 < n; t0 += (uint64_t)gridDim.x * PART_TILE)
        for (uint64_t j = t0 + threadIdx.x; j < min(n, t0 + PART_TILE); j += blockDim.x)
            atomicAdd(&h[part_owner(req_in[j].vertex, shard_size, world)], 1u);
    __syncthreads();
    if (threadIdx.x < world) hist[(size_t)blockIdx.x * PART_MAX_WORLD + threadIdx.x] = h[threadIdx.x];
}

// one block: hist[row][owner] -> running offsets per owner over the rows; base[owner] = bucket start; send_counts
__global__ void part_scan_kernel(uint32_t *hist, int n_rows, int world, int64_t *base, int64_t *send_counts) {
    __shared__ int64_t tot[PART_MAX_WORLD];
    const int o = threadIdx.x;
    if (o < world) {
        uint32_t run = 0;
        for (int r = 0; r < n_rows; ++r) {
            uint32_t *cell = hist + (size_t)r * PART_MAX_WORLD + o;
            const uint32_t v = *cell;
            *cell = run;
            run += v;
        }
        tot[o] = run;
        send_counts[o] = run;
    }
    __syncthreads();
    if (o == 0) {
        int64_t acc = 0;
        for (int w = 0; w < world; ++w) {
            base[w] = acc;
            acc += tot[w];
        }
        base[world] = acc;
        send_counts[world] = acc;
    }
}

__global__ void __launch_bounds__(PART_THREADS) part_scatter_kernel(const PartRequest *__restrict__ req_in,
                                                                     const unsigned long long *n_req, int64_t shard_size,
                                                                     int world, const uint32_t *hist, const int64_t *base,
                                                                     PartRequest *req_out, uint32_t *req_pos,
                                                                     const int64_t *rstate_in, int64_t *rstate_out) {
    __shared__ uint32_t cur[PART_MAX_WORLD];
    if (threadIdx.x < world)
        cur[threadIdx.x] = (uint32_t)base[threadIdx.x] + hist[(size_t)blockIdx.x * PART_MAX_WORLD + threadIdx.x];
    __syncthreads();
    const uint64_t n = *n_req;
    for (uint64_t t0 = (uint64_t)blockIdx.x * PART_TILE; t0 < n; t0 += (uint64_t)gridDim.x * PART_TILE)
        for (uint64_t j = t0 + threadIdx.x; j < min(n, t0 + PART_TILE); j += blockDim.x) {
            const PartRequest r = req_in[j];
            const uint32_t pos = atomicAdd(&cur[part_owner(r.vertex, shard_size, world)], 1u);
            req_out[pos] = r;
            req_pos[j] = pos;
            if (rstate_out) rstate_out[pos] = rstate_in[j];
        }
}

// world == 1: the frontier order IS the request order (part_requests_kernel wrote the send buffer itself)
__global__ void part_identity_kernel(const unsigned long long *n_req, uint32_t *req_pos, int64_t *send_counts) {
    const uint64_t n = *n_req;
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (uint64_t)gridDim.x * blockDim.x)
        req_pos[j] = (uint32_t)j;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        send_counts[0] = (int64_t)n;
        send_counts[1] = (int64_t)n;
    }
}

// ---------------------------------------------------------------- owner side
__host__ __device__ inline int part_reply_words(int format) { return format == TG_PART_REPLY_PACKED_STATE ? 2 : format; }
__host__ __device__ inline bool part_reply_packed(int format) {
    return format == TG_PART_REPLY_PACKED || format == TG_PART_REPLY_PACKED_STATE;
}
__host__ __device__ inline bool part_reply_has_state(int format) {
    return format == TG_PART_REPLY_TRIPLES || format == TG_PART_REPLY_PACKED_STATE;
}

struct PartOwnerParams {
    const int64_t *ptrs, *indices;
    const uint32_t *indices32; // optional u32 shadow of `indices`
    const uint32_t *ptrs32;    // optional u32 shadow of `ptrs` (a shard of < 2^32 edges)
    int64_t n_major, v_lo, e_lo;
    const PartRequest *req;
    const int64_t *m_dev; // number of requests (device)
    int32_t k, replace, world, packed;
    uint64_t seed;
    int64_t seg_off[PART_MAX_WORLD + 1]; // requests of requesting rank p: [seg_off[p], seg_off[p+1])
    uint64_t seg_call0[PART_MAX_WORLD];  // that rank's first call id
    uint32_t *cnt;                       // [m]
    const int64_t *off;                  // [m] exclusive prefix of cnt
    int64_t *reply;                      // [sum cnt][2], or [sum cnt] packed words
    int64_t *reply_counts;               // [world + 1] reply entries per requesting rank, total
};

__global__ void part_count_kernel(const PartOwnerParams p) {
    const int64_t m = *p.m_dev;
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < m; j += (int64_t)gridDim.x * blockDim.x) {
        const int64_t w = p.req[j].vertex - p.v_lo;
        int64_t deg = 0;
        if (w >= 0 && w < p.n_major) deg = p.ptrs[w + 1] - p.ptrs[w];
        p.cnt[j] = (deg <= 0) ? 0u : (p.replace ? (uint32_t)p.k : (uint32_t)min(deg, (int64_t)p.k));
    }
}

// reply entries per requesting rank, from the prefix (one thread per rank)
__global__ void part_reply_counts_kernel(const PartOwnerParams p) {
    const int t = threadIdx.x;
    const int64_t m = *p.m_dev;
    auto at = [&](int64_t j) -> int64_t { return p.off[j >= m ? m : j]; };
    if (t < p.world) p.reply_counts[t] = at(p.seg_off[t + 1]) - at(p.seg_off[t]);
    if (t == 0) p.reply_counts[p.world] = at(m);
}

template <int KMAX> __global__ void part_sample_kernel(const PartOwnerParams p) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
    const int k = p.k;
    const size_t wave_bytes = 64 * sizeof(int64_t) + (size_t)64 * k * sizeof(uint32_t) + (((size_t)64 * k + 15) & ~(size_t)15);
    unsigned char *wbase = smem + (size_t)wave * wave_bytes;
    int64_t *ebase = reinterpret_cast<int64_t *>(wbase);
    uint32_t *spos = reinterpret_cast<uint32_t *>(wbase + 64 * sizeof(int64_t));
    uint8_t *slane = reinterpret_cast<uint8_t *>(wbase + 64 * sizeof(int64_t) + (size_t)64 * k * sizeof(uint32_t));
    const int64_t m = *p.m_dev;
    const int64_t n_chunks = (m + 63) >> 6;
    for (int64_t c = (int64_t)blockIdx.x * n_waves + wave; c < n_chunks; c += (int64_t)gridDim.x * n_waves) {
        const int64_t j0 = c << 6, j = j0 + lane;
        int64_t e0 = 0;
        uint32_t cnt = 0, n = 0;
        uint64_t call = 0, did = 0;
        if (j < m) {
            const PartRequest r = p.req[j];
            const int64_t w = r.vertex - p.v_lo;
            cnt = p.cnt[j];
            if (cnt) {
                e0 = p.ptrs[w];
                n = (uint32_t)(p.ptrs[w + 1] - e0);
            }
            int src = 0; // requesting rank of request j: the segment that holds j
            while (src + 1 < p.world && p.seg_off[src + 1] <= j) ++src;
            call = p.seg_call0[src] + (uint64_t)r.batch;
            did = (uint64_t)r.slot;
        }
        const uint32_t incl = wave_inclusive_scan(cnt);
        const uint32_t excl = incl - cnt;
        const uint32_t total = __shfl(incl, 63, 64);
        const int64_t out0 = (j0 < m) ? p.off[j0] : 0; // the chunk's replies are contiguous from here
        ebase[lane] = e0;
        if (cnt > 0) {
            const CallKey ck = call_key(p.seed, call, TAG_NS_HOMO);
            if (p.replace) { // sampling.rs:57-69
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
        // gathers issued four per lane before the first store (unconditional loads, as in ns_homo.hip's emit)
        for (uint32_t q0 = 0; q0 < total; q0 += 256) {
            int64_t ep[4], v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t q = q0 + (uint32_t)(u * 64 + lane);
                const uint32_t qq = q < total ? q : 0u;
                ep[u] = ebase[slane[qq]] + (int64_t)spos[qq];
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
                    if (p.packed) { // one word per sample: halves what the reply all-to-all moves
                        p.reply[out0 + (int64_t)q] = (int64_t)((uint64_t)v[u] | ((uint64_t)(ep[u] + p.e_lo) << 32));
                    } else {
                        int64_t *o = p.reply + (out0 + (int64_t)q) * 2;
                        o[0] = v[u];
                        o[1] = ep[u] + p.e_lo;
                    }
                }
            }
        }
        wave_lds_handoff();
    }
}

// ---------------------------------------------------------------- owner side, window-ordered (many requests per hop)
// part_sample_kernel walks the requests in arrival order: its `indices[edge_ptr]` gathers go all over the shard, one
// 128-byte line request each (the 55 G/s ceiling of DESIGN.md 4.1b: 2.6 of the 4.8 ms of a 4 096-batch call).  With a
// workspace (tg_part_sample_ws) the requests are first counting-sorted BY THE WINDOW OF THEIR COLUMN -- found from the
// vertex id through a vertex -> window table, no look-up needed -- into a permutation; the XCDs then sweep contiguous
// eighths of the sorted order, so that at any time an XCD works inside a few windows that its L2 holds.  A request's
// replies still go to ITS place (off[j], request order), so the replies the origin receives are the same words.
constexpr int PSORT_BLOCKS = 512, PSORT_THREADS = 512, PSORT_MAX_WINDOWS = 4096, PSORT_SCAN_GROUPS = 16;
struct PartSortParams {
    const PartRequest *req;
    const int64_t *m_dev;
    const int64_t *ptrs;
    int64_t v_lo, n_major;
    int32_t shift, n_windows;
    uint32_t *vtab, *hist, *base, *perm;
    struct PartSorted *sorted; // optional: the requests themselves in window order (slot replies) instead of `perm`
};
struct PartSorted { // 16 bytes: a request moved into window order
    uint32_t v;     // vertex rebased to the shard (0xffffffff: not this shard's)
    uint32_t j;     // its index in arrival order = where its slot goes
    uint32_t batch, slot;
};
static_assert(sizeof(PartSorted) == 16, "sorted request layout");
__global__ void psort_vtab_kernel(const PartSortParams p) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.n_windows) return;
    const int64_t target = (int64_t)i << p.shift;
    int64_t lo = 0, hi = p.n_major; // first local vertex whose column starts at or beyond window i
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (p.ptrs[mid] >= target)
            hi = mid;
        else
            lo = mid + 1;
    }
    p.vtab[i] = (uint32_t)lo;
}
__device__ __forceinline__ int psort_window(const uint32_t *vtab, int n_windows, int64_t w, int64_t n_major) {
    if (w < 0 || w >= n_major) return 0; // not this shard's: sampled as empty wherever it lands
    int lo = 0, hi = n_windows;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if ((int64_t)vtab[mid] <= w)
            lo = mid;
        else
            hi = mid;
    }
    return lo;
}
__global__ void __launch_bounds__(PSORT_THREADS) psort_hist_kernel(const PartSortParams p) {
    __shared__ uint32_t h[PSORT_MAX_WINDOWS], lvtab[PSORT_MAX_WINDOWS];
    for (int i = threadIdx.x; i < p.n_windows; i += blockDim.x) {
        h[i] = 0;
        lvtab[i] = p.vtab[i];
    }
    __syncthreads();
    const int64_t m = *p.m_dev;
    for (int64_t t0 = (int64_t)blockIdx.x * 4096; t0 < m; t0 += (int64_t)gridDim.x * 4096)
        for (int64_t j = t0 + threadIdx.x; j < min(m, t0 + 4096); j += blockDim.x)
            atomicAdd(&h[psort_window(lvtab, p.n_windows, p.req[j].vertex - p.v_lo, p.n_major)], 1u);
    __syncthreads();
    for (int i = threadIdx.x; i < p.n_windows; i += blockDim.x) p.hist[(size_t)blockIdx.x * p.n_windows + i] = h[i];
}
// per window: exclusive running sum over the rows (64 windows x 16 row groups per workgroup), totals -> base
__global__ void __launch_bounds__(64 * PSORT_SCAN_GROUPS) psort_colscan_kernel(const PartSortParams p, int n_rows) {
    __shared__ uint32_t part[PSORT_SCAN_GROUPS][64];
    const int bl = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int w = blockIdx.x * 64 + bl;
    const int rows_per = (n_rows + PSORT_SCAN_GROUPS - 1) / PSORT_SCAN_GROUPS;
    const int r0 = g * rows_per, r1 = min(n_rows, r0 + rows_per);
    uint32_t sum = 0;
    if (w < p.n_windows)
        for (int r = r0; r < r1; ++r) sum += p.hist[(size_t)r * p.n_windows + w];
    part[g][bl] = sum;
    __syncthreads();
    uint32_t run = 0;
    for (int gg = 0; gg < g; ++gg) run += part[gg][bl];
    if (w < p.n_windows) {
        for (int r = r0; r < r1; ++r) {
            uint32_t *cell = p.hist + (size_t)r * p.n_windows + w;
            const uint32_t v = *cell;
            *cell = run;
            run += v;
        }
        if (g == PSORT_SCAN_GROUPS - 1) p.base[w] = run;
    }
}
__global__ void psort_basescan_kernel(const PartSortParams p) { // one workgroup: exclusive scan of the window totals
    __shared__ uint32_t wave_tot[16];
    __shared__ uint32_t carry_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int c0 = 0; c0 < p.n_windows; c0 += blockDim.x) {
        const int i = c0 + tid;
        const uint32_t v = i < p.n_windows ? p.base[i] : 0u;
        const uint32_t incl = wave_inclusive_scan(v);
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        uint32_t off = carry_s;
        for (int w = 0; w < wave; ++w) off += wave_tot[w];
        if (i < p.n_windows) p.base[i] = off + incl - v;
        __syncthreads();
        if (tid == 0) {
            uint32_t t = carry_s;
            for (int w = 0; w < nw; ++w) t += wave_tot[w];
            carry_s = t;
        }
        __syncthreads();
    }
}
__global__ void __launch_bounds__(PSORT_THREADS) psort_scatter_kernel(const PartSortParams p) {
    __shared__ uint32_t cur[PSORT_MAX_WINDOWS], lvtab[PSORT_MAX_WINDOWS];
    for (int i = threadIdx.x; i < p.n_windows; i += blockDim.x) {
        cur[i] = p.base[i] + p.hist[(size_t)blockIdx.x * p.n_windows + i];
        lvtab[i] = p.vtab[i];
    }
    __syncthreads();
    const int64_t m = *p.m_dev;
    for (int64_t t0 = (int64_t)blockIdx.x * 4096; t0 < m; t0 += (int64_t)gridDim.x * 4096)
        for (int64_t j = t0 + threadIdx.x; j < min(m, t0 + 4096); j += blockDim.x) {
            const PartRequest r = p.req[j];
            const int64_t w = r.vertex - p.v_lo;
            const uint32_t at = atomicAdd(&cur[psort_window(lvtab, p.n_windows, w, p.n_major)], 1u);
            if (p.sorted)
                p.sorted[at] = PartSorted{(w >= 0 && w < p.n_major) ? (uint32_t)w : 0xffffffffu, (uint32_t)j, r.batch, r.slot};
            else
                p.perm[at] = (uint32_t)j;
        }
}

// part_sample_kernel over the SORTED order: lane <- request perm[i]; the replies of a request go to off[request]
template <int KMAX> __global__ void part_sample_sorted_kernel(const PartOwnerParams p, const uint32_t *__restrict__ perm) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
    const int k = p.k;
    const size_t wave_bytes = 128 * sizeof(int64_t) + (size_t)64 * k * sizeof(uint32_t) + (((size_t)64 * k + 15) & ~(size_t)15);
    unsigned char *wbase = smem + (size_t)wave * wave_bytes;
    int64_t *ebase = reinterpret_cast<int64_t *>(wbase);
    int64_t *obase = ebase + 64;
    uint32_t *spos = reinterpret_cast<uint32_t *>(wbase + 128 * sizeof(int64_t));
    uint8_t *slane = reinterpret_cast<uint8_t *>(wbase + 128 * sizeof(int64_t) + (size_t)64 * k * sizeof(uint32_t));
    const int64_t m = *p.m_dev;
    const int64_t n_chunks = (m + 63) >> 6;
    // blocks with equal blockIdx % 8 share an XCD (observed; a speed matter only): group x sweeps the x-th eighth of the
    // sorted order, in order
    const int x = blockIdx.x & 7, local = blockIdx.x >> 3, per_x = gridDim.x >> 3;
    const int64_t c_lo = n_chunks * x / 8, c_hi = n_chunks * (x + 1) / 8;
    for (int64_t c = c_lo + (int64_t)local * n_waves + wave; c < c_hi; c += (int64_t)per_x * n_waves) {
        const int64_t i = (c << 6) + lane;
        int64_t e0 = 0, out0 = 0;
        uint32_t cnt = 0, n = 0;
        uint64_t call = 0, did = 0;
        if (i < m) {
            const int64_t j = perm[i];
            const PartRequest r = p.req[j];
            const int64_t w = r.vertex - p.v_lo;
            cnt = p.cnt[j];
            if (cnt) {
                e0 = p.ptrs[w];
                n = (uint32_t)(p.ptrs[w + 1] - e0);
                out0 = p.off[j];
            }
            int src = 0;
            while (src + 1 < p.world && p.seg_off[src + 1] <= j) ++src;
            call = p.seg_call0[src] + (uint64_t)r.batch;
            did = (uint64_t)r.slot;
        }
        const uint32_t incl = wave_inclusive_scan(cnt);
        const uint32_t excl = incl - cnt;
        const uint32_t total = __shfl(incl, 63, 64);
        ebase[lane] = e0;
        obase[lane] = out0 - (int64_t)excl; // + the running index of the staged entry = the reply position
        if (cnt > 0) {
            const CallKey ck = call_key(p.seed, call, TAG_NS_HOMO);
            if (p.replace) { // sampling.rs:57-69
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
        for (uint32_t q0 = 0; q0 < total; q0 += 256) {
            int64_t ep[4], v[4], o[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t q = q0 + (uint32_t)(u * 64 + lane);
                const uint32_t qq = q < total ? q : 0u;
                const int l = slane[qq];
                ep[u] = ebase[l] + (int64_t)spos[qq];
                o[u] = obase[l] + (int64_t)qq;
            }
            if (p.indices32) {
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = (int64_t)p.indices32[ep[u]];
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = p.indices[ep[u]];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t q = q0 + (uint32_t)(u * 64 + lane);
                if (q < total) {
                    if (p.packed) {
                        p.reply[o[u]] = (int64_t)((uint64_t)v[u] | ((uint64_t)(ep[u] + p.e_lo) << 32));
                    } else {
                        int64_t *dst = p.reply + o[u] * 2;
                        dst[0] = v[u];
                        dst[1] = ep[u] + p.e_lo;
                    }
                }
            }
        }
        wave_lds_handoff();
    }
}

// ---------------------------------------------------------------- owner side, general form (temporal filters, weights)
// The flat hops of ns_hop_scan.hip sample a frontier given as arrays (vertex, draw id, call id, state); here the
// received requests are unpacked into that shape (vertices rebased to the shard, -1 = not mine / padding) and the hop's
// compact outputs are packed into the reply format.
__global__ void part_unpack_kernel(const PartOwnerParams p, int64_t m_cap, int64_t *vertices, int64_t *ids,
                                   int64_t *call_ids) {
    const int64_t m = *p.m_dev;
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < m_cap; j += (int64_t)gridDim.x * blockDim.x) {
        int64_t v = -1, id = 0, call = 0;
        if (j < m) {
            const PartRequest r = p.req[j];
            const int64_t w = r.vertex - p.v_lo;
            if (w >= 0 && w < p.n_major) v = w;
            int src = 0;
            while (src + 1 < p.world && p.seg_off[src + 1] <= j) ++src;
            id = (int64_t)r.slot;
            call = (int64_t)(p.seg_call0[src] + (uint64_t)r.batch);
        }
        vertices[j] = v;
        ids[j] = id;
        call_ids[j] = call;
    }
}

// hop outputs (cnt i64 [m_cap], offsets [m_cap + 1], neighbours / local edge pointers / states compact) -> reply
__global__ void part_pack_kernel(const PartOwnerParams p, int64_t m_cap, const int64_t *hop_cnt, const int64_t *hop_off,
                                 const int64_t *nbr, const int64_t *ep, const int64_t *st_out, int32_t format) {
    const int64_t m = *p.m_dev;
    const int64_t total = hop_off[m_cap];
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < m; j += (int64_t)gridDim.x * blockDim.x)
        p.cnt[j] = (uint32_t)hop_cnt[j];
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (int64_t)gridDim.x * blockDim.x) {
        int64_t *o = p.reply + q * part_reply_words(format);
        if (part_reply_packed(format)) {
            o[0] = (int64_t)((uint64_t)nbr[q] | ((uint64_t)(ep[q] + p.e_lo) << 32));
            if (format == TG_PART_REPLY_PACKED_STATE) o[1] = st_out[q];
        } else {
            o[0] = nbr[q];
            o[1] = ep[q] + p.e_lo;
            if (format == TG_PART_REPLY_TRIPLES) o[2] = st_out[q];
        }
    }
    if (blockIdx.x == 0 && threadIdx.x <= p.world) { // reply entries per requesting rank (padding slots count 0)
        const int t = threadIdx.x;
        auto at = [&](int64_t j) -> int64_t { return hop_off[j >= m ? m : j]; };
        p.reply_counts[t] = (t < p.world) ? at(p.seg_off[t + 1]) - at(p.seg_off[t]) : at(m);
    }
}

// ---------------------------------------------------------------- origin side: emit
struct PartEmitParams {
    int64_t n_seeds, cap_nodes, cap_edges;
    int64_t *samples, *rows, *cols, *edge_index, *layer_offsets, *counts;
    PartState *state;
    const uint32_t *req_pos, *cnt; // cnt: per request, request order (as returned)
    const int64_t *poff;           // exclusive prefix of cnt
    const int64_t *reply;          // pairs (or triples with the sample's filter state), compact, request order
    int64_t *states;               // filter-state slab, written when the reply carries states
    int32_t k, hop, n_hops, reply_format;
};

__global__ void part_emit_kernel(const PartEmitParams p) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n_waves = blockDim.x >> 6;
    const int64_t b = blockIdx.x;
    const int k = p.k;
    uint32_t *chunk_off = reinterpret_cast<uint32_t *>(smem);
    unsigned char *wbase = smem + ((((size_t)(PART_CHUNKS_PER_ROUND + 1) * 4) + 15) & ~(size_t)15) +
                           (size_t)wave * (64 * sizeof(int64_t) + 2 * (((size_t)64 * k + 15) & ~(size_t)15));
    int64_t *rbase = reinterpret_cast<int64_t *>(wbase); // reply offset of each lane's request
    uint8_t *eslot = wbase + 64 * sizeof(int64_t);
    uint8_t *elane = eslot + (((size_t)64 * k + 15) & ~(size_t)15);
    int64_t *samples = p.samples + b * p.cap_nodes;
    int64_t *rows = p.rows + b * p.cap_edges, *cols = p.cols + b * p.cap_edges, *eidx = p.edge_index + b * p.cap_edges;
    const PartState st = p.state[b];
    const int64_t begin = st.begin, end = st.end, fbase = st.fbase;
    int64_t ne = st.ne;
    const int64_t n_seeds = p.n_seeds;
    const int words = part_reply_words(p.reply_format);
    const bool packed = part_reply_packed(p.reply_format), has_state = part_reply_has_state(p.reply_format);
    if (tid == 0) { // neighbor_sampling.rs:193
        int64_t *lo = p.layer_offsets + (b * p.n_hops + p.hop) * 3;
        lo[0] = n_seeds + ne;
        lo[1] = ne;
        lo[2] = n_seeds + ne;
    }
    for (int64_t round_begin = begin; round_begin < end; round_begin += (int64_t)PART_CHUNKS_PER_ROUND * 64) {
        const int64_t round_end = min(end, round_begin + (int64_t)PART_CHUNKS_PER_ROUND * 64);
        const int nc = (int)((round_end - round_begin + 63) >> 6);
        for (int c = wave; c < nc; c += n_waves) {
            const int64_t i = round_begin + (int64_t)c * 64 + lane;
            const uint32_t cnt = (i < round_end) ? p.cnt[p.req_pos[fbase + (i - begin)]] : 0u;
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
            int64_t ro = 0;
            if (i < round_end) {
                const uint32_t rp = p.req_pos[fbase + (i - begin)];
                cnt = p.cnt[rp];
                ro = p.poff[rp];
            }
            const uint32_t incl = wave_inclusive_scan(cnt);
            const uint32_t excl = incl - cnt;
            const uint32_t total = __shfl(incl, 63, 64);
            rbase[lane] = ro;
            for (uint32_t s = 0; s < cnt; ++s) {
                eslot[excl + s] = (uint8_t)s;
                elane[excl + s] = (uint8_t)lane;
            }
            wave_lds_handoff();
            const int64_t e_chunk = ne + (int64_t)chunk_off[c];
            for (uint32_t q = lane; q < total; q += 64) {
                const int l = elane[q];
                const int64_t *r = p.reply + (rbase[l] + (int64_t)eslot[q]) * words;
                const int64_t e = e_chunk + q;
                int64_t nbr = r[0], ep;
                if (packed) {
                    ep = (int64_t)((uint64_t)nbr >> 32);
                    nbr = (int64_t)((uint64_t)nbr & 0xffffffffull);
                } else {
                    ep = r[1];
                }
                samples[n_seeds + e] = nbr;                                 // :215 (the next hop's frontier)
                if (has_state) p.states[b * p.cap_nodes + n_seeds + e] = r[words - 1]; // :216 filter.mutate()
                __builtin_nontemporal_store(n_seeds + e, &rows[e]);        // :217
                __builtin_nontemporal_store(i0 + (int64_t)l, &cols[e]);
                __builtin_nontemporal_store(ep, &eidx[e]);
            }
            wave_lds_handoff();
        }
        __syncthreads();
        ne += chunk_off[nc];
        __syncthreads();
    }
    if (tid == 0) { // :221-222
        p.state[b] = PartState{end, n_seeds + ne, ne, fbase};
        p.counts[b * 2 + 0] = n_seeds + ne;
        p.counts[b * 2 + 1] = ne;
    }
}

#include "partition_slots.inl"

// ---------------------------------------------------------------- exclusive prefix of u32 counts, length on the device
// (the host only knows an upper bound of the number of requests; scanning that bound would cost 5x the work)
constexpr int PSCAN_BLOCKS = PSCAN_BLOCKS_HOST, PSCAN_THREADS = 256;

__global__ void __launch_bounds__(PSCAN_THREADS) pscan_sums_kernel(const uint32_t *__restrict__ v, const int64_t *n_dev,
                                                                    int64_t *block_sum) {
    __shared__ int64_t ws[PSCAN_THREADS / 64];
    const int64_t n = *n_dev;
    const int64_t per = (n + PSCAN_BLOCKS - 1) / PSCAN_BLOCKS;
    const int64_t lo = min(n, (int64_t)blockIdx.x * per), hi = min(n, lo + per);
    int64_t s = 0;
    for (int64_t j = lo + threadIdx.x; j < hi; j += PSCAN_THREADS) s += v[j];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        int64_t t = 0;
        for (int w = 0; w < PSCAN_THREADS / 64; ++w) t += ws[w];
        block_sum[blockIdx.x] = t;
    }
}

__global__ void __launch_bounds__(PSCAN_BLOCKS) pscan_top_kernel(int64_t *block_sum) { // -> exclusive, total at [BLOCKS]
    __shared__ int64_t ws[PSCAN_BLOCKS / 64];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int64_t v = block_sum[t];
    const int64_t incl = wave_inclusive_scan(v);
    if (lane == 63) ws[wave] = incl;
    __syncthreads();
    int64_t before = 0;
    for (int w = 0; w < wave; ++w) before += ws[w];
    block_sum[t] = before + incl - v;
    if (t == PSCAN_BLOCKS - 1) block_sum[PSCAN_BLOCKS] = before + incl;
}

__global__ void __launch_bounds__(PSCAN_THREADS) pscan_apply_kernel(const uint32_t *__restrict__ v, const int64_t *n_dev,
                                                                     const int64_t *block_sum, int64_t *out) {
    __shared__ int64_t ws[PSCAN_THREADS / 64];
    __shared__ int64_t carry_s;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int64_t n = *n_dev;
    const int64_t per = (n + PSCAN_BLOCKS - 1) / PSCAN_BLOCKS;
    const int64_t lo = min(n, (int64_t)blockIdx.x * per), hi = min(n, lo + per);
    if (t == 0) carry_s = block_sum[blockIdx.x];
    __syncthreads();
    for (int64_t j0 = lo; j0 < hi; j0 += PSCAN_THREADS) {
        const int64_t j = j0 + t;
        const int64_t x = j < hi ? (int64_t)v[j] : 0;
        const int64_t incl = wave_inclusive_scan(x);
        if (lane == 63) ws[wave] = incl;
        __syncthreads();
        int64_t before = carry_s;
        for (int w = 0; w < wave; ++w) before += ws[w];
        if (j < hi) out[j] = before + incl - x;
        __syncthreads();
        if (t == PSCAN_THREADS - 1) carry_s = before + incl;
        __syncthreads();
    }
    if (blockIdx.x == 0 && t == 0) out[n] = block_sum[PSCAN_BLOCKS]; // total behind the last entry
}

static int part_scan(const uint32_t *v, const int64_t *n_dev, int64_t *block_sum, int64_t *out, hipStream_t s) {
    hipLaunchKernelGGL(pscan_sums_kernel, dim3(PSCAN_BLOCKS), dim3(PSCAN_THREADS), 0, s, v, n_dev, block_sum);
    hipLaunchKernelGGL(pscan_top_kernel, dim3(1), dim3(PSCAN_BLOCKS), 0, s, block_sum);
    hipLaunchKernelGGL(pscan_apply_kernel, dim3(PSCAN_BLOCKS), dim3(PSCAN_THREADS), 0, s, v, n_dev, block_sum, out);
    TG_LAUNCH_CHECK();
    return TG_OK;
}

static inline unsigned part_grid(int64_t n, int threads, int64_t cap) {
    int64_t g = (n + threads - 1) / threads;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (unsigned)g;
}

} // namespace tg

extern "C" int tg_part_workspace_bytes(int64_t n_batches, int64_t request_cap, int32_t world, int64_t *bytes) {
    TG_REQUIRE(bytes && n_batches >= 0 && request_cap >= 0 && request_cap < ((int64_t)1 << 32) && world >= 1 &&
                   world <= tg::PART_MAX_WORLD,
               "tg_part_workspace_bytes: bad arguments (world <= %d, request_cap < 2^32)", tg::PART_MAX_WORLD);
    *bytes = (int64_t)tg::part_layout(n_batches, request_cap, world).total;
    return TG_OK;
}

extern "C" int tg_part_begin(const int64_t *seeds, const int64_t *seeds_state, int64_t n_batches, int64_t n_seeds,
                             int32_t n_hops, const tg_ns_out *out, int64_t request_cap, int32_t world, void *workspace,
                             void *stream) {
    TG_REQUIRE(out && workspace && n_batches >= 0 && n_seeds >= 0 && (seeds || n_seeds == 0) && world >= 1 &&
                   world <= tg::PART_MAX_WORLD && n_hops >= 0 && n_hops <= TG_MAX_HOPS,
               "tg_part_begin: bad arguments");
    TG_REQUIRE(out->samples && out->counts && out->cap_nodes >= n_seeds, "tg_part_begin: samples slab too small");
    TG_REQUIRE(!seeds_state || out->states, "tg_part_begin: filter states need the `states` slab");
    TG_REQUIRE(((uintptr_t)workspace & 255) == 0, "tg_part_begin: workspace must be 256-byte aligned");
    if (n_batches == 0) return TG_OK;
    const tg::PartLayout L = tg::part_layout(n_batches, request_cap, world);
    unsigned char *w = static_cast<unsigned char *>(workspace);
    hipLaunchKernelGGL(tg::part_init_kernel, dim3((unsigned)n_batches), dim3(256), 0, (hipStream_t)stream, seeds, seeds_state,
                       n_batches, n_seeds, out->samples, out->states, out->cap_nodes,
                       reinterpret_cast<tg::PartState *>(w + L.state), out->counts, n_hops);
    TG_LAUNCH_CHECK();
    return TG_OK;
}

extern "C" int tg_part_requests(const tg_ns_out *out, int64_t n_batches, int64_t request_cap, int64_t shard_size,
                                int32_t world, void *workspace, void *requests, int64_t *request_states,
                                int64_t *send_counts, void *stream) {
    TG_REQUIRE(out && workspace && requests && send_counts && n_batches >= 1 && shard_size >= 1 && world >= 1 &&
                   world <= tg::PART_MAX_WORLD,
               "tg_part_requests: bad arguments");
    using namespace tg;
    hipStream_t s = (hipStream_t)stream;
    const PartLayout L = part_layout(n_batches, request_cap, world);
    unsigned char *w = static_cast<unsigned char *>(workspace);
    PartState *state = reinterpret_cast<PartState *>(w + L.state);
    unsigned long long *n_req = reinterpret_cast<unsigned long long *>(w + L.n_req);
    PartRequest *req_in = reinterpret_cast<PartRequest *>(w + L.req_in);
    uint32_t *req_pos = reinterpret_cast<uint32_t *>(w + L.req_pos);
    uint32_t *hist = reinterpret_cast<uint32_t *>(w + L.hist);
    int64_t *base = reinterpret_cast<int64_t *>(w + L.base);
    TG_REQUIRE(!request_states || out->states, "tg_part_requests: request states need the `states` slab");
    int64_t *rstate_in = request_states ? reinterpret_cast<int64_t *>(w + L.rstate_in) : nullptr;
    TG_HIP(hipMemsetAsync(n_req, 0, sizeof(unsigned long long), s));
    if (world == 1) { // the frontier order IS the request order: straight into the send buffer
        hipLaunchKernelGGL(part_requests_kernel, dim3((unsigned)n_batches), dim3(256), 0, s, out->samples, out->states,
                           out->cap_nodes, state, n_req, static_cast<PartRequest *>(requests), request_states);
        hipLaunchKernelGGL(part_identity_kernel, dim3(part_grid(request_cap, 256, 4096)), dim3(256), 0, s, n_req, req_pos,
                           send_counts);
    } else {
        hipLaunchKernelGGL(part_requests_kernel, dim3((unsigned)n_batches), dim3(256), 0, s, out->samples, out->states,
                           out->cap_nodes, state, n_req, req_in, rstate_in);
        hipLaunchKernelGGL(part_hist_kernel, dim3(PART_BLOCKS), dim3(PART_THREADS), 0, s, req_in, n_req, shard_size,
                           (int)world, hist);
        hipLaunchKernelGGL(part_scan_kernel, dim3(1), dim3(64), 0, s, hist, PART_BLOCKS, (int)world, base, send_counts);
        hipLaunchKernelGGL(part_scatter_kernel, dim3(PART_BLOCKS), dim3(PART_THREADS), 0, s, req_in, n_req, shard_size,
                           (int)world, hist, base, static_cast<PartRequest *>(requests), req_pos, rstate_in, request_states);
    }
    // a copy of the sizes stays in the workspace for tg_part_emit
    TG_HIP(hipMemcpyAsync(w + L.send_counts, send_counts, sizeof(int64_t) * (size_t)(world + 1), hipMemcpyDeviceToDevice, s));
    TG_LAUNCH_CHECK();
    return TG_OK;
}

static int part_owner_params(tg::PartOwnerParams &p, const tg_graph *shard, int64_t v_lo, int64_t e_lo, const void *requests,
                             const int64_t *m_dev, int32_t world, const int64_t *seg_off, const uint64_t *seg_call0,
                             int32_t fanout, int32_t sampler, uint64_t seed, int64_t m_cap, bool general = false) {
    TG_REQUIRE(shard && shard->ptrs && (shard->indices || shard->n_edges == 0), "tg_part: null shard");
    TG_REQUIRE(fanout >= 1 && (general || fanout <= TG_MAX_FANOUT), "tg_part: fanout %d outside [1, %d]", fanout, TG_MAX_FANOUT);
    TG_REQUIRE(general || sampler == TG_SAMPLER_UNIFORM || sampler == TG_SAMPLER_UNIFORM_REPL,
               "tg_part_count / tg_part_sample: unweighted samplers only (filters and weights: tg_part_unpack + a flat hop + tg_part_pack)");
    TG_REQUIRE(world >= 1 && world <= tg::PART_MAX_WORLD && seg_off && seg_call0 && m_dev && (requests || m_cap == 0),
               "tg_part: bad owner arguments"); // (a rank that received no request may pass no buffer)
    p.ptrs = shard->ptrs;
    p.indices = shard->indices;
    p.indices32 = shard->indices32;
    p.ptrs32 = shard->ptrs32;
    p.n_major = shard->n_major;
    p.v_lo = v_lo;
    p.e_lo = e_lo;
    p.req = static_cast<const tg::PartRequest *>(requests);
    p.m_dev = m_dev;
    p.k = fanout;
    p.replace = sampler == TG_SAMPLER_UNIFORM_REPL;
    p.world = world;
    p.seed = seed;
    for (int i = 0; i <= world; ++i) p.seg_off[i] = seg_off[i];
    for (int i = 0; i < world; ++i) p.seg_call0[i] = seg_call0[i];
    return TG_OK;
}

extern "C" int tg_part_count(const tg_graph *shard, int64_t v_lo, const void *requests, const int64_t *m_dev, int64_t m_cap,
                             int32_t world, const int64_t *seg_off, const uint64_t *seg_call0, int32_t fanout,
                             int32_t sampler, uint32_t *cnt, int64_t *off, int64_t *reply_counts, void *scan_tmp,
                             int64_t scan_tmp_bytes, void *stream) {
    tg::PartOwnerParams p;
    int rc = part_owner_params(p, shard, v_lo, 0, requests, m_dev, world, seg_off, seg_call0, fanout, sampler, 0, m_cap);
    if (rc != TG_OK) return rc;
    TG_REQUIRE(cnt && off && reply_counts && m_cap >= 0, "tg_part_count: null buffers");
    if (m_cap == 0) {
        TG_HIP(hipMemsetAsync(reply_counts, 0, sizeof(int64_t) * (size_t)(world + 1), (hipStream_t)stream));
        return TG_OK;
    }
    p.cnt = cnt;
    p.off = off;
    p.reply = nullptr;
    p.reply_counts = reply_counts;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(tg::part_count_kernel, dim3(tg::part_grid(m_cap, 256, 4096)), dim3(256), 0, s, p);
    TG_REQUIRE(scan_tmp && scan_tmp_bytes >= (int64_t)((tg::PSCAN_BLOCKS + 1) * sizeof(int64_t)),
               "tg_part_count: scan workspace too small");
    rc = tg::part_scan(cnt, m_dev, static_cast<int64_t *>(scan_tmp), off, s); // off[0 .. m], off[m] = total
    if (rc != TG_OK) return rc;
    hipLaunchKernelGGL(tg::part_reply_counts_kernel, dim3(1), dim3(64), 0, s, p);
    TG_LAUNCH_CHECK();
    return TG_OK;
}

extern "C" int tg_part_scan_workspace_bytes(int64_t n, int64_t *bytes) {
    TG_REQUIRE(bytes && n >= 0, "tg_part_scan_workspace_bytes: bad arguments");
    *bytes = (int64_t)((tg::PSCAN_BLOCKS + 1) * sizeof(int64_t));
    return TG_OK;
}

extern "C" int tg_part_sample(const tg_graph *shard, int64_t v_lo, int64_t e_lo, const void *requests, const int64_t *m_dev,
                              int64_t m_cap, int32_t world, const int64_t *seg_off, const uint64_t *seg_call0,
                              int32_t fanout, int32_t sampler, uint64_t seed, const uint32_t *cnt, const int64_t *off,
                              int64_t *reply, int32_t reply_format, void *stream) {
    tg::PartOwnerParams p;
    int rc = part_owner_params(p, shard, v_lo, e_lo, requests, m_dev, world, seg_off, seg_call0, fanout, sampler, seed, m_cap);
    if (rc != TG_OK) return rc;
    TG_REQUIRE(reply_format == TG_PART_REPLY_PAIRS || reply_format == TG_PART_REPLY_PACKED,
               "tg_part_sample: reply_format is TG_PART_REPLY_PAIRS or TG_PART_REPLY_PACKED");
    if (m_cap == 0) return TG_OK;
    TG_REQUIRE(cnt && off && reply, "tg_part_sample: null buffers");
    p.packed = reply_format == TG_PART_REPLY_PACKED;
    p.cnt = const_cast<uint32_t *>(cnt);
    p.off = off;
    p.reply = reply;
    p.reply_counts = nullptr;
    const int n_waves = 4;
    const size_t lds = (size_t)n_waves * (64 * sizeof(int64_t) + (size_t)64 * fanout * 4 + (((size_t)64 * fanout + 15) & ~(size_t)15));
    const unsigned grid = tg::part_grid((m_cap + 63) / 64, n_waves, 256 * 16);
    if (fanout <= 16)
        hipLaunchKernelGGL(tg::part_sample_kernel<16>, dim3(grid), dim3(64 * n_waves), lds, (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL(tg::part_sample_kernel<32>, dim3(grid), dim3(64 * n_waves), lds, (hipStream_t)stream, p);
    TG_LAUNCH_CHECK();
    return TG_OK;
}

struct PartSortLayout {
    size_t vtab, hist, base, perm, sorted, keys, total;
};
static PartSortLayout part_sort_layout(int64_t m_cap) {
    PartSortLayout L;
    size_t at = 0;
    auto take = [&](size_t bytes) {
        const size_t here = at;
        at += (bytes + 255) & ~(size_t)255;
        return here;
    };
    L.vtab = take((size_t)tg::PSORT_MAX_WINDOWS * 4);
    L.hist = take((size_t)tg::PSORT_BLOCKS * tg::PSORT_MAX_WINDOWS * 4);
    L.base = take((size_t)(tg::PSORT_MAX_WINDOWS + 8) * 4);
    L.perm = take((size_t)(m_cap > 0 ? m_cap : 1) * 4);
    L.sorted = take((size_t)(m_cap > 0 ? m_cap : 1) * 16);
    L.keys = take((size_t)tg::PART_MAX_WORLD * 4096 * 8); // PART_KEY_CAP call keys per requesting rank
    L.total = at;
    return L;
}

static int part_env_int(const char *name, int dflt) {
    const char *v = getenv(name);
    return v && *v ? atoi(v) : dflt;
}

// the window order of a hop's requests (psort_*): -> the permutation in the workspace
static int part_order_requests(const tg::PartOwnerParams &p, const tg_graph *shard, int64_t v_lo, const int64_t *m_dev,
                               void *workspace, const PartSortLayout &L, hipStream_t s, const uint32_t **perm,
                               const tg::PartSorted **sorted = nullptr) {
    unsigned char *w = static_cast<unsigned char *>(workspace);
    tg::PartSortParams sp;
    sp.req = p.req;
    sp.m_dev = m_dev;
    sp.ptrs = shard->ptrs;
    sp.v_lo = v_lo;
    sp.n_major = shard->n_major;
    const int elem = shard->indices32 ? 4 : 8;
    int shift = 0;
    static const int64_t window_bytes = (int64_t)std::max(16, part_env_int("TG_PART_WINDOW_KIB", 512)) << 10;
    while (((int64_t)elem << shift) < window_bytes) ++shift; // 512 KB of the gathered array per window ...
    while (((shard->n_edges >> shift) + 1) > tg::PSORT_MAX_WINDOWS) ++shift; // ... as far as the counters go
    sp.shift = shift;
    sp.n_windows = (int32_t)((shard->n_edges >> shift) + 1);
    sp.vtab = reinterpret_cast<uint32_t *>(w + L.vtab);
    sp.hist = reinterpret_cast<uint32_t *>(w + L.hist);
    sp.base = reinterpret_cast<uint32_t *>(w + L.base);
    sp.perm = reinterpret_cast<uint32_t *>(w + L.perm);
    sp.sorted = sorted ? reinterpret_cast<tg::PartSorted *>(w + L.sorted) : nullptr;
    hipLaunchKernelGGL(tg::psort_vtab_kernel, dim3((sp.n_windows + 255) / 256), dim3(256), 0, s, sp);
    hipLaunchKernelGGL(tg::psort_hist_kernel, dim3(tg::PSORT_BLOCKS), dim3(tg::PSORT_THREADS), 0, s, sp);
    hipLaunchKernelGGL(tg::psort_colscan_kernel, dim3((sp.n_windows + 63) / 64), dim3(64 * tg::PSORT_SCAN_GROUPS), 0, s, sp,
                       tg::PSORT_BLOCKS);
    hipLaunchKernelGGL(tg::psort_basescan_kernel, dim3(1), dim3(1024), 0, s, sp);
    hipLaunchKernelGGL(tg::psort_scatter_kernel, dim3(tg::PSORT_BLOCKS), dim3(tg::PSORT_THREADS), 0, s, sp);
    TG_LAUNCH_CHECK();
    if (perm) *perm = sp.perm;
    if (sorted) *sorted = sp.sorted;
    return TG_OK;
}

static int64_t g_part_order_min_requests = (int64_t)1 << 21, g_part_order_min_edges = (int64_t)1 << 24;
extern "C" int tg_part_sample_order_thresholds(int64_t min_requests, int64_t min_edges) {
    if (min_requests >= 0) g_part_order_min_requests = min_requests;
    if (min_edges >= 0) g_part_order_min_edges = min_edges;
    return TG_OK;
}

extern "C" int tg_part_sample_workspace_bytes(int64_t m_cap, int64_t *bytes) {
    TG_REQUIRE(bytes && m_cap >= 0, "tg_part_sample_workspace_bytes: bad arguments");
    *bytes = (int64_t)part_sort_layout(m_cap).total;
    return TG_OK;
}

extern "C" int tg_part_sample_ws(const tg_graph *shard, int64_t v_lo, int64_t e_lo, const void *requests, const int64_t *m_dev,
                                 int64_t m_cap, int32_t world, const int64_t *seg_off, const uint64_t *seg_call0,
                                 int32_t fanout, int32_t sampler, uint64_t seed, const uint32_t *cnt, const int64_t *off,
                                 int64_t *reply, int32_t reply_format, void *workspace, int64_t workspace_bytes,
                                 void *stream) {
    // worth it when the hop's gathers revisit lines: many requests against a shard far larger than the L2s
    const bool ordered = workspace && m_cap >= g_part_order_min_requests && m_cap > 0 && m_cap < ((int64_t)1 << 32) && shard &&
                         shard->n_edges >= g_part_order_min_edges && shard->n_major < ((int64_t)1 << 32);
    if (!ordered)
        return tg_part_sample(shard, v_lo, e_lo, requests, m_dev, m_cap, world, seg_off, seg_call0, fanout, sampler, seed, cnt,
                              off, reply, reply_format, stream);
    tg::PartOwnerParams p;
    int rc = part_owner_params(p, shard, v_lo, e_lo, requests, m_dev, world, seg_off, seg_call0, fanout, sampler, seed, m_cap);
    if (rc != TG_OK) return rc;
    TG_REQUIRE(reply_format == TG_PART_REPLY_PAIRS || reply_format == TG_PART_REPLY_PACKED,
               "tg_part_sample_ws: reply_format is TG_PART_REPLY_PAIRS or TG_PART_REPLY_PACKED");
    TG_REQUIRE(cnt && off && reply, "tg_part_sample_ws: null buffers");
    const PartSortLayout L = part_sort_layout(m_cap);
    TG_REQUIRE(workspace_bytes >= (int64_t)L.total && ((uintptr_t)workspace & 255) == 0,
               "tg_part_sample_ws: workspace too small or not 256-byte aligned");
    p.packed = reply_format == TG_PART_REPLY_PACKED;
    p.cnt = const_cast<uint32_t *>(cnt);
    p.off = off;
    p.reply = reply;
    p.reply_counts = nullptr;
    hipStream_t s = (hipStream_t)stream;
    const uint32_t *perm = nullptr;
    rc = part_order_requests(p, shard, v_lo, m_dev, workspace, L, s, &perm);
    if (rc != TG_OK) return rc;
    const int n_waves = 4;
    const size_t lds = (size_t)n_waves * (128 * sizeof(int64_t) + (size_t)64 * fanout * 4 + (((size_t)64 * fanout + 15) & ~(size_t)15));
    const unsigned grid = 1024; // a multiple of 8: eight groups of equal size
    if (fanout <= 16)
        hipLaunchKernelGGL(tg::part_sample_sorted_kernel<16>, dim3(grid), dim3(64 * n_waves), lds, s, p, perm);
    else
        hipLaunchKernelGGL(tg::part_sample_sorted_kernel<32>, dim3(grid), dim3(64 * n_waves), lds, s, p, perm);
    TG_LAUNCH_CHECK();
    return TG_OK;
}

extern "C" int tg_part_unpack(int64_t v_lo, int64_t n_major, const void *requests, const int64_t *m_dev, int64_t m_cap,
                              int32_t world, const int64_t *seg_off, const uint64_t *seg_call0, int64_t *vertices, int64_t *ids,
                              int64_t *call_ids, void *stream) {
    TG_REQUIRE(m_dev && seg_off && seg_call0 && m_cap >= 0 && world >= 1 && world <= tg::PART_MAX_WORLD &&
                   (m_cap == 0 || (requests && vertices && ids && call_ids)),
               "tg_part_unpack: bad arguments");
    if (m_cap == 0) return TG_OK;
    tg::PartOwnerParams p{};
    p.req = static_cast<const tg::PartRequest *>(requests);
    p.m_dev = m_dev;
    p.v_lo = v_lo;
    p.n_major = n_major;
    p.world = world;
    for (int i = 0; i <= world; ++i) p.seg_off[i] = seg_off[i];
    for (int i = 0; i < world; ++i) p.seg_call0[i] = seg_call0[i];
    hipLaunchKernelGGL(tg::part_unpack_kernel, dim3(tg::part_grid(m_cap, 256, 4096)), dim3(256), 0, (hipStream_t)stream, p, m_cap,
                       vertices, ids, call_ids);
    TG_LAUNCH_CHECK();
    return TG_OK;
}

extern "C" int tg_part_pack(const tg_hop_out *hop, const int64_t *states_out, const int64_t *m_dev, int64_t m_cap, int64_t e_lo,
                            int32_t world, const int64_t *seg_off, uint32_t *cnt, int64_t *reply, int32_t reply_format,
                            int64_t *reply_counts, void *stream) {
    TG_REQUIRE(hop && hop->cnt && hop->offsets && hop->neighbors && hop->edge_ptrs && m_dev && seg_off && cnt && reply &&
                   reply_counts && m_cap >= 0 && world >= 1 && world <= tg::PART_MAX_WORLD,
               "tg_part_pack: bad arguments");
    TG_REQUIRE(reply_format >= TG_PART_REPLY_PACKED && reply_format <= TG_PART_REPLY_PACKED_STATE &&
                   (!tg::part_reply_has_state(reply_format) || states_out),
               "tg_part_pack: reply_format is one of TG_PART_REPLY_*, the ones with a state need states_out");
    if (m_cap == 0) {
        TG_HIP(hipMemsetAsync(reply_counts, 0, sizeof(int64_t) * (size_t)(world + 1), (hipStream_t)stream));
        return TG_OK;
    }
    tg::PartOwnerParams p{};
    p.m_dev = m_dev;
    p.e_lo = e_lo;
    p.world = world;
    for (int i = 0; i <= world; ++i) p.seg_off[i] = seg_off[i];
    p.cnt = cnt;
    p.reply = reply;
    p.reply_counts = reply_counts;
    hipLaunchKernelGGL(tg::part_pack_kernel, dim3(tg::part_grid(m_cap * 4, 256, 4096)), dim3(256), 0, (hipStream_t)stream, p,
                       m_cap, hop->cnt, hop->offsets, hop->neighbors, hop->edge_ptrs, states_out, reply_format);
    TG_LAUNCH_CHECK();
    return TG_OK;
}

extern "C" int tg_part_emit(const tg_ns_out *out, int64_t n_batches, int64_t n_seeds, int64_t request_cap, int64_t hop_cap,
                            int32_t world, int32_t fanout, int32_t hop, int32_t n_hops, void *workspace,
                            const uint32_t *cnt, const int64_t *cnt_prefix, const int64_t *reply, int32_t reply_format,
                            void *stream) {
    TG_REQUIRE(out && workspace && cnt && n_batches >= 1 && fanout >= 1 && fanout <= 255 && hop >= 0 && hop < n_hops &&
                   n_hops <= TG_MAX_HOPS,
               "tg_part_emit: bad arguments");
    TG_REQUIRE(reply_format >= TG_PART_REPLY_PACKED && reply_format <= TG_PART_REPLY_PACKED_STATE &&
                   (!tg::part_reply_has_state(reply_format) || out->states),
               "tg_part_emit: reply_format is one of TG_PART_REPLY_*, the ones with a state need a `states` slab");
    TG_REQUIRE(out->samples && out->rows && out->cols && out->edge_index && out->layer_offsets && out->counts,
               "tg_part_emit: null output slabs");
    using namespace tg;
    hipStream_t s = (hipStream_t)stream;
    const PartLayout L = part_layout(n_batches, request_cap, world);
    unsigned char *w = static_cast<unsigned char *>(workspace);
    const int64_t *poff = cnt_prefix;
    TG_REQUIRE(hop_cap >= 0 && hop_cap <= request_cap, "tg_part_emit: hop_cap outside [0, request_cap]");
    if (!poff) { // reply offset of every request: prefix of the returned counts (the number of requests is on the device)
        int64_t *mine = reinterpret_cast<int64_t *>(w + L.poff);
        const int64_t *n_dev = reinterpret_cast<const int64_t *>(w + L.send_counts) + world; // total written by tg_part_requests
        int rc = part_scan(cnt, n_dev, reinterpret_cast<int64_t *>(w + L.scan_tmp), mine, s);
        if (rc != TG_OK) return rc;
        poff = mine;
    }
    PartEmitParams p;
    p.n_seeds = n_seeds;
    p.cap_nodes = out->cap_nodes;
    p.cap_edges = out->cap_edges;
    p.samples = out->samples;
    p.rows = out->rows;
    p.cols = out->cols;
    p.edge_index = out->edge_index;
    p.layer_offsets = out->layer_offsets;
    p.counts = out->counts;
    p.state = reinterpret_cast<PartState *>(w + L.state);
    p.req_pos = reinterpret_cast<uint32_t *>(w + L.req_pos);
    p.cnt = cnt;
    p.poff = poff;
    p.reply = reply;
    p.states = out->states;
    p.reply_format = reply_format;
    p.k = fanout;
    p.hop = hop;
    p.n_hops = n_hops;
    const int threads = 512;
    const size_t lds = ((((size_t)(PART_CHUNKS_PER_ROUND + 1) * 4) + 15) & ~(size_t)15) +
                       (size_t)(threads / 64) * (64 * sizeof(int64_t) + 2 * (((size_t)64 * fanout + 15) & ~(size_t)15));
    hipLaunchKernelGGL(part_emit_kernel, dim3((unsigned)n_batches), dim3(threads), lds, s, p);
    TG_LAUNCH_CHECK();
    return TG_OK;
}

// ---------------------------------------------------------------- slot replies (partition_slots.inl)
static int part_slot_format(int32_t fanout, int32_t vertex_bits, int32_t position_bits, tg::StageBits *sb, int *words) {
    TG_REQUIRE(fanout >= 1 && vertex_bits >= 1 && vertex_bits <= 32 && position_bits >= 1 && position_bits <= 32,
               "tg_part slots: fan-out >= 1, 1 <= vertex_bits, position_bits <= 32");
    sb->bv = vertex_bits;
    sb->bp = position_bits;
    const int bits = tg::stage_slot_bits((int)fanout, *sb);
    *words = fanout > 16 ? 0 : bits <= 512 ? 16 : bits <= 1024 ? 32 : 0;
    return TG_OK;
}

extern "C" int tg_part_slot_words(int32_t fanout, int32_t vertex_bits, int32_t position_bits, int32_t *words) {
    TG_REQUIRE(words, "tg_part_slot_words: null result");
    tg::StageBits sb;
    int w = 0;
    int rc = part_slot_format(fanout, vertex_bits, position_bits, &sb, &w);
    if (rc != TG_OK) return rc;
    *words = w;
    return TG_OK;
}

template <int W, int KMAX>
static void part_slot_sample_launch(const tg::PartOwnerParams &p, const tg::PartSorted *sorted, const tg::CallKey *keys,
                                    tg::StageBits sb, uint32_t *slots, int64_t m_cap, hipStream_t s) {
    static const int threads = std::min(512, std::max(64, part_env_int("TG_PART_SAMPLE_THREADS", 256) & ~63));
    static const int chunks = std::max(1, part_env_int("TG_PART_SAMPLE_CHUNKS", 1)); // per wave
    const size_t lds = (size_t)(threads / 64) * (64 * (W + 1) + 64) * sizeof(uint32_t);
    const int64_t per_block = (int64_t)(threads / 64) * chunks;
    const int64_t per_x = (((m_cap + 63) / 64 + 7) / 8 + per_block - 1) / per_block + 1; // eight groups of equal size
    const unsigned blocks = (unsigned)(8 * per_x);
    if (p.replace)
        hipLaunchKernelGGL((tg::part_slot_sample_kernel<W, KMAX, true>), dim3(blocks), dim3(threads), lds, s, p, sorted, keys, sb,
                           slots, chunks);
    else
        hipLaunchKernelGGL((tg::part_slot_sample_kernel<W, KMAX, false>), dim3(blocks), dim3(threads), lds, s, p, sorted, keys, sb,
                           slots, chunks);
}

extern "C" int tg_part_sample_slots(const tg_graph *shard, int64_t v_lo, const void *requests, const int64_t *m_dev,
                                    int64_t m_cap, int32_t world, const int64_t *seg_off, const uint64_t *seg_call0,
                                    int32_t fanout, int32_t sampler, uint64_t seed, int32_t vertex_bits,
                                    int32_t position_bits, int32_t order_scale, void *slots, void *workspace,
                                    int64_t workspace_bytes, void *stream) {
    tg::PartOwnerParams p;
    int rc = part_owner_params(p, shard, v_lo, 0, requests, m_dev, world, seg_off, seg_call0, fanout, sampler, seed, m_cap);
    if (rc != TG_OK) return rc;
    tg::StageBits sb;
    int W = 0;
    rc = part_slot_format(fanout, vertex_bits, position_bits, &sb, &W);
    if (rc != TG_OK) return rc;
    TG_REQUIRE(W != 0, "tg_part_sample_slots: fan-out %d with %d + %d bits per sample does not fit a slot (tg_part_slot_words)",
               fanout, vertex_bits, position_bits);
    TG_REQUIRE(shard->n_edges < ((int64_t)1 << 32) && m_cap < ((int64_t)1 << 32), "tg_part_sample_slots: a shard of < 2^32 edges, < 2^32 requests");
    TG_REQUIRE((shard->n_major <= 1 || vertex_bits >= 1) && (slots || m_cap == 0) && ((uintptr_t)slots & 63) == 0,
               "tg_part_sample_slots: slots null or not 64-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const tg::CallKey *keys = nullptr;
    if (workspace && m_cap > 0) { // the call keys of the first PART_KEY_CAP batches of every requesting rank
        const PartSortLayout L = part_sort_layout(m_cap);
        TG_REQUIRE(workspace_bytes >= (int64_t)L.total && ((uintptr_t)workspace & 255) == 0,
                   "tg_part_sample_slots: workspace too small or not 256-byte aligned");
        tg::CallKey *k = reinterpret_cast<tg::CallKey *>(static_cast<unsigned char *>(workspace) + L.keys);
        hipLaunchKernelGGL(tg::part_call_keys_kernel, dim3((unsigned)(world * tg::PART_KEY_CAP + 255) / 256), dim3(256), 0, s, p, k);
        keys = k;
    }
    const tg::PartSorted *perm = nullptr; // the requests in window order, or NULL: arrival order
    const bool ordered = workspace && m_cap >= g_part_order_min_requests * std::max(order_scale, 1) &&
                         shard->n_edges >= g_part_order_min_edges &&
                         shard->n_major < ((int64_t)0xffffffff);
    if (ordered) {
        const PartSortLayout L = part_sort_layout(m_cap);
        TG_REQUIRE(workspace_bytes >= (int64_t)L.total && ((uintptr_t)workspace & 255) == 0,
                   "tg_part_sample_slots: workspace too small or not 256-byte aligned");
        rc = part_order_requests(p, shard, v_lo, m_dev, workspace, L, s, nullptr, &perm);
        if (rc != TG_OK) return rc;
    }
    if (m_cap == 0) return TG_OK;
    uint32_t *out = static_cast<uint32_t *>(slots);
    if (W == 16) {
        if (fanout <= 12)
            part_slot_sample_launch<16, 12>(p, perm, keys, sb, out, m_cap, s);
        else
            part_slot_sample_launch<16, 16>(p, perm, keys, sb, out, m_cap, s);
    } else {
        if (fanout <= 12)
            part_slot_sample_launch<32, 12>(p, perm, keys, sb, out, m_cap, s);
        else
            part_slot_sample_launch<32, 16>(p, perm, keys, sb, out, m_cap, s);
    }
    TG_LAUNCH_CHECK();
    return TG_OK;
}

template <int W, int KMAX>
static void part_slot_emit_launch(const tg::PartSlotEmitParams &p, tg::StageBits sb, int64_t n_batches, hipStream_t s) {
    static const int waves = part_env_int("TG_PART_EMIT_WAVES", 2);
    const size_t lds = (size_t)waves * tg::part_slot_emit_wave_bytes(W, p.k);
    if (waves == 1)
        hipLaunchKernelGGL((tg::part_slot_emit_kernel<W, KMAX, 1>), dim3((unsigned)n_batches), dim3(64), lds, s, p, sb);
    else if (waves == 4)
        hipLaunchKernelGGL((tg::part_slot_emit_kernel<W, KMAX, 4>), dim3((unsigned)n_batches), dim3(256), lds, s, p, sb);
    else
        hipLaunchKernelGGL((tg::part_slot_emit_kernel<W, KMAX, 2>), dim3((unsigned)n_batches), dim3(128), lds, s, p, sb);
}

extern "C" int tg_part_emit_slots(const tg_ns_out *out, int64_t n_batches, int64_t n_seeds, int64_t request_cap, int32_t world,
                                  int32_t fanout, int32_t hop, int32_t n_hops, void *workspace, const void *slots,
                                  int32_t vertex_bits, int32_t position_bits, const int64_t *e_lo_of, void *stream) {
    TG_REQUIRE(out && workspace && slots && e_lo_of && n_batches >= 1 && hop >= 0 && hop < n_hops && n_hops <= TG_MAX_HOPS &&
                   world >= 1 && world <= tg::PART_MAX_WORLD,
               "tg_part_emit_slots: bad arguments");
    TG_REQUIRE(out->samples && out->rows && out->cols && out->edge_index && out->layer_offsets && out->counts,
               "tg_part_emit_slots: null output slabs");
    using namespace tg;
    StageBits sb;
    int W = 0;
    int rc = part_slot_format(fanout, vertex_bits, position_bits, &sb, &W);
    if (rc != TG_OK) return rc;
    TG_REQUIRE(W != 0, "tg_part_emit_slots: no slot format for fan-out %d (tg_part_slot_words)", fanout);
    const PartLayout L = part_layout(n_batches, request_cap, world);
    unsigned char *w = static_cast<unsigned char *>(workspace);
    PartSlotEmitParams p;
    p.n_seeds = n_seeds;
    p.cap_nodes = out->cap_nodes;
    p.cap_edges = out->cap_edges;
    p.samples = out->samples;
    p.rows = out->rows;
    p.cols = out->cols;
    p.edge_index = out->edge_index;
    p.layer_offsets = out->layer_offsets;
    p.counts = out->counts;
    p.state = reinterpret_cast<PartState *>(w + L.state);
    p.req_pos = reinterpret_cast<uint32_t *>(w + L.req_pos);
    p.slots = static_cast<const uint32_t *>(slots);
    p.base = reinterpret_cast<const int64_t *>(w + L.base);
    for (int i = 0; i < world; ++i) p.e_lo_of[i] = e_lo_of[i];
    p.k = fanout;
    p.hop = hop;
    p.n_hops = n_hops;
    p.world = world;
    hipStream_t s = (hipStream_t)stream;
    if (W == 16) {
        if (fanout <= 12)
            part_slot_emit_launch<16, 12>(p, sb, n_batches, s);
        else
            part_slot_emit_launch<16, 16>(p, sb, n_batches, s);
    } else {
        if (fanout <= 12)
            part_slot_emit_launch<32, 12>(p, sb, n_batches, s);
        else
            part_slot_emit_launch<32, 16>(p, sb, n_batches, s);
    }
    TG_LAUNCH_CHECK();
    return TG_OK;
}
