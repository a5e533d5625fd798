// Device side of neighbor_sampling_homogenous over a RANGE-PARTITIONED CSC (SURVEY.md 8(e) mode 2, BASELINE cfg5).
// The origin rank keeps the ordinary per-batch output slabs of tg_ns_homo_batched (samples | rows | cols |
// edge_index | layer_offsets | counts) plus a small per-batch state; per hop
//
//   tg_part_requests   origin: every frontier slot of every batch becomes a request (vertex, call id, slot), bucketed
//                      by the rank that owns the vertex's column (histogram + wave-aggregated scatter); req_pos
//                      remembers where each frontier slot's request went.
//   [all-to-all]       requests travel to their owners (host: torch.distributed / RCCL).
//   tg_part_sample     owner: each request is sampled with the REQUESTER's draws (tag NS_HOMO, id = slot, call id =
//                      the requester's batch) -- the same draws the replicated-graph sampler uses, so results are
//                      equal bit for bit -- into a fixed-stride reply: k (neighbour, global edge pointer) pairs per
//                      request, -1 padded.  Fixed stride = no size exchange, no owner-side read-back.
//   [all-to-all]       replies travel back; they arrive in request order.
//   tg_part_emit       origin: one workgroup per batch compacts its frontier's replies in slot order (the reference's
//                      output order, neighbor_sampling.rs:195-218) into the slabs: counts -> LDS scan -> LDS staging ->
//                      coalesced writes, then advances the batch state.
// Only the bucket sizes (world integers) reach the host per hop.
#include "ns_tickets.h"
#include "tg_device.h"
#include "tg_host.h"

namespace tg {

constexpr int PART_MAX_WORLD = 64;
constexpr int PART_CHUNKS_PER_ROUND = 1024;

// state[b*4 ..] = {frontier begin, frontier end, edges so far, -}
__global__ void part_init_kernel(const int64_t *__restrict__ seeds, int64_t n_batches, int64_t n_seeds, int64_t *samples,
                                 int64_t cap_nodes, int64_t *state) {
    const int64_t total = n_batches * n_seeds;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = t / n_seeds, i = t - b * n_seeds;
        samples[b * cap_nodes + i] = seeds[t];
    }
    for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b < n_batches; b += (int64_t)gridDim.x * blockDim.x) {
        state[b * 4 + 0] = 0;
        state[b * 4 + 1] = n_seeds;
        state[b * 4 + 2] = 0;
        state[b * 4 + 3] = 0;
    }
}

// batch_off[b] = sum of frontier sizes of the batches before b; batch_off[n_batches] = number of requests.
// One workgroup.  Also clears the bucket histogram and cursors.
__global__ void part_sizes_kernel(const int64_t *__restrict__ state, int64_t n_batches, int64_t *batch_off, int64_t *hist,
                                  int64_t *cursor, int world) {
    __shared__ int64_t wave_tot[16];
    __shared__ int64_t carry_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n_waves = blockDim.x >> 6;
    if (tid < world) {
        hist[tid] = 0;
        cursor[tid] = 0;
    }
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int64_t base = 0; base < n_batches; base += blockDim.x) {
        const int64_t b = base + tid;
        const int64_t v = (b < n_batches) ? state[b * 4 + 1] - state[b * 4 + 0] : 0;
        const int64_t incl = wave_inclusive_scan(v);
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        int64_t before = carry_s;
        for (int w = 0; w < wave; ++w) before += wave_tot[w];
        if (b < n_batches) batch_off[b] = before + incl - v;
        __syncthreads();
        if (tid == 0) {
            int64_t t = carry_s;
            for (int w = 0; w < n_waves; ++w) t += wave_tot[w];
            carry_s = t;
        }
        __syncthreads();
    }
    if (tid == 0) batch_off[n_batches] = carry_s;
}

struct PartReqParams {
    const int64_t *samples, *state, *batch_off;
    int64_t cap_nodes, n_batches, shard_size, first_call_id;
    int world;
    int64_t *hist, *cursor;
    int64_t *req;     // [M][3] vertex, call id, slot -- bucketed by owner
    int64_t *req_pos; // [M] flat frontier index -> position of its request
};

__device__ __forceinline__ void part_locate(const PartReqParams &p, int64_t f, int64_t &b, int64_t &i) {
    int64_t lo = 0, hi = p.n_batches; // last b with batch_off[b] <= f
    while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (p.batch_off[mid] <= f)
            lo = mid;
        else
            hi = mid;
    }
    b = lo;
    i = p.state[b * 4 + 0] + (f - p.batch_off[b]);
}

template <bool SCATTER> __global__ void part_bucket_kernel(const PartReqParams p) {
    __shared__ int64_t base[PART_MAX_WORLD];
    if (SCATTER) { // bucket bases = exclusive prefix of the histogram (world values: every workgroup recomputes them)
        if (threadIdx.x == 0) {
            int64_t acc = 0;
            for (int o = 0; o < p.world; ++o) {
                base[o] = acc;
                acc += p.hist[o];
            }
        }
        __syncthreads();
    }
    const int lane = threadIdx.x & 63;
    const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const int64_t M = p.batch_off[p.n_batches];
    const int64_t step = (int64_t)gridDim.x * blockDim.x;
    for (int64_t f0 = (int64_t)blockIdx.x * blockDim.x + (threadIdx.x & ~63); f0 < M; f0 += step) {
        const int64_t f = f0 + lane;
        const bool valid = f < M;
        int64_t b = 0, i = 0, v = 0;
        int owner = -1;
        if (valid) {
            part_locate(p, f, b, i);
            v = p.samples[b * p.cap_nodes + i];
            const int64_t o = v / p.shard_size;
            owner = (int)(o < (int64_t)p.world - 1 ? o : (int64_t)p.world - 1);
        }
        // one atomic per (wavefront, owner): lanes of the same owner are ranked by ballot
        uint64_t todo = __ballot(valid);
        while (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            const int o = __shfl(owner, leader, 64);
            const uint64_t same = __ballot(valid && owner == o);
            int64_t start = 0;
            if (lane == leader)
                start = (int64_t)atomicAdd(reinterpret_cast<unsigned long long *>(SCATTER ? &p.cursor[o] : &p.hist[o]),
                                           (unsigned long long)__popcll(same));
            if (SCATTER) {
                start = __shfl(start, leader, 64);
                if (valid && owner == o) {
                    const int64_t pos = base[o] + start + (int64_t)__popcll(same & lt_mask);
                    p.req[pos * 3 + 0] = v;
                    p.req[pos * 3 + 1] = p.first_call_id + b;
                    p.req[pos * 3 + 2] = i;
                    p.req_pos[f] = pos;
                }
            }
            todo &= ~same;
        }
    }
}

// ---------------------------------------------------------------- owner side
struct PartSampleParams {
    const int64_t *ptrs, *indices;
    int64_t n_major, v_lo, e_lo;
    const int64_t *req; // [m][3]
    int64_t m;
    int32_t k, replace;
    uint64_t seed;
    int64_t *reply; // [m][k][2]
};

template <int KMAX> __global__ void part_sample_kernel(const PartSampleParams p) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
    const int k = p.k;
    const size_t wave_bytes = 64 * sizeof(int64_t) + (size_t)64 * k * sizeof(uint32_t) + (((size_t)64 * k + 15) & ~(size_t)15);
    unsigned char *wbase = smem + (size_t)wave * wave_bytes;
    int64_t *ebase = reinterpret_cast<int64_t *>(wbase);
    uint32_t *spos = reinterpret_cast<uint32_t *>(wbase + 64 * sizeof(int64_t));
    uint8_t *scratch_lane = reinterpret_cast<uint8_t *>(wbase + 64 * sizeof(int64_t) + (size_t)64 * k * sizeof(uint32_t));
    const int64_t n_chunks = (p.m + 63) >> 6;
    for (int64_t c = (int64_t)blockIdx.x * n_waves + wave; c < n_chunks; c += (int64_t)gridDim.x * n_waves) {
        const int64_t j0 = c << 6, j = j0 + lane;
        int64_t e0 = 0, deg = 0, call = 0, slot = 0;
        if (j < p.m) {
            const int64_t w = p.req[j * 3 + 0] - p.v_lo;
            call = p.req[j * 3 + 1];
            slot = p.req[j * 3 + 2];
            if (w >= 0 && w < p.n_major) {
                e0 = p.ptrs[w];
                deg = p.ptrs[w + 1] - e0;
            }
        }
        const uint32_t cnt = (deg <= 0) ? 0u : (p.replace ? (uint32_t)k : (uint32_t)min(deg, (int64_t)k));
        ebase[lane] = e0;
        const uint32_t mine = (uint32_t)lane * (uint32_t)k; // fixed stride: slot s of this request at mine + s
        for (uint32_t s = cnt; s < (uint32_t)k; ++s) spos[mine + s] = 0xffffffffu;
        if (cnt > 0) {
            const CallKey ck = call_key(p.seed, (uint64_t)call, TAG_NS_HOMO);
            const uint64_t did = (uint64_t)slot;
            const uint32_t n = (uint32_t)deg;
            if (p.replace) { // sampling.rs:57-69
                Draw d;
                for (int s = 0; s < k; ++s) {
                    if ((s & 1) == 0) d = draw(ck, did, (uint32_t)(s >> 1), D1_REPLACE);
                    spos[mine + s] = bounded32(d.half(s & 1), n);
                }
            } else if (deg <= k) { // sampling.rs:12-15
                for (uint32_t s = 0; s < cnt; ++s) spos[mine + s] = s;
            } else {
                sample_tickets<KMAX>(ck, did, n, k, spos, scratch_lane, mine, lane);
            }
        }
        wave_lds_handoff();
        const uint32_t total = (uint32_t)(min((int64_t)64, p.m - j0) * k);
        for (uint32_t q = lane; q < total; q += 64) {
            const uint32_t l = q / (uint32_t)k;
            const uint32_t pos = spos[q];
            int64_t nbr = -1, ep = -1;
            if (pos != 0xffffffffu) {
                const int64_t e = ebase[l] + (int64_t)pos;
                nbr = __builtin_nontemporal_load(&p.indices[e]);
                ep = e + p.e_lo;
            }
            int64_t *o = p.reply + ((j0 * k) + q) * 2;
            o[0] = nbr;
            o[1] = ep;
        }
        wave_lds_handoff();
    }
}

// ---------------------------------------------------------------- origin side: emit
struct PartEmitParams {
    int64_t n_seeds, cap_nodes, cap_edges;
    int64_t *samples, *rows, *cols, *edge_index, *layer_offsets, *counts, *state;
    const int64_t *batch_off, *req_pos, *reply;
    int32_t k, hop, n_hops;
};

__global__ void part_emit_kernel(const PartEmitParams p) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n_waves = blockDim.x >> 6;
    const int64_t b = blockIdx.x;
    const int k = p.k;
    uint32_t *chunk_off = reinterpret_cast<uint32_t *>(smem);
    unsigned char *wbase = smem + ((((size_t)(PART_CHUNKS_PER_ROUND + 1) * 4) + 15) & ~(size_t)15) +
                           (size_t)wave * 2 * (((size_t)64 * k + 15) & ~(size_t)15);
    uint8_t *eslot = wbase;
    uint8_t *elane = wbase + (((size_t)64 * k + 15) & ~(size_t)15);
    int64_t *samples = p.samples + b * p.cap_nodes;
    int64_t *rows = p.rows + b * p.cap_edges, *cols = p.cols + b * p.cap_edges, *eidx = p.edge_index + b * p.cap_edges;
    const int64_t begin = p.state[b * 4 + 0], end = p.state[b * 4 + 1];
    int64_t ne = p.state[b * 4 + 2];
    const int64_t n_seeds = p.n_seeds, off_b = p.batch_off[b];
    if (tid == 0) { // neighbor_sampling.rs:193
        int64_t *lo = p.layer_offsets + (b * p.n_hops + p.hop) * 3;
        lo[0] = n_seeds + ne;
        lo[1] = ne;
        lo[2] = n_seeds + ne;
    }
    auto count_of = [&](int64_t i) -> uint32_t { // replies are a prefix of valid pairs, then -1 padding
        const int64_t *r = p.reply + p.req_pos[off_b + (i - begin)] * (int64_t)k * 2;
        uint32_t c = 0;
        while (c < (uint32_t)k && r[2 * c] >= 0) ++c;
        return c;
    };
    for (int64_t round_begin = begin; round_begin < end; round_begin += (int64_t)PART_CHUNKS_PER_ROUND * 64) {
        const int64_t round_end = min(end, round_begin + (int64_t)PART_CHUNKS_PER_ROUND * 64);
        const int nc = (int)((round_end - round_begin + 63) >> 6);
        for (int c = wave; c < nc; c += n_waves) {
            const int64_t i = round_begin + (int64_t)c * 64 + lane;
            const uint32_t cnt = (i < round_end) ? count_of(i) : 0u;
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
            const uint32_t cnt = (i < round_end) ? count_of(i) : 0u;
            const uint32_t incl = wave_inclusive_scan(cnt);
            const uint32_t excl = incl - cnt;
            const uint32_t total = __shfl(incl, 63, 64);
            for (uint32_t s = 0; s < cnt; ++s) {
                eslot[excl + s] = (uint8_t)s;
                elane[excl + s] = (uint8_t)lane;
            }
            wave_lds_handoff();
            const int64_t e_chunk = ne + (int64_t)chunk_off[c];
            for (uint32_t q = lane; q < total; q += 64) {
                const int l = elane[q];
                const int64_t *r = p.reply + (p.req_pos[off_b + (i0 + l - begin)] * (int64_t)k + eslot[q]) * 2;
                const int64_t e = e_chunk + q;
                samples[n_seeds + e] = r[0]; // :215
                rows[e] = n_seeds + e;       // :217
                cols[e] = i0 + l;
                eidx[e] = r[1];
            }
            wave_lds_handoff();
        }
        __syncthreads();
        ne += chunk_off[nc];
        __syncthreads();
    }
    if (tid == 0) { // :221-222
        p.state[b * 4 + 0] = end;
        p.state[b * 4 + 1] = n_seeds + ne;
        p.state[b * 4 + 2] = ne;
        p.counts[b * 2 + 0] = n_seeds + ne;
        p.counts[b * 2 + 1] = ne;
    }
}

static inline unsigned part_grid(int64_t n, int threads, int64_t cap) {
    int64_t g = (n + threads - 1) / threads;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (unsigned)g;
}

} // namespace tg

extern "C" int tg_part_workspace_bytes(int64_t n_batches, int32_t world, int64_t *bytes) {
    TG_REQUIRE(bytes && n_batches >= 0 && world >= 1 && world <= tg::PART_MAX_WORLD,
               "tg_part_workspace_bytes: bad arguments (world <= %d)", tg::PART_MAX_WORLD);
    *bytes = (n_batches * 4 + (n_batches + 1) + 2 * (int64_t)world) * (int64_t)sizeof(int64_t);
    return TG_OK;
}

// workspace layout: state[n_batches*4] | batch_off[n_batches+1] | hist[world] | cursor[world]
static inline int64_t *ws_state(void *ws) { return reinterpret_cast<int64_t *>(ws); }
static inline int64_t *ws_batch_off(void *ws, int64_t nb) { return ws_state(ws) + nb * 4; }
static inline int64_t *ws_hist(void *ws, int64_t nb) { return ws_batch_off(ws, nb) + nb + 1; }

extern "C" int tg_part_begin(const int64_t *seeds, int64_t n_batches, int64_t n_seeds, const tg_ns_out *out, void *workspace,
                             void *stream) {
    TG_REQUIRE(out && workspace && n_batches >= 0 && n_seeds >= 0 && (seeds || n_seeds == 0), "tg_part_begin: bad arguments");
    TG_REQUIRE(out->samples && out->cap_nodes >= n_seeds, "tg_part_begin: samples slab too small");
    if (n_batches == 0) return TG_OK;
    hipLaunchKernelGGL(tg::part_init_kernel, dim3(tg::part_grid(n_batches * (n_seeds > 0 ? n_seeds : 1), 256, 4096)), dim3(256), 0,
                       (hipStream_t)stream, seeds, n_batches, n_seeds, out->samples, out->cap_nodes, ws_state(workspace));
    TG_LAUNCH_CHECK();
    return TG_OK;
}

extern "C" int tg_part_requests(const tg_ns_out *out, int64_t n_batches, int64_t request_cap, int64_t shard_size,
                                int32_t world, uint64_t first_call_id, void *workspace, int64_t *requests, int64_t *req_pos,
                                void *stream) {
    TG_REQUIRE(out && workspace && requests && req_pos && n_batches >= 1 && shard_size >= 1 && world >= 1 &&
                   world <= tg::PART_MAX_WORLD,
               "tg_part_requests: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    int64_t *hist = ws_hist(workspace, n_batches);
    hipLaunchKernelGGL(tg::part_sizes_kernel, dim3(1), dim3(1024), 0, s, ws_state(workspace), n_batches,
                       ws_batch_off(workspace, n_batches), hist, hist + world, (int)world);
    tg::PartReqParams p;
    p.samples = out->samples;
    p.state = ws_state(workspace);
    p.batch_off = ws_batch_off(workspace, n_batches);
    p.cap_nodes = out->cap_nodes;
    p.n_batches = n_batches;
    p.shard_size = shard_size;
    p.first_call_id = (int64_t)first_call_id;
    p.world = world;
    p.hist = hist;
    p.cursor = hist + world;
    p.req = requests;
    p.req_pos = req_pos;
    const unsigned grid = tg::part_grid(request_cap, 256, 2048);
    hipLaunchKernelGGL(tg::part_bucket_kernel<false>, dim3(grid), dim3(256), 0, s, p);
    hipLaunchKernelGGL(tg::part_bucket_kernel<true>, dim3(grid), dim3(256), 0, s, p);
    TG_LAUNCH_CHECK();
    return TG_OK;
}

extern "C" int tg_part_sample(const tg_graph *shard, int64_t v_lo, int64_t e_lo, const int64_t *requests, int64_t m,
                              int32_t fanout, int32_t sampler, uint64_t seed, int64_t *reply, void *stream) {
    TG_REQUIRE(shard && shard->ptrs && (shard->indices || shard->n_edges == 0), "tg_part_sample: null shard");
    TG_REQUIRE(m >= 0 && fanout >= 1 && fanout <= TG_MAX_FANOUT, "tg_part_sample: fanout %d outside [1, %d]", fanout,
               TG_MAX_FANOUT);
    TG_REQUIRE(sampler == TG_SAMPLER_UNIFORM || sampler == TG_SAMPLER_UNIFORM_REPL, "tg_part_sample: unweighted samplers only");
    if (m == 0) return TG_OK;
    TG_REQUIRE(requests && reply, "tg_part_sample: null buffers");
    tg::PartSampleParams p;
    p.ptrs = shard->ptrs;
    p.indices = shard->indices;
    p.n_major = shard->n_major;
    p.v_lo = v_lo;
    p.e_lo = e_lo;
    p.req = requests;
    p.m = m;
    p.k = fanout;
    p.replace = sampler == TG_SAMPLER_UNIFORM_REPL;
    p.seed = seed;
    p.reply = reply;
    const int n_waves = 4;
    const size_t lds = (size_t)n_waves * (64 * sizeof(int64_t) + (size_t)64 * fanout * 4 + (((size_t)64 * fanout + 15) & ~(size_t)15));
    const unsigned grid = tg::part_grid((m + 63) / 64, n_waves, 256 * 32);
    if (fanout <= 16)
        hipLaunchKernelGGL(tg::part_sample_kernel<16>, dim3(grid), dim3(64 * n_waves), lds, (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL(tg::part_sample_kernel<32>, dim3(grid), dim3(64 * n_waves), lds, (hipStream_t)stream, p);
    TG_LAUNCH_CHECK();
    return TG_OK;
}

extern "C" int tg_part_emit(const tg_ns_out *out, int64_t n_batches, int64_t n_seeds, int32_t fanout, int32_t hop,
                            int32_t n_hops, void *workspace, const int64_t *req_pos, const int64_t *reply, void *stream) {
    TG_REQUIRE(out && workspace && req_pos && n_batches >= 1 && fanout >= 1 && fanout <= TG_MAX_FANOUT && hop >= 0 &&
                   hop < n_hops && n_hops <= TG_MAX_HOPS,
               "tg_part_emit: bad arguments");
    TG_REQUIRE(out->samples && out->rows && out->cols && out->edge_index && out->layer_offsets && out->counts,
               "tg_part_emit: null output slabs");
    tg::PartEmitParams p;
    p.n_seeds = n_seeds;
    p.cap_nodes = out->cap_nodes;
    p.cap_edges = out->cap_edges;
    p.samples = out->samples;
    p.rows = out->rows;
    p.cols = out->cols;
    p.edge_index = out->edge_index;
    p.layer_offsets = out->layer_offsets;
    p.counts = out->counts;
    p.state = ws_state(workspace);
    p.batch_off = ws_batch_off(workspace, n_batches);
    p.req_pos = req_pos;
    p.reply = reply;
    p.k = fanout;
    p.hop = hop;
    p.n_hops = n_hops;
    const int threads = 512;
    const size_t lds = ((((size_t)(tg::PART_CHUNKS_PER_ROUND + 1) * 4) + 15) & ~(size_t)15) +
                       (size_t)(threads / 64) * 2 * (((size_t)64 * fanout + 15) & ~(size_t)15);
    hipLaunchKernelGGL(tg::part_emit_kernel, dim3((unsigned)n_batches), dim3(threads), lds, (hipStream_t)stream, p);
    TG_LAUNCH_CHECK();
    return TG_OK;
}
