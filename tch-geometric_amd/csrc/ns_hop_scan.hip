// One hop of the UNIFORM samplers under a temporal filter over a flat frontier, spread over the whole chip
// (the filtered counterpart of ns_hop.hip; reference: neighbor_sampling.rs:36-77 filter, :195-218 hop body).
// A column has to be inspected edge by edge, and columns differ by five orders of magnitude, so the work is cut
// into GROUPS of 512 consecutive edges of one column, independent of which vertex or batch they belong to:
//   groups   lane per frontier vertex: ceil(deg / 512) -> device scan -> first group of every vertex
//   count    wavefront per run of 32 groups: stream the timestamps (8 x 512 B in flight per lane), ballot +
//            popcount -> admissible edges per group                                     [the HBM-bound kernel]
//   scan     device scan of the group counts -> rank of every group's first admissible edge
//   select   wavefront per vertex: n = its admissible edges; ranks to keep = all of them when n <= k, k draws of
//            U[0,n) with replacement, or the reservoir's ticket chain; each rank is located by
//            a binary search over the vertex's groups and a re-read of that one group -> parked edge pointers
//   emit     device scan of the per-vertex counts, then coalesced gather / write of (neighbour, edge pointer,
//            parent, new filter state)
// Same draws and same per-vertex output order as ns_homo_scan.hip, hence the same results.  No synchronisation;
// if the frontier needs group_cap groups or more, `status[0]` is set to 1 and nothing is sampled.
#include <cstring>

#include <rocprim/device/device_scan.hpp>

#include <algorithm>

#include "tg_device.h"
#include "tg_host.h"
#include "tg_scan.h"

namespace tg {

constexpr int HS_GROUP = 512;  // edges per group = 8 chunks of 64
constexpr int HS_RUN = 32;     // groups per wavefront work unit, at most (fewer when the frontier has few groups)
constexpr int HS_CHUNKS = 8;

// The frontier may be the concatenation of up to HS_MAX_SEG SEGMENTS, each with its own graph, fan-out and draw tag
// (all relations of one heterogeneous hop in one set of launches): vertex i belongs to the last segment whose `begin`
// is <= i.  The single-graph entry points use one segment.
constexpr int HS_MAX_SEG = TG_HOP_MAX_SEGMENTS;
struct HsSeg {
    const int64_t *ptrs, *indices, *timestamps;
    const double *weights;
    int64_t begin;
    int32_t k;
    uint32_t tag;
};
struct HopScanParams {
    HsSeg seg[HS_MAX_SEG];
    int32_t n_seg;
    // optional device-side layout [n_seg + 1]: the segments' real starts, then the frontier's real length (<= m); the
    // host-side `begin`s and `m` are then only the worst case the launches are sized for
    const int64_t *layout_dev;
    const int64_t *vertices, *states, *ids, *call_ids;
    int64_t m, id_base;
    int32_t k, replace; // k: the largest fan-out of the segments = the stride of `park`
    int32_t filter_mode, forward;
    int64_t win_lo, win_hi;
    uint64_t seed, call_id;
    int64_t group_cap;
    // workspace
    int64_t *vgroups; // [m + 1] groups per vertex, then (in place) first group of every vertex; [m] = total
    uint32_t *gcount; // [group_cap]
    uint64_t *gchunk; // [group_cap] admissible edges of each of the group's 8 chunks, one byte each (<= 64)
    int64_t *gpref;   // [group_cap + 1] exclusive prefix of gcount
    int64_t *park;    // [m * k]
    double *gtot;     // weighted sampler, group form: [group_cap * 8] chunk totals -> carries before the chunks
    uint32_t *grank;  // [group_cap] candidates of the vertex before the group
    unsigned long long *slot_best; // [m * k] (rank << 32 | position in the column) of a slot's last accepted candidate
    uint32_t *vck;    // [m * 2] the call key of every vertex's draws
    double *long_tot; // weighted sampler, long columns: per workgroup HW_LONG_CHUNKS chunk totals -> carries ...
    uint32_t *long_cnt; // ... and admissible edges per chunk -> candidates before the chunk
    int32_t *status;  // [0] overflow flag
    // outputs
    int64_t *cnt, *offsets, *neighbors, *edge_ptrs, *parents, *states_out;
};

// the segment table goes from the kernel arguments to LDS once per workgroup (dynamic indexing of by-value kernel
// arguments would go through scratch memory)
__device__ __forceinline__ int64_t hs_m(const HopScanParams &p) { // the frontier's length
    return p.layout_dev ? min(p.m, p.layout_dev[p.n_seg]) : p.m;
}
__device__ __forceinline__ int64_t hs_load_segs(const HopScanParams &p, HsSeg *S) {
#pragma unroll
    for (int j = 0; j < HS_MAX_SEG; ++j)
        if ((int)threadIdx.x == j && j < p.n_seg) {
            S[j] = p.seg[j];
            if (p.layout_dev) S[j].begin = p.layout_dev[j];
        }
    __syncthreads();
    return hs_m(p);
}
__device__ __forceinline__ int hs_seg_of(const HsSeg *S, int n_seg, int64_t v) {
    int s = 0;
    for (int j = 1; j < n_seg; ++j) s += (v >= S[j].begin) ? 1 : 0;
    return s;
}

__device__ __forceinline__ bool hs_pass(const HopScanParams &p, int64_t state, int64_t t) { // neighbor_sampling.rs:55-67
    if (p.filter_mode == TG_FILTER_NONE) return true;
    const int64_t x = (p.filter_mode == TG_FILTER_STATIC) ? t : (p.forward ? (t - state) : -(t - state));
    return p.win_lo <= x && x <= p.win_hi;
}

__global__ void hs_groups_kernel(const HopScanParams p) {
    __shared__ HsSeg S[HS_MAX_SEG];
    const int64_t m = hs_load_segs(p, S);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t w = p.vertices[i];
        int64_t g = 0;
        if (w >= 0) {
            const int64_t *ptrs = S[hs_seg_of(S, p.n_seg, i)].ptrs;
            g = (ptrs[w + 1] - ptrs[w] + HS_GROUP - 1) / HS_GROUP;
        }
        p.vgroups[i + 1] = g;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) p.vgroups[0] = 0;
}
// after the inclusive scan: vgroups[i] = first group of vertex i, vgroups[m] = number of groups
__global__ void hs_check_kernel(const HopScanParams p) {
    const int64_t m = hs_m(p);
    if (threadIdx.x == 0 && blockIdx.x == 0 && p.vgroups[m] >= p.group_cap) p.status[0] = 1;
}

// The three prefix sums of a hop in one launch each when the frontier is short (tg_scan.h): a per-call hop is bound by
// its number of launches.  groups1 = hs_groups_kernel + scan + hs_check_kernel.
__global__ void __launch_bounds__(SCAN1_THREADS) hs_groups1_kernel(const HopScanParams p) {
    __shared__ HsSeg S[HS_MAX_SEG];
    const int64_t m = hs_load_segs(p, S);
    for (int64_t i0 = threadIdx.x; i0 < m; i0 += 8 * SCAN1_THREADS) { // eight independent gathers in flight per lane
        int64_t w[8], a[8], b[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int64_t i = i0 + (int64_t)u * SCAN1_THREADS;
            w[u] = (i < m) ? p.vertices[i] : -1;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int64_t *ptrs = S[hs_seg_of(S, p.n_seg, i0 + (int64_t)u * SCAN1_THREADS)].ptrs;
            a[u] = (w[u] >= 0) ? ptrs[w[u]] : 0;
            b[u] = (w[u] >= 0) ? ptrs[w[u] + 1] : 0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int64_t i = i0 + (int64_t)u * SCAN1_THREADS;
            if (i < m) p.vgroups[i + 1] = (b[u] - a[u] + HS_GROUP - 1) / HS_GROUP;
        }
    }
    __syncthreads();
    block_scan_exclusive_plus1(m, [&](int64_t i) { return p.vgroups[i + 1]; }, p.vgroups);
    __syncthreads();
    if (threadIdx.x == 0 && p.vgroups[m] >= p.group_cap) p.status[0] = 1;
}
// gpref[0 .. n_groups]: only the groups the frontier really has are scanned (their number is on the device)
__global__ void __launch_bounds__(SCAN1_THREADS) hs_gscan1_kernel(const HopScanParams p) {
    const int64_t m = hs_m(p);
    const int64_t gt = p.vgroups[m];
    if (gt >= p.group_cap) return;
    block_scan_exclusive_plus1(gt, [&](int64_t i) { return p.gcount[i]; }, p.gpref);
}
__global__ void __launch_bounds__(SCAN1_THREADS) hs_oscan1_kernel(const HopScanParams p) {
    const int64_t m = hs_m(p);
    block_scan_exclusive_plus1(m, [&](int64_t i) { return p.cnt[i]; }, p.offsets);
}

// wavefront per run of consecutive groups
__global__ void hs_count_kernel(const HopScanParams p) {
    __shared__ HsSeg S[HS_MAX_SEG];
    const int64_t m = hs_load_segs(p, S);
    const int lane = threadIdx.x & 63;
    const int64_t wave_id = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int64_t gt = p.vgroups[m];
    if (gt >= p.group_cap) return;
    // a run's groups are read one after the other (each a dependent chain of loads): long runs only when there are
    // more groups than wavefronts to spread them over
    const int64_t run = max((int64_t)1, min((int64_t)HS_RUN, (gt + n_waves - 1) / n_waves));
    for (int64_t g0 = wave_id * run; g0 < gt; g0 += n_waves * run) {
        // vertex owning group g0: last v with vgroups[v] <= g0
        int64_t lo = 0, hi = m - 1;
        while (lo < hi) {
            const int64_t mid = (lo + hi + 1) >> 1;
            if (p.vgroups[mid] <= g0)
                lo = mid;
            else
                hi = mid - 1;
        }
        int64_t v = lo;
        const int64_t g1 = min(g0 + run, gt);
        for (int64_t g = g0; g < g1; ++g) {
            while (g >= p.vgroups[v + 1]) ++v; // skip to the owner (vertices without groups own nothing)
            const int64_t w = p.vertices[v];
            const int64_t st = p.states[v];
            const HsSeg &sg = S[hs_seg_of(S, p.n_seg, v)];
            const int64_t e1 = sg.ptrs[w + 1];
            const int64_t gb = sg.ptrs[w] + (g - p.vgroups[v]) * HS_GROUP;
            int64_t tsv[HS_CHUNKS];
#pragma unroll
            for (int u = 0; u < HS_CHUNKS; ++u) {
                const int64_t e = gb + u * 64 + lane;
                tsv[u] = (e < e1) ? __builtin_nontemporal_load(&sg.timestamps[e]) : 0;
            }
            uint32_t c = 0;
            uint64_t per_chunk = 0;
#pragma unroll
            for (int u = 0; u < HS_CHUNKS; ++u) {
                const int64_t e = gb + u * 64 + lane;
                const uint32_t cu = (uint32_t)__popcll(__ballot(e < e1 && hs_pass(p, st, tsv[u])));
                c += cu;
                per_chunk |= (uint64_t)cu << (8 * u);
            }
            if (lane == 0) {
                p.gcount[g] = c;
                p.gchunk[g] = per_chunk;
            }
        }
    }
}

// wavefront per frontier vertex.  Per wavefront in LDS: ranks[k] (the ranks to fetch, slot order) and, for fan-outs
// above 64, the ticket chain's displaced entries keys[k] / vals[k].
constexpr int HS_MAX_FANOUT = 1024;
constexpr int HS_LOCATE = 8; // slots whose chunk reads are in flight together
__global__ void hs_select_kernel(const HopScanParams p) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t wave_id = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    __shared__ HsSeg S[HS_MAX_SEG];
    const int64_t m = hs_load_segs(p, S);
    const bool overflow = p.vgroups[m] >= p.group_cap;
    const int kmax = p.k;
    uint32_t *ranks = reinterpret_cast<uint32_t *>(smem) + (size_t)wave * 3 * kmax, *keys = ranks + kmax, *vals = keys + kmax;
    for (int64_t v = wave_id; v < m; v += n_waves) {
        const int64_t w = p.vertices[v];
        const HsSeg &sg = S[hs_seg_of(S, p.n_seg, v)];
        const int k = sg.k;
        const int64_t gfirst = p.vgroups[v], glast = p.vgroups[v + 1];
        uint32_t n = 0;
        int64_t base_rank = 0;
        if (w >= 0 && !overflow && glast > gfirst) {
            base_rank = p.gpref[gfirst];
            n = (uint32_t)(p.gpref[glast] - base_rank);
        }
        const uint32_t cnt_sel = p.replace ? (n > 0 ? (uint32_t)k : 0u) : min(n, (uint32_t)k);
        if (lane == 0) p.cnt[v] = cnt_sel;
        if (cnt_sel > 0) {
            const uint64_t did = p.ids ? (uint64_t)p.ids[v] : (uint64_t)(p.id_base + v);
            const CallKey ck = call_key(p.seed, p.call_ids ? (uint64_t)p.call_ids[v] : p.call_id, sg.tag);
            if (p.replace) { // sampling.rs:57-69
                for (uint32_t s = lane; s < (uint32_t)k; s += 64)
                    ranks[s] = slot_draw(ck, did, s, D1_REPLACE, n);
            } else if (n <= (uint32_t)k) { // every candidate, in order
                for (uint32_t s = lane; s < cnt_sel; s += 64) ranks[s] = s;
            } else if (k <= 64) { // reservoir by tickets, lane s owns slot s
                uint32_t myK = 0xffffffffu, myV = 0, myrank = 0;
                Draw d;
                for (int s = 0; s < k; ++s) {
                    const uint32_t mm = (n - 1u) - (uint32_t)s;
                    if ((s & 3) == 0) d = draw(ck, did, (uint32_t)(s >> 2), 0u);
                    const uint32_t r = slot_draw_from(d, ck, did, (uint32_t)s, 0u, mm), last = mm - 1u;
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
                if (lane < k) ranks[lane] = myrank;
            } else { // the same chain with its displaced entries in LDS, searched by all lanes
                Draw d;
                for (int s = 0; s < k; ++s) {
                    const uint32_t mm = (n - 1u) - (uint32_t)s;
                    if ((s & 3) == 0) d = draw(ck, did, (uint32_t)(s >> 2), 0u);
                    const uint32_t r = slot_draw_from(d, ck, did, (uint32_t)s, 0u, mm), last = mm - 1u;
                    int jr = -1, jl = -1;
                    for (int j = lane; j < s; j += 64) {
                        const uint32_t key = keys[j];
                        jr = (key == r) ? j : jr;
                        jl = (key == last) ? j : jl;
                    }
#pragma unroll
                    for (int off = 32; off > 0; off >>= 1) {
                        jr = max(jr, __shfl_xor(jr, off, 64));
                        jl = max(jl, __shfl_xor(jl, off, 64));
                    }
                    const uint32_t tr = jr >= 0 ? vals[jr] : r, tl = jl >= 0 ? vals[jl] : last;
                    if (lane == 0) {
                        keys[s] = r;
                        vals[s] = tl;
                        ranks[s] = (tr < n - (uint32_t)k) ? (uint32_t)k + tr : (uint32_t)s;
                    }
                    wave_lds_handoff();
                }
            }
        }
        wave_lds_handoff();
        // locate every kept rank.  Lane = slot: binary search over the vertex's groups, then the group's per-chunk counts
        // name the 64-edge chunk and the rank inside it -- all slots side by side.  Then the wavefront reads the chunks of
        // HS_LOCATE slots at once (one load per lane and slot, all in flight together) and a ballot names each edge.
        if (cnt_sel > 0) {
            const int64_t st = p.states[v];
            const int64_t e0 = sg.ptrs[w], e1 = sg.ptrs[w + 1];
            for (uint32_t s = lane; s < cnt_sel; s += 64) {
                const int64_t target = base_rank + (int64_t)ranks[s];
                int64_t lo = gfirst, hi = glast - 1; // last group g with gpref[g] <= target
                while (lo < hi) {
                    const int64_t mid = (lo + hi + 1) >> 1;
                    if (p.gpref[mid] <= target)
                        lo = mid;
                    else
                        hi = mid - 1;
                }
                uint32_t r = (uint32_t)(target - p.gpref[lo]);
                const uint64_t per_chunk = p.gchunk[lo];
                uint32_t u = 0;
#pragma unroll
                for (int c = 0; c < HS_CHUNKS - 1; ++c) {
                    const uint32_t cu = (uint32_t)(per_chunk >> (8 * c)) & 0xffu;
                    const bool next = (u == (uint32_t)c) && r >= cu;
                    r -= next ? cu : 0u;
                    u += next ? 1u : 0u;
                }
                keys[s] = (uint32_t)(lo - gfirst) * HS_CHUNKS + u; // chunk of the column
                vals[s] = r;                                        // rank among the chunk's admissible edges
            }
            wave_lds_handoff();
            for (uint32_t s0 = 0; s0 < cnt_sel; s0 += HS_LOCATE) {
                int64_t tsv[HS_LOCATE];
#pragma unroll
                for (int j = 0; j < HS_LOCATE; ++j) {
                    const uint32_t s = min(s0 + (uint32_t)j, cnt_sel - 1u);
                    const int64_t e = e0 + (int64_t)keys[s] * 64 + lane;
                    tsv[j] = (e < e1) ? sg.timestamps[e] : 0;
                }
#pragma unroll
                for (int j = 0; j < HS_LOCATE; ++j) {
                    const uint32_t s = s0 + (uint32_t)j;
                    if (s < cnt_sel) {
                        const int64_t e = e0 + (int64_t)keys[s] * 64 + lane;
                        const bool ok = e < e1 && hs_pass(p, st, tsv[j]);
                        const uint64_t mask = __ballot(ok);
                        if (ok && (uint32_t)__popcll(mask & lt_mask) == vals[s]) p.park[v * kmax + s] = e;
                    }
                }
            }
        }
        for (uint32_t s = cnt_sel + lane; s < (uint32_t)kmax; s += 64) p.park[v * kmax + s] = -1;
        wave_lds_handoff();
    }
}

// ---------------------------------------------------------------- weighted sampler (sampling.rs:28-55)
// The reference's weighted reservoir is a per-candidate algorithm over a running weight sum.  Columns are independent:
// one wavefront per frontier vertex, all over the device -- except LONG columns (> HW_LONG_EDGES), whose single wavefront
// used to set the time of the whole hop (245 us for the longest column of cfg4): with philox-mode's blocked running sum
// (tg_device.h) the 64-edge chunks of a column only meet in the left-to-right sum of their totals, so a whole workgroup
// takes such a column (hw_select_long_kernel).  status[0] |= 2 where the reference panics (running sum <= 0).
constexpr int64_t HW_LONG_EDGES = 16384;  // columns longer than this go to hw_select_long_kernel
constexpr int HW_LONG_BLOCKS = 256;       // its workgroups (each with its own scratch)
constexpr int64_t HW_GROUP_FORM_MIN = 1024; // group_cap from which the weighted sampler takes its group form
constexpr int64_t HW_LONG_CHUNKS = 8192;  // chunks of 64 edges its scratch holds per workgroup (512 K edges); beyond: one
                                          // wavefront walks the column as before
// the long columns of the frontier -> vgroups[1 ...], their number -> vgroups[0] (zeroed by the launcher)
__global__ void hw_list_long_kernel(const HopScanParams p) {
    __shared__ HsSeg S[HS_MAX_SEG];
    const int64_t m = hs_load_segs(p, S);
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < m; v += (int64_t)gridDim.x * blockDim.x) {
        const int64_t w = p.vertices[v];
        if (w < 0) continue;
        const HsSeg &sg = S[hs_seg_of(S, p.n_seg, v)];
        if (sg.ptrs[w + 1] - sg.ptrs[w] > HW_LONG_EDGES)
            p.vgroups[1 + atomicAdd(reinterpret_cast<unsigned long long *>(p.vgroups), 1ull)] = v;
    }
}

__device__ __forceinline__ void hw_select_short(const HopScanParams &p, unsigned char *smem, const HsSeg *S, int64_t m,
                                                int64_t block, int64_t n_blocks) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int kmax = p.k;
    int64_t *slot_ptr = reinterpret_cast<int64_t *>(smem) + (size_t)wave * (2 * kmax + 64);
    uint32_t *slot_rank = reinterpret_cast<uint32_t *>(slot_ptr + kmax);
    // (64 doubles behind slot_ptr used to hold the serial prefix; the blocked running sum needs no scratch)
    const int64_t wave_id = (block * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = (n_blocks * blockDim.x) >> 6;
    const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    for (int64_t v = wave_id; v < m; v += n_waves) {
        const int64_t w = p.vertices[v];
        const HsSeg &sg = S[hs_seg_of(S, p.n_seg, v)];
        const int k = sg.k;
        uint32_t n = 0;
        for (int sl = lane; sl < k; sl += 64) slot_rank[sl] = 0;
        wave_lds_handoff();
        if (w >= 0) {
            const int64_t st = p.states ? p.states[v] : 0;
            const int64_t e0 = sg.ptrs[w], e1 = sg.ptrs[w + 1];
            if (e1 - e0 > HW_LONG_EDGES) continue; // a whole workgroup takes it (hw_list_long_kernel listed it)
            const uint64_t did = p.ids ? (uint64_t)p.ids[v] : (uint64_t)(p.id_base + v);
            const CallKey ck = call_key(p.seed, p.call_ids ? (uint64_t)p.call_ids[v] : p.call_id, sg.tag);
            double w_sum = 0.0;
            for (int64_t gbase = e0; gbase < e1; gbase += 64 * HS_CHUNKS) {
                int64_t tsv[HS_CHUNKS];
                double wvv[HS_CHUNKS];
#pragma unroll
                for (int u = 0; u < HS_CHUNKS; ++u) {
                    const int64_t e = gbase + u * 64 + lane;
                    tsv[u] = (p.filter_mode != TG_FILTER_NONE && e < e1) ? __builtin_nontemporal_load(&sg.timestamps[e]) : 0;
                    wvv[u] = (e < e1) ? __builtin_nontemporal_load(&sg.weights[e]) : 0.0;
                }
#pragma unroll
                for (int u = 0; u < HS_CHUNKS; ++u) {
                    const int64_t base = gbase + u * 64;
                    if (base >= e1) break;
                    const int64_t e = base + lane;
                    const bool ok = e < e1 && hs_pass(p, st, tsv[u]);
                    const uint64_t mask = __ballot(ok);
                    const uint32_t rank = n + (uint32_t)__popcll(mask & lt_mask);
                    const double wv = ok ? wvv[u] : 0.0; // x + 0.0 == x: excluded edges leave the running sum alone
                    double tot;
                    const double pref = wave_blocked_prefix_f64(wv, w_sum, &tot); // blocked running sum, sampling.rs:40,48
                    w_sum = tot;
                    uint32_t hit_slot = 0xffffffffu;
                    if (ok && rank >= (uint32_t)k) {
                        if (!(0.0 < pref)) {
                            atomicOr(p.status, 2);
                        } else {
                            const Draw d = draw(ck, did, rank, D1_WEIGHTED);
                            const double j = u64_to_f64_01(d.a()) * pref + 0.0;
                            if (j < wv) hit_slot = (uint32_t)bounded64(d.b(), (uint64_t)k);
                        }
                    }
                    const bool fill = ok && rank < (uint32_t)k;
                    if (__ballot(fill || hit_slot != 0xffffffffu) != 0ull) {
                        if (fill) slot_ptr[rank] = e;
                        if (hit_slot != 0xffffffffu) atomicMax(&slot_rank[hit_slot], rank);
                        wave_lds_handoff();
                        if (hit_slot != 0xffffffffu && slot_rank[hit_slot] == rank) slot_ptr[hit_slot] = e; // last hit wins
                        wave_lds_handoff();
                    }
                    n += (uint32_t)__popcll(mask);
                }
            }
        }
        wave_lds_handoff();
        const uint32_t cnt_sel = min(n, (uint32_t)k);
        if (lane == 0) p.cnt[v] = cnt_sel;
        for (int sl = lane; sl < kmax; sl += 64) p.park[v * kmax + sl] = ((uint32_t)sl < cnt_sel) ? slot_ptr[sl] : -1;
        wave_lds_handoff();
    }
}

// One workgroup per long column: (A) every wavefront forms the weight totals and admissible counts of its chunks, (B) one
// lane turns them into carries (left to right: the defined order of the blocked sum) and candidate ranks, (C) every
// wavefront draws for its chunks -- candidates of rank < k fill slot rank, an accepted later candidate raises its slot to
// (rank, position) with a 64-bit atomic max, so that the LAST accepted candidate of a slot wins as in the reference's loop
// whatever order the chunks run in -- (D) the slots are read out.  Same draws (named by (call, vertex, rank)), same result.
__device__ __forceinline__ void hw_select_long(const HopScanParams &p, unsigned char *smem, const HsSeg *S,
                                               uint32_t &n_total_s, int64_t block, int64_t n_blocks) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n_waves = blockDim.x >> 6;
    const int kmax = p.k;
    int64_t *slot_ptr = reinterpret_cast<int64_t *>(smem);
    unsigned long long *slot_best = reinterpret_cast<unsigned long long *>(slot_ptr + kmax);
    double *tot = p.long_tot + (size_t)block * HW_LONG_CHUNKS;
    uint32_t *cnt = p.long_cnt + (size_t)block * HW_LONG_CHUNKS;
    const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const int64_t n_long = p.vgroups[0];
    for (int64_t li = block; li < n_long; li += n_blocks) {
        const int64_t v = p.vgroups[1 + li];
        const int64_t w = p.vertices[v];
        const HsSeg &sg = S[hs_seg_of(S, p.n_seg, v)];
        const int k = sg.k;
        const int64_t st = p.states ? p.states[v] : 0;
        const int64_t e0 = sg.ptrs[w], e1 = sg.ptrs[w + 1];
        const int64_t nc = (e1 - e0 + 63) >> 6;
        const uint64_t did = p.ids ? (uint64_t)p.ids[v] : (uint64_t)(p.id_base + v);
        const CallKey ck = call_key(p.seed, p.call_ids ? (uint64_t)p.call_ids[v] : p.call_id, sg.tag);
        const bool filtered = p.filter_mode != TG_FILTER_NONE;
        for (int sl = tid; sl < k; sl += blockDim.x) slot_best[sl] = 0ull;
        __syncthreads();
        auto chunk = [&](int64_t c, bool &ok, uint64_t &mask, double &wv, int64_t &e) {
            e = e0 + c * 64 + lane;
            const int64_t t = (filtered && e < e1) ? __builtin_nontemporal_load(&sg.timestamps[e]) : 0;
            const double x = (e < e1) ? __builtin_nontemporal_load(&sg.weights[e]) : 0.0;
            ok = e < e1 && hs_pass(p, st, t);
            mask = __ballot(ok);
            wv = ok ? x : 0.0; // x + 0.0 == x: excluded edges leave the running sum alone
        };
        if (nc > HW_LONG_CHUNKS) { // beyond the scratch: wavefront 0 walks the column chunk by chunk (the old form)
            if (wave == 0) {
                double w_sum = 0.0;
                uint32_t n = 0;
                for (int64_t c = 0; c < nc; ++c) {
                    bool ok;
                    uint64_t mask;
                    double wv, t2;
                    int64_t e;
                    chunk(c, ok, mask, wv, e);
                    const uint32_t rank = n + (uint32_t)__popcll(mask & lt_mask);
                    const double pref = wave_blocked_prefix_f64(wv, w_sum, &t2);
                    w_sum = t2;
                    if (ok && rank < (uint32_t)k) slot_ptr[rank] = e;
                    if (ok && rank >= (uint32_t)k) {
                        if (!(0.0 < pref)) {
                            atomicOr(p.status, 2);
                        } else {
                            const Draw d = draw(ck, did, rank, D1_WEIGHTED);
                            const double j = u64_to_f64_01(d.a()) * pref + 0.0;
                            if (j < wv)
                                atomicMax(&slot_best[bounded64(d.b(), (uint64_t)k)],
                                          ((unsigned long long)rank << 32) | (unsigned long long)(e - e0));
                        }
                    }
                    n += (uint32_t)__popcll(mask);
                }
                if (lane == 0) n_total_s = n;
            }
            __syncthreads();
        } else {
            constexpr int U = 4; // chunks whose loads a wavefront keeps in flight
            for (int64_t c0 = (int64_t)wave * U; c0 < nc; c0 += (int64_t)n_waves * U) { // (A) chunk totals and counts
                bool ok[U];
                uint64_t mask[U];
                double wv[U];
                int64_t e[U];
#pragma unroll
                for (int u = 0; u < U; ++u) chunk(min(c0 + u, nc - 1), ok[u], mask[u], wv[u], e[u]);
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (c0 + u >= nc) break; // uniform
                    double t2;
                    (void)wave_blocked_prefix_f64(wv[u], 0.0, &t2);
                    if (lane == 0) {
                        tot[c0 + u] = t2;
                        cnt[c0 + u] = (uint32_t)__popcll(mask[u]);
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __syncthreads();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            { // (B) carries and ranks before every chunk, added left to right: tiles of 1 024 chunks through LDS, where one
              // lane's dependent chain costs ~10 ns per chunk (in global memory every step of it waited for a load: 0.5 us
              // per chunk, 0.8 ms for a column of 10^5 edges -- the whole call's time)
                __shared__ double tile_t[1024];
                __shared__ uint32_t tile_c[1024];
                __shared__ double run_s;
                __shared__ uint32_t n_s;
                if (tid == 0) {
                    run_s = 0.0;
                    n_s = 0;
                }
                for (int64_t c0 = 0; c0 < nc; c0 += 1024) {
                    const int64_t c = c0 + tid;
                    if (tid < 1024 && c < nc) {
                        tile_t[tid] = tot[c];
                        tile_c[tid] = cnt[c];
                    }
                    __syncthreads();
                    if (tid == 0) {
                        double run = run_s;
                        uint32_t n = n_s;
                        const int len = (int)min((int64_t)1024, nc - c0);
                        for (int i = 0; i < len; ++i) {
                            const double t2 = tile_t[i];
                            const uint32_t q = tile_c[i];
                            tile_t[i] = run;
                            tile_c[i] = n;
                            run = run + t2;
                            n += q;
                        }
                        run_s = run;
                        n_s = n;
                    }
                    __syncthreads();
                    if (tid < 1024 && c < nc) {
                        tot[c] = tile_t[tid];
                        cnt[c] = tile_c[tid];
                    }
                    __syncthreads();
                }
                if (tid == 0) n_total_s = n_s;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __syncthreads();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            for (int64_t c0 = (int64_t)wave * U; c0 < nc; c0 += (int64_t)n_waves * U) { // (C) draws
                bool ok[U];
                uint64_t mask[U];
                double wv[U], carry[U];
                int64_t e[U];
                uint32_t before[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int64_t c = min(c0 + u, nc - 1);
                    chunk(c, ok[u], mask[u], wv[u], e[u]);
                    carry[u] = tot[c];
                    before[u] = cnt[c];
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (c0 + u >= nc) break; // uniform
                    double t2;
                    const uint32_t rank = before[u] + (uint32_t)__popcll(mask[u] & lt_mask);
                    const double pref = wave_blocked_prefix_f64(wv[u], carry[u], &t2); // blocked running sum, sampling.rs:40,48
                    if (ok[u] && rank < (uint32_t)k) slot_ptr[rank] = e[u]; // sampling.rs:37-45
                    if (ok[u] && rank >= (uint32_t)k) {
                        if (!(0.0 < pref)) {
                            atomicOr(p.status, 2);
                        } else {
                            const Draw d = draw(ck, did, rank, D1_WEIGHTED);
                            const double j = u64_to_f64_01(d.a()) * pref + 0.0;
                            if (j < wv[u])
                                atomicMax(&slot_best[bounded64(d.b(), (uint64_t)k)],
                                          ((unsigned long long)rank << 32) | (unsigned long long)(e[u] - e0));
                        }
                    }
                }
            }
            __syncthreads();
        }
        const uint32_t cnt_sel = min(n_total_s, (uint32_t)k); // (D)  (rank >= k >= 1: a slot_best of 0 = never hit)
        if (tid == 0) p.cnt[v] = cnt_sel;
        for (int sl = tid; sl < kmax; sl += blockDim.x) {
            int64_t ep = -1;
            if ((uint32_t)sl < cnt_sel) ep = slot_best[sl] ? e0 + (int64_t)(slot_best[sl] & 0xffffffffull) : slot_ptr[sl];
            p.park[v * kmax + sl] = ep;
        }
        __syncthreads();
    }
}

__global__ void hw_select_kernel(const HopScanParams p) { // the short columns only (any workgroup size)
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ HsSeg S[HS_MAX_SEG];
    const int64_t m = hs_load_segs(p, S);
    hw_select_short(p, smem, S, m, blockIdx.x, gridDim.x);
}
__global__ void __launch_bounds__(1024) hw_select_long_kernel(const HopScanParams p) { // the long columns only
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ HsSeg S[HS_MAX_SEG];
    __shared__ uint32_t n_total_s;
    hs_load_segs(p, S);
    hw_select_long(p, smem, S, n_total_s, blockIdx.x, gridDim.x);
}
// Both in ONE launch of 1 024-thread workgroups: the first HW_LONG_BLOCKS take the long columns, the others the short ones,
// side by side -- the long columns' two passes (2 x 100 us on cfg4) no longer wait for the short ones to finish.
__global__ void __launch_bounds__(1024) hw_select_all_kernel(const HopScanParams p) {
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ HsSeg S[HS_MAX_SEG];
    __shared__ uint32_t n_total_s;
    const int64_t m = hs_load_segs(p, S);
    if (blockIdx.x < HW_LONG_BLOCKS)
        hw_select_long(p, smem, S, n_total_s, blockIdx.x, HW_LONG_BLOCKS);
    else
        hw_select_short(p, smem, S, m, (int64_t)blockIdx.x - HW_LONG_BLOCKS, (int64_t)gridDim.x - HW_LONG_BLOCKS);
}

// thread per (frontier vertex, slot): the slot's output position is offsets[v] + s -- no search, two rounds of loads
__global__ void hs_emit_kernel(const HopScanParams p) {
    __shared__ HsSeg S[HS_MAX_SEG];
    const int64_t m = hs_load_segs(p, S);
    const int64_t n = m * p.k;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t v = t / p.k, s = t - v * p.k;
        if (s >= p.cnt[v]) continue;
        const int64_t o = p.offsets[v] + s;
        const int64_t ep = p.park[t];
        const HsSeg &sg = S[hs_seg_of(S, p.n_seg, v)];
        p.neighbors[o] = sg.indices[ep];
        p.edge_ptrs[o] = ep;
        p.parents[o] = v;
        if (p.filter_mode != TG_FILTER_NONE)
            p.states_out[o] = (p.filter_mode == TG_FILTER_DYNAMIC) ? sg.timestamps[ep] : p.states[v]; // :69-76
    }
}


// ---------------------------------------------------------------- weighted sampler, GROUP FORM
// The same algorithm with the work cut like the filtered path's: a hub column of 10^5 edges no longer sits on one wavefront
// (or one workgroup) while the device idles.  (1) hw_gtotals: FLAT over the 512-edge groups of all frontier columns --
// admissible counts and the blocked sum's chunk totals; (2) hw_gcarry: a wavefront per vertex turns its chunk totals into
// the carries before every chunk (left to right: the defined order) and its group counts into candidate ranks, sets the
// vertex's count and clears its slots; (3) hw_gdraw: FLAT over the groups again -- candidates of rank < k fill slot rank,
// an accepted later candidate raises its slot to (rank, position) with a 64-bit atomic max: the LAST accepted candidate of
// a slot wins, as in the reference's loop, whatever order the groups run in; (4) hw_gresolve reads the slots out.  Same
// draws (named by (call, vertex, rank)), same running sums, same result as the column-at-a-time kernels above.
template <typename Body>
__device__ __forceinline__ void hs_for_groups(const HopScanParams &p, const HsSeg *S, int64_t m, Body &&body) {
    const int64_t wave_id = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int64_t gt = p.vgroups[m];
    if (gt >= p.group_cap) return;
    const int64_t run = max((int64_t)1, min((int64_t)HS_RUN, (gt + n_waves - 1) / n_waves));
    for (int64_t g0 = wave_id * run; g0 < gt; g0 += n_waves * run) {
        int64_t lo = 0, hi = m - 1; // vertex owning group g0: last v with vgroups[v] <= g0
        while (lo < hi) {
            const int64_t mid = (lo + hi + 1) >> 1;
            if (p.vgroups[mid] <= g0)
                lo = mid;
            else
                hi = mid - 1;
        }
        int64_t v = lo;
        const int64_t g1 = min(g0 + run, gt);
        for (int64_t g = g0; g < g1; ++g) {
            while (g >= p.vgroups[v + 1]) ++v; // skip to the owner (vertices without groups own nothing)
            body(g, v, S[hs_seg_of(S, p.n_seg, v)]);
        }
    }
}
__global__ void hw_gtotals_kernel(const HopScanParams p) {
    __shared__ HsSeg S[HS_MAX_SEG];
    const int64_t m = hs_load_segs(p, S);
    const int lane = threadIdx.x & 63;
    const bool filtered = p.filter_mode != TG_FILTER_NONE;
    hs_for_groups(p, S, m, [&](int64_t g, int64_t v, const HsSeg &sg) {
        const int64_t w = p.vertices[v];
        const int64_t st = p.states ? p.states[v] : 0;
        const int64_t e1 = sg.ptrs[w + 1];
        const int64_t gb = sg.ptrs[w] + (g - p.vgroups[v]) * HS_GROUP;
        int64_t tsv[HS_CHUNKS];
        double wvv[HS_CHUNKS];
#pragma unroll
        for (int u = 0; u < HS_CHUNKS; ++u) {
            const int64_t e = gb + u * 64 + lane;
            tsv[u] = (filtered && e < e1) ? __builtin_nontemporal_load(&sg.timestamps[e]) : 0;
            wvv[u] = (e < e1) ? __builtin_nontemporal_load(&sg.weights[e]) : 0.0;
        }
        uint32_t c = 0;
        uint64_t per_chunk = 0;
        double mine = 0.0; // lane u keeps chunk u's total
#pragma unroll
        for (int u = 0; u < HS_CHUNKS; ++u) {
            const int64_t e = gb + u * 64 + lane;
            const bool ok = e < e1 && hs_pass(p, st, tsv[u]);
            const uint32_t cu = (uint32_t)__popcll(__ballot(ok));
            c += cu;
            per_chunk |= (uint64_t)cu << (8 * u);
            double tot;
            (void)wave_blocked_prefix_f64(ok ? wvv[u] : 0.0, 0.0, &tot); // x + 0.0 == x: excluded edges add nothing
            if (lane == u) mine = tot;
        }
        if (lane < HS_CHUNKS) p.gtot[g * HS_CHUNKS + lane] = mine;
        if (lane == 0) {
            p.gcount[g] = c;
            p.gchunk[g] = per_chunk;
        }
    });
}
// wavefront per vertex
__global__ void hw_gcarry_kernel(const HopScanParams p) {
    __shared__ HsSeg S[HS_MAX_SEG];
    __shared__ double chain_s[4][64];
    const int64_t m = hs_load_segs(p, S);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t wave_id = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const bool overflow = p.vgroups[m] >= p.group_cap;
    const int kmax = p.k;
    for (int64_t v = wave_id; v < m; v += n_waves) {
        const HsSeg &sg = S[hs_seg_of(S, p.n_seg, v)];
        const int64_t g0 = p.vgroups[v], g1 = overflow ? g0 : p.vgroups[v + 1];
        for (int sl = lane; sl < kmax; sl += 64) p.slot_best[v * kmax + sl] = 0ull;
        if (lane == 0) {
            const CallKey ck = call_key(p.seed, p.call_ids ? (uint64_t)p.call_ids[v] : p.call_id, sg.tag);
            p.vck[2 * v] = ck.k0;
            p.vck[2 * v + 1] = ck.k1;
        }
        double carry = 0.0;
        const int64_t c_begin = g0 * HS_CHUNKS, c_end = g1 * HS_CHUNKS;
        for (int64_t c0 = c_begin; c0 < c_end; c0 += 64) { // carries before the chunks: the totals added left to right
            const int64_t c = c0 + lane;
            const double t = c < c_end ? p.gtot[c] : 0.0;
            double total;
            const double incl = wave_serial_prefix_f64(t, carry, &total, chain_s[wave]);
            const double before = __shfl_up(incl, 1, 64);
            if (c < c_end) p.gtot[c] = lane == 0 ? carry : before;
            carry = total;
        }
        uint32_t n = 0;
        for (int64_t gq = g0; gq < g1; gq += 64) { // candidates before every group
            const int64_t g = gq + lane;
            const uint32_t cg = g < g1 ? p.gcount[g] : 0u;
            const uint32_t incl = wave_inclusive_scan(cg);
            if (g < g1) p.grank[g] = n + incl - cg;
            n += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        }
        if (lane == 0) p.cnt[v] = min(n, (uint32_t)sg.k);
    }
}
__global__ void hw_gdraw_kernel(const HopScanParams p) {
    __shared__ HsSeg S[HS_MAX_SEG];
    const int64_t m = hs_load_segs(p, S);
    const int lane = threadIdx.x & 63;
    const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const bool filtered = p.filter_mode != TG_FILTER_NONE;
    const int kmax = p.k;
    hs_for_groups(p, S, m, [&](int64_t g, int64_t v, const HsSeg &sg) {
        const int64_t w = p.vertices[v];
        const int64_t st = p.states ? p.states[v] : 0;
        const int k = sg.k;
        const int64_t e0 = sg.ptrs[w], e1 = sg.ptrs[w + 1];
        const int64_t gb = e0 + (g - p.vgroups[v]) * HS_GROUP;
        const uint64_t did = p.ids ? (uint64_t)p.ids[v] : (uint64_t)(p.id_base + v);
        const CallKey ck{p.vck[2 * v], p.vck[2 * v + 1]};
        int64_t tsv[HS_CHUNKS];
        double wvv[HS_CHUNKS];
#pragma unroll
        for (int u = 0; u < HS_CHUNKS; ++u) {
            const int64_t e = gb + u * 64 + lane;
            tsv[u] = (filtered && e < e1) ? __builtin_nontemporal_load(&sg.timestamps[e]) : 0;
            wvv[u] = (e < e1) ? __builtin_nontemporal_load(&sg.weights[e]) : 0.0;
        }
        const double my_carry = lane < HS_CHUNKS ? p.gtot[g * HS_CHUNKS + lane] : 0.0;
        uint32_t n = p.grank[g];
#pragma unroll
        for (int u = 0; u < HS_CHUNKS; ++u) {
            if (gb + u * 64 >= e1) break; // uniform
            const int64_t e = gb + u * 64 + lane;
            const bool ok = e < e1 && hs_pass(p, st, tsv[u]);
            const uint64_t mask = __ballot(ok);
            const uint32_t rank = n + (uint32_t)__popcll(mask & lt_mask);
            const double wv = ok ? wvv[u] : 0.0;
            double tot;
            const double pref = wave_blocked_prefix_f64(wv, __shfl(my_carry, u, 64), &tot); // blocked running sum, sampling.rs:40,48
            if (ok && rank < (uint32_t)k) p.park[v * kmax + rank] = e; // sampling.rs:37-45
            if (ok && rank >= (uint32_t)k) {
                if (!(0.0 < pref)) {
                    atomicOr(p.status, 2);
                } else {
                    const Draw d = draw(ck, did, rank, D1_WEIGHTED);
                    const double j = u64_to_f64_01(d.a()) * pref + 0.0;
                    if (j < wv)
                        atomicMax(&p.slot_best[v * kmax + (int64_t)bounded64(d.b(), (uint64_t)k)],
                                  ((unsigned long long)rank << 32) | (unsigned long long)(e - e0));
                }
            }
            n += (uint32_t)__popcll(mask);
        }
    });
}
// one lane per (vertex, slot)  (rank >= k >= 1: a slot_best of 0 = never hit)
__global__ void hw_gresolve_kernel(const HopScanParams p) {
    __shared__ HsSeg S[HS_MAX_SEG];
    const int64_t m = hs_load_segs(p, S);
    const int kmax = p.k;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < m * kmax; q += (int64_t)gridDim.x * blockDim.x) {
        const int64_t v = q / kmax;
        const int sl = (int)(q - v * kmax);
        if ((uint32_t)sl >= (uint32_t)p.cnt[v]) {
            p.park[q] = -1;
            continue;
        }
        const unsigned long long sb = p.slot_best[q];
        if (sb) p.park[q] = S[hs_seg_of(S, p.n_seg, v)].ptrs[p.vertices[v]] + (int64_t)(sb & 0xffffffffull);
    }
}

static inline size_t hs_align(size_t x) { return (x + 255) & ~(size_t)255; }
static size_t hs_scan_temp(int64_t n) {
    size_t a = 0, b = 0;
    (void)rocprim::inclusive_scan(nullptr, a, (int64_t *)nullptr, (int64_t *)nullptr, (size_t)(n > 0 ? n : 1),
                                  rocprim::plus<int64_t>(), (hipStream_t)0, false);
    (void)rocprim::exclusive_scan(nullptr, b, (uint32_t *)nullptr, (int64_t *)nullptr, (int64_t)0, (size_t)(n > 0 ? n : 1),
                                  rocprim::plus<int64_t>(), (hipStream_t)0, false);
    return a > b ? a : b;
}

} // namespace tg

extern "C" int tg_ns_hop_scan_workspace_bytes(int64_t m, int32_t fanout, int64_t group_cap, int64_t *bytes) {
    TG_REQUIRE(m >= 0 && fanout >= 1 && group_cap >= 1 && bytes, "tg_ns_hop_scan_workspace_bytes: bad arguments");
    using namespace tg;
    const int64_t big = group_cap + 1 > m + 1 ? group_cap + 1 : m + 1;
    *bytes = (int64_t)(hs_align(8 * (size_t)(m + 1)) + hs_align(4 * (size_t)group_cap) + hs_align(8 * (size_t)group_cap) +
                       hs_align(8 * (size_t)(group_cap + 1)) +
                       hs_align(8 * (size_t)(m > 0 ? m : 1) * fanout) + hs_align(hs_scan_temp(big)) + 512 +
                       hs_align(12 * (size_t)HW_LONG_BLOCKS * (size_t)HW_LONG_CHUNKS)); // long columns of the weighted sampler
    return TG_OK;
}
// the weighted sampler's group form needs 64 more bytes per group (chunk totals), ranks before the groups, slots and call
// keys -- and no group prefix; sized apart so that the filtered hop's workspace stays at 20 bytes per group
extern "C" int tg_ns_hop_weighted_workspace_bytes(int64_t m, int32_t fanout, int64_t group_cap, int64_t *bytes) {
    TG_REQUIRE(m >= 0 && fanout >= 1 && group_cap >= 1 && bytes, "tg_ns_hop_weighted_workspace_bytes: bad arguments");
    using namespace tg;
    int64_t base = 0;
    const int rc = tg_ns_hop_scan_workspace_bytes(m, fanout, group_cap, &base);
    if (rc != TG_OK) return rc;
    *bytes = base + (int64_t)(hs_align(8 * (size_t)group_cap * HS_CHUNKS) + hs_align(4 * (size_t)group_cap) +
                              hs_align(8 * (size_t)(m > 0 ? m : 1) * fanout) + hs_align(8 * (size_t)(m > 0 ? m : 1)));
    return TG_OK;
}

// ---------------------------------------------------------------- host side: one launcher for every entry point
namespace tg {
struct HsCall {
    HsSeg seg[HS_MAX_SEG];
    int n_seg;
    int kmax;
    bool weighted;
    const int64_t *layout_dev;
};

static int hs_run(const char *who, const HsCall &c, const tg_hop_in *in, const tg_hop_filter *flt, const tg_rng *rng,
                  const tg_hop_out *out, int64_t *states_out, int32_t *status, void *workspace, int64_t workspace_bytes,
                  int64_t group_cap, hipStream_t stream) {
    const int filter_mode = flt ? flt->filter_mode : TG_FILTER_NONE;
    TG_REQUIRE(in->m >= 0 && c.kmax >= 1 && c.kmax <= HS_MAX_FANOUT, "%s: bad frontier size or fan-out (<= %d)", who,
               HS_MAX_FANOUT);
    TG_REQUIRE(out->cnt && out->offsets && group_cap >= 1, "%s: null outputs", who);
    if (in->m == 0) {
        TG_HIP(hipMemsetAsync(out->offsets, 0, sizeof(int64_t), stream));
        return TG_OK;
    }
    // the weighted sampler takes the GROUP FORM whenever the caller sized the workspace for the frontier's groups (the
    // same bound as the filtered path's; reached: status |= 1, all counts 0, retry with more) -- with the historical
    // group_cap = 1 it keeps the column-at-a-time kernels
    const bool weighted_groups = c.weighted && group_cap >= HW_GROUP_FORM_MIN;
    int64_t need = 0;
    int rc = weighted_groups ? tg_ns_hop_weighted_workspace_bytes(in->m, c.kmax, group_cap, &need)
                             : tg_ns_hop_scan_workspace_bytes(in->m, c.kmax, group_cap, &need);
    if (rc != TG_OK) return rc;
    TG_REQUIRE(workspace && workspace_bytes >= need, "%s: workspace too small", who);
    TG_REQUIRE(in->vertices && out->neighbors && out->edge_ptrs && out->parents, "%s: null buffers", who);
    HopScanParams p{};
    for (int j = 0; j < c.n_seg; ++j) p.seg[j] = c.seg[j];
    p.n_seg = c.n_seg;
    p.layout_dev = c.layout_dev;
    p.vertices = in->vertices;
    p.states = filter_mode == TG_FILTER_NONE ? nullptr : flt->states;
    p.ids = in->ids;
    p.call_ids = in->call_ids;
    p.m = in->m;
    p.id_base = in->id_base;
    p.k = c.kmax;
    p.replace = in->sampler == TG_SAMPLER_UNIFORM_REPL;
    p.filter_mode = filter_mode;
    p.forward = flt ? flt->forward : 0;
    p.win_lo = flt ? flt->win_lo : 0;
    p.win_hi = flt ? flt->win_hi : 0;
    p.seed = rng->seed;
    p.call_id = rng->call_id;
    p.group_cap = group_cap;
    unsigned char *base = reinterpret_cast<unsigned char *>(workspace);
    size_t off = 0;
    auto take = [&](size_t bytes) {
        unsigned char *q = base + off;
        off += hs_align(bytes);
        return q;
    };
    p.vgroups = reinterpret_cast<int64_t *>(take(8 * (size_t)(p.m + 1)));
    p.gcount = reinterpret_cast<uint32_t *>(take(4 * (size_t)group_cap));
    p.gchunk = reinterpret_cast<uint64_t *>(take(8 * (size_t)group_cap));
    p.gpref = reinterpret_cast<int64_t *>(take(8 * (size_t)(group_cap + 1)));
    p.park = reinterpret_cast<int64_t *>(take(8 * (size_t)p.m * p.k));
    {
        unsigned char *ls = take(12 * (size_t)HW_LONG_BLOCKS * (size_t)HW_LONG_CHUNKS);
        p.long_tot = reinterpret_cast<double *>(ls);
        p.long_cnt = reinterpret_cast<uint32_t *>(ls + 8 * (size_t)HW_LONG_BLOCKS * (size_t)HW_LONG_CHUNKS);
    }
    if (weighted_groups) {
        p.gtot = reinterpret_cast<double *>(take(8 * (size_t)group_cap * HS_CHUNKS));
        p.grank = reinterpret_cast<uint32_t *>(take(4 * (size_t)group_cap));
        p.slot_best = reinterpret_cast<unsigned long long *>(take(8 * (size_t)p.m * p.k));
        p.vck = reinterpret_cast<uint32_t *>(take(8 * (size_t)p.m));
    }
    void *temp = base + off;
    size_t temp_bytes = (size_t)workspace_bytes - off;
    p.status = status;
    p.cnt = out->cnt;
    p.offsets = out->offsets;
    p.neighbors = out->neighbors;
    p.edge_ptrs = out->edge_ptrs;
    p.parents = out->parents;
    p.states_out = states_out;

    auto grid = [](int64_t n, int threads) {
        int64_t g = (n + threads - 1) / threads;
        if (g < 1) g = 1;
        if (g > 256 * 32) g = 256 * 32;
        return dim3((unsigned)g);
    };
    // short frontiers (the per-call operators): every prefix sum is one single-workgroup launch and nothing is memset
    const bool short_m = p.m <= SCAN1_MAX, short_g = group_cap <= SCAN1_GROUPS_MAX;
    TG_REQUIRE(!c.layout_dev || (short_m && (c.weighted || short_g)),
               "%s: a device-side layout needs a frontier bound <= %lld and a group bound <= %lld", who, (long long)SCAN1_MAX,
               (long long)SCAN1_GROUPS_MAX);
    size_t st = temp_bytes;
    if (weighted_groups) {
        if (short_m) {
            hipLaunchKernelGGL(hs_groups1_kernel, dim3(1), dim3(SCAN1_THREADS), 0, stream, p);
        } else {
            hipLaunchKernelGGL(hs_groups_kernel, grid(p.m, 256), dim3(256), 0, stream, p);
            TG_HIP(rocprim::inclusive_scan(temp, st, p.vgroups + 1, p.vgroups + 1, (size_t)p.m, rocprim::plus<int64_t>(),
                                           stream, false));
            hipLaunchKernelGGL(hs_check_kernel, dim3(1), dim3(64), 0, stream, p);
        }
        hipLaunchKernelGGL(hw_gtotals_kernel, dim3(256 * 8), dim3(256), 0, stream, p);
        hipLaunchKernelGGL(hw_gcarry_kernel, grid(p.m * 64, 256), dim3(256), 0, stream, p);
        hipLaunchKernelGGL(hw_gdraw_kernel, dim3(256 * 8), dim3(256), 0, stream, p);
        hipLaunchKernelGGL(hw_gresolve_kernel, grid(p.m * p.k, 256), dim3(256), 0, stream, p);
    } else if (c.weighted) {
        int n_waves = 4;
        while (n_waves > 1 && (size_t)n_waves * (2 * p.k + 64) * sizeof(int64_t) > 60 * 1024) n_waves >>= 1;
        const size_t lds = (size_t)n_waves * (2 * p.k + 64) * sizeof(int64_t);
        int64_t blocks = (p.m + n_waves - 1) / n_waves;
        if (blocks > 256 * 32) blocks = 256 * 32;
        TG_HIP(hipMemsetAsync(p.vgroups, 0, sizeof(int64_t), stream)); // the list of long columns: [0] = how many
        hipLaunchKernelGGL(hw_list_long_kernel, grid(p.m, 256), dim3(256), 0, stream, p);
        const size_t lds_all = std::max((size_t)16 * (2 * p.k + 64) * sizeof(int64_t), (size_t)p.k * 16);
        if (lds_all <= 60 * 1024) { // both roles in one launch
            int64_t short_blocks = (p.m + 15) / 16;
            if (short_blocks > 256 * 8) short_blocks = 256 * 8;
            hipLaunchKernelGGL(hw_select_all_kernel, dim3((unsigned)(HW_LONG_BLOCKS + short_blocks)), dim3(1024), lds_all,
                               stream, p);
        } else {
            hipLaunchKernelGGL(hw_select_kernel, dim3((unsigned)blocks), dim3(64 * n_waves), lds, stream, p);
            hipLaunchKernelGGL(hw_select_long_kernel, dim3(HW_LONG_BLOCKS), dim3(1024), (size_t)p.k * 16, stream, p);
        }
    } else {
        if (short_m) {
            hipLaunchKernelGGL(hs_groups1_kernel, dim3(1), dim3(SCAN1_THREADS), 0, stream, p);
        } else {
            hipLaunchKernelGGL(hs_groups_kernel, grid(p.m, 256), dim3(256), 0, stream, p);
            TG_HIP(rocprim::inclusive_scan(temp, st, p.vgroups + 1, p.vgroups + 1, (size_t)p.m, rocprim::plus<int64_t>(),
                                           stream, false));
            hipLaunchKernelGGL(hs_check_kernel, dim3(1), dim3(64), 0, stream, p);
        }
        if (!short_g) TG_HIP(hipMemsetAsync(p.gcount, 0, 4 * (size_t)group_cap, stream)); // groups beyond the frontier's count as empty
        hipLaunchKernelGGL(hs_count_kernel, dim3(256 * 8), dim3(256), 0, stream, p);
        if (short_g) {
            hipLaunchKernelGGL(hs_gscan1_kernel, dim3(1), dim3(SCAN1_THREADS), 0, stream, p);
        } else {
            st = temp_bytes;
            TG_HIP(rocprim::exclusive_scan(temp, st, p.gcount, p.gpref, (int64_t)0, (size_t)group_cap,
                                           rocprim::plus<int64_t>(), stream, false));
        }
        // the frontier uses fewer than group_cap groups (else status = 1), so gpref[n_groups] is inside the scanned range
        hipLaunchKernelGGL(hs_select_kernel, grid(p.m * 64, 256), dim3(256), (size_t)4 * 3 * p.k * sizeof(uint32_t), stream,
                           p);
    }
    if (short_m) {
        hipLaunchKernelGGL(hs_oscan1_kernel, dim3(1), dim3(SCAN1_THREADS), 0, stream, p);
    } else {
        TG_HIP(hipMemsetAsync(p.offsets, 0, 8, stream));
        st = temp_bytes;
        TG_HIP(rocprim::inclusive_scan(temp, st, p.cnt, p.offsets + 1, (size_t)p.m, rocprim::plus<int64_t>(), stream, false));
    }
    hipLaunchKernelGGL(hs_emit_kernel, grid(p.m * p.k, 256), dim3(256), 0, stream, p);
    TG_LAUNCH_CHECK();
    return TG_OK;
}
} // namespace tg

extern "C" int tg_ns_hop_scan(const tg_graph *csc, const tg_hop_in *in, const tg_hop_filter *flt, const tg_rng *rng,
                              const tg_hop_out *out, int64_t *states_out, int32_t *status, void *workspace,
                              int64_t workspace_bytes, int64_t group_cap, void *stream_) {
    using namespace tg;
    TG_REQUIRE(csc && csc->ptrs && in && flt && rng && out && status, "tg_ns_hop_scan: null argument");
    TG_REQUIRE(csc->timestamps, "tg_ns_hop_scan: the graph has no edge timestamps");
    TG_REQUIRE(in->sampler == TG_SAMPLER_UNIFORM || in->sampler == TG_SAMPLER_UNIFORM_REPL,
               "tg_ns_hop_scan: only the unweighted samplers");
    TG_REQUIRE(flt->filter_mode >= TG_FILTER_STATIC && flt->filter_mode <= TG_FILTER_DYNAMIC, "tg_ns_hop_scan: bad filter");
    TG_REQUIRE(in->m == 0 || (flt->states && states_out), "tg_ns_hop_scan: null buffers");
    HsCall c{};
    c.seg[0] = HsSeg{csc->ptrs, csc->indices, csc->timestamps, nullptr, 0, in->fanout, in->rng_tag ? in->rng_tag : TG_TAG_NS_HOMO};
    c.n_seg = 1;
    c.kmax = in->fanout;
    c.weighted = false;
    return hs_run("tg_ns_hop_scan", c, in, flt, rng, out, states_out, status, workspace, workspace_bytes, group_cap,
                  (hipStream_t)stream_);
}

extern "C" int tg_ns_hop_weighted(const tg_graph *csc, const tg_hop_in *in, const tg_hop_filter *flt, const tg_rng *rng,
                                  const tg_hop_out *out, int64_t *states_out, int32_t *status, void *workspace,
                                  int64_t workspace_bytes, void *stream_) {
    using namespace tg;
    TG_REQUIRE(csc && csc->ptrs && in && rng && out && status, "tg_ns_hop_weighted: null argument");
    TG_REQUIRE(csc->weights, "tg_ns_hop_weighted: the graph has no edge weights");
    const int filter_mode = flt ? flt->filter_mode : TG_FILTER_NONE;
    TG_REQUIRE(filter_mode >= TG_FILTER_NONE && filter_mode <= TG_FILTER_DYNAMIC, "tg_ns_hop_weighted: bad filter");
    TG_REQUIRE(filter_mode == TG_FILTER_NONE || (csc->timestamps && flt->states && states_out),
               "tg_ns_hop_weighted: the filter needs edge timestamps and states");
    HsCall c{};
    c.seg[0] = HsSeg{csc->ptrs, csc->indices, csc->timestamps, csc->weights, 0, in->fanout, in->rng_tag ? in->rng_tag : TG_TAG_NS_HOMO};
    c.n_seg = 1;
    c.kmax = in->fanout;
    c.weighted = true;
    return hs_run("tg_ns_hop_weighted", c, in, flt, rng, out, states_out, status, workspace, workspace_bytes, 1,
                  (hipStream_t)stream_);
}
// the same hop in the weighted sampler's GROUP FORM (chunk totals and draws flat over the 512-edge groups of all columns)
extern "C" int tg_ns_hop_weighted_groups(const tg_graph *csc, const tg_hop_in *in, const tg_hop_filter *flt, const tg_rng *rng,
                                         const tg_hop_out *out, int64_t *states_out, int32_t *status, void *workspace,
                                         int64_t workspace_bytes, int64_t group_cap, void *stream_) {
    using namespace tg;
    TG_REQUIRE(csc && csc->ptrs && in && rng && out && status, "tg_ns_hop_weighted_groups: null argument");
    TG_REQUIRE(csc->weights, "tg_ns_hop_weighted_groups: the graph has no edge weights");
    const int filter_mode = flt ? flt->filter_mode : TG_FILTER_NONE;
    TG_REQUIRE(filter_mode >= TG_FILTER_NONE && filter_mode <= TG_FILTER_DYNAMIC, "tg_ns_hop_weighted_groups: bad filter");
    TG_REQUIRE(filter_mode == TG_FILTER_NONE || (csc->timestamps && flt->states && states_out),
               "tg_ns_hop_weighted_groups: the filter needs edge timestamps and states");
    HsCall c{};
    c.seg[0] = HsSeg{csc->ptrs, csc->indices, csc->timestamps, csc->weights, 0, in->fanout, in->rng_tag ? in->rng_tag : TG_TAG_NS_HOMO};
    c.n_seg = 1;
    c.kmax = in->fanout;
    c.weighted = true;
    return hs_run("tg_ns_hop_weighted_groups", c, in, flt, rng, out, states_out, status, workspace, workspace_bytes, group_cap,
                  (hipStream_t)stream_);
}

extern "C" int tg_ns_hop_segments(const tg_hop_segment *segments, int32_t n_segments, const tg_hop_in *in,
                                  const int64_t *layout_dev, const tg_hop_filter *flt, const tg_rng *rng,
                                  const tg_hop_out *out, int64_t *states_out, int32_t *status, void *workspace,
                                  int64_t workspace_bytes, int64_t group_cap, void *stream_) {
    using namespace tg;
    TG_REQUIRE(segments && in && rng && out && status, "tg_ns_hop_segments: null argument");
    TG_REQUIRE(n_segments >= 1 && n_segments <= HS_MAX_SEG, "tg_ns_hop_segments: 1 .. %d segments", HS_MAX_SEG);
    const bool weighted = in->sampler == TG_SAMPLER_WEIGHTED;
    const int filter_mode = flt ? flt->filter_mode : TG_FILTER_NONE;
    TG_REQUIRE(filter_mode >= TG_FILTER_NONE && filter_mode <= TG_FILTER_DYNAMIC, "tg_ns_hop_segments: bad filter");
    TG_REQUIRE(weighted || filter_mode != TG_FILTER_NONE,
               "tg_ns_hop_segments: the unweighted samplers come here under a temporal filter only");
    TG_REQUIRE(filter_mode == TG_FILTER_NONE || in->m == 0 || (flt->states && states_out),
               "tg_ns_hop_segments: the filter needs states");
    HsCall c{};
    c.n_seg = n_segments;
    c.weighted = weighted;
    c.layout_dev = layout_dev;
    for (int j = 0; j < n_segments; ++j) {
        const tg_hop_segment &g = segments[j];
        TG_REQUIRE(g.graph && g.graph->ptrs, "tg_ns_hop_segments: segment %d has no graph", j);
        TG_REQUIRE(g.begin >= 0 && g.begin <= in->m && (j == 0 ? g.begin == 0 : g.begin >= segments[j - 1].begin),
                   "tg_ns_hop_segments: segment starts are ascending from 0");
        TG_REQUIRE(g.fanout >= 1 && g.fanout <= HS_MAX_FANOUT, "tg_ns_hop_segments: bad fan-out in segment %d", j);
        const bool no_edges = g.graph->n_edges == 0; // an empty relation: its per-edge arrays are never touched
        TG_REQUIRE(!weighted || g.graph->weights || no_edges, "tg_ns_hop_segments: segment %d has no edge weights", j);
        TG_REQUIRE(filter_mode == TG_FILTER_NONE || g.graph->timestamps || no_edges,
                   "tg_ns_hop_segments: segment %d has no edge timestamps", j);
        c.seg[j] = HsSeg{g.graph->ptrs, g.graph->indices, g.graph->timestamps, g.graph->weights, g.begin, g.fanout,
                         g.rng_tag ? g.rng_tag : TG_TAG_NS_HOMO};
        if (g.fanout > c.kmax) c.kmax = g.fanout;
    }
    return hs_run("tg_ns_hop_segments", c, in, flt, rng, out, states_out, status, workspace, workspace_bytes, group_cap,
                  (hipStream_t)stream_); // weighted: the group form from group_cap >= 1024 on, column at a time below
}
