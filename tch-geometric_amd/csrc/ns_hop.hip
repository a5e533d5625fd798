// One hop of (unweighted, unfiltered) neighbor sampling over a FLAT frontier, spread over the whole chip:
//   count   lane per frontier vertex: min(deg, k) (k with replacement) -> cnt[]
//   scan    device-wide inclusive scan (rocPRIM) -> offsets[]
//   emit    wavefront per 64 frontier vertices: ticket positions per lane, LDS staging in output order,
//           coalesced gather of indices[edge_ptr] and write of (neighbour, edge pointer, parent)
// It is the building block where the frontier is not "one workgroup per seed batch": the owner side of the
// range-partitioned sampler (every vertex carries its requester's draw id and call id), and relation-hops of
// the heterogeneous sampler.  Same draws, same per-vertex output order as ns_homo.hip, hence the same results.
// No synchronisation: the caller sizes outputs for the worst case m*k and reads offsets[m] when it wants to.
#include <cstring>

#include <rocprim/device/device_scan.hpp>

#include "ns_tickets.h"
#include "tg_device.h"
#include "tg_host.h"
#include "tg_scan.h"

namespace tg {

struct HopParams {
    const int64_t *ptrs, *indices;
    const uint32_t *ptrs32, *indices32;
    const int64_t *vertices, *ids, *call_ids;
    int64_t m, id_base;
    int32_t k, replace;
    uint32_t tag;
    uint64_t seed, call_id;
    int64_t *cnt, *offsets, *neighbors, *edge_ptrs, *parents;
    TG_BOUNDS_FIELDS
};

__device__ __forceinline__ void hop_range(const HopParams &p, int64_t w, int64_t &e0, int64_t &deg) {
    TG_CHECK_VERTEX(p, w);
    if (p.ptrs32) {
        e0 = (int64_t)p.ptrs32[w];
        deg = (int64_t)p.ptrs32[w + 1] - e0;
    } else {
        e0 = p.ptrs[w];
        deg = p.ptrs[w + 1] - e0;
    }
}

__global__ void hop_count_kernel(const HopParams p) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < p.m; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t w = p.vertices[i];
        int64_t c = 0;
        if (w >= 0) {
            int64_t e0, deg;
            hop_range(p, w, e0, deg);
            c = (deg <= 0) ? 0 : (p.replace ? (int64_t)p.k : min(deg, (int64_t)p.k));
        }
        p.cnt[i] = c;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) p.offsets[0] = 0;
}

// short frontiers (a per-call hop is a few thousand vertices): counts and their prefix in ONE launch of one workgroup
// instead of a count kernel + the library's scan launches
__global__ void __launch_bounds__(SCAN1_THREADS) hop_count_scan1_kernel(const HopParams p) {
    block_scan_exclusive_plus1(
        p.m,
        [&](int64_t i) {
            const int64_t w = p.vertices[i];
            int64_t c = 0;
            if (w >= 0) {
                int64_t e0, deg;
                hop_range(p, w, e0, deg);
                c = (deg <= 0) ? 0 : (p.replace ? (int64_t)p.k : min(deg, (int64_t)p.k));
            }
            p.cnt[i] = c;
            return c;
        },
        p.offsets);
}

template <int KMAX>
__global__ void hop_emit_kernel(const HopParams p) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
    const int k = p.k;
    const size_t wave_bytes = 64 * sizeof(int64_t) + (size_t)64 * k * sizeof(uint32_t) + (((size_t)64 * k + 15) & ~(size_t)15) +
                              (KMAX == 0 ? (size_t)64 * 2 * k * sizeof(uint32_t) : 0);
    unsigned char *wbase = smem + (size_t)wave * wave_bytes;
    int64_t *ebase = reinterpret_cast<int64_t *>(wbase);
    uint32_t *spos = reinterpret_cast<uint32_t *>(wbase + 64 * sizeof(int64_t));
    uint8_t *slane = reinterpret_cast<uint8_t *>(wbase + 64 * sizeof(int64_t) + (size_t)64 * k * sizeof(uint32_t));
    uint32_t *strip = reinterpret_cast<uint32_t *>(wbase + 64 * sizeof(int64_t) + (size_t)64 * k * sizeof(uint32_t) +
                                                   (((size_t)64 * k + 15) & ~(size_t)15));
    (void)strip;
    const CallKey ck0 = call_key(p.seed, p.call_id, p.tag);
    const int64_t n_chunks = (p.m + 63) >> 6;
    for (int64_t c = (int64_t)blockIdx.x * n_waves + wave; c < n_chunks; c += (int64_t)gridDim.x * n_waves) {
        const int64_t i0 = c << 6, i = i0 + lane;
        int64_t e0 = 0, deg = 0, w = -1;
        if (i < p.m) {
            w = p.vertices[i];
            if (w >= 0) hop_range(p, w, e0, deg);
        }
        const uint32_t cnt = (deg <= 0) ? 0u : (p.replace ? (uint32_t)k : (uint32_t)min(deg, (int64_t)k));
        const int64_t chunk_off = p.offsets[i0];
        const uint32_t excl = (i < p.m) ? (uint32_t)(p.offsets[i] - chunk_off) : 0u;
        const int64_t last = min(i0 + 64, p.m);
        const uint32_t total = (uint32_t)(p.offsets[last] - chunk_off);
        ebase[lane] = e0;
        if (cnt > 0) {
            const uint64_t did = p.ids ? (uint64_t)p.ids[i] : (uint64_t)(p.id_base + i);
            const CallKey ck = p.call_ids ? call_key(p.seed, (uint64_t)p.call_ids[i], p.tag) : ck0;
            const uint32_t n = (uint32_t)deg;
            if (p.replace) { // sampling.rs:57-69
                sample_replace_any(ck, did, n, k, spos, slane, excl, lane);
            } else if (deg <= k) { // sampling.rs:12-15
                for (uint32_t s = 0; s < cnt; ++s) {
                    spos[excl + s] = s;
                    slane[excl + s] = (uint8_t)lane;
                }
            } else if constexpr (KMAX == 0) {
                sample_tickets_lds(ck, did, n, k, spos, slane, excl, lane, strip);
            } else {
                sample_tickets<KMAX>(ck, did, n, k, spos, slane, excl, lane);
            }
        }
        wave_lds_handoff();
#pragma unroll 4
        for (uint32_t q = lane; q < total; q += 64) {
            const int l = slane[q];
            const int64_t ep = ebase[l] + (int64_t)spos[q];
            const int64_t v = p.indices32 ? (int64_t)__builtin_nontemporal_load(&p.indices32[ep])
                                          : __builtin_nontemporal_load(&p.indices[ep]);
            const int64_t o = chunk_off + q;
            p.neighbors[o] = v;
            p.edge_ptrs[o] = ep;
            p.parents[o] = i0 + l;
        }
        wave_lds_handoff();
    }
}

// Fan-outs above 255: one WAVEFRONT per frontier vertex.  The ticket chain is inherently
// slot-after-slot; here the shuffle's displaced entries (keys, vals) live in LDS and all 64 lanes search them for
// the latest entry of the two keys a slot needs, so a slot costs ceil(s / 64) LDS probes per lane.  Same draws and
// the same positions as sample_tickets / the oracle.  Columns with deg <= k (the usual case at such fan-outs) are
// copied whole.
constexpr int HOP_BIGK_MAX = 4096;
__global__ void hop_emit_bigk_kernel(const HopParams p) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int k = p.k;
    uint32_t *keys = reinterpret_cast<uint32_t *>(smem), *vals = keys + k, *pos = vals + k;
    const CallKey ck0 = call_key(p.seed, p.call_id, p.tag);
    for (int64_t i = blockIdx.x; i < p.m; i += gridDim.x) {
        const int64_t w = p.vertices[i];
        if (w < 0) continue;
        int64_t e0, deg;
        hop_range(p, w, e0, deg);
        if (deg <= 0) continue;
        const int64_t o = p.offsets[i];
        const uint32_t n = (uint32_t)deg;
        const uint32_t cnt = p.replace ? (uint32_t)k : (uint32_t)min(deg, (int64_t)k);
        const uint64_t did = p.ids ? (uint64_t)p.ids[i] : (uint64_t)(p.id_base + i);
        const CallKey ck = p.call_ids ? call_key(p.seed, (uint64_t)p.call_ids[i], p.tag) : ck0;
        if (p.replace) { // sampling.rs:57-69: slot s draws its own position
            for (uint32_t s = lane; s < cnt; s += 64) pos[s] = slot_draw(ck, did, s, D1_REPLACE, n);
        } else if (deg <= k) { // sampling.rs:12-15
            for (uint32_t s = lane; s < cnt; s += 64) pos[s] = s;
        } else {
            Draw d;
            for (int s = 0; s < k; ++s) {
                const uint32_t m = (n - 1u) - (uint32_t)s;
                if ((s & 3) == 0) d = draw(ck, did, (uint32_t)(s >> 2), 0u);
                const uint32_t r = slot_draw_from(d, ck, did, (uint32_t)s, 0u, m), last = m - 1u;
                int jr = -1, jl = -1; // latest displaced entry of r / of last
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
                    pos[s] = (tr < n - (uint32_t)k) ? (uint32_t)k + tr : (uint32_t)s;
                }
                wave_lds_handoff();
            }
        }
        wave_lds_handoff();
        for (uint32_t q = lane; q < cnt; q += 64) {
            const int64_t ep = e0 + (int64_t)pos[q];
            p.neighbors[o + q] = p.indices32 ? (int64_t)p.indices32[ep] : p.indices[ep];
            p.edge_ptrs[o + q] = ep;
            p.parents[o + q] = i;
        }
        wave_lds_handoff();
    }
}

template <int KMAX>
static int launch_hop_emit(const HopParams &p, hipStream_t stream) {
    const int k = p.k;
    const size_t wave_bytes = 64 * sizeof(int64_t) + (size_t)64 * k * sizeof(uint32_t) + (((size_t)64 * k + 15) & ~(size_t)15) +
                              (KMAX == 0 ? (size_t)64 * 2 * k * sizeof(uint32_t) : 0);
    int n_waves = 4;
    while (n_waves > 1 && n_waves * wave_bytes > 64 * 1024) n_waves >>= 1;
    const size_t lds = n_waves * wave_bytes;
    if (lds > 160 * 1024) return fail(TG_ERR_UNSUPPORTED, "tg_ns_hop: fan-out %d needs %zu B of LDS per wavefront", k, lds);
    if (lds > 64 * 1024)
        TG_HIP(hipFuncSetAttribute((const void *)hop_emit_kernel<KMAX>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)lds));
    const int64_t n_chunks = (p.m + 63) >> 6;
    int64_t blocks = (n_chunks + n_waves - 1) / n_waves;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(hop_emit_kernel<KMAX>, dim3((unsigned)blocks), dim3(64 * n_waves), lds, stream, p);
    TG_LAUNCH_CHECK();
    return TG_OK;
}

} // namespace tg

extern "C" int tg_ns_hop_workspace_bytes(int64_t m, int64_t *bytes) {
    TG_REQUIRE(m >= 0 && bytes, "tg_ns_hop_workspace_bytes: bad arguments");
    size_t st = 0;
    hipError_t e = rocprim::inclusive_scan(nullptr, st, (int64_t *)nullptr, (int64_t *)nullptr, (size_t)(m > 0 ? m : 1),
                                           rocprim::plus<int64_t>(), (hipStream_t)0, false);
    if (e != hipSuccess) return tg::fail(TG_ERR_HIP, "rocprim::inclusive_scan size query failed: %s", hipGetErrorString(e));
    *bytes = (int64_t)st + 256;
    return TG_OK;
}

extern "C" int tg_ns_hop(const tg_graph *csc, const tg_hop_in *in, const tg_rng *rng, const tg_hop_out *out,
                         void *workspace, int64_t workspace_bytes, void *stream_) {
    using namespace tg;
    TG_REQUIRE(csc && csc->ptrs && in && rng && out, "tg_ns_hop: null argument");
    TG_REQUIRE(in->m >= 0 && in->fanout >= 1 && in->fanout <= HOP_BIGK_MAX, "tg_ns_hop: frontier size or fan-out (1..%d)",
               HOP_BIGK_MAX);
    TG_REQUIRE(in->sampler == TG_SAMPLER_UNIFORM || in->sampler == TG_SAMPLER_UNIFORM_REPL,
               "tg_ns_hop: only the unweighted samplers");
    TG_REQUIRE(out->cnt && out->offsets, "tg_ns_hop: null outputs");
    hipStream_t stream = (hipStream_t)stream_;
    HopParams p;
    p.ptrs = csc->ptrs;
    p.indices = csc->indices;
    p.ptrs32 = csc->ptrs32;
    p.indices32 = csc->indices32;
    p.vertices = in->vertices;
    p.ids = in->ids;
    p.call_ids = in->call_ids;
    p.m = in->m;
    p.id_base = in->id_base;
    p.k = in->fanout;
    p.replace = in->sampler == TG_SAMPLER_UNIFORM_REPL;
    p.tag = in->rng_tag ? in->rng_tag : TG_TAG_NS_HOMO;
    TG_BOUNDS_INIT(p, csc);
    p.seed = rng->seed;
    p.call_id = rng->call_id;
    p.cnt = out->cnt;
    p.offsets = out->offsets;
    p.neighbors = out->neighbors;
    p.edge_ptrs = out->edge_ptrs;
    p.parents = out->parents;
    if (p.m == 0) {
        TG_HIP(hipMemsetAsync(out->offsets, 0, sizeof(int64_t), stream));
        return TG_OK;
    }
    TG_REQUIRE(in->vertices && out->neighbors && out->edge_ptrs && out->parents && workspace, "tg_ns_hop: null buffers");
    if (p.m <= 16384) { // one workgroup beats count + the library's launches up to about here (measured with HGT's scans)
        hipLaunchKernelGGL(hop_count_scan1_kernel, dim3(1), dim3(SCAN1_THREADS), 0, stream, p);
    } else {
        int64_t g = (p.m + 255) / 256;
        if (g > 8192) g = 8192;
        hipLaunchKernelGGL(hop_count_kernel, dim3((unsigned)g), dim3(256), 0, stream, p);
        size_t need = 0;
        TG_HIP(rocprim::inclusive_scan(nullptr, need, p.cnt, p.offsets + 1, (size_t)p.m, rocprim::plus<int64_t>(), stream,
                                       false));
        TG_REQUIRE((size_t)workspace_bytes >= need, "tg_ns_hop: workspace too small (%lld < %zu)", (long long)workspace_bytes,
                   need);
        size_t st = (size_t)workspace_bytes;
        TG_HIP(rocprim::inclusive_scan(workspace, st, p.cnt, p.offsets + 1, (size_t)p.m, rocprim::plus<int64_t>(), stream,
                                       false));
    }
    if (p.k <= 16) return launch_hop_emit<16>(p, stream);
    if (p.k <= TG_MAX_FANOUT) return launch_hop_emit<32>(p, stream);
    if (p.k <= 128) return launch_hop_emit<0>(p, stream); // LDS ticket strips, lane per vertex
    int64_t blocks = p.m < 256 * 32 ? p.m : 256 * 32;      // wavefront per vertex
    hipLaunchKernelGGL(hop_emit_bigk_kernel, dim3((unsigned)blocks), dim3(64), (size_t)p.k * 12, stream, p);
    TG_LAUNCH_CHECK();
    return TG_OK;
}
