// Calibration kernel for the measurement harness: the chip's ceiling for
// independent random 8-byte gathers from a table much larger than the
// Infinity Cache -- the access pattern of `indices[edge_ptr]` in neighbor
// sampling.  Not part of the sampling path.
#include "tg_device.h"
#include "tg_host.h"

namespace tg {
__global__ void gather_probe_kernel(const int64_t *table, int64_t n_table, int64_t per_thread, uint64_t seed,
                                    int64_t *sink) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const CallKey ck = call_key(seed, 0, 0xBEu);
    int64_t acc = 0;
    for (int64_t i = 0; i < per_thread; i += 8) {
        // 8 independent gathers in flight per lane, 4 Philox blocks
        int64_t v[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const Draw d = draw(ck, (uint64_t)t, (uint32_t)(i / 2 + j), 0u);
            v[2 * j] = table[bounded64(d.a(), (uint64_t)n_table)];
            v[2 * j + 1] = table[bounded64(d.b(), (uint64_t)n_table)];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) acc ^= v[j];
    }
    sink[t] = acc;
}

// Speed of light of neighbor_sampling_homogenous's OUTPUT CONTRACT: the algorithmic bytes of a finished launch -- 8 B
// read + 8 B written per seed, 24 B read per expanded frontier slot, 8 B read + 32 B written per sampled edge
// -- moved as pure streams through the same per-batch slabs, with precomputed contents: no draws, no random access, no
// ordering.  One workgroup per batch; 16-byte streaming loads / stores.  Lengths come from `src` (a finished launch).
__global__ void ns_sol_kernel(const tg_ns_out src, const tg_ns_out dst, const int64_t *seeds, int64_t n_seeds,
                              int32_t n_hops, int64_t *sink) {
    typedef long long v2 __attribute__((ext_vector_type(2)));
    const int64_t b = blockIdx.x;
    const int64_t ne = src.counts[b * 2 + 1];
    const int64_t nf = n_hops > 0 ? src.layer_offsets[(b * n_hops + (n_hops - 1)) * 3] : 0; // slots expanded
    int64_t *samples = dst.samples + b * dst.cap_nodes, *rows = dst.rows + b * dst.cap_edges;
    int64_t *cols = dst.cols + b * dst.cap_edges, *eidx = dst.edge_index + b * dst.cap_edges;
    const int64_t *r_s = src.samples + b * src.cap_nodes, *r_a = src.rows + b * src.cap_edges;
    const int64_t *r_b = src.cols + b * src.cap_edges, *r_c = src.edge_index + b * src.cap_edges;
    long long acc = 0;
    for (int64_t i = 2 * (int64_t)threadIdx.x; i < n_seeds; i += 2 * blockDim.x) { // 16 B per seed
        if (i + 1 < n_seeds)
            __builtin_nontemporal_store(__builtin_nontemporal_load(reinterpret_cast<const v2 *>(seeds + b * n_seeds + i)),
                                        reinterpret_cast<v2 *>(samples + i));
        else
            samples[i] = seeds[b * n_seeds + i];
    }
    const int64_t nf2 = nf < src.cap_edges ? nf : src.cap_edges;
    for (int64_t i = 2 * (int64_t)threadIdx.x; i + 1 < nf2; i += 2 * blockDim.x) { // 24 B per frontier slot, read only
        const v2 x = __builtin_nontemporal_load(reinterpret_cast<const v2 *>(r_a + i));
        const v2 y = __builtin_nontemporal_load(reinterpret_cast<const v2 *>(r_b + i));
        const v2 z = __builtin_nontemporal_load(reinterpret_cast<const v2 *>(r_c + i));
        acc ^= x.x ^ x.y ^ y.x ^ y.y ^ z.x ^ z.y;
    }
    const int64_t head = ((uintptr_t)(rows) >> 3) & 1; // slabs with an odd pitch: odd batches are 8-byte aligned
    for (int64_t e = head + 2 * (int64_t)threadIdx.x; e + 1 < ne; e += 2 * blockDim.x) { // 8 B read + 32 B written per edge
        const int64_t v0 = __builtin_nontemporal_load(r_s + n_seeds + e), v1 = __builtin_nontemporal_load(r_s + n_seeds + e + 1);
        v2 s = {v0, v1}, r = {n_seeds + e, n_seeds + e + 1}, c = {e >> 3, (e + 1) >> 3}, x = {v0 + 1, v1 + 1};
        if (((n_seeds + e) & 1) == (((uintptr_t)samples >> 3) & 1)) { // samples[n_seeds + e] 16-byte aligned
            __builtin_nontemporal_store(s, reinterpret_cast<v2 *>(samples + n_seeds + e));
        } else {
            __builtin_nontemporal_store(v0, samples + n_seeds + e);
            __builtin_nontemporal_store(v1, samples + n_seeds + e + 1);
        }
        __builtin_nontemporal_store(r, reinterpret_cast<v2 *>(rows + e));
        __builtin_nontemporal_store(c, reinterpret_cast<v2 *>(cols + e));
        __builtin_nontemporal_store(x, reinterpret_cast<v2 *>(eidx + e));
    }
    if (acc == 0x7fffffffffffffffll) sink[b] = acc; // keeps the frontier reads alive
}
} // namespace tg

extern "C" TG_API int tg_probe_ns_sol(const tg_ns_out *src, const tg_ns_out *dst, const int64_t *seeds, int64_t n_batches,
                                      int64_t n_seeds, int32_t n_hops, int64_t *sink, void *stream) {
    TG_REQUIRE(src && dst && seeds && sink && n_batches > 0 && n_seeds > 0 && n_hops >= 0 && n_hops <= TG_MAX_HOPS,
               "tg_probe_ns_sol: bad arguments");
    TG_REQUIRE(src->cap_nodes == dst->cap_nodes && src->cap_edges == dst->cap_edges && src->samples != dst->samples,
               "tg_probe_ns_sol: src and dst must be two slab sets of equal pitch");
    hipLaunchKernelGGL(tg::ns_sol_kernel, dim3((unsigned)n_batches), dim3(256), 0, (hipStream_t)stream, *src, *dst, seeds,
                       n_seeds, n_hops, sink);
    TG_LAUNCH_CHECK();
    return TG_OK;
}

extern "C" TG_API int tg_probe_random_gather(const int64_t *table, int64_t n_table, int64_t n_threads,
                                             int64_t per_thread, uint64_t seed, int64_t *sink, void *stream) {
    TG_REQUIRE(table && sink && n_table > 0 && n_threads > 0 && n_threads % 256 == 0 && per_thread % 8 == 0,
               "tg_probe_random_gather: bad arguments");
    hipLaunchKernelGGL(tg::gather_probe_kernel, dim3((unsigned)(n_threads / 256)), dim3(256), 0, (hipStream_t)stream,
                       table, n_table, per_thread, seed, sink);
    TG_LAUNCH_CHECK();
    return TG_OK;
}
