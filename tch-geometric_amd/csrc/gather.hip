// Row gather: dst[i, :] = src[index[i], :].  The step after the sampling path -- turning the `samples` /
// `edge_index` tensors of a mini-batch into feature rows (what the reference's examples delegate to PyG's
// `filter_data`, examples/neighbor_sampling.py:24,36,48) and composing `edge_index` with the ingest `perm`.
//
// HBM-bound byte mover.  Rows are handled as vectors of V bytes (16 when base pointers, stride and row length allow,
// else 8 / 4 / 1); a workgroup of 256 threads is cut into groups of L = 2^j lanes, one row per group per pass, four
// passes unrolled so every lane keeps four independent loads in flight.  Reads of one row are contiguous (whole
// 128-byte lines when the row is >= 128 B), the output is written once with non-temporal stores so it does not evict
// feature rows of hub vertices that the next batches will ask for again.
#include "tg_device.h"
#include "tg_host.h"

namespace tg {

template <typename V> __device__ __forceinline__ V zero_vec();
template <> __device__ __forceinline__ uint4 zero_vec<uint4>() { return make_uint4(0u, 0u, 0u, 0u); }
template <> __device__ __forceinline__ uint2 zero_vec<uint2>() { return make_uint2(0u, 0u); }
template <> __device__ __forceinline__ uint32_t zero_vec<uint32_t>() { return 0u; }
template <> __device__ __forceinline__ uint8_t zero_vec<uint8_t>() { return 0; }

template <typename V> __device__ __forceinline__ void store_nt(V *p, V v) { __builtin_nontemporal_store(v, p); }
template <> __device__ __forceinline__ void store_nt<uint4>(uint4 *p, uint4 v) {
    uint32_t *q = reinterpret_cast<uint32_t *>(p); // one 16-byte non-temporal store
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    u32x4 t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<u32x4 *>(q));
}
template <> __device__ __forceinline__ void store_nt<uint2>(uint2 *p, uint2 v) {
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    u32x2 t = {v.x, v.y};
    __builtin_nontemporal_store(t, reinterpret_cast<u32x2 *>(p));
}

constexpr int GATHER_THREADS = 256;
constexpr int GATHER_UNROLL = 4;

// lanes_log2: log2 of the lanes that share one row; vpr: vectors per row; strides in vectors
template <typename V>
__global__ __launch_bounds__(GATHER_THREADS) void gather_rows_kernel(const V *__restrict__ src, int64_t n_src_rows,
                                                                     int64_t src_stride, const int64_t *__restrict__ index,
                                                                     int64_t n, V *__restrict__ dst, int64_t vpr,
                                                                     int lanes_log2, int32_t *status) {
    const int lanes = 1 << lanes_log2;
    const int lane = threadIdx.x & (lanes - 1);
    const int rows_per_pass = GATHER_THREADS >> lanes_log2;
    const int64_t tile = (int64_t)rows_per_pass * GATHER_UNROLL;
    bool bad = false;
    for (int64_t base = (int64_t)blockIdx.x * tile; base < n; base += (int64_t)gridDim.x * tile) {
        const int64_t r0 = base + (threadIdx.x >> lanes_log2);
        int64_t from[GATHER_UNROLL];
#pragma unroll
        for (int u = 0; u < GATHER_UNROLL; ++u) {
            const int64_t r = r0 + (int64_t)u * rows_per_pass;
            int64_t f = -1;
            if (r < n) {
                f = index[r];
                if (f < 0 || f >= n_src_rows) {
                    bad = true;
                    f = -2; // written as zeros
                }
            }
            from[u] = f;
        }
        for (int64_t c = lane; c < vpr; c += lanes) {
            V v[GATHER_UNROLL];
#pragma unroll
            for (int u = 0; u < GATHER_UNROLL; ++u) v[u] = from[u] >= 0 ? src[from[u] * src_stride + c] : zero_vec<V>();
#pragma unroll
            for (int u = 0; u < GATHER_UNROLL; ++u)
                if (from[u] != -1) store_nt<V>(dst + (r0 + (int64_t)u * rows_per_pass) * vpr + c, v[u]);
        }
    }
    if (status && __ballot(bad) != 0ull && (threadIdx.x & 63) == 0) atomicOr(status, 1);
}

template <typename V>
static int launch_gather(const void *src, int64_t n_src_rows, int64_t row_bytes, int64_t src_stride_bytes,
                         const int64_t *index, int64_t n, void *dst, int32_t *status, hipStream_t stream) {
    const int64_t vpr = row_bytes / (int64_t)sizeof(V);
    int lanes_log2 = 0;
    while (lanes_log2 < 6 && (1ll << lanes_log2) < vpr) ++lanes_log2; // up to one wavefront per row
    const int64_t tile = (int64_t)(GATHER_THREADS >> lanes_log2) * GATHER_UNROLL;
    int64_t grid = (n + tile - 1) / tile;
    if (grid > 256 * 64) grid = 256 * 64;
    hipLaunchKernelGGL(gather_rows_kernel<V>, dim3((unsigned)grid), dim3(GATHER_THREADS), 0, stream,
                       reinterpret_cast<const V *>(src), n_src_rows, src_stride_bytes / (int64_t)sizeof(V), index, n,
                       reinterpret_cast<V *>(dst), vpr, lanes_log2, status);
    TG_LAUNCH_CHECK();
    return TG_OK;
}

// Per-batch output slabs of tg_ns_homo_batched -> flat, batch-major arrays (what a loader hands on): batch b's
// samples go to flat_samples[node_off[b] ..], its rows / cols / edge pointers to flat_*[edge_off[b] ..].
__global__ void ns_compact_kernel(tg_ns_out o, const int64_t *__restrict__ node_off, const int64_t *__restrict__ edge_off,
                                  int64_t *flat_samples, int64_t *flat_rows, int64_t *flat_cols, int64_t *flat_eidx) {
    const int64_t b = blockIdx.x;
    const int64_t n_nodes = o.counts[b * 2], n_edges = o.counts[b * 2 + 1];
    const int64_t step = (int64_t)gridDim.y * blockDim.x, first = (int64_t)blockIdx.y * blockDim.x + threadIdx.x;
    const int64_t *s = o.samples + b * o.cap_nodes;
    const int64_t *r = o.rows + b * o.cap_edges, *c = o.cols + b * o.cap_edges, *e = o.edge_index + b * o.cap_edges;
    const int64_t no = node_off[b], eo = edge_off[b];
    for (int64_t i = first; i < n_nodes; i += step) flat_samples[no + i] = s[i];
    for (int64_t i = first; i < n_edges; i += step) {
        flat_rows[eo + i] = r[i];
        flat_cols[eo + i] = c[i];
        flat_eidx[eo + i] = e[i];
    }
}

// ragged rows of an int64 slab -> one flat array: dst[off[r] + i] = src[r * pitch + i] for i < lens[r * lens_stride]
__global__ void compact_rows_kernel(const int64_t *__restrict__ src, int64_t pitch, const int64_t *__restrict__ lens,
                                    int64_t lens_stride, const int64_t *__restrict__ off, int64_t *dst) {
    const int64_t r = blockIdx.x, n = lens[r * lens_stride], o = off[r];
    const int64_t *s = src + r * pitch;
    for (int64_t i = (int64_t)blockIdx.y * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.y * blockDim.x) dst[o + i] = s[i];
}

} // namespace tg

extern "C" int tg_compact_rows(const int64_t *src, int64_t pitch, const int64_t *lens, int64_t lens_stride,
                               const int64_t *offsets, int64_t n_rows, int64_t *dst, void *stream) {
    TG_REQUIRE(n_rows >= 0 && n_rows <= 0x7fffffff && pitch >= 0 && lens_stride >= 1, "tg_compact_rows: bad sizes");
    if (n_rows == 0) return TG_OK;
    TG_REQUIRE(src && lens && offsets && dst, "tg_compact_rows: null buffers");
    hipLaunchKernelGGL(tg::compact_rows_kernel, dim3((unsigned)n_rows, 8), dim3(256), 0, (hipStream_t)stream, src, pitch, lens,
                       lens_stride, offsets, dst);
    TG_LAUNCH_CHECK();
    return TG_OK;
}

extern "C" int tg_ns_homo_compact(const tg_ns_out *out, int64_t n_batches, const int64_t *node_off, const int64_t *edge_off,
                                  int64_t *flat_samples, int64_t *flat_rows, int64_t *flat_cols, int64_t *flat_edge_index,
                                  void *stream) {
    TG_REQUIRE(out && n_batches >= 0 && n_batches <= 0x7fffffff, "tg_ns_homo_compact: bad arguments");
    if (n_batches == 0) return TG_OK;
    TG_REQUIRE(out->samples && out->counts && node_off && edge_off && flat_samples, "tg_ns_homo_compact: null buffers");
    TG_REQUIRE(out->cap_edges == 0 || (out->rows && out->cols && out->edge_index && flat_rows && flat_cols && flat_edge_index),
               "tg_ns_homo_compact: null edge buffers");
    hipLaunchKernelGGL(tg::ns_compact_kernel, dim3((unsigned)n_batches, 8), dim3(256), 0, (hipStream_t)stream, *out, node_off,
                       edge_off, flat_samples, flat_rows, flat_cols, flat_edge_index);
    TG_LAUNCH_CHECK();
    return TG_OK;
}

extern "C" int tg_gather_rows(const void *src, int64_t n_src_rows, int64_t row_bytes, int64_t src_stride_bytes,
                              const int64_t *index, int64_t n, void *dst, int32_t *status, void *stream) {
    TG_REQUIRE(n >= 0 && n_src_rows >= 0 && row_bytes >= 0, "tg_gather_rows: negative size");
    TG_REQUIRE(src_stride_bytes >= row_bytes, "tg_gather_rows: source stride %lld smaller than the row (%lld bytes)",
               (long long)src_stride_bytes, (long long)row_bytes);
    if (n == 0 || row_bytes == 0) return TG_OK;
    TG_REQUIRE(index && dst && (src || n_src_rows == 0), "tg_gather_rows: null buffer");
    const uintptr_t align = (uintptr_t)src | (uintptr_t)dst | (uintptr_t)row_bytes | (uintptr_t)src_stride_bytes;
    hipStream_t s = (hipStream_t)stream;
    if ((align & 15) == 0) return tg::launch_gather<uint4>(src, n_src_rows, row_bytes, src_stride_bytes, index, n, dst, status, s);
    if ((align & 7) == 0) return tg::launch_gather<uint2>(src, n_src_rows, row_bytes, src_stride_bytes, index, n, dst, status, s);
    if ((align & 3) == 0) return tg::launch_gather<uint32_t>(src, n_src_rows, row_bytes, src_stride_bytes, index, n, dst, status, s);
    return tg::launch_gather<uint8_t>(src, n_src_rows, row_bytes, src_stride_bytes, index, n, dst, status, s);
}
