// Synthetic inputs of the measurement harness and the device
// half of graph ingest (ind2ptr, src/data/storage.rs:67-101).
#include "tg_device.h"
#include "tg_host.h"

namespace tg {

// R-MAT quadrant thresholds on a 32-bit word: a, a+b, a+b+c of (0.57, 0.19, 0.19, 0.05)
constexpr uint32_t RMAT_T_A = 2448131358u;   // floor(0.57 * 2^32)
constexpr uint32_t RMAT_T_AB = 3264175144u;  // floor(0.76 * 2^32)
constexpr uint32_t RMAT_T_ABC = 4080218931u; // floor(0.95 * 2^32)

// rectangular form: row bits and column bits are drawn to their own depths (level `bit` contributes a row bit
// while bit < row_scale and a column bit while bit < col_scale); square when both scales are equal
__global__ void rmat_edges_kernel(int row_scale, int col_scale, int64_t n_edges, uint64_t seed, int64_t *row,
                                  int64_t *col) {
    const CallKey ck = call_key(seed, 0, TAG_RMAT);
    const int scale = max(row_scale, col_scale);
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n_edges;
         e += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = 0, c = 0;
        Draw d;
        for (int bit = 0; bit < scale; ++bit) { // most significant bit first
            if ((bit & 3) == 0) d = draw(ck, (uint64_t)e, (uint32_t)(bit >> 2), 0u);
            const uint32_t u = d.w[bit & 3];
            const int rb = u >= RMAT_T_AB;                       // quadrants c, d
            const int cb = (u >= RMAT_T_A && u < RMAT_T_AB) || u >= RMAT_T_ABC; // quadrants b, d
            if (bit < row_scale) r = (r << 1) | rb;
            if (bit < col_scale) c = (c << 1) | cb;
        }
        row[e] = r;
        col[e] = c;
    }
}

__global__ void seed_batches_kernel(uint64_t seed, int64_t first_batch, int64_t n_batches, int64_t n_seeds,
                                    int64_t n_nodes, int64_t *out) {
    const CallKey ck = call_key(seed, 0, TAG_SEEDS);
    const int64_t total = n_batches * n_seeds;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = t / n_seeds, i = t - b * n_seeds;
        const Draw d = draw(ck, (uint64_t)(first_batch + b), (uint32_t)i, (uint32_t)((uint64_t)i >> 32));
        out[t] = (int64_t)bounded64(d.a(), (uint64_t)n_nodes);
    }
}

// out[j] = number of entries of the sorted `ind` that are < j  (j = 0..m)
__global__ void ind2ptr_kernel(const int64_t *ind, int64_t numel, int64_t m, int64_t *out) {
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j <= m; j += (int64_t)gridDim.x * blockDim.x) {
        int64_t lo = 0, hi = numel;
        while (lo < hi) {
            const int64_t mid = lo + ((hi - lo) >> 1);
            if (ind[mid] < j)
                lo = mid + 1;
            else
                hi = mid;
        }
        out[j] = lo;
    }
}

// flag[0] |= 1 when any value lies outside [lo, hi)
__global__ void check_range_kernel(const int64_t *__restrict__ v, int64_t n, int64_t lo, int64_t hi, int32_t *flag) {
    bool bad = false;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        bad |= (v[i] < lo) | (v[i] >= hi);
    if (__ballot(bad) != 0ull && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

// out[i] = v[i] when it lies in [lo, hi), else lo; flag[0] |= 1 if any did not
__global__ void sanitize_range_kernel(const int64_t *__restrict__ v, int64_t n, int64_t lo, int64_t hi, int64_t *out,
                                      int64_t *flag) {
    bool bad = false;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t x = v[i];
        const bool b = (x < lo) | (x >= hi);
        out[i] = b ? lo : x;
        bad |= b;
    }
    if (__ballot(bad) != 0ull && (threadIdx.x & 63) == 0) atomicOr(reinterpret_cast<unsigned long long *>(flag), 1ull);
}

static inline unsigned grid_for(int64_t n, int threads) {
    int64_t g = (n + threads - 1) / threads;
    if (g < 1) g = 1;
    if (g > 256 * 32) g = 256 * 32;
    return (unsigned)g;
}

} // namespace tg

extern "C" int tg_rmat_edges(int32_t scale, int64_t n_edges, uint64_t seed, int64_t *row, int64_t *col, void *stream) {
    TG_REQUIRE(scale >= 1 && scale <= 40, "tg_rmat_edges: scale %d outside [1, 40]", scale);
    TG_REQUIRE(n_edges >= 0 && (n_edges == 0 || (row && col)), "tg_rmat_edges: bad buffers");
    if (n_edges == 0) return TG_OK;
    hipLaunchKernelGGL(tg::rmat_edges_kernel, dim3(tg::grid_for(n_edges, 256)), dim3(256), 0, (hipStream_t)stream,
                       scale, scale, n_edges, seed, row, col);
    TG_LAUNCH_CHECK();
    return TG_OK;
}

extern "C" int tg_rmat_edges_rect(int32_t row_scale, int32_t col_scale, int64_t n_edges, uint64_t seed, int64_t *row,
                                  int64_t *col, void *stream) {
    TG_REQUIRE(row_scale >= 1 && row_scale <= 40 && col_scale >= 1 && col_scale <= 40, "tg_rmat_edges_rect: bad scales");
    TG_REQUIRE(n_edges >= 0 && (n_edges == 0 || (row && col)), "tg_rmat_edges_rect: bad buffers");
    if (n_edges == 0) return TG_OK;
    hipLaunchKernelGGL(tg::rmat_edges_kernel, dim3(tg::grid_for(n_edges, 256)), dim3(256), 0, (hipStream_t)stream,
                       row_scale, col_scale, n_edges, seed, row, col);
    TG_LAUNCH_CHECK();
    return TG_OK;
}

extern "C" int tg_seed_batches(uint64_t seed, int64_t first_batch, int64_t n_batches, int64_t n_seeds, int64_t n_nodes,
                               int64_t *out, void *stream) {
    TG_REQUIRE(n_batches >= 0 && n_seeds >= 0 && n_nodes >= 1, "tg_seed_batches: bad sizes");
    if (n_batches * n_seeds == 0) return TG_OK;
    TG_REQUIRE(out, "tg_seed_batches: null output");
    hipLaunchKernelGGL(tg::seed_batches_kernel, dim3(tg::grid_for(n_batches * n_seeds, 256)), dim3(256), 0,
                       (hipStream_t)stream, seed, first_batch, n_batches, n_seeds, n_nodes, out);
    TG_LAUNCH_CHECK();
    return TG_OK;
}

extern "C" int tg_ind2ptr(const int64_t *ind, int64_t numel, int64_t m, int64_t *out, void *stream) {
    TG_REQUIRE(numel >= 0 && m >= 0 && out && (ind || numel == 0), "tg_ind2ptr: bad arguments");
    hipLaunchKernelGGL(tg::ind2ptr_kernel, dim3(tg::grid_for(m + 1, 256)), dim3(256), 0, (hipStream_t)stream, ind,
                       numel, m, out);
    TG_LAUNCH_CHECK();
    return TG_OK;
}

extern "C" int tg_sanitize_range(const int64_t *values, int64_t n, int64_t lo, int64_t hi, int64_t *out, int64_t *flag,
                                 void *stream) {
    TG_REQUIRE(n >= 0 && flag && lo < hi && ((values && out) || n == 0), "tg_sanitize_range: bad arguments (empty range?)");
    if (n == 0) return TG_OK;
    hipLaunchKernelGGL(tg::sanitize_range_kernel, dim3(tg::grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, values, n,
                       lo, hi, out, flag);
    TG_LAUNCH_CHECK();
    return TG_OK;
}

extern "C" int tg_check_range(const int64_t *values, int64_t n, int64_t lo, int64_t hi, int32_t *flag, void *stream) {
    TG_REQUIRE(n >= 0 && flag && (values || n == 0), "tg_check_range: bad arguments");
    if (n == 0) return TG_OK;
    hipLaunchKernelGGL(tg::check_range_kernel, dim3(tg::grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, values, n,
                       lo, hi, flag);
    TG_LAUNCH_CHECK();
    return TG_OK;
}
