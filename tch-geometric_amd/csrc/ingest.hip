// COO -> CSR / CSC on the device -- replaces src/data/storage.rs:103-126 (reference: tch argsort of
// row*size1+col / col*size0+row, then the serial ind2ptr of :67-101).
//   keys  = major * size_minor + minor, values = edge position
//   sort  = rocPRIM stable LSD radix sort over only the bits the key range needs (6 passes for RMAT-24)
//   perm  = sorted values; indices = minor[perm]; ptrs[j] = lower_bound(sorted keys, j * size_minor)
// Duplicate edges keep their input order (stable), so `perm` is well defined where the reference's argsort is not.
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "tg_device.h"
#include "tg_host.h"

namespace tg {

__global__ void csx_keys_kernel(const int64_t *__restrict__ row, const int64_t *__restrict__ col, int64_t nnz,
                                int64_t size0, int64_t size1, int csc, int64_t *keys, int64_t *vals) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < nnz; e += (int64_t)gridDim.x * blockDim.x) {
        keys[e] = csc ? col[e] * size0 + row[e] : row[e] * size1 + col[e]; // storage.rs:119 / :112
        vals[e] = e;
    }
}
__global__ void csx_indices_kernel(const int64_t *__restrict__ sorted_keys, int64_t nnz, int64_t size_minor,
                                   int64_t *indices) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < nnz; e += (int64_t)gridDim.x * blockDim.x)
        indices[e] = sorted_keys[e] % size_minor;
}
// ptrs[j] = number of sorted keys < j * size_minor
__global__ void csx_ptrs_kernel(const int64_t *__restrict__ sorted_keys, int64_t nnz, int64_t size_minor, int64_t m,
                                int64_t *ptrs) {
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j <= m; j += (int64_t)gridDim.x * blockDim.x) {
        const int64_t target = j * size_minor;
        int64_t lo = 0, hi = nnz;
        while (lo < hi) {
            const int64_t mid = lo + ((hi - lo) >> 1);
            if (sorted_keys[mid] < target)
                lo = mid + 1;
            else
                hi = mid;
        }
        ptrs[j] = lo;
    }
}
// largest column (row) length of a CSC (CSR): one 8-byte atomic max per workgroup
__global__ void max_degree_kernel(const int64_t *__restrict__ ptrs, const uint32_t *__restrict__ ptrs32, int64_t m,
                                  unsigned long long *out) {
    __shared__ unsigned long long part[4];
    unsigned long long best = 0;
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < m; j += (int64_t)gridDim.x * blockDim.x) {
        const unsigned long long d = ptrs32 ? (unsigned long long)(ptrs32[j + 1] - ptrs32[j])
                                            : (unsigned long long)(ptrs[j + 1] - ptrs[j]);
        best = d > best ? d : best;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = __shfl_xor(best, off, 64);
        best = o > best ? o : best;
    }
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) best = part[w] > best ? part[w] : best;
        atomicMax(out, best);
    }
}
static inline unsigned csx_grid(int64_t n) {
    int64_t g = (n + 255) / 256;
    if (g < 1) g = 1;
    if (g > 256 * 32) g = 256 * 32;
    return (unsigned)g;
}
static inline int key_bits(int64_t size0, int64_t size1) {
    unsigned __int128 range = (unsigned __int128)size0 * (unsigned __int128)size1;
    int bits = 1;
    while (bits < 64 && ((unsigned __int128)1 << bits) < range) ++bits;
    return bits;
}
static size_t csx_sort_temp(int64_t nnz, int bits) {
    size_t st = 0;
    (void)rocprim::radix_sort_pairs(nullptr, st, (int64_t *)nullptr, (int64_t *)nullptr, (int64_t *)nullptr,
                                    (int64_t *)nullptr, (size_t)(nnz > 0 ? nnz : 1), 0, bits, (hipStream_t)0, false);
    return st;
}

} // namespace tg

extern "C" int tg_coo_to_csx_workspace_bytes(int64_t nnz, int64_t size0, int64_t size1, int64_t *bytes) {
    TG_REQUIRE(nnz >= 0 && size0 >= 1 && size1 >= 1 && bytes, "tg_coo_to_csx_workspace_bytes: bad arguments");
    TG_REQUIRE((unsigned __int128)size0 * (unsigned __int128)size1 < ((unsigned __int128)1 << 63),
               "tg_coo_to_csx: size0 * size1 overflows the int64 sort key (as in the reference)");
    const size_t n = (size_t)(nnz > 0 ? nnz : 1);
    *bytes = (int64_t)(3 * 8 * n + tg::csx_sort_temp(nnz, tg::key_bits(size0, size1)) + 512); // keys, vals, sorted keys
    return TG_OK;
}

extern "C" int tg_coo_to_csx(const int64_t *row, const int64_t *col, int64_t nnz, int64_t size0, int64_t size1,
                             int32_t csc, int64_t *ptrs, int64_t *indices, int64_t *perm, void *workspace,
                             int64_t workspace_bytes, void *stream_) {
    using namespace tg;
    TG_REQUIRE(nnz >= 0 && size0 >= 1 && size1 >= 1 && ptrs, "tg_coo_to_csx: bad arguments");
    int64_t need = 0;
    int rc = tg_coo_to_csx_workspace_bytes(nnz, size0, size1, &need);
    if (rc != TG_OK) return rc;
    hipStream_t stream = (hipStream_t)stream_;
    const int64_t m = csc ? size1 : size0, size_minor = csc ? size0 : size1;
    if (nnz == 0) {
        TG_HIP(hipMemsetAsync(ptrs, 0, sizeof(int64_t) * (size_t)(m + 1), stream));
        return TG_OK;
    }
    TG_REQUIRE(row && col && indices && perm && workspace && workspace_bytes >= need,
               "tg_coo_to_csx: null buffers or workspace too small");
    int64_t *keys = reinterpret_cast<int64_t *>(workspace), *vals = keys + nnz, *skeys = vals + nnz;
    void *temp = skeys + nnz;
    const int bits = key_bits(size0, size1);
    size_t st = (size_t)workspace_bytes - 3 * 8 * (size_t)nnz;
    hipLaunchKernelGGL(csx_keys_kernel, dim3(csx_grid(nnz)), dim3(256), 0, stream, row, col, nnz, size0, size1, csc, keys,
                       vals);
    TG_HIP(rocprim::radix_sort_pairs(temp, st, keys, skeys, vals, perm, (size_t)nnz, 0, bits, stream, false));
    hipLaunchKernelGGL(csx_indices_kernel, dim3(csx_grid(nnz)), dim3(256), 0, stream, skeys, nnz, size_minor, indices);
    hipLaunchKernelGGL(csx_ptrs_kernel, dim3(csx_grid(m + 1)), dim3(256), 0, stream, skeys, nnz, size_minor, m, ptrs);
    TG_LAUNCH_CHECK();
    return TG_OK;
}

/* tg_graph.max_degree for a caller that does not know it: *max_degree_dev (device int64) = the longest column. */
extern "C" int tg_graph_max_degree(const tg_graph *g, int64_t *max_degree_dev, void *stream_) {
    using namespace tg;
    TG_REQUIRE(g && max_degree_dev && (g->ptrs || g->ptrs32) && g->n_major >= 0, "tg_graph_max_degree: bad arguments");
    hipStream_t stream = (hipStream_t)stream_;
    TG_HIP(hipMemsetAsync(max_degree_dev, 0, sizeof(int64_t), stream));
    if (g->n_major == 0) return TG_OK;
    hipLaunchKernelGGL(max_degree_kernel, dim3(csx_grid(g->n_major)), dim3(256), 0, stream, g->ptrs, g->ptrs32, g->n_major,
                       reinterpret_cast<unsigned long long *>(max_degree_dev));
    TG_LAUNCH_CHECK();
    return TG_OK;
}
