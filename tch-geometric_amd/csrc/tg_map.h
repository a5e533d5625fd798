// Open-addressing device hash map (int64 key >= 0 -> int64 value) and small utility kernels shared by
// the negative-sampling and HGT kernels.  Keys are claimed with atomicCAS; what a value means (min item
// position, max input slot, entry index) is up to the caller's atomicMin / atomicMax / plain store.
#pragma once
#include "tg_device.h"

namespace tg {

constexpr int64_t MAP_EMPTY = -1; // node ids are >= 0

__device__ __forceinline__ uint64_t map_hash(int64_t key) {
    uint64_t x = (uint64_t)key * 0x9E3779B97F4A7C15ull;
    return x ^ (x >> 29);
}
// claims (or finds) the slot of `key`
__device__ __forceinline__ int64_t map_slot_insert(int64_t *keys, int64_t mask, int64_t key) {
    int64_t s = (int64_t)(map_hash(key) & (uint64_t)mask);
    for (;;) {
        const unsigned long long prev = atomicCAS(reinterpret_cast<unsigned long long *>(&keys[s]),
                                                  (unsigned long long)MAP_EMPTY, (unsigned long long)key);
        if ((int64_t)prev == MAP_EMPTY || (int64_t)prev == key) return s;
        s = (s + 1) & mask;
    }
}
__device__ __forceinline__ int64_t map_slot_find(const int64_t *keys, int64_t mask, int64_t key) {
    int64_t s = (int64_t)(map_hash(key) & (uint64_t)mask);
    for (;;) {
        const int64_t k = keys[s];
        if (k == key) return s;
        if (k == MAP_EMPTY) return -1;
        s = (s + 1) & mask;
    }
}

static __global__ void fill_i64_kernel(int64_t *p, int64_t n, int64_t v) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        p[i] = v;
}

static inline int64_t pow2_at_least(int64_t n) {
    int64_t c = 64;
    while (c < n) c <<= 1;
    return c;
}
static inline unsigned grid_1d(int64_t n, int threads = 256) {
    int64_t g = (n + threads - 1) / threads;
    if (g < 1) g = 1;
    if (g > 8192) g = 8192;
    return (unsigned)g;
}

} // namespace tg
