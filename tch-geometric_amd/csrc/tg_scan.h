// Prefix sums of short device arrays in ONE launch of ONE workgroup.
// The per-call operators (one mini-batch per call) are bound by the number of launches, not by bytes: a library device
// scan is two or three launches however short the array; here the frontier-sized scans of a flat hop (a few thousand to
// ~10^5 elements) take one launch whose length may live on the device.  Long arrays stay with the multi-block scans.
#pragma once
#include "tg_device.h"

namespace tg {

constexpr int SCAN1_THREADS = 1024;
constexpr int64_t SCAN1_MAX = (int64_t)1 << 17;        // above this a multi-block scan is the better tool
constexpr int64_t SCAN1_GROUPS_MAX = (int64_t)1 << 20; // bound of a scan whose real length is on the device and usually far below

// out[0] = 0, out[i + 1] = out[i] + load(i) for i in [0, n): every wavefront owns a contiguous segment, sums it with
// coalesced loads, the 16 segment totals are scanned through LDS, then the segment is swept again with a running carry.
// Elements are < 2^32 and so is the sum of any 64 consecutive ones (counts of edges / groups of 64 vertices): the scan
// inside a 64-element chunk runs in 32 bits on DPP adds, the carry between chunks in 64 bits.
// `out` may alias the array `load` reads at i + 1 (the in-place inclusive form): element i is loaded before out[i + 1]
// is written by the same lane.  Call from every thread of a SCAN1_THREADS-wide workgroup.
template <typename Load>
__device__ __forceinline__ void block_scan_exclusive_plus1(int64_t n, Load load, int64_t *out) {
    __shared__ int64_t seg_total[SCAN1_THREADS / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int n_waves = SCAN1_THREADS / 64;
    const int64_t chunks = (n + 63) >> 6;
    const int64_t per = (chunks + n_waves - 1) / n_waves;
    const int64_t c0 = min(chunks, (int64_t)wave * per), c1 = min(chunks, c0 + per);
    constexpr int U = 8; // chunk loads in flight per lane
    int64_t acc = 0;
    for (int64_t c = c0; c < c1; c += U) {
        int64_t v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = ((c + u) << 6) + lane;
            v[u] = (c + u < c1 && i < n) ? (int64_t)load(i) : 0;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u];
    }
    acc = wave_sum(acc);
    if (lane == 0) seg_total[wave] = acc;
    __syncthreads();
    int64_t carry = 0;
    for (int w = 0; w < wave; ++w) carry += seg_total[w];
    if (threadIdx.x == 0) out[0] = 0;
    for (int64_t c = c0; c < c1; c += U) {
        int64_t v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = ((c + u) << 6) + lane;
            v[u] = (c + u < c1 && i < n) ? (int64_t)load(i) : 0;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = ((c + u) << 6) + lane;
            if (c + u >= c1) break; // uniform
            const uint32_t incl = wave_inclusive_scan_u32_dpp((uint32_t)v[u]);
            if (i < n) out[i + 1] = carry + (int64_t)incl;
            carry += (int64_t)(uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        }
    }
}

} // namespace tg
