// What does moving N short runs through a random permutation cost on this chip -- PUSHED (gathered values written as
// scattered runs: what win_gather_kernel does with its 80-byte sample runs) or PULLED (runs read from random places,
// written as one coalesced stream)?  And what does a partial 32-byte sector cost a scattered write?
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/probe_permute tools/probe_permute.hip && tools/bin/probe_permute [bits]
//
// N = 2^bits runs; run r lives at slot perm(r) (a 4-round Feistel bijection: no two runs collide, every slot is hit once).
// push<T,L,P>: lanes walk (run, word) pairs consecutively -- 64 lanes cover 64/L runs -- and write word w of run r to
//              dst[perm(r) * P + w] (P = pitch in words >= L).  Source: registers.
// pull<T,L,P>: dst[q] = src[perm(q / L) * P + q % L]: coalesced writes, scattered run reads.
// Prints one JSON line per variant: ms (best of 3), runs/s, payload GB/s.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x)                                                                                                          \
    do {                                                                                                               \
        hipError_t e_ = (x);                                                                                           \
        if (e_ != hipSuccess) {                                                                                        \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                                                    \
            exit(1);                                                                                                   \
        }                                                                                                              \
    } while (0)

__device__ __forceinline__ uint32_t feistel(uint32_t x, int half_bits) {
    const uint32_t mask = (1u << half_bits) - 1u;
    uint32_t a = x >> half_bits, b = x & mask;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const uint32_t f = ((b * 0x9E3779B1u + 0x7F4A7C15u * (uint32_t)(r + 1)) >> 9) ^ (b * 0x85EBCA6Bu >> 3);
        const uint32_t t = a ^ (f & mask);
        a = b;
        b = t;
    }
    return (a << half_bits) | b;
}

template <typename T, int L, int P, bool NT>
__global__ void push_kernel(T *dst, uint64_t n_words, int half_bits) {
    for (uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n_words; q += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t r = (uint32_t)(q / L), w = (uint32_t)(q % L);
        T *at = dst + (uint64_t)feistel(r, half_bits) * P + w;
        if (NT)
            __builtin_nontemporal_store((T)q, at);
        else
            *at = (T)q;
    }
}

template <typename T, int L, int P>
__global__ void pull_kernel(const T *__restrict__ src, T *__restrict__ dst, uint64_t n_words, int half_bits) {
    for (uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n_words; q += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t r = (uint32_t)(q / L), w = (uint32_t)(q % L);
        __builtin_nontemporal_store(src[(uint64_t)feistel(r, half_bits) * P + w], dst + q);
    }
}

// pull of u32 runs widened to i64 on the way out (what a pulling last pass of the sampler would do)
template <int L, int P>
__global__ void pull_widen_kernel(const uint32_t *__restrict__ src, int64_t *__restrict__ dst, uint64_t n_words, int half_bits) {
    for (uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n_words; q += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t r = (uint32_t)(q / L), w = (uint32_t)(q % L);
        __builtin_nontemporal_store((int64_t)src[(uint64_t)feistel(r, half_bits) * P + w], dst + q);
    }
}

__global__ void fill_kernel(int64_t *dst, uint64_t n) {
    typedef long long v2 __attribute__((ext_vector_type(2)));
    for (uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; 2 * q + 1 < n; q += (uint64_t)gridDim.x * blockDim.x) {
        v2 x = {(long long)q, (long long)q};
        __builtin_nontemporal_store(x, reinterpret_cast<v2 *>(dst) + q);
    }
}

template <typename F> static float best_ms(F launch) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(a, 0));
        launch();
        CK(hipEventRecord(b, 0));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        if (rep > 0 && ms < best) best = ms;
    }
    return best;
}

static void report(const char *name, int word_bytes, int L, int P, uint64_t n_runs, float ms) {
    printf("{\"variant\": \"%s\", \"run_bytes\": %d, \"pitch_bytes\": %d, \"runs\": %llu, \"ms\": %.3f, \"G_runs_per_s\": %.2f, "
           "\"payload_GBps\": %.0f}\n",
           name, L * word_bytes, P * word_bytes, (unsigned long long)n_runs, ms, n_runs / ms / 1e6,
           (double)n_runs * L * word_bytes / ms / 1e6);
    fflush(stdout);
}

int main(int argc, char **argv) {
    const int bits = argc > 1 ? atoi(argv[1]) : 26; // even
    const int half = bits / 2;
    const uint64_t N = 1ull << (2 * half);
    const int blocks = 256 * 16, threads = 256;
    int64_t *big = nullptr, *lin = nullptr;
    const uint64_t big_words = N * 16; // up to 128-byte pitch
    CK(hipMalloc(&big, big_words * 8));
    CK(hipMalloc(&lin, N * 16 * 8));
    CK(hipMemset(big, 0, big_words * 8));
    CK(hipMemset(lin, 0, N * 16 * 8));

    report("fill_streaming_16B", 8, 10, 10, N, best_ms([&] { hipLaunchKernelGGL(fill_kernel, dim3(blocks), dim3(threads), 0, 0, big, N * 10); }));
#define PUSH(T, L, P, NT, NAME)                                                                                         \
    report(NAME, (int)sizeof(T), L, P, N, best_ms([&] {                                                                 \
               hipLaunchKernelGGL((push_kernel<T, L, P, NT>), dim3(blocks), dim3(threads), 0, 0, (T *)big, N * L, half); \
           }))
    PUSH(int64_t, 10, 10, false, "push_i64_80B_dense");          // K4's shape: 80-byte runs, 8-byte aligned, lines shared
    PUSH(int64_t, 10, 10, true, "push_i64_80B_dense_nt");
    PUSH(int64_t, 10, 16, false, "push_i64_80B_in_128B_slots");  // same runs, each alone in an aligned line
    PUSH(int64_t, 12, 12, false, "push_i64_96B_dense");          // whole 32-byte sectors only
    PUSH(int64_t, 16, 16, false, "push_i64_128B_lines");         // whole lines
    PUSH(int64_t, 8, 8, false, "push_i64_64B");                  // whole half lines
    PUSH(int64_t, 4, 4, false, "push_i64_32B");                  // whole sectors
    PUSH(int64_t, 1, 1, false, "push_i64_8B");                   // single words
    PUSH(int64_t, 5, 5, false, "push_i64_40B_dense");
    PUSH(uint32_t, 10, 10, false, "push_u32_40B_dense");         // the same runs as u32
    PUSH(uint32_t, 1, 1, false, "push_u32_4B");                  // a 4-byte word per item (an inverse-permutation table)
#define PULL(T, L, P, NAME)                                                                                             \
    report(NAME, (int)sizeof(T), L, P, N, best_ms([&] {                                                                 \
               hipLaunchKernelGGL((pull_kernel<T, L, P>), dim3(blocks), dim3(threads), 0, 0, (const T *)big, (T *)lin,  \
                                  N * L, half);                                                                         \
           }))
    PULL(int64_t, 10, 10, "pull_i64_80B_dense");
    PULL(uint32_t, 10, 10, "pull_u32_40B_dense");
    PULL(int64_t, 16, 16, "pull_i64_128B_lines");
    PULL(int64_t, 1, 1, "pull_i64_8B");
    report("pull_u32_40B_widen_to_i64", 4, 10, 10, N, best_ms([&] {
               hipLaunchKernelGGL((pull_widen_kernel<10, 10>), dim3(blocks), dim3(threads), 0, 0, (const uint32_t *)big, lin,
                                  N * 10, half);
           }));
    return 0;
}
