// The packed slot of one sampled frontier item as a bit stream: shared by the staged pipeline of
// tg_ns_homo_batched (ns_homo_stage.inl) and the slot replies of the partitioned sampler (partition.hip).
//   { e0 : 32, cnt : 8, cnt x { neighbour : bv bits, position : bp bits } }   -- k pairs always present, unused ones zero
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace tg {

struct StageBits {
    int32_t bv, bp;
};
__host__ __device__ inline int stage_bits_of(uint64_t max_value) { // bits that hold 0 .. max_value
    int b = 1;
    while (b < 64 && (max_value >> b) != 0) ++b;
    return b;
}
__host__ __device__ inline int stage_slot_bits(int k, StageBits sb) { return 40 + k * (sb.bv + sb.bp); }

// LDS bit-stream writer / reader over a lane's row of the tile (uniform control flow: bv, bp and the unrolled slot index
// are the same for every lane, so `fill` and the word index are scalars)
struct BitWriter {
    uint32_t *row;
    uint64_t acc;
    int fill, w;
    __device__ __forceinline__ void push(uint32_t v, int bits) { // bits <= 32, v < 2^bits
        acc |= (uint64_t)v << fill;
        fill += bits;
        if (fill >= 32) {
            row[w++] = (uint32_t)acc;
            acc >>= 32;
            fill -= 32;
        }
    }
    __device__ __forceinline__ void finish(int n_words) {
        if (fill > 0) row[w++] = (uint32_t)acc;
        for (; w < n_words; ++w) row[w] = 0u;
    }
};
struct BitReader {
    const uint32_t *row;
    uint64_t acc;
    int fill, w;
    __device__ __forceinline__ uint32_t pop(int bits) { // bits <= 32
        if (fill < bits) {
            acc |= (uint64_t)row[w++] << fill;
            fill += 32;
        }
        const uint32_t v = (uint32_t)acc & (bits >= 32 ? 0xffffffffu : ((1u << bits) - 1u));
        acc >>= bits;
        fill -= bits;
        return v;
    }
};

} // namespace tg
