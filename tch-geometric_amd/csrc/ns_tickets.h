// Device-side "reservoir by tickets" samplers shared by the neighbor-sampling kernels.
#pragma once
#include "tg_device.h"

namespace tg {

// Reservoir by tickets for one vertex with n > k candidates: slot s receives
// position k+ticket or keeps position s on a blank.  The shuffle's
// displaced entries live in registers; loops are fully unrolled so that no
// array is indexed dynamically.
template <int KMAX>
__device__ __forceinline__ void sample_tickets(CallKey ck, uint64_t id, uint32_t n, int k, uint32_t *spos,
                                               uint8_t *slane, uint32_t out_base, int lane) {
    uint32_t keys[KMAX > 0 ? KMAX : 1], vals[KMAX > 0 ? KMAX : 1];
    Draw d;
#pragma unroll
    for (int s = 0; s < KMAX; ++s) {
        if (s < k) {
            const uint32_t m = (n - 1u) - (uint32_t)s;
            if ((s & 1) == 0) d = draw(ck, id, (uint32_t)(s >> 1), 0u);
            const uint32_t r = bounded32(d.half(s & 1), m);
            const uint32_t last = m - 1u;
            uint32_t tr = r, tl = last;
#pragma unroll
            for (int j = 0; j < s; ++j) {
                tr = (keys[j] == r) ? vals[j] : tr;
                tl = (keys[j] == last) ? vals[j] : tl;
            }
            keys[s] = r;
            vals[s] = tl;
            const uint32_t pos = (tr < n - (uint32_t)k) ? (uint32_t)k + tr : (uint32_t)s;
            spos[out_base + s] = pos;
            slane[out_base + s] = (uint8_t)lane;
        }
    }
}

// Same law for any fan-out (KMAX == 0 instantiation): the shuffle's displaced entries live in LDS, one
// 2k-word strip per lane.  Slower than the register form; used above TG_MAX_FANOUT.
__device__ __forceinline__ void sample_tickets_lds(CallKey ck, uint64_t id, uint32_t n, int k, uint32_t *spos,
                                                   uint8_t *slane, uint32_t out_base, int lane, uint32_t *strip) {
    uint32_t *keys = strip + (size_t)lane * 2 * k, *vals = keys + k;
    Draw d;
    for (int s = 0; s < k; ++s) {
        const uint32_t m = (n - 1u) - (uint32_t)s;
        if ((s & 1) == 0) d = draw(ck, id, (uint32_t)(s >> 1), 0u);
        const uint32_t r = bounded32(d.half(s & 1), m), last = m - 1u;
        uint32_t tr = r, tl = last;
        for (int j = 0; j < s; ++j) {
            tr = (keys[j] == r) ? vals[j] : tr;
            tl = (keys[j] == last) ? vals[j] : tl;
        }
        keys[s] = r;
        vals[s] = tl;
        spos[out_base + s] = (tr < n - (uint32_t)k) ? (uint32_t)k + tr : (uint32_t)s;
        slane[out_base + s] = (uint8_t)lane;
    }
}

} // namespace tg
