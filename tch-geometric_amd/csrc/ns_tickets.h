// Device-side "reservoir by tickets" samplers shared by the neighbor-sampling kernels.
#pragma once
#include "tg_device.h"

namespace tg {

// The k bounded draws of one vertex's slots, left in registers: r[s] in [0, n - 1 - s) for the ticket sampler, in [0, n)
// with replacement (REPL; sampling.rs:57-69).  One 32-bit Philox word and one multiply per slot (tg_device.h slot draw);
// the exact-rejection test is folded into ONE rarely taken branch for all slots: `suspect` collects "low half < range",
// and only then are the slots redone with the full test (and the 64-bit fallback draw where a word is rejected).
template <int KMAX, bool REPL>
__device__ __forceinline__ void slot_draws(CallKey ck, uint64_t id, uint32_t n, int k, uint32_t (&r)[KMAX]) {
    constexpr uint32_t d1 = REPL ? D1_REPLACE : 0u;
    Draw d;
    bool suspect = false;
#pragma unroll
    for (int s = 0; s < KMAX; ++s) {
        if (s < k) {
            if ((s & 3) == 0) d = draw(ck, id, (uint32_t)(s >> 2), d1);
            const uint32_t m = REPL ? n : (n - 1u) - (uint32_t)s;
            const uint64_t p = (uint64_t)d.w[s & 3] * m;
            r[s] = (uint32_t)(p >> 32);
            suspect |= (uint32_t)p < m;
        }
    }
    if (__builtin_expect(suspect, 0)) {
#pragma nounroll
        for (int s = 0; s < k; ++s) {
            const uint32_t m = REPL ? n : (n - 1u) - (uint32_t)s;
            const uint32_t v = slot_draw(ck, id, (uint32_t)s, d1, m);
#pragma unroll
            for (int j = 0; j < KMAX; ++j) r[j] = (j == s) ? v : r[j];
        }
    }
}

// The ticket chain over given draws: slot s receives position k + ticket, or keeps position s on a blank.
// The shuffle's displaced entries live in registers; loops are fully unrolled so that no array is indexed dynamically.
template <int KMAX>
__device__ __forceinline__ void ticket_chain(const uint32_t (&r_in)[KMAX], uint32_t n, int k, uint32_t (&pos)[KMAX]) {
    uint32_t keys[KMAX], vals[KMAX];
#pragma unroll
    for (int s = 0; s < KMAX; ++s) {
        if (s < k) {
            const uint32_t m = (n - 1u) - (uint32_t)s;
            const uint32_t r = r_in[s];
            const uint32_t last = m - 1u;
            uint32_t tr = r, tl = last;
#pragma unroll
            for (int j = 0; j < s; ++j) {
                tr = (keys[j] == r) ? vals[j] : tr;
                tl = (keys[j] == last) ? vals[j] : tl;
            }
            keys[s] = r;
            vals[s] = tl;
            pos[s] = (tr < n - (uint32_t)k) ? (uint32_t)k + tr : (uint32_t)s;
        }
    }
}

// Reservoir by tickets for one vertex with n > k candidates, positions left in registers.
template <int KMAX>
__device__ __forceinline__ void sample_tickets_reg(CallKey ck, uint64_t id, uint32_t n, int k, uint32_t (&pos)[KMAX]) {
    uint32_t r[KMAX];
    slot_draws<KMAX, false>(ck, id, n, k, r);
    ticket_chain<KMAX>(r, n, k, pos);
}

// ... staged in LDS in output order: positions at spos[out_base + s], the drawing lane beside them
template <int KMAX>
__device__ __forceinline__ void sample_tickets_given(const uint32_t (&r)[KMAX], uint32_t n, int k, uint32_t *spos,
                                                     uint8_t *slane, uint32_t out_base, int lane) {
    uint32_t pos[KMAX];
    ticket_chain<KMAX>(r, n, k, pos);
#pragma unroll
    for (int s = 0; s < KMAX; ++s) {
        if (s < k) {
            spos[out_base + s] = pos[s];
            slane[out_base + s] = (uint8_t)lane;
        }
    }
}
template <int KMAX>
__device__ __forceinline__ void sample_tickets(CallKey ck, uint64_t id, uint32_t n, int k, uint32_t *spos,
                                               uint8_t *slane, uint32_t out_base, int lane) {
    uint32_t r[KMAX];
    slot_draws<KMAX, false>(ck, id, n, k, r);
    sample_tickets_given<KMAX>(r, n, k, spos, slane, out_base, lane);
}

// k draws of U[0, n) (sampling.rs:57-69), staged like the tickets
template <int KMAX>
__device__ __forceinline__ void sample_replace(CallKey ck, uint64_t id, uint32_t n, int k, uint32_t *spos,
                                               uint8_t *slane, uint32_t out_base, int lane) {
    uint32_t r[KMAX];
    slot_draws<KMAX, true>(ck, id, n, k, r);
#pragma unroll
    for (int s = 0; s < KMAX; ++s) {
        if (s < k) {
            spos[out_base + s] = r[s];
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
        if ((s & 3) == 0) d = draw(ck, id, (uint32_t)(s >> 2), 0u);
        const uint32_t r = slot_draw_from(d, ck, id, (uint32_t)s, 0u, m), last = m - 1u;
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
// ... and k draws of U[0, n) for any fan-out
__device__ __forceinline__ void sample_replace_any(CallKey ck, uint64_t id, uint32_t n, int k, uint32_t *spos,
                                                   uint8_t *slane, uint32_t out_base, int lane) {
    Draw d;
    for (int s = 0; s < k; ++s) {
        if ((s & 3) == 0) d = draw(ck, id, (uint32_t)(s >> 2), D1_REPLACE);
        spos[out_base + s] = slot_draw_from(d, ck, id, (uint32_t)s, D1_REPLACE, n);
        slane[out_base + s] = (uint8_t)lane;
    }
}

} // namespace tg
