// Device-side building blocks shared by the gfx950 kernels: counter-addressed
// Philox4x32-10 draws, bounded integers, wave64 scans.  Written for CDNA4 only
// (64-lane wavefronts are assumed everywhere).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tg {

// operator tags of the draw address (must match the spec in DESIGN.md "Randomness")
constexpr uint32_t TAG_NS_HOMO = 1u;
constexpr uint32_t TAG_NS_HETERO = 2u;
constexpr uint32_t TAG_RW = 3u;
constexpr uint32_t TAG_RW_TEMPO = 4u;
constexpr uint32_t TAG_NEG_HOMO = 5u;
constexpr uint32_t TAG_NEG_HETERO = 6u;
constexpr uint32_t TAG_HGT = 7u;
constexpr uint32_t TAG_RMAT = 8u;
constexpr uint32_t TAG_SEEDS = 9u;

constexpr uint32_t D1_REPLACE = 0x52455000u;  // "REP"
constexpr uint32_t D1_LITERAL = 0x4C495400u;  // "LIT"
constexpr uint32_t D1_WEIGHTED = 0x57475400u; // "WGT"
constexpr uint32_t D1_RESTART = 0x52535400u;  // "RST"
constexpr uint32_t D1_CHUNK = 0x43484B00u;    // "CHK"
constexpr uint32_t D1_FALLBACK = 0x46u;       // 'F', or-ed into d1: the 64-bit draw that replaces a rejected 32-bit word

struct Draw {
    uint32_t w[4];
    __device__ __forceinline__ uint64_t a() const { return (uint64_t)w[0] | ((uint64_t)w[1] << 32); }
    __device__ __forceinline__ uint64_t b() const { return (uint64_t)w[2] | ((uint64_t)w[3] << 32); }
    __device__ __forceinline__ uint64_t half(int h) const { return h ? b() : a(); }
};

struct CallKey {
    uint32_t k0, k1;
};

__device__ __forceinline__ Draw philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c1 = (uint32_t)p1;
        c3 = (uint32_t)p0;
        c0 = n0;
        c2 = n2;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    Draw d;
    d.w[0] = c0;
    d.w[1] = c1;
    d.w[2] = c2;
    d.w[3] = c3;
    return d;
}

// per-call key: Philox(key = seed, ctr = (call_id, tag, "tchg")) words 0,1
__device__ __forceinline__ CallKey call_key(uint64_t seed, uint64_t call_id, uint32_t tag) {
    const Draw d = philox4x32_10((uint32_t)call_id, (uint32_t)(call_id >> 32), tag, 0x74636867u, (uint32_t)seed,
                                 (uint32_t)(seed >> 32));
    return CallKey{d.w[0], d.w[1]};
}

// the draw named (call key, id, d0, d1)
__device__ __forceinline__ Draw draw(CallKey ck, uint64_t id, uint32_t d0, uint32_t d1) {
    return philox4x32_10(d0, d1, (uint32_t)id, (uint32_t)(id >> 32), ck.k0, ck.k1);
}

// floor(x * range / 2^64)
__device__ __forceinline__ uint64_t bounded64(uint64_t x, uint64_t range) { return __umul64hi(x, range); }
// same value for range < 2^32, with 32-bit partial products
__device__ __forceinline__ uint32_t bounded32(uint64_t x, uint32_t range) {
    const uint64_t lo = (x & 0xffffffffull) * range;
    const uint64_t hi = (x >> 32) * range;
    return (uint32_t)((hi + (lo >> 32)) >> 32);
}

// ---- the bounded draw of SLOT s of a reservoir / replacement sample (philox-mode since round 4; the CPU checker's
// orc_slot_draw) -----------------------------------------------------------------------------------------------------
// Slot s owns ONE 32-bit word: word s & 3 of the block (id, s >> 2, d1).  It is bounded by Lemire's multiply-shift --
// ONE 32 x 32 -> 64 multiply -- with the method's exact rejection test; a rejected word (probability < range / 2^32, so a
// branch no wavefront usually takes) is replaced by the 64-bit multiply-shift draw of the slot's own block
// (id, s, d1 | D1_FALLBACK).  Until round 3 a slot took half a block: five blocks and ten two-multiply draws per vertex at
// k = 10, where three blocks and ten one-multiply draws do now -- the draws are what bounds the sampling kernels.
__device__ __forceinline__ uint32_t bounded_word(uint32_t w, uint32_t range, bool &ok) {
    const uint64_t m = (uint64_t)w * range;
    const uint32_t lo = (uint32_t)m;
    ok = true;
    if (__builtin_expect(lo < range, 0)) ok = lo >= (0u - range) % range; // 2^32 mod range
    return (uint32_t)(m >> 32);
}
__device__ __forceinline__ uint32_t slot_fallback(CallKey ck, uint64_t id, uint32_t s, uint32_t d1, uint32_t range) {
    return bounded32(draw(ck, id, s, d1 | D1_FALLBACK).a(), range);
}
// word i of a block, i possibly different from lane to lane
__device__ __forceinline__ uint32_t draw_word(const Draw &d, uint32_t i) {
    const uint32_t lo = (i & 1u) ? d.w[1] : d.w[0], hi = (i & 1u) ? d.w[3] : d.w[2];
    return (i & 2u) ? hi : lo;
}
// slot s from the block `d` = draw(ck, id, s >> 2, d1) the caller holds
__device__ __forceinline__ uint32_t slot_draw_from(const Draw &d, CallKey ck, uint64_t id, uint32_t s, uint32_t d1,
                                                   uint32_t range) {
    bool ok;
    uint32_t r = bounded_word(draw_word(d, s), range, ok);
    if (__builtin_expect(!ok, 0)) r = slot_fallback(ck, id, s, d1, range);
    return r;
}
// slot s, block computed here (a lane per slot)
__device__ __forceinline__ uint32_t slot_draw(CallKey ck, uint64_t id, uint32_t s, uint32_t d1, uint32_t range) {
    return slot_draw_from(draw(ck, id, s >> 2, d1), ck, id, s, d1, range);
}

__device__ __forceinline__ float u32_to_f32_01(uint32_t w) { return __uint_as_float(0x3F800000u | (w >> 9)) - 1.0f; }
__device__ __forceinline__ double u64_to_f64_01(uint64_t x) {
    return __longlong_as_double((long long)(0x3FF0000000000000ull | (x >> 12))) - 1.0;
}

// ---- -DTG_DEBUG_BOUNDS: range-checked frontier ids ------------------------------------------------------------------
// The multi-hop kernels index `ptrs[w]` with ids that came out of `indices` one hop earlier and trust them (the C ABI's
// contract: ids of the adjacency are < n_major; the host module validates a graph once).  An EXPERIMENT that drops or
// alters a hop's gathers breaks that contract -- round 2 lost a box to exactly that -- so experiments
// run on the debug build (`make dbg` -> lib/libtchgeo_hip_dbg.so): every such id is compared with n_major first; an
// offender raises bit 0 of the word registered with tg_debug_bounds_set_flag and is replaced by vertex 0.
#ifdef TG_DEBUG_BOUNDS
struct DebugBounds {
    unsigned int *flag;
    long long n_major;
};
#define TG_BOUNDS_FIELDS tg::DebugBounds dbg;
#define TG_BOUNDS_INIT(p, graph)                                                                                       \
    do {                                                                                                               \
        (p).dbg.flag = tg::debug_bounds_flag();                                                                        \
        (p).dbg.n_major = (graph)->n_major;                                                                            \
    } while (0)
#define TG_CHECK_VERTEX(p, w)                                                                                          \
    do {                                                                                                               \
        if ((unsigned long long)(w) >= (unsigned long long)(p).dbg.n_major) {                                          \
            if ((p).dbg.flag) atomicOr((p).dbg.flag, 1u);                                                              \
            (w) = 0;                                                                                                   \
        }                                                                                                              \
    } while (0)
#else
#define TG_BOUNDS_FIELDS
#define TG_BOUNDS_INIT(p, graph)                                                                                       \
    do {                                                                                                               \
    } while (0)
#define TG_CHECK_VERTEX(p, w)                                                                                          \
    do {                                                                                                               \
    } while (0)
#endif
unsigned int *debug_bounds_flag(); // host side (api.hip): the registered device word, or NULL

// ---- wave64 helpers -------------------------------------------------------
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

template <typename T> __device__ __forceinline__ T wave_inclusive_scan(T v) {
    const int lane = lane_id();
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const T u = __shfl_up(v, off, 64);
        if (lane >= off) v += u;
    }
    return v;
}
// Inclusive prefix sum of one u32 per lane with DPP row shifts / row broadcasts: six v_add_u32 with a DPP operand
// instead of six ds_bpermute round trips through the LDS crossbar.  All 64 lanes must be active.
__device__ __forceinline__ uint32_t wave_inclusive_scan_u32_dpp(uint32_t v) {
    // row_shr:n = 0x110 + n (inside each row of 16 lanes; lanes without a source keep `old` = 0)
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
    // row_bcast:15 (0x142) into rows 1 and 3, then row_bcast:31 (0x143) into rows 2 and 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
    return v;
}
template <typename T> __device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
// 32-bit counts (every hot kernel's per-chunk scan) take the DPP form; call sites are wave-uniform
template <> __device__ __forceinline__ uint32_t wave_inclusive_scan<uint32_t>(uint32_t v) {
    return wave_inclusive_scan_u32_dpp(v);
}
template <> __device__ __forceinline__ uint32_t wave_sum<uint32_t>(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_readlane((int)wave_inclusive_scan_u32_dpp(v), 63);
}
// orders LDS traffic between lanes of ONE wave (the lanes run in lockstep; this
// only stops the compiler from moving accesses across the hand-off)
__device__ __forceinline__ void wave_lds_handoff() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// philox-mode's BLOCKED running weight sum (the CPU checker restates it as orc_blocked_prefix): inclusive prefix of one f64 per
// lane as a Kogge-Stone scan -- for d = 1, 2, 4, 8, 16, 32: v += (lane >= d ? v of lane - d : nothing), every lane at
// once -- then `carry` (the left-to-right sum of the earlier chunks' totals) is added.  Six dependent adds per chunk of 64
// candidates instead of 64: the reference's literal left-to-right sum made the weighted sampler's time the length of the
// longest column's chain.  *total = the value after lane 63.  All 64 lanes must be active.
__device__ __forceinline__ double wave_blocked_prefix_f64(double v, double carry, double *total) {
    const int lane = lane_id();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const double u = __shfl_up(v, d, 64);
        if (lane >= d) v = v + u;
    }
    *total = carry + __shfl(v, 63, 64);
    return carry + v;
}

// Left-to-right inclusive prefix of one f64 per lane starting from `carry`: lane l receives
// (((carry + v0) + v1) + ... + vl) with exactly the reference's rounding (sampling.rs:40,48 sums weights in
// candidate order).  The chain is inherently serial, so it is kept as short as the hardware allows: the 64 values
// go to LDS (`buf`, 64 doubles owned by this wavefront), ONE lane runs the dependent v_add_f64 chain over them
// (the LDS reads pipeline ahead of the adds, the writes trail them), every lane then reads its prefix.
// *total = the value after lane 63.
__device__ __forceinline__ double wave_serial_prefix_f64(double v, double carry, double *total, double *buf) {
    const int lane = lane_id();
    buf[lane] = v;
    wave_lds_handoff();
    if (lane == 0) {
        double r = carry;
#pragma unroll
        for (int l = 0; l < 64; ++l) {
            r = r + buf[l];
            buf[l] = r;
        }
    }
    wave_lds_handoff();
    const double mine = buf[lane];
    *total = buf[63];
    wave_lds_handoff();
    return mine;
}

} // namespace tg
