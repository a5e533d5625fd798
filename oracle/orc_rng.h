/*
 * oracle/orc_rng.h -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).
 *
 * Two interchangeable random sources for the CPU restatement of
 * tch-geometric's samplers:
 *
 *  ORC_RNG_REF    the reference's own generator: rand 0.8.5 `SmallRng`
 *                 (= Xoshiro256++ on 64-bit targets) with rand's
 *                 `gen_range` semantics.  rand 0.8.5 / rand_core 0.6.3 are
 *                 crates.io dependencies pinned at Cargo.lock:619-642 and are
 *                 NOT under /root/reference; their published algorithm is
 *                 restated here.  Call sites it serves:
 *                 src/utils/sampling.rs:19,49,51,64; src/algo/random_walk.rs:53,54,146;
 *                 src/algo/negative_sampling.rs:34,104,111; src/utils/random.rs:10,16,22.
 *
 *  ORC_RNG_PHILOX counter-addressed Philox4x32-10.  Every draw is named by
 *                 (seed, call_id, tag, id, d0, d1) instead of by its position in
 *                 a sequential stream, so a device can evaluate draws in any
 *                 order and still equal this sequential code bit for bit.
 *
 * PARITY UNPINNED for sampled values: the reference holds no golden vectors
 * for its samplers (all its tests are invariant checks) and no Rust
 * toolchain exists here to produce any.  What pins this file: the
 * Xoshiro256++ / SplitMix64 known answers and the Random123 Philox known
 * answers in tests/test_oracle_rng.py.
 */
#ifndef ORC_RNG_H
#define ORC_RNG_H

#include <stdint.h>
#include <string.h>

#define ORC_RNG_REF 0
#define ORC_RNG_PHILOX 1

/* stream tags (philox mode): one per operator so that two operators called
 * with the same (seed, call_id) never share draws */
#define ORC_TAG_NS_HOMO 1u
#define ORC_TAG_NS_HETERO 2u /* | relation index << 8 */
#define ORC_TAG_RW 3u
#define ORC_TAG_RW_TEMPO 4u
#define ORC_TAG_NEG_HOMO 5u
#define ORC_TAG_NEG_HETERO 6u /* | node-type index << 8 */
#define ORC_TAG_HGT 7u
#define ORC_TAG_RMAT 8u
#define ORC_TAG_SEEDS 9u
#define ORC_TAG_BUDGET 10u /* | node-type index << 8 */
#define ORC_TAG_RW_BIASED 11u

typedef struct {
    int32_t mode;
    int32_t _pad;
    uint64_t s[4];      /* ORC_RNG_REF: Xoshiro256++ state */
    uint64_t seed;      /* ORC_RNG_PHILOX */
    uint64_t call_id;   /* ORC_RNG_PHILOX */
    uint64_t raw_draws; /* statistics: raw u64 words consumed (REF) / philox blocks (PHILOX) */
} orc_rng;

/* ------------------------------------------------------------------ */
/* Xoshiro256++  (rand 0.8.5 rand::rngs::SmallRng on 64-bit)          */
/* ------------------------------------------------------------------ */
static inline uint64_t orc_rotl64(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }

static inline uint64_t orc_xoshiro_next_u64(orc_rng *r) {
    uint64_t *s = r->s;
    uint64_t result = orc_rotl64(s[0] + s[3], 23) + s[0];
    uint64_t t = s[1] << 17;
    s[2] ^= s[0];
    s[3] ^= s[1];
    s[1] ^= s[2];
    s[0] ^= s[3];
    s[2] ^= t;
    s[3] = orc_rotl64(s[3], 45);
    r->raw_draws++;
    return result;
}
/* rand 0.8.5 xoshiro256plusplus.rs: next_u32 takes the UPPER 32 bits */
static inline uint32_t orc_xoshiro_next_u32(orc_rng *r) { return (uint32_t)(orc_xoshiro_next_u64(r) >> 32); }

/* SeedableRng::seed_from_u64 for Xoshiro256++: SplitMix64 x4 */
static inline void orc_xoshiro_seed_from_u64(orc_rng *r, uint64_t state) {
    for (int i = 0; i < 4; i++) {
        state += 0x9e3779b97f4a7c15ULL;
        uint64_t z = state;
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
        z = z ^ (z >> 31);
        r->s[i] = z;
    }
}
/* SeedableRng::from_seed([u8;32]): little-endian 4xu64; all-zero -> seed_from_u64(0) */
static inline void orc_xoshiro_from_seed(orc_rng *r, const uint8_t seed[32]) {
    int all_zero = 1;
    for (int i = 0; i < 32; i++) all_zero &= (seed[i] == 0);
    memset(r, 0, sizeof(*r));
    r->mode = ORC_RNG_REF;
    if (all_zero) {
        orc_xoshiro_seed_from_u64(r, 0);
        return;
    }
    for (int i = 0; i < 4; i++) {
        uint64_t v = 0;
        for (int b = 7; b >= 0; b--) v = (v << 8) | seed[i * 8 + b];
        r->s[i] = v;
    }
}
/* SeedableRng::from_rng(parent) -- src/utils/random.rs:19-22 `rng_get` */
static inline void orc_xoshiro_from_rng(orc_rng *child, orc_rng *parent) {
    uint8_t seed[32];
    for (int i = 0; i < 4; i++) {
        uint64_t v = orc_xoshiro_next_u64(parent);
        for (int b = 0; b < 8; b++) seed[i * 8 + b] = (uint8_t)(v >> (8 * b));
    }
    orc_xoshiro_from_seed(child, seed);
}

/* rand 0.8.5 UniformInt<u64/usize/i64>::sample_single(0, range): widening
 * multiply with the leading-zeros rejection zone. range > 0. */
static inline uint64_t orc_ref_gen_range_u64(orc_rng *r, uint64_t range) {
    uint64_t zone = (range << __builtin_clzll(range)) - 1;
    for (;;) {
        uint64_t v = orc_xoshiro_next_u64(r);
        unsigned __int128 m = (unsigned __int128)v * range;
        uint64_t hi = (uint64_t)(m >> 64), lo = (uint64_t)m;
        if (lo <= zone) return hi;
    }
}
/* rand 0.8.5 UniformFloat<f32>::sample_single(0.0, high) */
static inline float orc_ref_gen_range_f32(orc_rng *r, float high) {
    float scale = high;
    for (;;) {
        uint32_t u = orc_xoshiro_next_u32(r) >> 9;
        uint32_t bits = 0x3F800000u | u;
        float v12;
        memcpy(&v12, &bits, 4);
        float res = (v12 - 1.0f) * scale + 0.0f;
        if (res < high) return res;
    }
}
/* rand 0.8.5 UniformFloat<f64>::sample_single(0.0, high) */
static inline double orc_ref_gen_range_f64(orc_rng *r, double high) {
    double scale = high;
    for (;;) {
        uint64_t u = orc_xoshiro_next_u64(r) >> 12;
        uint64_t bits = 0x3FF0000000000000ULL | u;
        double v12;
        memcpy(&v12, &bits, 8);
        double res = (v12 - 1.0) * scale + 0.0;
        if (res < high) return res;
    }
}

/* ------------------------------------------------------------------ */
/* Philox4x32-10 (Salmon et al., SC'11; Random123 constants)          */
/* ------------------------------------------------------------------ */
static inline void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int round = 0; round < 10; round++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

typedef struct { uint32_t k[2]; } orc_callkey;

/* per-call key: Philox(key = seed, ctr = (call_id, tag, "tchg")) words 0,1 */
static inline orc_callkey orc_philox_callkey(uint64_t seed, uint64_t call_id, uint32_t tag) {
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    uint32_t ctr[4] = {(uint32_t)call_id, (uint32_t)(call_id >> 32), tag, 0x74636867u};
    uint32_t out[4];
    orc_philox4x32_10(ctr, key, out);
    orc_callkey ck = {{out[0], out[1]}};
    return ck;
}

typedef struct { uint64_t a, b; uint32_t w[4]; } orc_draw;

/* the draw named (callkey, id, d0, d1): a = words 0,1 ; b = words 2,3 */
static inline orc_draw orc_philox_draw(orc_callkey ck, uint64_t id, uint32_t d0, uint32_t d1) {
    uint32_t ctr[4] = {d0, d1, (uint32_t)id, (uint32_t)(id >> 32)};
    orc_draw d;
    orc_philox4x32_10(ctr, ck.k, d.w);
    d.a = (uint64_t)d.w[0] | ((uint64_t)d.w[1] << 32);
    d.b = (uint64_t)d.w[2] | ((uint64_t)d.w[3] << 32);
    return d;
}
/* multiply-shift bounded integer in [0, range) (bias <= range / 2^64) */
static inline uint64_t orc_bounded(uint64_t x, uint64_t range) {
    return (uint64_t)(((unsigned __int128)x * range) >> 64);
}
/* Bounded integer from ONE 32-bit word: Lemire's multiply-shift with its exact rejection test ("Fast Random Integer
 * Generation in an Interval", 2019).  *ok = 0 when the word must be rejected (probability < range / 2^32); an accepted
 * word gives an exactly uniform value in [0, range).  0 < range < 2^32. */
static inline uint32_t orc_bounded_word(uint32_t w, uint32_t range, int *ok) {
    uint64_t m = (uint64_t)w * range;
    uint32_t lo = (uint32_t)m;
    *ok = 1;
    if (lo < range) {
        uint32_t t = (uint32_t)(0u - range) % range; /* 2^32 mod range */
        if (lo < t) *ok = 0;
    }
    return (uint32_t)(m >> 32);
}
#define ORC_D1_FALLBACK 0x46u /* 'F': or-ed into d1 for the 64-bit draw that replaces a rejected word */

static inline float orc_u32_to_f32_01(uint32_t w) {
    uint32_t bits = 0x3F800000u | (w >> 9);
    float v;
    memcpy(&v, &bits, 4);
    return v - 1.0f;
}
static inline double orc_u64_to_f64_01(uint64_t x) {
    uint64_t bits = 0x3FF0000000000000ULL | (x >> 12);
    double v;
    memcpy(&v, &bits, 8);
    return v - 1.0;
}

#endif
