/*
 * oracle/orc_sampling.h -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).
 *
 * CPU restatement of src/utils/sampling.rs (reference) over a *position space*
 * [0, n): the reference's functions consume an iterator of candidates; since
 * its filters never touch the RNG, "collect the candidates, then sample
 * positions" consumes the RNG identically.
 *
 * In ORC_RNG_REF mode every function is the reference loop, draw for draw.
 * In ORC_RNG_PHILOX mode each draw is addressed by a counter (orc_rng.h) and
 * `orc_reservoir` offers two algorithms:
 *   ORC_RES_LITERAL  the reference loop with one addressed draw per item m>=k;
 *   ORC_RES_TICKETS  a closed form of the SAME output distribution that needs
 *                    k draws instead of n-k (derivation in DESIGN.md
 *                    "Reservoir by tickets"; equality of the two distributions is
 *                    checked exhaustively in tests/test_reservoir_equivalence.py).
 */
#ifndef ORC_SAMPLING_H
#define ORC_SAMPLING_H

#include "orc_rng.h"

#define ORC_RES_TICKETS 0
#define ORC_RES_LITERAL 1
#define ORC_RES_CHUNKED 2 /* one slot only: one draw per chunk of 64 raw positions (orc_reservoir_one_chunked) */
#define ORC_RES_AUTO (-1) /* what the device uses: tickets for neighbor sampling, chunked for the temporal walk */

typedef struct {
    orc_rng *rng;
    orc_callkey ck; /* philox mode */
} orc_ctx;

static inline void orc_ctx_init(orc_ctx *c, orc_rng *rng, uint32_t tag) {
    c->rng = rng;
    c->ck.k[0] = c->ck.k[1] = 0;
    if (rng->mode == ORC_RNG_PHILOX) c->ck = orc_philox_callkey(rng->seed, rng->call_id, tag);
}
static inline orc_draw orc_ctx_draw(orc_ctx *c, uint64_t id, uint32_t d0, uint32_t d1) {
    c->rng->raw_draws++;
    return orc_philox_draw(c->ck, id, d0, d1);
}

/* Bounded draw of SLOT s of a reservoir / replacement sample (philox-mode's definition since round 4; the kernels'
 * tg::slot_draw): slot s owns ONE 32-bit word -- word (s & 3) of the Philox block (id, d0_base + (s >> 2), d1) -- bounded
 * by Lemire's multiply-shift with its exact rejection test; a rejected word (probability < range / 2^32) is replaced by
 * the 64-bit multiply-shift draw of a block of the slot's own, (id, d0_base + s, d1 | ORC_D1_FALLBACK).  Exactly uniform
 * up to the 2^-64 * range bias every 64-bit draw of this file has.  Before round 4 a slot took half a block (64 bits):
 * five blocks per vertex at k = 10 where three do now -- the ticket draws are what bounds the sampling kernels
 * (DESIGN.md 4.1).  Ranges of 2^32 and more take the 64-bit draw directly.  *blk caches the current block. */
static uint64_t orc_slot_fallbacks = 0; /* statistics for the tests (single-threaded use): 64-bit fallback draws taken */
static inline uint64_t orc_slot_draw(orc_ctx *c, uint64_t id, uint32_t d0_base, uint32_t d1, int64_t s, orc_draw *blk,
                                     uint64_t range) {
    if ((s & 3) == 0) *blk = orc_ctx_draw(c, id, d0_base + (uint32_t)(s >> 2), d1); /* slots are drawn in order from 0 */
    if (range < ((uint64_t)1 << 32)) {
        int ok;
        uint32_t r = orc_bounded_word(blk->w[s & 3], (uint32_t)range, &ok);
        if (ok) return r;
    }
    orc_slot_fallbacks++;
    orc_draw f = orc_ctx_draw(c, id, d0_base + (uint32_t)s, d1 | ORC_D1_FALLBACK);
    return orc_bounded(f.a, range);
}

/* Reservoir by tickets (philox mode). n > k >= 1.
 * The reference loop of sampling.rs:17-24 (item i >= k draws j from 0..i and
 * overwrites slot j when j < k) leaves in slot s the LAST item that hit it.
 * Conditioning slot by slot, the last hit of slot s is uniform over the
 * (n-k) positions k..n-1 not claimed by slots 0..s-1, and "never hit" has
 * weight (k-1) minus the blanks already used -- i.e. the k slots draw k
 * tickets WITHOUT replacement from an urn of (n-k) position tickets and
 * (k-1) blanks (derivation: DESIGN.md "Reservoir by tickets"; exhaustive
 * check: tests/test_reservoir_equivalence.py).  A blank leaves item s in
 * slot s.  The ordered draw is a partial Fisher-Yates over ticket indices
 * [0, n-1): ticket tau < n-k is position k+tau, the rest are blanks.
 * One bounded draw per slot (orc_slot_draw: one 32-bit Philox word each).
 * `scratch` holds 2k int64 (the displaced-entry list of the shuffle). */
static inline void orc_reservoir_tickets(orc_ctx *c, uint64_t id, uint32_t d0_base, int64_t n, int64_t k,
                                         int64_t *dst, int64_t *scratch) {
    int64_t *keys = scratch, *vals = scratch + k;
    orc_draw d = {0};
    for (int64_t s = 0; s < k; s++) {
        int64_t m = (n - 1) - s; /* tickets left in the urn */
        int64_t r = (int64_t)orc_slot_draw(c, id, d0_base, 0, s, &d, (uint64_t)m), last = m - 1;
        int64_t tr = r, tl = last;
        for (int64_t j = 0; j < s; j++) { /* latest entry for an index wins */
            if (keys[j] == r) tr = vals[j];
            if (keys[j] == last) tl = vals[j];
        }
        keys[s] = r; /* the last ticket of the urn moves into the hole */
        vals[s] = tl;
        dst[s] = (tr < n - k) ? k + tr : s;
    }
}

/* src/utils/sampling.rs:6-26 reservoir_sampling over positions [0,n) into k
 * slots; returns min(n,k).  NOTE the reference quirk kept on purpose: the
 * index for item i is drawn from 0..i (not 0..=i), sampling.rs:19. */
static inline int64_t orc_reservoir(orc_ctx *c, uint64_t id, uint32_t d0_base, int64_t n, int64_t k, int64_t *dst,
                                    int64_t *scratch, int algo) {
    int64_t filled = n < k ? n : k;
    for (int64_t i = 0; i < filled; i++) dst[i] = i; /* sampling.rs:12-15 */
    if (n <= k) return filled;
    if (c->rng->mode == ORC_RNG_REF) {
        for (int64_t i = k; i < n; i++) { /* sampling.rs:17-24 */
            uint64_t j = orc_ref_gen_range_u64(c->rng, (uint64_t)i);
            if (j < (uint64_t)k) dst[j] = i;
        }
    } else if (algo == ORC_RES_LITERAL) {
        for (int64_t i = k; i < n; i++) {
            orc_draw d = orc_ctx_draw(c, id, (uint32_t)i, 0x4C495400u);
            uint64_t j = orc_bounded(d.a, (uint64_t)i);
            if (j < (uint64_t)k) dst[j] = i;
        }
    } else {
        orc_reservoir_tickets(c, id, d0_base, n, k, dst, scratch);
    }
    return filled;
}

/* The ONE-SLOT reservoir of the temporal walk (sampling.rs:12-24 with k = 1) with one draw per CHUNK of 64 raw row
 * positions instead of one per candidate -- philox-mode's definition since round 3 (the kernel forms it with a handful of
 * wave-wide operations per chunk; a Philox block per candidate was half of its time).  The law is the literal loop's:
 * with n candidates the loop ends on candidate 0 iff n == 1 and on each of candidates 1..n-1 with probability 1/(n-1)
 * (item i draws from 0..i, so item 1 always replaces item 0: the quirk noted above).  Here the candidates of rank >= 1
 * are ELIGIBLE; a chunk holding m of them, after M eligible ones in earlier chunks, takes the slot with probability
 * m / (M + m) (always when M = 0) and then holds each of its m with probability 1/m: every eligible candidate ends up
 * chosen with probability 1/(n-1).  Draw of a chunk: (id, d0 = chunk index, d1 = "CHK"); half a decides the take, half b
 * the candidate.  raw_pos[i] = candidate i's raw position (ascending).  Returns the chosen candidate, -1 when n == 0. */
static inline int64_t orc_reservoir_one_chunked(orc_ctx *c, uint64_t id, int64_t n, const int64_t *raw_pos) {
    if (n == 0) return -1;
    int64_t pick = 0, seen = 0; /* eligible candidates before the current chunk */
    int64_t i = 1;
    while (i < n) {
        const int64_t chunk = raw_pos[i] >> 6;
        int64_t j = i;
        while (j < n && (raw_pos[j] >> 6) == chunk) j++;
        const int64_t m = j - i;
        const orc_draw d = orc_ctx_draw(c, id, (uint32_t)chunk, 0x43484B00u);
        if (seen == 0 || orc_bounded(d.a, (uint64_t)(seen + m)) < (uint64_t)m) pick = i + (int64_t)orc_bounded(d.b, (uint64_t)m);
        seen += m;
        i = j;
    }
    return pick;
}

/* src/utils/sampling.rs:57-69 replacement_sampling: k draws from [0,n), n>0 */
static inline int64_t orc_replacement(orc_ctx *c, uint64_t id, int64_t n, int64_t k, int64_t *dst) {
    orc_draw d = {0};
    for (int64_t s = 0; s < k; s++) {
        if (c->rng->mode == ORC_RNG_REF)
            dst[s] = (int64_t)orc_ref_gen_range_u64(c->rng, (uint64_t)n);
        else
            dst[s] = (int64_t)orc_slot_draw(c, id, 0, 0x52455000u, s, &d, (uint64_t)n);
    }
    return k;
}

/* philox-mode's running weight sum: BLOCKED, so that a wavefront can form it without a 64-deep dependent chain.
 * The candidates' RAW positions (position in the column / in the live list; excluded positions count as weight 0.0,
 * which an IEEE sum passes through unchanged) are cut into chunks of 64 from position 0; inside a chunk the inclusive
 * prefix is a Kogge-Stone scan -- for d = 1, 2, 4, 8, 16, 32: y[i] += y[i - d] (i >= d), all lanes at once -- in f64; the
 * sum before a chunk (`carry`) is the left-to-right sum of the previous chunks' totals y[63]; the running sum at raw
 * position r is carry + y[r mod 64].  ref-mode keeps the reference's literal left-to-right sum (sampling.rs:40,48); the two
 * differ in the last bits of a double only, and the kernels implement exactly this order (tg_device.h
 * wave_blocked_prefix_f64). */
static inline void orc_blocked_prefix(const double *x, int64_t n_raw, double *out) {
    double carry = 0.0;
    for (int64_t base = 0; base < n_raw; base += 64) {
        double y[64], t[64];
        for (int i = 0; i < 64; i++) y[i] = (base + i < n_raw) ? x[base + i] : 0.0;
        for (int d = 1; d < 64; d <<= 1) {
            for (int i = 0; i < 64; i++) t[i] = (i >= d) ? y[i] + y[i - d] : y[i];
            for (int i = 0; i < 64; i++) y[i] = t[i];
        }
        for (int i = 0; i < 64 && base + i < n_raw; i++) out[base + i] = carry + y[i];
        carry = carry + y[63];
    }
}

/* the same in f32 (biased_tempo_random_walk's one-slot reservoir, random_walk.rs:264-271 with f32 weights) */
static inline void orc_blocked_prefix_f32(const float *x, int64_t n_raw, float *out) {
    float carry = 0.0f;
    for (int64_t base = 0; base < n_raw; base += 64) {
        float y[64], t[64];
        for (int i = 0; i < 64; i++) y[i] = (base + i < n_raw) ? x[base + i] : 0.0f;
        for (int d = 1; d < 64; d <<= 1) {
            for (int i = 0; i < 64; i++) t[i] = (i >= d) ? y[i] + y[i - d] : y[i];
            for (int i = 0; i < 64; i++) y[i] = t[i];
        }
        for (int i = 0; i < 64 && base + i < n_raw; i++) out[base + i] = carry + y[i];
        carry = carry + y[63];
    }
}

/* src/utils/sampling.rs:28-55 reservoir_sampling_weighted over positions
 * [0,n) with weights w[pos]; returns min(n,k), or -1 where the reference
 * panics (empty float range: running weight sum <= 0, sampling.rs:49).
 * raw_pos[i] (ascending; NULL = i) is candidate i's raw position among n_raw, which philox-mode's blocked running sum is
 * defined on (above); ref-mode sums left to right as the reference does. */
static inline int64_t orc_reservoir_weighted(orc_ctx *c, uint64_t id, int64_t n, int64_t k, const double *w,
                                             int64_t *dst, const int64_t *raw_pos, int64_t n_raw) {
    int64_t filled = 0;
    double w_sum = 0.0;
    double *blocked = NULL;
    if (c->rng->mode != ORC_RNG_REF && n > 0) {
        if (!raw_pos) n_raw = n;
        double *x = (double *)calloc((size_t)n_raw, sizeof(double));
        blocked = (double *)malloc(sizeof(double) * (size_t)n_raw);
        for (int64_t i = 0; i < n; i++) x[raw_pos ? raw_pos[i] : i] = w[i];
        orc_blocked_prefix(x, n_raw, blocked);
        free(x);
    }
    for (int64_t i = 0; i < k && i < n; i++) { /* sampling.rs:37-45 */
        dst[i] = i;
        w_sum = w_sum + w[i];
        filled++;
    }
    for (int64_t i = k; i < n; i++) { /* sampling.rs:47-53 */
        w_sum = blocked ? blocked[raw_pos ? raw_pos[i] : i] : w_sum + w[i];
        if (!(0.0 < w_sum)) {
            free(blocked);
            return -1;
        }
        double j;
        uint64_t slot_x = 0;
        if (c->rng->mode == ORC_RNG_REF) {
            j = orc_ref_gen_range_f64(c->rng, w_sum);
        } else {
            orc_draw d = orc_ctx_draw(c, id, (uint32_t)i, 0x57475400u);
            j = orc_u64_to_f64_01(d.a) * w_sum + 0.0;
            slot_x = d.b;
        }
        if (j < w[i]) {
            uint64_t slot = (c->rng->mode == ORC_RNG_REF) ? orc_ref_gen_range_u64(c->rng, (uint64_t)k)
                                                          : orc_bounded(slot_x, (uint64_t)k);
            dst[slot] = i;
        }
    }
    free(blocked);
    return filled;
}

#endif
