/*
 * oracle/tg_oracle.c -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).
 *
 * CPU restatement, in plain C, of tch-geometric's mini-batch construction
 * path.  Each function cites the reference file:line it follows.  Nothing in
 * the shipped package links, loads or calls this file; only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
 *
 * PARITY UNPINNED for sampled values (no golden vectors exist in the
 * reference and its Rust crate cannot be built here -- SURVEY.md 8(c)).
 * Pinned exactly: orc_ind2ptr / orc_to_csx against src/data/storage.rs:152-184.
 *
 * Canonical iteration order.  Where the reference iterates a std HashMap
 * (order differs from process to process: neighbor_sampling.rs:294,345;
 * negative_sampling.rs:99; hgt_sampling.rs:47,167,183,201,227,247) this
 * restatement visits relations in `edge_types` order and node types in
 * `node_types` order.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "orc_sampling.h"

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------ */
/* RNG entry points for tests                                          */
/* ------------------------------------------------------------------ */
ORC_API void orc_rng_ref_from_seed(orc_rng *r, const uint8_t *seed32) { orc_xoshiro_from_seed(r, seed32); }
ORC_API void orc_rng_ref_from_state(orc_rng *r, const uint64_t *s4) {
    memset(r, 0, sizeof(*r));
    r->mode = ORC_RNG_REF;
    memcpy(r->s, s4, 32);
}
/* src/utils/random.rs:19-22 rng_get(): child = SmallRng::from_rng(global) */
ORC_API void orc_rng_ref_child(orc_rng *child, orc_rng *parent) { orc_xoshiro_from_rng(child, parent); }
ORC_API void orc_rng_philox(orc_rng *r, uint64_t seed, uint64_t call_id) {
    memset(r, 0, sizeof(*r));
    r->mode = ORC_RNG_PHILOX;
    r->seed = seed;
    r->call_id = call_id;
}
ORC_API uint64_t orc_rng_next_u64(orc_rng *r) { return orc_xoshiro_next_u64(r); }
ORC_API uint64_t orc_rng_gen_range_u64(orc_rng *r, uint64_t range) { return orc_ref_gen_range_u64(r, range); }
ORC_API float orc_rng_gen_range_f32(orc_rng *r, float high) { return orc_ref_gen_range_f32(r, high); }
ORC_API double orc_rng_gen_range_f64(orc_rng *r, double low, double high) {
    /* rand 0.8.5 UniformFloat<f64>::sample_single(low, high) */
    double scale = high - low;
    for (;;) {
        uint64_t u = orc_xoshiro_next_u64(r) >> 12;
        uint64_t bits = 0x3FF0000000000000ULL | u;
        double v12;
        memcpy(&v12, &bits, 8);
        double res = (v12 - 1.0) * scale + low;
        if (res < high) return res;
    }
}
ORC_API void orc_philox_raw(const uint32_t *ctr4, const uint32_t *key2, uint32_t *out4) {
    orc_philox4x32_10(ctr4, key2, out4);
}
ORC_API void orc_philox_named_draw(uint64_t seed, uint64_t call_id, uint32_t tag, uint64_t id, uint32_t d0,
                                   uint32_t d1, uint32_t *out4) {
    orc_draw d = orc_philox_draw(orc_philox_callkey(seed, call_id, tag), id, d0, d1);
    memcpy(out4, d.w, 16);
}

/* the slot draw's primitives exposed for tests/test_oracle_rng.py: the bounded word with its rejection flag, and the whole
 * slot draw (a fresh context per call, so `s` need not start at 0 here: the block is recomputed) */
ORC_API uint32_t orc_test_bounded_word(uint32_t w, uint32_t range, int *ok) { return orc_bounded_word(w, range, ok); }
ORC_API uint64_t orc_test_slot_draw(orc_rng *rng, uint32_t tag, uint64_t id, uint32_t d1, int64_t s, uint64_t range,
                                    int *fell_back) {
    orc_ctx c;
    orc_ctx_init(&c, rng, tag);
    orc_draw blk = orc_philox_draw(c.ck, id, (uint32_t)(s >> 2), d1);
    int ok = 0;
    if (range < ((uint64_t)1 << 32)) (void)orc_bounded_word(blk.w[s & 3], (uint32_t)range, &ok);
    *fell_back = !ok;
    orc_draw cache = blk; /* orc_slot_draw recomputes it only when s is a multiple of 4: hand it the right block */
    return orc_slot_draw(&c, id, 0, d1, s, &cache, range);
}

ORC_API uint64_t orc_slot_fallback_count(int reset) {
    uint64_t v = orc_slot_fallbacks;
    if (reset) orc_slot_fallbacks = 0;
    return v;
}

/* reservoir primitive exposed for the equivalence tests */
ORC_API int64_t orc_reservoir_positions(orc_rng *rng, uint32_t tag, uint64_t id, int64_t n, int64_t k, int algo,
                                        int64_t *dst) {
    orc_ctx c;
    orc_ctx_init(&c, rng, tag);
    int64_t *scratch = (int64_t *)malloc(sizeof(int64_t) * (size_t)(k > 0 ? 2 * k : 2));
    int64_t r = orc_reservoir(&c, id, 0, n, k, dst, scratch, algo);
    free(scratch);
    return r;
}

/* ------------------------------------------------------------------ */
/* small growable vector                                               */
/* ------------------------------------------------------------------ */
typedef struct {
    int64_t *p;
    int64_t n, cap;
} vec64;
static void vpush(vec64 *v, int64_t x) {
    if (v->n == v->cap) {
        v->cap = v->cap ? v->cap * 2 : 64;
        v->p = (int64_t *)realloc(v->p, sizeof(int64_t) * (size_t)v->cap);
    }
    v->p[v->n++] = x;
}
static void vfree(vec64 *v) {
    free(v->p);
    v->p = NULL;
    v->n = v->cap = 0;
}

/* ------------------------------------------------------------------ */
/* Graph ingest: src/data/storage.rs                                   */
/* ------------------------------------------------------------------ */
/* storage.rs:67-101 ind2ptr (sorted `ind`, m rows) -> out[m+1] */
ORC_API void orc_ind2ptr(const int64_t *ind, int64_t numel, int64_t m, int64_t *out) {
    if (numel == 0) {
        for (int64_t i = 0; i <= m; i++) out[i] = 0;
        return;
    }
    for (int64_t i = 0; i <= ind[0]; i++) out[i] = 0;
    int64_t idx = ind[0];
    for (int64_t i = 0; i < numel - 1; i++) {
        int64_t next_idx = ind[i + 1];
        for (int64_t x = idx; x < next_idx; x++) out[x + 1] = i + 1;
        idx = next_idx;
    }
    for (int64_t i = ind[numel - 1] + 1; i < m + 1; i++) out[i] = numel;
}

typedef struct {
    int64_t key, pos;
} keypos;
static int keypos_cmp(const void *a, const void *b) {
    const keypos *x = (const keypos *)a, *y = (const keypos *)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    return x->pos < y->pos ? -1 : (x->pos > y->pos);
}
/* storage.rs:103-126: csr: perm = argsort(row*size1 + col); csc: perm =
 * argsort(col*size0 + row).  The reference's argsort is not documented as
 * stable; ties (duplicate edges) are broken here by original position. */
ORC_API void orc_to_csx(const int64_t *row, const int64_t *col, int64_t nnz, int64_t size0, int64_t size1, int csc,
                        int64_t *ptrs, int64_t *indices, int64_t *perm) {
    keypos *kp = (keypos *)malloc(sizeof(keypos) * (size_t)(nnz ? nnz : 1));
    for (int64_t e = 0; e < nnz; e++) {
        kp[e].key = csc ? col[e] * size0 + row[e] : row[e] * size1 + col[e];
        kp[e].pos = e;
    }
    qsort(kp, (size_t)nnz, sizeof(keypos), keypos_cmp);
    int64_t *major = (int64_t *)malloc(sizeof(int64_t) * (size_t)(nnz ? nnz : 1));
    for (int64_t e = 0; e < nnz; e++) {
        perm[e] = kp[e].pos;
        major[e] = csc ? col[kp[e].pos] : row[kp[e].pos];
        indices[e] = csc ? row[kp[e].pos] : col[kp[e].pos];
    }
    orc_ind2ptr(major, nnz, csc ? size1 : size0, ptrs);
    free(major);
    free(kp);
}

/* src/data/graph.rs:80-83 has_edge: binary search of x's (sorted) row */
static int orc_has_edge(const int64_t *ptrs, const int64_t *indices, int64_t x, int64_t y) {
    int64_t lo = ptrs[x], hi = ptrs[x + 1];
    while (lo < hi) {
        int64_t mid = lo + (hi - lo) / 2;
        int64_t v = indices[mid];
        if (v == y) return 1;
        if (v < y)
            lo = mid + 1;
        else
            hi = mid;
    }
    return 0;
}

/* ------------------------------------------------------------------ */
/* Neighbor sampling: src/algo/neighbor_sampling.rs                    */
/* ------------------------------------------------------------------ */
#define ORC_SAMPLER_UNIFORM 0     /* UnweightedSampler<false> neighbor_sampling.rs:124-127 */
#define ORC_SAMPLER_UNIFORM_REPL 1 /* UnweightedSampler<true>  neighbor_sampling.rs:111-123 */
#define ORC_SAMPLER_WEIGHTED 2    /* WeightedSampler<f64>     neighbor_sampling.rs:141-158 */

#define ORC_FILTER_NONE (-1) /* IdentityFilter neighbor_sampling.rs:22-30 */
#define ORC_FILTER_STATIC 0
#define ORC_FILTER_RELATIVE 1
#define ORC_FILTER_DYNAMIC 2

typedef struct {
    int32_t sampler;
    int32_t filter_mode;
    int32_t forward;
    int32_t reservoir_algo;
    int64_t win_lo, win_hi;    /* inclusive window, python.rs:150 */
    const double *weights;     /* per edge ptr (f64, python.rs:214) */
    const int64_t *timestamps; /* per edge ptr (i64, python.rs:149) */
    int64_t id_base;           /* philox mode: draw id of slot i is id_base + i */
} orc_ns_cfg;

/* neighbor_sampling.rs:55-67 TemporalFilter::filter */
static inline int orc_filter_pass(const orc_ns_cfg *cfg, int64_t state, int64_t e) {
    if (cfg->filter_mode == ORC_FILTER_NONE) return 1;
    int64_t t = cfg->timestamps[e];
    int64_t x;
    if (cfg->filter_mode == ORC_FILTER_STATIC)
        x = t;
    else
        x = cfg->forward ? (t - state) : -(t - state);
    return cfg->win_lo <= x && x <= cfg->win_hi;
}
/* neighbor_sampling.rs:69-76 TemporalFilter::mutate */
static inline int64_t orc_filter_mutate(const orc_ns_cfg *cfg, int64_t state, int64_t e) {
    if (cfg->filter_mode == ORC_FILTER_DYNAMIC) return cfg->timestamps[e];
    return state;
}

typedef struct {
    int64_t *cand;  /* candidate edge ptrs (filtered) */
    double *cw;     /* their weights */
    int64_t cap;
    int64_t *dst;   /* k positions */
    int64_t *scratch;
    int64_t k;
} orc_sampler_state;

static void sst_init(orc_sampler_state *st, int64_t k) {
    memset(st, 0, sizeof(*st));
    st->k = k;
    st->dst = (int64_t *)malloc(sizeof(int64_t) * (size_t)(k > 0 ? k : 1));
    st->scratch = (int64_t *)malloc(sizeof(int64_t) * (size_t)(k > 0 ? 2 * k : 2));
}
static void sst_free(orc_sampler_state *st) {
    free(st->cand);
    free(st->cw);
    free(st->dst);
    free(st->scratch);
}

/* One frontier vertex: neighbor_sampling.rs:199-208 (range, empty check,
 * filter, Sampler::sample).  Returns the number of selected edge ptrs (written
 * to st->cand-indexed st->dst as edge ptrs in out_eptr), or -1 on the
 * reference's panic. `id` = the vertex's slot in `samples` (philox address). */
static int64_t orc_sample_vertex(orc_ctx *c, const orc_ns_cfg *cfg, orc_sampler_state *st, const int64_t *ptrs,
                                 int64_t w, int64_t w_state, uint64_t id, int64_t *out_eptr) {
    int64_t b = ptrs[w], e = ptrs[w + 1];
    if (e <= b) return 0; /* neighbor_sampling.rs:200-202 */
    int64_t deg = e - b, k = st->k, n;
    int need_list = (cfg->filter_mode != ORC_FILTER_NONE) || cfg->sampler == ORC_SAMPLER_WEIGHTED;
    if (need_list) {
        if (st->cap < deg) {
            st->cap = deg * 2;
            st->cand = (int64_t *)realloc(st->cand, sizeof(int64_t) * (size_t)st->cap);
            st->cw = (double *)realloc(st->cw, sizeof(double) * (size_t)st->cap);
        }
        n = 0;
        for (int64_t p = b; p < e; p++)
            if (orc_filter_pass(cfg, w_state, p)) {
                st->cand[n] = p;
                if (cfg->sampler == ORC_SAMPLER_WEIGHTED) st->cw[n] = cfg->weights[p];
                n++;
            }
    } else {
        n = deg;
    }
    int64_t cnt;
    switch (cfg->sampler) {
    case ORC_SAMPLER_UNIFORM_REPL:
        cnt = (n > 0) ? orc_replacement(c, id, n, k, st->dst) : 0; /* neighbor_sampling.rs:118-122 */
        break;
    case ORC_SAMPLER_WEIGHTED:
        /* raw positions = positions in the column (st->cand holds edge pointers, b the column start) */
        if (cfg->filter_mode != ORC_FILTER_NONE) {
            int64_t *raw = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n ? n : 1));
            for (int64_t q = 0; q < n; q++) raw[q] = st->cand[q] - b;
            cnt = orc_reservoir_weighted(c, id, n, k, st->cw, st->dst, raw, deg);
            free(raw);
        } else
            cnt = orc_reservoir_weighted(c, id, n, k, st->cw, st->dst, NULL, n);
        if (cnt < 0) return -1;
        break;
    default: {
        /* philox-mode spec: the k-draw ticket form, with or without a filter (under a filter the device counts
         * the admissible edges first, then fetches the drawn ranks) */
        int algo = cfg->reservoir_algo;
        if (algo == ORC_RES_AUTO) algo = ORC_RES_TICKETS;
        cnt = orc_reservoir(c, id, 0, n, k, st->dst, st->scratch, algo);
    }
    }
    for (int64_t s = 0; s < cnt; s++) out_eptr[s] = need_list ? st->cand[st->dst[s]] : b + st->dst[s];
    return cnt;
}

/* neighbor_sampling.rs:162-230 neighbor_sampling_homogenous.
 * Caller-allocated outputs: samples[cap_nodes], rows/cols/edge_index[cap_edges],
 * layer_offsets[3*H], counts[2] = {n_samples, n_edges}.
 * Returns 0, -1 on the reference's panic, -2 on capacity overflow. */
ORC_API int orc_ns_homo(const int64_t *ptrs, const int64_t *indices, const int64_t *inputs, int64_t B,
                        const int64_t *fanout, int32_t H, const orc_ns_cfg *cfg, const int64_t *inputs_state,
                        orc_rng *rng, int64_t *samples, int64_t cap_nodes, int64_t *rows, int64_t *cols,
                        int64_t *edge_index, int64_t cap_edges, int64_t *layer_offsets, int64_t *counts) {
    orc_ctx c;
    orc_ctx_init(&c, rng, ORC_TAG_NS_HOMO);
    int has_state = cfg->filter_mode != ORC_FILTER_NONE;
    int64_t *states = has_state ? (int64_t *)malloc(sizeof(int64_t) * (size_t)(cap_nodes ? cap_nodes : 1)) : NULL;
    if (B > cap_nodes) return -2;
    int64_t ns = 0, ne = 0;
    for (int64_t i = 0; i < B; i++) { /* :184-185 */
        samples[ns] = inputs[i];
        if (has_state) states[ns] = inputs_state[i];
        ns++;
    }
    int64_t begin = 0, end = ns, rc = 0;
    for (int32_t h = 0; h < H && rc == 0; h++) { /* :188 */
        int64_t k = fanout[h];
        orc_sampler_state st;
        sst_init(&st, k); /* :190 */
        int64_t *eptr = (int64_t *)malloc(sizeof(int64_t) * (size_t)(k > 0 ? k : 1));
        layer_offsets[3 * h + 0] = ns; /* :193 */
        layer_offsets[3 * h + 1] = ne;
        layer_offsets[3 * h + 2] = ns;
        for (int64_t i = begin; i < end; i++) { /* :195 */
            int64_t w = samples[i];
            int64_t w_state = has_state ? states[i] : 0;
            int64_t cnt = orc_sample_vertex(&c, cfg, &st, ptrs, w, w_state, (uint64_t)(cfg->id_base + i), eptr);
            if (cnt < 0) {
                rc = -1;
                break;
            }
            if (ns + cnt > cap_nodes || ne + cnt > cap_edges) {
                rc = -2;
                break;
            }
            for (int64_t s = 0; s < cnt; s++) { /* :210-218 */
                int64_t ep = eptr[s];
                int64_t v = indices[ep];
                int64_t j = ns;
                samples[ns] = v;
                if (has_state) states[ns] = orc_filter_mutate(cfg, w_state, ep);
                ns++;
                rows[ne] = j;
                cols[ne] = i;
                edge_index[ne] = ep;
                ne++;
            }
        }
        begin = end; /* :221-222 */
        end = ns;
        free(eptr);
        sst_free(&st);
    }
    counts[0] = ns;
    counts[1] = ne;
    free(states);
    return (int)rc;
}

/* neighbor_sampling.rs:233-356 neighbor_sampling_heterogenous (canonical
 * relation order = r = 0..R-1, the caller's `edge_types` order).
 * rel_src[r], rel_dst[r]: node-type indices. fanout[r*H + hop].
 * Results are returned through an opaque handle (sizes are data dependent). */
typedef struct {
    int32_t T, R, H;
    vec64 *samples;       /* [T] */
    vec64 *ts;            /* [T] sample timestamps (hgt_sampling only) */
    vec64 *rows, *cols, *eidx; /* [R] */
    int64_t *layer_offsets;    /* [R*H*3] */
} orc_het_out;

ORC_API void orc_het_free(orc_het_out *o) {
    if (!o) return;
    for (int t = 0; t < o->T; t++) {
        vfree(&o->samples[t]);
        vfree(&o->ts[t]);
    }
    for (int r = 0; r < o->R; r++) {
        vfree(&o->rows[r]);
        vfree(&o->cols[r]);
        vfree(&o->eidx[r]);
    }
    free(o->samples);
    free(o->ts);
    free(o->rows);
    free(o->cols);
    free(o->eidx);
    free(o->layer_offsets);
    free(o);
}
ORC_API int64_t orc_het_num_samples(const orc_het_out *o, int t) { return o->samples[t].n; }
ORC_API int64_t orc_het_num_edges(const orc_het_out *o, int r) { return o->rows[r].n; }
ORC_API void orc_het_copy_samples(const orc_het_out *o, int t, int64_t *dst) {
    memcpy(dst, o->samples[t].p, sizeof(int64_t) * (size_t)o->samples[t].n);
}
ORC_API void orc_het_copy_sample_ts(const orc_het_out *o, int t, int64_t *dst) {
    memcpy(dst, o->ts[t].p, sizeof(int64_t) * (size_t)o->ts[t].n);
}
ORC_API void orc_het_copy_edges(const orc_het_out *o, int r, int64_t *rows, int64_t *cols, int64_t *eidx) {
    size_t nb = sizeof(int64_t) * (size_t)o->rows[r].n;
    memcpy(rows, o->rows[r].p, nb);
    memcpy(cols, o->cols[r].p, nb);
    if (eidx) memcpy(eidx, o->eidx[r].p, nb);
}
ORC_API void orc_het_copy_layer_offsets(const orc_het_out *o, int64_t *dst) {
    memcpy(dst, o->layer_offsets, sizeof(int64_t) * (size_t)(o->R * o->H * 3));
}

static orc_het_out *het_alloc(int32_t T, int32_t R, int32_t H) {
    orc_het_out *o = (orc_het_out *)calloc(1, sizeof(*o));
    o->T = T;
    o->R = R;
    o->H = H;
    o->samples = (vec64 *)calloc((size_t)T, sizeof(vec64));
    o->ts = (vec64 *)calloc((size_t)T, sizeof(vec64));
    o->rows = (vec64 *)calloc((size_t)R, sizeof(vec64));
    o->cols = (vec64 *)calloc((size_t)R, sizeof(vec64));
    o->eidx = (vec64 *)calloc((size_t)R, sizeof(vec64));
    o->layer_offsets = (int64_t *)calloc((size_t)(R * (H > 0 ? H : 1) * 3), sizeof(int64_t));
    return o;
}

/* cfgs[r]: per-relation sampler/filter data (weights/timestamps differ per
 * relation, python.rs:123-128,154-162). inputs_state[t] may be NULL. */
ORC_API orc_het_out *orc_ns_hetero(int32_t T, int32_t R, const int32_t *rel_src, const int32_t *rel_dst,
                                   const int64_t *const *ptrs, const int64_t *const *indices,
                                   const int64_t *const *inputs, const int64_t *n_inputs, const int64_t *fanout,
                                   int32_t H, const orc_ns_cfg *cfgs, const int64_t *const *inputs_state,
                                   orc_rng *rng, int32_t *status) {
    orc_het_out *o = het_alloc(T, R, H);
    int has_state = R > 0 && cfgs[0].filter_mode != ORC_FILTER_NONE;
    vec64 *states = (vec64 *)calloc((size_t)T, sizeof(vec64));
    int64_t *sl_begin = (int64_t *)calloc((size_t)T, sizeof(int64_t));
    int64_t *sl_end = (int64_t *)calloc((size_t)T, sizeof(int64_t));
    *status = 0;
    for (int t = 0; t < T; t++) { /* :264-278 */
        for (int64_t i = 0; i < n_inputs[t]; i++) {
            vpush(&o->samples[t], inputs[t][i]);
            if (has_state) vpush(&states[t], inputs_state[t] ? inputs_state[t][i] : 0);
        }
        sl_begin[t] = 0; /* :288-290 */
        sl_end[t] = o->samples[t].n;
    }
    for (int32_t ell = 0; ell < H && *status == 0; ell++) { /* :292 */
        for (int32_t r = 0; r < R && *status == 0; r++) { /* :294 (canonical order) */
            int64_t k = fanout[r * H + ell];
            int32_t st_ = rel_src[r], dt = rel_dst[r];
            const orc_ns_cfg *cfg = &cfgs[r];
            orc_ctx c;
            orc_ctx_init(&c, rng, ORC_TAG_NS_HETERO | ((uint32_t)r << 8));
            orc_sampler_state st;
            sst_init(&st, k); /* :301 */
            int64_t *eptr = (int64_t *)malloc(sizeof(int64_t) * (size_t)(k > 0 ? k : 1));
            int64_t *lo = &o->layer_offsets[(r * H + ell) * 3]; /* :314-315 */
            lo[0] = o->samples[st_].n;
            lo[1] = o->rows[r].n;
            lo[2] = o->samples[dt].n;
            for (int64_t i = sl_begin[dt]; i < sl_end[dt]; i++) { /* :317-318 */
                int64_t w = o->samples[dt].p[i];
                int64_t w_state = has_state ? states[dt].p[i] : 0;
                int64_t cnt = orc_sample_vertex(&c, cfg, &st, ptrs[r], w, w_state, (uint64_t)i, eptr);
                if (cnt < 0) {
                    *status = -1;
                    break;
                }
                for (int64_t s = 0; s < cnt; s++) { /* :333-341 */
                    int64_t ep = eptr[s];
                    int64_t v = indices[r][ep];
                    int64_t j = o->samples[st_].n;
                    vpush(&o->samples[st_], v);
                    if (has_state) vpush(&states[st_], orc_filter_mutate(cfg, w_state, ep));
                    vpush(&o->rows[r], j);
                    vpush(&o->cols[r], i);
                    vpush(&o->eidx[r], ep);
                }
            }
            free(eptr);
            sst_free(&st);
        }
        for (int t = 0; t < T; t++) { /* :345-348 */
            sl_begin[t] = sl_end[t];
            sl_end[t] = o->samples[t].n;
        }
    }
    for (int t = 0; t < T; t++) vfree(&states[t]);
    free(states);
    free(sl_begin);
    free(sl_end);
    return o;
}

/* ------------------------------------------------------------------ */
/* Random walks: src/algo/random_walk.rs                               */
/* ------------------------------------------------------------------ */
/* random_walk.rs:10-75 random_walk (node2vec rejection sampling).
 * walks: [n, walk_length+1], filled with -1 first (:19-23). */
ORC_API void orc_random_walk(const int64_t *ptrs, const int64_t *indices, const int64_t *start, int64_t n,
                             int64_t walk_length, float p, float q, orc_rng *rng, int64_t *walks) {
    orc_ctx c;
    orc_ctx_init(&c, rng, ORC_TAG_RW);
    int64_t L = walk_length + 1;
    for (int64_t i = 0; i < n * L; i++) walks[i] = -1;
    /* :29-36, all in f32 */
    float inv_p = 1.0f / p, inv_q = 1.0f / q;
    float max_prob = inv_p;
    if (1.0f >= max_prob) max_prob = 1.0f;
    if (inv_q >= max_prob) max_prob = inv_q;
    float prob0 = 1.0f / p / max_prob;
    float prob1 = 1.0f / max_prob;
    float prob2 = 1.0f / q / max_prob;
    for (int64_t i = 0; i < n; i++) { /* :38 */
        int64_t prev = -1, cur = start[i];
        walks[i * L] = cur;
        for (int64_t l = 0; l < walk_length; l++) { /* :43 */
            int64_t b = ptrs[cur], e = ptrs[cur + 1];
            if (e <= b) break; /* :45-47 */
            uint64_t deg = (uint64_t)(e - b);
            int64_t next;
            for (uint32_t attempt = 0;; attempt++) { /* :52-66 */
                float r;
                if (rng->mode == ORC_RNG_REF) {
                    next = indices[b + (int64_t)orc_ref_gen_range_u64(rng, deg)];
                    r = orc_ref_gen_range_f32(rng, 1.0f);
                } else {
                    orc_draw d = orc_ctx_draw(&c, (uint64_t)i, (uint32_t)l, attempt);
                    next = indices[b + (int64_t)orc_bounded(d.a, deg)];
                    r = orc_u32_to_f32_01(d.w[2]);
                }
                if (next == prev) {
                    if (r < prob0) break;
                } else if (prev >= 0 && orc_has_edge(ptrs, indices, next, prev)) {
                    if (r < prob1) break;
                } else if (r < prob2) {
                    break;
                }
            }
            prev = cur; /* :68-70 */
            cur = next;
            walks[i * L + l + 1] = cur;
        }
    }
}
/* NOTE on `prev >= 0` above: with prev = -1 (first step) the reference calls
 * has_edge(next, -1) = binary search for -1 in a row of non-negative ids,
 * which is always false; the guard states the same result. */

#define ORC_NAN_TS (-1) /* random_walk.rs:77 */

/* random_walk.rs:80-158 tempo_random_walk; walks, walks_ts: [n, walk_length] */
ORC_API void orc_tempo_random_walk(const int64_t *ptrs, const int64_t *indices, const int64_t *node_ts,
                                   const int64_t *edge_ts, const int64_t *start, const int64_t *start_ts, int64_t n,
                                   int64_t walk_length, int64_t win0, int64_t win1, int32_t reservoir_algo,
                                   orc_rng *rng, int64_t *walks, int64_t *walks_ts) {
    orc_ctx c;
    orc_ctx_init(&c, rng, ORC_TAG_RW_TEMPO);
    int64_t L = walk_length;
    for (int64_t i = 0; i < n * L; i++) {
        walks[i] = -1;
        walks_ts[i] = -1;
    }
    vec64 cn = {0}, ct = {0}, cr = {0}; /* candidates: neighbour, time, raw position in the row */
    for (int64_t i = 0; i < n; i++) { /* :108 */
        int64_t cur = start[i];
        int64_t it = start_ts[i];
        int64_t wlo = it + win0, whi = it + win1; /* half open :111 */
        if (L > 0) {
            walks[i * L] = cur;
            walks_ts[i * L] = it;
        }
        for (int64_t l = 0; l < walk_length - 1; l++) { /* :117 */
            cn.n = ct.n = cr.n = 0;
            for (int64_t e = ptrs[cur]; e < ptrs[cur + 1]; e++) { /* :118-139 */
                int64_t v = indices[e];
                int64_t t = edge_ts[e] != ORC_NAN_TS ? edge_ts[e] : node_ts[v];
                int ok = (t == ORC_NAN_TS || it == ORC_NAN_TS) || (wlo <= t && t < whi);
                if (ok) {
                    vpush(&cn, v);
                    vpush(&ct, t);
                    vpush(&cr, e - ptrs[cur]);
                }
            }
            int64_t pos, scratch[2];
            const uint64_t step_id = (uint64_t)i * (uint64_t)L + (uint64_t)l; /* philox address of this step */
            int algo = reservoir_algo == ORC_RES_AUTO ? ORC_RES_CHUNKED : reservoir_algo;
            int64_t success;
            if (rng->mode != ORC_RNG_REF && algo == ORC_RES_CHUNKED) { /* what the kernel computes */
                pos = orc_reservoir_one_chunked(&c, step_id, cn.n, cr.p);
                success = cn.n > 0;
            } else {
                success = orc_reservoir(&c, step_id, 0, cn.n, 1, &pos, scratch, algo == ORC_RES_CHUNKED ? ORC_RES_LITERAL : algo);
            }
            int64_t next, next_t;
            if (success == 0) { /* :144-148 restart from an earlier position */
                uint64_t rr;
                if (rng->mode == ORC_RNG_REF)
                    rr = orc_ref_gen_range_u64(rng, (uint64_t)(l + 1));
                else
                    rr = orc_bounded(orc_ctx_draw(&c, step_id, 0, 0x52535400u).a, (uint64_t)(l + 1));
                next_t = walks_ts[i * L + (int64_t)rr];
                next = walks[i * L + (int64_t)rr];
            } else {
                next = cn.p[pos];
                next_t = ct.p[pos];
            }
            cur = next; /* :150-153 */
            walks[i * L + l + 1] = cur;
            walks_ts[i * L + l + 1] = next_t;
        }
    }
    vfree(&cn);
    vfree(&ct);
    vfree(&cr);
}

/* ------------------------------------------------------------------ */
/* random_walk.rs:160-288 biased_tempo_random_walk (CTDNE-style walk)  */
/* ------------------------------------------------------------------ */
#include "orc_exp_table.h"

#define ORC_BIAS_UNIFORM 0
#define ORC_BIAS_LINEAR 1
#define ORC_BIAS_EXPONENTIAL 2

typedef struct {
    int32_t time;
    int32_t pos;
} orc_timepos;
/* descending time, ties by ascending position: `times.argsort(0, true)` (random_walk.rs:171) made deterministic.
 * tch 0.7.2's argsort is the non-stable at::argsort, whose tie order the reference leaves to libtorch. */
static int orc_timepos_desc(const void *a, const void *b) {
    const orc_timepos *x = (const orc_timepos *)a, *y = (const orc_timepos *)b;
    if (x->time != y->time) return x->time > y->time ? -1 : 1;
    return x->pos < y->pos ? -1 : (x->pos > y->pos);
}

/* BiasType::apply (random_walk.rs:165-182) over the candidates' i32 times.  Arithmetic the reference delegates to
 * libtorch and that therefore has no order / rounding fixed by the reference's own sources is pinned here as:
 *   Linear       weight[c] = (f32)perm[c] / (f32)(n(n-1)/2), perm = argsort above (n == 1 gives 0/0 = NaN, as there);
 *   Exponential  softmax of the i32 (wrapping) differences: e[c] = exp(-(m[c])) from the correctly rounded table,
 *                m[c] = max_delta - delta[c] evaluated in f32 as softmax does, weight 0 from m >= 104;
 *                denominator = sum over m = 0..103 of count[m] * e(m) accumulated in f64 in that order, rounded to
 *                f32 (libtorch's vectorised summation order is not part of the reference). */
static void orc_bias_weights(int bias, const int32_t *times, int64_t n, int64_t t, int forward, float *w) {
    if (bias == ORC_BIAS_UNIFORM) {
        for (int64_t c = 0; c < n; c++) w[c] = 1.0f;
    } else if (bias == ORC_BIAS_LINEAR) {
        orc_timepos *tp = (orc_timepos *)malloc(sizeof(orc_timepos) * (size_t)(n > 0 ? n : 1));
        for (int64_t c = 0; c < n; c++) {
            tp[c].time = times[c];
            tp[c].pos = (int32_t)c;
        }
        qsort(tp, (size_t)n, sizeof(orc_timepos), orc_timepos_desc);
        const float den = (float)(n * (n - 1) / 2);
        for (int64_t c = 0; c < n; c++) w[c] = (float)tp[c].pos / den;
        free(tp);
    } else {
        const uint32_t t32 = (uint32_t)(uint64_t)t;
        float mx = 0.0f;
        for (int64_t c = 0; c < n; c++) {
            const int32_t d = (int32_t)(forward ? t32 - (uint32_t)times[c] : (uint32_t)times[c] - t32);
            if (c == 0 || (float)d > mx) mx = (float)d;
        }
        int64_t hist[ORC_EXP_NEG_BITS_N];
        memset(hist, 0, sizeof(hist));
        for (int64_t c = 0; c < n; c++) {
            const int32_t d = (int32_t)(forward ? t32 - (uint32_t)times[c] : (uint32_t)times[c] - t32);
            const float m = mx - (float)d;
            float e = 0.0f;
            if (m < (float)ORC_EXP_NEG_BITS_N) {
                memcpy(&e, &orc_exp_neg_bits[(int)m], 4);
                hist[(int)m]++;
            }
            w[c] = e;
        }
        double acc = 0.0;
        for (int m = 0; m < ORC_EXP_NEG_BITS_N; m++) {
            float e;
            memcpy(&e, &orc_exp_neg_bits[m], 4);
            acc = acc + (double)hist[m] * (double)e;
        }
        const float den = (float)acc;
        for (int64_t c = 0; c < n; c++) w[c] = w[c] / den;
    }
}

/* walks, walks_ts: [n, walk_length].  Returns 0, or -1 where the reference panics (gen_range over an empty float range,
 * sampling.rs:49: every weight so far underflowed to zero).  Philox address of candidate c of step l of attempt a of
 * walker i: id = (i * retry_count + a) * walk_length + l, d0 = c, d1 = "WGT". */
ORC_API int32_t orc_biased_tempo_random_walk(const int64_t *ptrs, const int64_t *indices, const int64_t *node_ts,
                                             const int64_t *edge_ts, const int64_t *start, const int64_t *start_ts,
                                             int64_t n, int64_t walk_length, int32_t bias, int32_t forward,
                                             int64_t retry_count, orc_rng *rng, int64_t *walks, int64_t *walks_ts) {
    orc_ctx c;
    orc_ctx_init(&c, rng, ORC_TAG_RW_BIASED);
    const int64_t L = walk_length;
    for (int64_t i = 0; i < n * L; i++) { /* :199-208 */
        walks[i] = -1;
        walks_ts[i] = -1;
    }
    vec64 cn = {0}, ct = {0}, cr = {0}; /* candidates: neighbour, time, raw position in the row */
    int32_t *times = NULL;
    float *w = NULL, *raw_w = NULL, *blocked = NULL;
    int64_t raw_cap = 0;
    int64_t cap = 0;
    int32_t status = 0;
    for (int64_t i = 0; i < n && status == 0; i++) {
        for (int64_t attempt = 0; attempt < retry_count && status == 0; attempt++) { /* :217 */
            int64_t cur = start[i], cur_ts = start_ts[i];
            walks[i * L] = cur;
            walks_ts[i * L] = cur_ts;
            for (int64_t l = 0; l < L - 1; l++) walks[i * L + l + 1] = -1; /* :223-225: timestamps are NOT reset */
            int restart = 0;
            for (int64_t l = 0; l < L - 1; l++) {
                cn.n = ct.n = cr.n = 0;
                for (int64_t e = ptrs[cur]; e < ptrs[cur + 1]; e++) { /* :228-251 */
                    const int64_t v = indices[e];
                    const int64_t t = edge_ts[e] != ORC_NAN_TS ? edge_ts[e] : node_ts[v];
                    if (t == ORC_NAN_TS || cur_ts == ORC_NAN_TS || cur_ts <= t) {
                        vpush(&cn, v);
                        vpush(&ct, t);
                        vpush(&cr, e - ptrs[cur]);
                    }
                }
                const int64_t m = cn.n;
                if (m > cap) {
                    cap = m * 2;
                    times = (int32_t *)realloc(times, sizeof(int32_t) * (size_t)cap);
                    w = (float *)realloc(w, sizeof(float) * (size_t)cap);
                }
                for (int64_t k = 0; k < m; k++) /* :253-256 */
                    times[k] = (int32_t)(uint32_t)(uint64_t)(ct.p[k] == ORC_NAN_TS ? cur_ts : ct.p[k]);
                orc_bias_weights(cur_ts == ORC_NAN_TS ? ORC_BIAS_UNIFORM : bias, times, m, cur_ts, forward, w); /* :258-262 */
                if (m == 0) { /* :273-276 */
                    restart = 1;
                    break;
                }
                /* reservoir_sampling_weighted with one slot and f32 weights (sampling.rs:28-55) */
                const uint64_t step_id = ((uint64_t)i * (uint64_t)retry_count + (uint64_t)attempt) * (uint64_t)L + (uint64_t)l;
                /* philox-mode, linear / exponential bias: the running sum is BLOCKED over the row's raw positions (chunks of 64,
                 * non-candidates add 0; orc_sampling.h) -- what the kernel computes; ref-mode and the uniform bias (whose
                 * sum of ones the kernel has in closed form) keep the reference's literal left-to-right sum */
                const int64_t n_raw = ptrs[cur + 1] - ptrs[cur];
                const int use_blocked = rng->mode != ORC_RNG_REF && cur_ts != ORC_NAN_TS && bias != ORC_BIAS_UNIFORM;
                if (use_blocked) {
                    if (n_raw > raw_cap) {
                        raw_cap = n_raw * 2;
                        raw_w = (float *)realloc(raw_w, sizeof(float) * (size_t)raw_cap);
                        blocked = (float *)realloc(blocked, sizeof(float) * (size_t)raw_cap);
                    }
                    for (int64_t r = 0; r < n_raw; r++) raw_w[r] = 0.0f;
                    for (int64_t k = 0; k < m; k++) raw_w[cr.p[k]] = w[k];
                    orc_blocked_prefix_f32(raw_w, n_raw, blocked);
                }
                int64_t pick = 0;
                float w_sum = 0.0f + w[0];
                for (int64_t k = 1; k < m; k++) {
                    w_sum = use_blocked ? blocked[cr.p[k]] : w_sum + w[k];
                    if (!(0.0f < w_sum)) {
                        status = -1;
                        break;
                    }
                    float j;
                    if (rng->mode == ORC_RNG_REF) {
                        j = orc_ref_gen_range_f32(rng, w_sum);
                    } else {
                        j = orc_u32_to_f32_01(orc_ctx_draw(&c, step_id, (uint32_t)k, 0x57475400u).w[0]) * w_sum + 0.0f;
                    }
                    if (j < w[k]) {
                        if (rng->mode == ORC_RNG_REF) (void)orc_ref_gen_range_u64(rng, 1); /* dst[gen_range(0..1)] */
                        pick = k;
                    }
                }
                if (status != 0) break;
                const int64_t next = cn.p[pick], next_ts = ct.p[pick]; /* :278-284 */
                cur = next;
                if (next_ts != -1) cur_ts = next_ts;
                walks[i * L + l + 1] = cur;
                walks_ts[i * L + l + 1] = next_ts;
            }
            if (!restart) break; /* :286 */
        }
    }
    vfree(&cn);
    vfree(&ct);
    vfree(&cr);
    free(times);
    free(w);
    free(raw_w);
    free(blocked);
    return status;
}

/* ------------------------------------------------------------------ */
/* Negative sampling: src/algo/negative_sampling.rs                    */
/* ------------------------------------------------------------------ */
/* tiny open-addressing map i64 -> i64 (first-seen order is kept by the
 * caller's vector, as the reference's HashMap + Vec pair does) */
typedef struct {
    int64_t *k, *v;
    int64_t cap, n;
} imap;
static void imap_init(imap *m, int64_t cap) {
    int64_t c = 64;
    while (c < cap * 2) c <<= 1;
    m->cap = c;
    m->n = 0;
    m->k = (int64_t *)malloc(sizeof(int64_t) * (size_t)c);
    m->v = (int64_t *)malloc(sizeof(int64_t) * (size_t)c);
    for (int64_t i = 0; i < c; i++) m->k[i] = INT64_MIN;
}
static void imap_free(imap *m) {
    free(m->k);
    free(m->v);
}
static int64_t *imap_slot(imap *m, int64_t key, int *found);
static void imap_grow(imap *m) {
    imap o = *m;
    imap_init(m, o.cap);
    for (int64_t i = 0; i < o.cap; i++)
        if (o.k[i] != INT64_MIN) {
            int f;
            int64_t *s = imap_slot(m, o.k[i], &f);
            *s = o.v[i];
        }
    imap_free(&o);
}
static int64_t *imap_slot(imap *m, int64_t key, int *found) {
    if ((m->n + 1) * 2 > m->cap) imap_grow(m);
    uint64_t h = (uint64_t)key * 0x9E3779B97F4A7C15ULL;
    int64_t i = (int64_t)(h >> 20) & (m->cap - 1);
    while (m->k[i] != INT64_MIN && m->k[i] != key) i = (i + 1) & (m->cap - 1);
    *found = m->k[i] == key;
    if (!*found) {
        m->k[i] = key;
        m->n++;
    }
    return &m->v[i];
}

/* negative_sampling.rs:6-48. samples cap = B + B*num_neg; rows/cols cap = B*num_neg.
 * counts = {n_samples, n_edges}; returns sample_count (= B). */
ORC_API int64_t orc_neg_homo(const int64_t *ptrs, const int64_t *indices, int64_t node_count, const int64_t *inputs,
                             int64_t B, int64_t num_neg, int64_t try_count, orc_rng *rng, int64_t *samples,
                             int64_t *rows, int64_t *cols, int64_t *counts) {
    orc_ctx c;
    orc_ctx_init(&c, rng, ORC_TAG_NEG_HOMO);
    imap m;
    imap_init(&m, B * (1 + num_neg));
    int64_t ns = 0, ne = 0;
    for (int64_t i = 0; i < B; i++) { /* :25-26: later duplicates overwrite earlier ones */
        samples[ns++] = inputs[i];
        int f;
        *imap_slot(&m, inputs[i], &f) = i;
    }
    for (int64_t i = 0; i < B; i++) { /* :31 */
        int64_t v = inputs[i];
        for (int64_t jn = 0; jn < num_neg; jn++) {
            for (int64_t t = 0; t < try_count; t++) { /* :33 */
                int64_t w;
                if (rng->mode == ORC_RNG_REF)
                    w = (int64_t)orc_ref_gen_range_u64(rng, (uint64_t)node_count);
                else
                    w = (int64_t)orc_bounded(orc_ctx_draw(&c, (uint64_t)i, (uint32_t)jn, (uint32_t)t).a,
                                             (uint64_t)node_count);
                if (!orc_has_edge(ptrs, indices, v, w) && v != w) { /* :35 */
                    int f;
                    int64_t *slot = imap_slot(&m, w, &f);
                    if (!f) {
                        samples[ns] = w;
                        *slot = ns;
                        ns++;
                    }
                    rows[ne] = i; /* :40 */
                    cols[ne] = *slot;
                    ne++;
                    break;
                }
            }
        }
    }
    imap_free(&m);
    counts[0] = ns;
    counts[1] = ne;
    return B;
}

/* negative_sampling.rs:50-131 (canonical order: node types t = 0..T-1 that
 * have inputs; relations of a type in edge_types order, :65-71).
 * sizes[r*2+1] = node_count of relation r (:105).  has_input[t] mirrors
 * `inputs.contains_key` (:88). Returns handle (layer_offsets unused);
 * sample_count[t] = number of inputs of type t (:96). */
ORC_API orc_het_out *orc_neg_hetero(int32_t T, int32_t R, const int32_t *rel_src, const int32_t *rel_dst,
                                    const int64_t *const *ptrs, const int64_t *const *indices, const int64_t *sizes,
                                    const int64_t *const *inputs, const int64_t *n_inputs, const int32_t *has_input,
                                    int64_t num_neg, int64_t try_count, int32_t inbound, orc_rng *rng,
                                    int64_t *sample_count, int32_t *status) {
    orc_het_out *o = het_alloc(T, R, 0);
    *status = 0;
    imap *maps = (imap *)calloc((size_t)T, sizeof(imap));
    for (int t = 0; t < T; t++) { /* :77-91 */
        imap_init(&maps[t], 64);
        if (has_input[t])
            for (int64_t i = 0; i < n_inputs[t]; i++) {
                vpush(&o->samples[t], inputs[t][i]);
                int f;
                *imap_slot(&maps[t], inputs[t][i], &f) = i;
            }
        sample_count[t] = o->samples[t].n; /* :96 */
    }
    for (int t = 0; t < T; t++) { /* :99 */
        if (!has_input[t]) continue;
        int32_t nrel = 0;
        int32_t *rels = (int32_t *)malloc(sizeof(int32_t) * (size_t)(R ? R : 1));
        for (int r = 0; r < R; r++)
            if (rel_src[r] == t) rels[nrel++] = r; /* :65-71 */
        orc_ctx c;
        orc_ctx_init(&c, rng, ORC_TAG_NEG_HETERO | ((uint32_t)t << 8));
        for (int64_t i = 0; i < n_inputs[t]; i++) { /* :102 */
            int64_t v = inputs[t][i];
            for (int64_t jn = 0; jn < num_neg; jn++) {
                /* :104 panics on nrel == 0 (empty range) -- callers must not do that */
                int32_t r;
                if (rng->mode == ORC_RNG_REF)
                    r = rels[orc_ref_gen_range_u64(rng, (uint64_t)nrel)];
                else
                    r = rels[orc_bounded(orc_ctx_draw(&c, (uint64_t)i, (uint32_t)jn, 0xFFFFFFFFu).a, (uint64_t)nrel)];
                int32_t dt = rel_dst[r];
                int64_t node_count = sizes[r * 2 + 1];
                for (int64_t tr = 0; tr < try_count; tr++) { /* :110 */
                    int64_t w;
                    if (rng->mode == ORC_RNG_REF)
                        w = (int64_t)orc_ref_gen_range_u64(rng, (uint64_t)node_count);
                    else
                        w = (int64_t)orc_bounded(orc_ctx_draw(&c, (uint64_t)i, (uint32_t)jn, (uint32_t)tr).a,
                                                 (uint64_t)node_count);
                    if (inbound && w >= sizes[r * 2 + 0]) { /* :113 indexes ptrs[w]: the reference panics */
                        *status = -1;
                        break;
                    }
                    int he = inbound ? orc_has_edge(ptrs[r], indices[r], w, v)
                                     : orc_has_edge(ptrs[r], indices[r], v, w); /* :112-115 */
                    if (!he && v != w) { /* :117 */
                        int f;
                        int64_t *slot = imap_slot(&maps[dt], w, &f);
                        if (!f) {
                            *slot = o->samples[dt].n;
                            vpush(&o->samples[dt], w);
                        }
                        vpush(&o->rows[r], i);
                        vpush(&o->cols[r], *slot);
                        vpush(&o->eidx[r], -1);
                        break;
                    }
                }
            }
        }
        free(rels);
    }
    for (int t = 0; t < T; t++) imap_free(&maps[t]);
    free(maps);
    return o;
}

/* ------------------------------------------------------------------ */
/* CPU-baseline driver (bench.py cpu_baseline leg)                     */
/* ------------------------------------------------------------------ */
#include <pthread.h>
#include <time.h>

typedef struct {
    const int64_t *ptrs, *indices, *seeds;
    int64_t B;
    const int64_t *fanout;
    int32_t H;
    int64_t batch_begin, batch_end;
    int64_t cap_nodes, cap_edges;
    int64_t total_edges;
} orc_bench_job;

static void *orc_bench_worker(void *arg) {
    orc_bench_job *j = (orc_bench_job *)arg;
    int64_t *samples = (int64_t *)malloc(sizeof(int64_t) * (size_t)j->cap_nodes);
    int64_t *rows = (int64_t *)malloc(sizeof(int64_t) * (size_t)j->cap_edges);
    int64_t *cols = (int64_t *)malloc(sizeof(int64_t) * (size_t)j->cap_edges);
    int64_t *eidx = (int64_t *)malloc(sizeof(int64_t) * (size_t)j->cap_edges);
    int64_t lo[64], counts[2];
    orc_ns_cfg cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.sampler = ORC_SAMPLER_UNIFORM;
    cfg.filter_mode = ORC_FILTER_NONE;
    j->total_edges = 0;
    for (int64_t b = j->batch_begin; b < j->batch_end; b++) {
        orc_rng rng;
        memset(&rng, 0, sizeof(rng));
        rng.mode = ORC_RNG_REF;
        orc_xoshiro_seed_from_u64(&rng, (uint64_t)b); /* one independent reference stream per batch */
        orc_ns_homo(j->ptrs, j->indices, j->seeds + b * j->B, j->B, j->fanout, j->H, &cfg, NULL, &rng, samples,
                    j->cap_nodes, rows, cols, eidx, j->cap_edges, lo, counts);
        j->total_edges += counts[1];
    }
    free(samples);
    free(rows);
    free(cols);
    free(eidx);
    return NULL;
}

/* Times the ref-mode restatement of neighbor_sampling_homogenous (default
 * sampler, no filter) over n_batches seed batches on n_threads threads, each
 * thread owning whole batches.  Returns seconds; *edges = sampled edges. */
ORC_API double orc_bench_ns_homo(const int64_t *ptrs, const int64_t *indices, const int64_t *seeds, int64_t B,
                                 int64_t n_batches, const int64_t *fanout, int32_t H, int32_t n_threads,
                                 int64_t *edges) {
    if (n_threads < 1) n_threads = 1;
    if (H > 20) return -1.0;
    int64_t cap = B, layer = B;
    for (int h = 0; h < H; h++) {
        layer *= fanout[h];
        cap += layer;
    }
    orc_bench_job *jobs = (orc_bench_job *)calloc((size_t)n_threads, sizeof(orc_bench_job));
    pthread_t *th = (pthread_t *)calloc((size_t)n_threads, sizeof(pthread_t));
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int t = 0; t < n_threads; t++) {
        jobs[t].ptrs = ptrs;
        jobs[t].indices = indices;
        jobs[t].seeds = seeds;
        jobs[t].B = B;
        jobs[t].fanout = fanout;
        jobs[t].H = H;
        jobs[t].batch_begin = n_batches * t / n_threads;
        jobs[t].batch_end = n_batches * (t + 1) / n_threads;
        jobs[t].cap_nodes = cap;
        jobs[t].cap_edges = cap;
        pthread_create(&th[t], NULL, orc_bench_worker, &jobs[t]);
    }
    int64_t tot = 0;
    for (int t = 0; t < n_threads; t++) {
        pthread_join(th[t], NULL);
        tot += jobs[t].total_edges;
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    *edges = tot;
    free(jobs);
    free(th);
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

/* ------------------------------------------------------------------ */
/* Synthetic inputs: CPU twins of the device generators */
/* ------------------------------------------------------------------ */
ORC_API void orc_rmat_edges(int32_t scale, int64_t n_edges, uint64_t seed, int64_t *row, int64_t *col) {
    const uint32_t T_A = 2448131358u, T_AB = 3264175144u, T_ABC = 4080218931u; /* 0.57, 0.76, 0.95 of 2^32 */
    orc_callkey ck = orc_philox_callkey(seed, 0, ORC_TAG_RMAT);
    for (int64_t e = 0; e < n_edges; e++) {
        int64_t r = 0, c = 0;
        orc_draw d = {0};
        for (int bit = 0; bit < scale; bit++) {
            if ((bit & 3) == 0) d = orc_philox_draw(ck, (uint64_t)e, (uint32_t)(bit >> 2), 0);
            uint32_t u = d.w[bit & 3];
            int rb = u >= T_AB;
            int cb = (u >= T_A && u < T_AB) || u >= T_ABC;
            r = (r << 1) | rb;
            c = (c << 1) | cb;
        }
        row[e] = r;
        col[e] = c;
    }
}
ORC_API void orc_seed_batches(uint64_t seed, int64_t first_batch, int64_t n_batches, int64_t n_seeds, int64_t n_nodes,
                              int64_t *out) {
    orc_callkey ck = orc_philox_callkey(seed, 0, ORC_TAG_SEEDS);
    for (int64_t b = 0; b < n_batches; b++)
        for (int64_t i = 0; i < n_seeds; i++) {
            orc_draw d = orc_philox_draw(ck, (uint64_t)(first_batch + b), (uint32_t)i, (uint32_t)((uint64_t)i >> 32));
            out[b * n_seeds + i] = (int64_t)orc_bounded(d.a, (uint64_t)n_nodes);
        }
}


/* ------------------------------------------------------------------ */
/* HGT sampling: src/algo/hgt_sampling.rs                              */
/* ------------------------------------------------------------------ */
#define ORC_HGT_MAX_NEIGHBORS 50 /* hgt_sampling.rs:10 */

/* NodeBudget (hgt_sampling.rs:13-24) with a canonical iteration order: entries are kept in the order
 * their key was first inserted (the reference iterates a std HashMap, whose order is not reproducible);
 * removal leaves a tombstone.  A removed key is never re-inserted: it is in to_local by then (:80). */
typedef struct {
    vec64 key, ts, alive;
    double *score;
    int64_t score_cap;
    imap slot; /* key -> entry index */
    int present; /* budget_dict.entry(type) has been created */
} orc_budget;

static void budget_add(orc_budget *b, int64_t v, double inv_deg, int64_t ts) { /* :95-97 */
    int f;
    int64_t *s = imap_slot(&b->slot, v, &f);
    if (!f) {
        *s = b->key.n;
        if (b->key.n == b->score_cap) {
            b->score_cap = b->score_cap ? b->score_cap * 2 : 64;
            b->score = (double *)realloc(b->score, sizeof(double) * (size_t)b->score_cap);
        }
        b->score[b->key.n] = 0.0;
        vpush(&b->key, v);
        vpush(&b->ts, 0);
        vpush(&b->alive, 1);
    }
    b->score[*s] += inv_deg;
    b->ts.p[*s] = ts;
}

typedef struct {
    int32_t T, R;
    const int32_t *rel_src, *rel_dst;
    const int64_t *const *ptrs, *const *indices, *const *rel_ts;
    int has_timerange;
    int64_t tr_lo, tr_hi; /* half open, python.rs:450 */
    imap *to_local;       /* [T] */
    orc_budget *budget;   /* [T] */
} orc_hgt_ctx;

/* hgt_sampling.rs:27-102 update_budget for the samples of node type `nt` */
static void hgt_update_budget(orc_hgt_ctx *h, int nt, const int64_t *samples, const int64_t *samples_ts, int64_t n) {
    if (n == 0) return; /* :38-40 */
    for (int r = 0; r < h->R; r++) { /* :47 canonical relation order */
        if (h->rel_dst[r] != nt) continue; /* :50-52 */
        int src = h->rel_src[r];
        imap *to_local_src = &h->to_local[src];
        orc_budget *sb = &h->budget[src];
        sb->present = 1; /* :55 entry(src).or_default() */
        const int64_t *ptrs = h->ptrs[r], *indices = h->indices[r];
        const int64_t *ets = h->rel_ts ? h->rel_ts[r] : NULL;
        for (int64_t j = 0; j < n; j++) { /* :58 */
            int64_t w = samples[j];
            int64_t b = ptrs[w], e = ptrs[w + 1];
            if (e <= b) continue; /* :60-62 */
            int64_t w_ts = samples_ts[j];
            /* :72 reservoir over 0..min(len,50) into 50 slots never draws: the first min(len,50) neighbours */
            int64_t cnt = e - b < ORC_HGT_MAX_NEIGHBORS ? e - b : ORC_HGT_MAX_NEIGHBORS;
            double inv_deg = 1.0 / (double)cnt; /* :73 */
            for (int64_t i = 0; i < cnt; i++) { /* :76 */
                int64_t v = indices[b + i];
                int f;
                /* :80 contains_key -- probe without inserting */
                {
                    uint64_t hh = (uint64_t)v * 0x9E3779B97F4A7C15ULL;
                    int64_t s = (int64_t)(hh >> 20) & (to_local_src->cap - 1);
                    while (to_local_src->k[s] != INT64_MIN && to_local_src->k[s] != v) s = (s + 1) & (to_local_src->cap - 1);
                    f = to_local_src->k[s] == v;
                }
                if (f) continue;
                int64_t v_ts = ets ? ets[b + i] : ORC_NAN_TS; /* :82 */
                if (v_ts == ORC_NAN_TS) v_ts = w_ts;           /* :83-85 */
                if (h->has_timerange && v_ts != ORC_NAN_TS && !(h->tr_lo <= v_ts && v_ts < h->tr_hi)) continue; /* :88-92 */
                budget_add(sb, v, inv_deg, v_ts);
            }
        }
    }
}

/* hgt_sampling.rs:138-278.  inputs[t] NULL / n_inputs[t] < 0: type absent from `inputs`.
 * input_ts NULL: no input timestamps at all (:172-178).  num_samples[t*H + layer].
 * philox addresses: sample_from of (layer, type) uses id = layer*T + type under ORC_TAG_HGT; the edge
 * reservoir of relation r uses id = dst slot under ORC_TAG_HGT | (r+1) << 8. */
ORC_API orc_het_out *orc_hgt(int32_t T, int32_t R, const int32_t *rel_src, const int32_t *rel_dst,
                             const int64_t *const *ptrs, const int64_t *const *indices, const int64_t *const *rel_ts,
                             const int64_t *const *inputs, const int64_t *n_inputs, const int64_t *const *input_ts,
                             const int64_t *num_samples, int32_t H, int32_t has_timerange, int64_t tr_lo,
                             int64_t tr_hi, int32_t edge_reservoir_algo, orc_rng *rng, int32_t *status) {
    orc_het_out *o = het_alloc(T, R, 0);
    *status = 0;
    orc_hgt_ctx h;
    memset(&h, 0, sizeof(h));
    h.T = T;
    h.R = R;
    h.rel_src = rel_src;
    h.rel_dst = rel_dst;
    h.ptrs = ptrs;
    h.indices = indices;
    h.rel_ts = rel_ts;
    h.has_timerange = has_timerange;
    h.tr_lo = tr_lo;
    h.tr_hi = tr_hi;
    h.to_local = (imap *)calloc((size_t)T, sizeof(imap));
    h.budget = (orc_budget *)calloc((size_t)T, sizeof(orc_budget));
    int *has_nodes = (int *)calloc((size_t)T, sizeof(int)); /* nodes_dict has an entry for the type */
    for (int t = 0; t < T; t++) {
        imap_init(&h.to_local[t], 64);
        imap_init(&h.budget[t].slot, 64);
    }
    /* :167-180 inputs become the first sampled nodes */
    for (int t = 0; t < T; t++) {
        if (n_inputs[t] < 0) continue;
        has_nodes[t] = 1;
        for (int64_t i = 0; i < n_inputs[t]; i++) {
            int f;
            *imap_slot(&h.to_local[t], inputs[t][i], &f) = o->samples[t].n; /* insert overwrites */
            vpush(&o->samples[t], inputs[t][i]);
            vpush(&o->ts[t], input_ts ? input_ts[t][i] : ORC_NAN_TS);
        }
    }
    /* :183-196 */
    for (int t = 0; t < T; t++)
        if (has_nodes[t]) hgt_update_budget(&h, t, o->samples[t].p, o->ts[t].p, o->samples[t].n);

    orc_ctx c;
    orc_ctx_init(&c, rng, ORC_TAG_HGT);
    vec64 *lay_s = (vec64 *)calloc((size_t)T, sizeof(vec64)), *lay_t = (vec64 *)calloc((size_t)T, sizeof(vec64));
    int *lay_has = (int *)calloc((size_t)T, sizeof(int));
    for (int32_t layer = 0; layer < H && *status == 0; layer++) { /* :198 */
        for (int t = 0; t < T; t++) {
            lay_s[t].n = lay_t[t].n = 0;
            lay_has[t] = 0;
        }
        for (int t = 0; t < T && *status == 0; t++) { /* :201 canonical type order */
            orc_budget *b = &h.budget[t];
            if (!b->present) continue;
            int64_t k = num_samples[t * H + layer];
            if (k < 0) { /* :202 missing key panics */
                *status = -1;
                break;
            }
            /* :104-135 sample_from: weighted reservoir over the live entries, weights score^2 */
            int64_t n = 0;
            for (int64_t i = 0; i < b->key.n; i++) n += b->alive.p[i];
            int64_t *live = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n ? n : 1));
            double *w = (double *)malloc(sizeof(double) * (size_t)(n ? n : 1));
            int64_t m = 0;
            for (int64_t i = 0; i < b->key.n; i++)
                if (b->alive.p[i]) {
                    live[m] = i;
                    w[m] = b->score[i] * b->score[i]; /* :110 */
                    m++;
                }
            int64_t *dst = (int64_t *)malloc(sizeof(int64_t) * (size_t)(k ? k : 1));
            int64_t cnt = 0;
            if (k > 0) cnt = orc_reservoir_weighted(&c, (uint64_t)(layer * T + t), n, k, w, dst, NULL, n);
            if (cnt < 0) {
                *status = -1;
                cnt = 0;
            }
            lay_has[t] = 1; /* :206-208 */
            has_nodes[t] = 1;
            for (int64_t s = 0; s < cnt; s++) { /* :216-221 */
                int64_t e = live[dst[s]];
                int64_t v = b->key.p[e], vt = b->ts.p[e];
                vpush(&lay_s[t], v);
                vpush(&lay_t[t], vt);
                int f;
                *imap_slot(&h.to_local[t], v, &f) = o->samples[t].n;
                vpush(&o->samples[t], v);
                vpush(&o->ts[t], vt);
            }
            for (int64_t s = 0; s < cnt; s++) b->alive.p[live[dst[s]]] = 0; /* :220 remove */
            free(live);
            free(w);
            free(dst);
        }
        if (layer < H - 1) /* :224-240 */
            for (int t = 0; t < T; t++)
                if (lay_has[t]) hgt_update_budget(&h, t, lay_s[t].p, lay_t[t].p, lay_s[t].n);
    }
    /* :244-268 rebuild the edges among the sampled nodes */
    int64_t res[ORC_HGT_MAX_NEIGHBORS], scratch[2 * ORC_HGT_MAX_NEIGHBORS];
    for (int r = 0; r < R && *status == 0; r++) {
        int src = rel_src[r], dst = rel_dst[r];
        orc_ctx ce;
        orc_ctx_init(&ce, rng, ORC_TAG_HGT | ((uint32_t)(r + 1) << 8));
        imap *tl = &h.to_local[src];
        for (int64_t i = 0; i < o->samples[dst].n; i++) { /* :254 */
            int64_t w = o->samples[dst].p[i];
            int64_t b = ptrs[r][w], e = ptrs[r][w + 1];
            int64_t len = e - b;
            int64_t k = len < ORC_HGT_MAX_NEIGHBORS ? len : ORC_HGT_MAX_NEIGHBORS; /* :258 */
            int64_t cnt = orc_reservoir(&ce, (uint64_t)i, 0, len, k, res, scratch,
                                        edge_reservoir_algo == ORC_RES_AUTO ? ORC_RES_TICKETS : edge_reservoir_algo);
            for (int64_t s = 0; s < cnt; s++) { /* :261-266 */
                int64_t ep = b + res[s];
                int64_t v = indices[r][ep];
                uint64_t hh = (uint64_t)v * 0x9E3779B97F4A7C15ULL;
                int64_t sl = (int64_t)(hh >> 20) & (tl->cap - 1);
                while (tl->k[sl] != INT64_MIN && tl->k[sl] != v) sl = (sl + 1) & (tl->cap - 1);
                if (tl->k[sl] == v) {
                    vpush(&o->rows[r], tl->v[sl]);
                    vpush(&o->cols[r], i);
                    vpush(&o->eidx[r], ep);
                }
            }
        }
    }
    for (int t = 0; t < T; t++) {
        imap_free(&h.to_local[t]);
        imap_free(&h.budget[t].slot);
        vfree(&h.budget[t].key);
        vfree(&h.budget[t].ts);
        vfree(&h.budget[t].alive);
        free(h.budget[t].score);
        vfree(&lay_s[t]);
        vfree(&lay_t[t]);
    }
    free(h.to_local);
    free(h.budget);
    free(has_nodes);
    free(lay_s);
    free(lay_t);
    free(lay_has);
    return o;
}


/* ------------------------------------------------------------------ */
/* Budget sampling: src/algo/budget_sampling.rs (SURVEY 8(f) "next")   */
/* ------------------------------------------------------------------ */
/* budget_sampling.rs:19-38 TemporalFilter (half-open window, python.rs:541-548) */
typedef struct {
    int32_t enabled, forward, relative, _pad;
    int64_t lo, hi;
} orc_budget_filter;
static inline int budget_filter_pass(const orc_budget_filter *f, int64_t state, int64_t t) { /* :20-29 */
    if (state == ORC_NAN_TS || t == ORC_NAN_TS) return 1;
    int64_t x = f->forward ? (t - state) : -(t - state);
    return f->lo <= x && x < f->hi;
}

/* budget_sampling.rs:155-265 budget_neighbor_sampling_heterogenous.  Canonical order: the budget of a node
 * lists relations in `edge_types` order (:81 iterates a HashMap).  num_neighbors[t*H + layer] (< 0: missing key
 * -> the reference panics at :226).  NOTE the reference stores the neighbour's index INSIDE the column as the
 * edge's `edge_index` (:116 `*i as EdgePtr`), not the CSC edge pointer; kept.  layer_offsets stay empty (:199-201).
 * philox address of a node's reservoir (:138): tag ORC_TAG_BUDGET | type << 8, id = slot of the node. */
ORC_API orc_het_out *orc_budget_sampling(int32_t T, int32_t R, const int32_t *rel_src, const int32_t *rel_dst,
                                const int64_t *const *ptrs, const int64_t *const *indices, const int64_t *const *rel_ts,
                                const int64_t *const *inputs, const int64_t *n_inputs, const int64_t *const *input_ts,
                                const int64_t *num_neighbors, int32_t H, const orc_budget_filter *filter, orc_rng *rng,
                                int32_t *status) {
    orc_het_out *o = het_alloc(T, R, 0);
    *status = 0;
    for (int t = 0; t < T; t++) /* :181-197 */
        for (int64_t i = 0; i < (n_inputs[t] > 0 ? n_inputs[t] : 0); i++) {
            vpush(&o->samples[t], inputs[t][i]);
            vpush(&o->ts[t], (input_ts && input_ts[t]) ? input_ts[t][i] : ORC_NAN_TS);
        }
    int64_t *begin = (int64_t *)calloc((size_t)T, sizeof(int64_t)), *end = (int64_t *)calloc((size_t)T, sizeof(int64_t));
    for (int t = 0; t < T; t++) end[t] = o->samples[t].n; /* :207-209 */
    /* a candidate: (relation, index in column, node, timestamp) */
    typedef struct {
        int32_t rel;
        int64_t i, v, ts;
    } cand_t;
    cand_t *cand = (cand_t *)malloc(sizeof(cand_t) * (size_t)(50 * (R > 0 ? R : 1)));
    for (int32_t layer = 0; layer < H && *status == 0; layer++) { /* :223 */
        for (int t = 0; t < T && *status == 0; t++) { /* :225 node_types order (a Vec in the reference) */
            int64_t k = num_neighbors[t * H + layer];
            if (k < 0) {
                *status = -1;
                break;
            }
            orc_ctx c;
            orc_ctx_init(&c, rng, ORC_TAG_BUDGET | ((uint32_t)t << 8));
            int64_t *dst = (int64_t *)malloc(sizeof(int64_t) * (size_t)(k > 0 ? k : 1));
            int64_t *scratch = (int64_t *)malloc(sizeof(int64_t) * (size_t)(k > 0 ? 2 * k : 2));
            for (int64_t j = begin[t]; j < end[t]; j++) { /* :230 */
                /* Budget::update for node j (:81-122): relations into type t, first min(deg,50) neighbours each */
                int64_t w = o->samples[t].p[j], w_t = o->ts[t].p[j], n = 0;
                for (int r = 0; r < R; r++) {
                    if (rel_dst[r] != t) continue;
                    int64_t b = ptrs[r][w], len = ptrs[r][w + 1] - b;
                    int64_t cnt = len < 50 ? len : 50; /* :100 no draws: a column prefix */
                    for (int64_t i = 0; i < cnt; i++) {
                        int64_t v_t = (rel_ts && rel_ts[r]) ? rel_ts[r][b + i] : ORC_NAN_TS; /* :103 */
                        if (v_t == ORC_NAN_TS) v_t = w_t;                                    /* :104-106 */
                        if (filter->enabled && !budget_filter_pass(filter, w_t, v_t)) continue; /* :108-112 */
                        cand[n].rel = r;
                        cand[n].i = i;
                        cand[n].v = indices[r][b + i];
                        cand[n].ts = filter->enabled ? (filter->relative ? w_t : v_t) : v_t; /* :117-119 */
                        n++;
                    }
                }
                /* Budget::sample (:137-151) */
                int64_t cnt = (k > 0) ? orc_reservoir(&c, (uint64_t)j, 0, n, k, dst, scratch, ORC_RES_TICKETS) : 0;
                for (int64_t s = 0; s < cnt; s++) {
                    const cand_t *q = &cand[dst[s]];
                    int st = rel_src[q->rel];
                    int64_t i_new = o->samples[st].n; /* :147 */
                    vpush(&o->samples[st], q->v);
                    vpush(&o->ts[st], q->ts);
                    vpush(&o->rows[q->rel], i_new); /* :150 push_edge(i, j, edge_ptr) */
                    vpush(&o->cols[q->rel], j);
                    vpush(&o->eidx[q->rel], q->i);
                }
            }
            free(dst);
            free(scratch);
        }
        for (int t = 0; t < T; t++) { /* :240-243 */
            begin[t] = end[t];
            end[t] = o->samples[t].n;
        }
    }
    free(cand);
    free(begin);
    free(end);
    return o;
}
