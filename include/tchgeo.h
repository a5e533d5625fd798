/*
 * tchgeo.h -- C ABI of the MI355X (gfx950) graph-sampling backend that replaces
 * tch-geometric's CPU samplers (reference: the files under src/algo) behind its Python
 * operator surface (reference: src/python.rs, tch_geometric/tch_geometric.pyi).
 *
 * Conventions
 *  - plain C: raw device pointers + sizes, no torch / HIP types in signatures
 *    (`stream` is a hipStream_t passed as void*; NULL = the default stream);
 *  - every entry point is stream-ordered and never synchronises, allocates or
 *    frees: the caller owns every buffer, including outputs and workspaces;
 *  - all indices are int64 (reference: src/utils/types.rs:4-9), weights are
 *    float64 (python.rs:214), timestamps int64 (python.rs:149);
 *  - return value: TG_OK or a TG_ERR_* code; tg_last_error() gives the text of
 *    the last failure on the calling thread;
 *  - randomness: counter-addressed Philox4x32-10.  A draw is named by
 *    (seed, call_id, operator tag, id, d0, d1); results do not depend on
 *    launch geometry, and equal the CPU oracle's philox-mode bit for bit.
 *    (The reference's own stream -- one sequential Xoshiro256++ threaded
 *    through every vertex with data-dependent rejection, utils/random.rs +
 *    rand 0.8.5 -- cannot be reproduced by any parallel device; DESIGN.md.)
 */
#ifndef TCHGEO_H
#define TCHGEO_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TG_API __attribute__((visibility("default")))

#define TG_OK 0
#define TG_ERR_INVALID 1     /* bad argument (null pointer, negative size, unsupported fan-out ...) */
#define TG_ERR_HIP 2         /* a HIP runtime call failed */
#define TG_ERR_UNSUPPORTED 3 /* valid request this build does not implement */

#define TG_MAX_HOPS 8
#define TG_MAX_FANOUT 32 /* per-hop fan-out handled by the register-resident sampler; larger fan-outs (<= 255 while
                            the LDS holds them) take a slower LDS-resident form with identical results */

/* Adjacency resident in HBM, borrowed for the call: replaces
 * SparseGraph{ptrs,indices} (src/data/graph.rs:34-38) and EdgeAttr
 * (graph.rs:104-120).  CSC for neighbor sampling / HGT, CSR for walks and
 * negative sampling; rows must be sorted ascending inside a column
 * (to_csc/to_csr guarantee it, src/data/storage.rs:112,119). */
typedef struct {
    const int64_t *ptrs;       /* [n_major + 1] */
    const int64_t *indices;    /* [n_edges] */
    const double *weights;     /* [n_edges] or NULL  (WeightedSampler, neighbor_sampling.rs:131-158) */
    const int64_t *timestamps; /* [n_edges] or NULL  (TemporalFilter, neighbor_sampling.rs:36-77) */
    int64_t n_major;
    int64_t n_edges;
    const uint32_t *indices32; /* optional u32 shadow of `indices` (all ids < 2^32): same values, half the bytes per
                                  gathered line; NULL = not provided.  Outputs stay int64 either way. */
    const uint32_t *ptrs32;    /* optional u32 shadow of `ptrs` (n_edges < 2^32): 4 B per offset, so the whole
                                  offset table of RMAT-24 (67 MB) stays in the Infinity Cache; NULL = not provided */
    int64_t max_degree;        /* optional: the longest column (row) of the graph, 0 = unknown (n_edges is assumed).  The
                                  window-ordered launch packs sampled positions into max_degree's bits; a
                                  value SMALLER than the truth is a contract violation like an out-of-range id.
                                  tg_graph_max_degree computes it on the device. */
} tg_graph;

typedef struct {
    uint64_t seed;
    uint64_t call_id; /* batch b of a batched call uses call_id + b */
} tg_rng;

/* Sampler variants of python.rs:210-216 */
#define TG_SAMPLER_UNIFORM 0      /* UnweightedSampler<false> -- the default (python.rs:215) */
#define TG_SAMPLER_UNIFORM_REPL 1 /* UnweightedSampler<true> */
#define TG_SAMPLER_WEIGHTED 2     /* WeightedSampler<f64> */
/* Filter variants of python.rs:218-249 */
#define TG_FILTER_NONE (-1)
#define TG_FILTER_STATIC 0   /* TEMPORAL_SAMPLE_STATIC   (neighbor_sampling.rs:32) */
#define TG_FILTER_RELATIVE 1 /* TEMPORAL_SAMPLE_RELATIVE */
#define TG_FILTER_DYNAMIC 2  /* TEMPORAL_SAMPLE_DYNAMIC */

typedef struct {
    int32_t sampler;     /* TG_SAMPLER_* */
    int32_t filter_mode; /* TG_FILTER_* */
    int32_t forward;     /* TemporalFilter FORWARD */
    uint32_t rng_tag;    /* operator tag of the draw address; 0 = the homogeneous default.  The heterogeneous
                            sampler runs one relation-hop at a time with TG_TAG_NS_HETERO | relation << 8 */
    int64_t win_lo, win_hi;       /* inclusive window (python.rs:150) */
    const int64_t *seeds_state;   /* [n_batches * n_seeds] initial filter state, or NULL */
    int64_t id_base;              /* draw id of slot i is id_base + i (slot of the vertex in its sample list) */
    /* Remote-frontier mode (range-partitioned graphs, n_hops == 1 only): seed i of batch b is some OTHER sampler's
     * frontier vertex; it is sampled with that sampler's draws: id = seed_ids[b*n_seeds+i], call id =
     * seed_call_ids[b*n_seeds+i] (instead of id_base + i and rng.call_id + b).  Both NULL = off. */
    const int64_t *seed_ids;
    const int64_t *seed_call_ids;
} tg_ns_config;

#define TG_TAG_NS_HOMO 1u
#define TG_TAG_NS_HETERO 2u

/* Per-batch output slabs of neighbor_sampling_homogenous.  Batch b owns
 * samples[b*cap_nodes ..], rows/cols/edge_index[b*cap_edges ..],
 * layer_offsets[b*n_hops*3 ..], counts[b*2 ..] = {n_samples, n_edges}.
 * `states` ([n_batches*cap_nodes]) is workspace for temporal filters, else NULL. */
typedef struct {
    int64_t *samples;
    int64_t *rows;
    int64_t *cols;
    int64_t *edge_index;
    int64_t *layer_offsets;
    int64_t *counts;
    int64_t *states;
    int64_t cap_nodes; /* >= tg_ns_homo_capacity() */
    int64_t cap_edges;
} tg_ns_out;

/* *max_degree_dev (device int64) = the longest column of `g` (reads ptrs32 when given, else ptrs).  Stream-ordered. */
TG_API int tg_graph_max_degree(const tg_graph *g, int64_t *max_degree_dev, void *stream);

TG_API const char *tg_version(void);
TG_API const char *tg_last_error(void);

/* Worst-case slab sizes for n_seeds seeds and fan-outs k_0..k_{H-1}:
 * cap_edges = sum_h n_seeds*k_0*...*k_h, cap_nodes = n_seeds + cap_edges. */
TG_API int tg_ns_homo_capacity(int64_t n_seeds, const int64_t *fanout, int32_t n_hops, int64_t *cap_nodes,
                        int64_t *cap_edges);

/* neighbor_sampling_homogenous (src/algo/neighbor_sampling.rs:162-230; binding
 * python.rs:187-271) for n_batches independent seed batches in one launch.
 * seeds: [n_batches * n_seeds] device int64.  Output layout per batch is the
 * reference's: samples = seeds ++ hop-1 samples ++ ..., rows[e] = n_seeds + e,
 * cols[e] = slot of the parent, edge_index[e] = CSC edge pointer,
 * layer_offsets[h] = (len(samples), len(edges), len(samples)) when hop h starts. */
TG_API int tg_ns_homo_batched(const tg_graph *csc, const int64_t *seeds, int64_t n_batches, int64_t n_seeds,
                       const int64_t *fanout, int32_t n_hops, const tg_ns_config *cfg, const tg_rng *rng,
                       const tg_ns_out *out, void *stream);

/* The same operator with a caller-provided workspace (tg_ns_homo_workspace_bytes; 256-byte aligned).  With it, a
 * launch of many batches (unweighted, unfiltered samplers, fan-outs <= TG_MAX_FANOUT) runs hop by hop over the whole
 * device and issues the `indices[edge_ptr]` gathers of a hop in ADDRESS order (items of the hop's frontier
 * counting-sorted by window of their column start, windows dealt to XCDs), so that each 128-byte line of `indices` is
 * fetched about once per hop instead of once per sampled neighbour.  Outputs are identical to tg_ns_homo_batched
 * (output positions are fixed by per-batch prefix sums, neighbor_sampling.rs:212-217); launches that do not qualify
 * fall through to it.  workspace == NULL is tg_ns_homo_batched.  `mode`: TG_NS_FORM_AUTO picks the form by launch size
 * (many batches against a graph far larger than the L2s), _WINDOWED takes it whenever the launch qualifies, _FUSED never. */
#define TG_NS_FORM_AUTO 0
#define TG_NS_FORM_WINDOWED 1
#define TG_NS_FORM_FUSED 2
#define TG_NS_FORM_WINDOWED_WIDE 3 /* _WINDOWED with the 24-byte work items that launches beyond 32-bit offsets use */
/* (the size includes the staged pipeline's stage slots only while tg_ns_win_tuning.staged is on at the time of the query;
 * a launch whose workspace lacks them takes the push form) */
TG_API int tg_ns_homo_workspace_bytes(int64_t n_batches, int64_t n_seeds, const int64_t *fanout, int32_t n_hops,
                               int64_t *n_bytes);
/* The same for a given graph: the staged pipeline's stage slots are then sized by the graph's bit widths (vertex ids and
 * tg_graph.max_degree: one 64-byte chunk per frontier vertex where the pairs {neighbour, position} fit, else two); without
 * a graph the larger size is assumed. */
TG_API int tg_ns_homo_workspace_bytes_for(const tg_graph *csc, int64_t n_batches, int64_t n_seeds, const int64_t *fanout,
                                   int32_t n_hops, int64_t *n_bytes);
/* The workspace for a launch with THIS sampler / filter configuration: the window-ordered form's for the plain samplers
 * (as above); for the weighted sampler or a temporal filter with at most 256 batches, the workspace of the FLAT path -- the
 * launch then runs hop by hop over the whole device (tg_ns_hop_scan / tg_ns_hop_weighted_groups on all batches' frontiers
 * at once) instead of one workgroup per batch: 64 weighted batches of 1 024 seeds on RMAT-24 53 -> 15 ms, temporal 5.5 ->
 * 2.3 ms; 8 batches 47 -> 2.2 / 5.2 -> 0.55 ms.  Results are tg_ns_homo_batched's bit for bit (the per-batch kernel runs behind the flat path whenever that could
 * not finish: a column-group bound reached, a non-positive weight sum).  0 bytes: the launch takes no workspace. */
TG_API int tg_ns_homo_batched_workspace_bytes(const tg_graph *csc, int64_t n_batches, int64_t n_seeds, const int64_t *fanout,
                                       int32_t n_hops, const tg_ns_config *cfg, int64_t *n_bytes);
TG_API int tg_ns_homo_batched_ws(const tg_graph *csc, const int64_t *seeds, int64_t n_batches, int64_t n_seeds,
                          const int64_t *fanout, int32_t n_hops, const tg_ns_config *cfg, const tg_rng *rng,
                          const tg_ns_out *out, void *workspace, int64_t workspace_bytes, int32_t mode, void *stream);

/* Which form a tg_ns_homo_batched_ws call with these arguments runs: *form = TG_NS_FORM_FUSED (the per-batch kernel; also
 * reported for the weighted / filtered samplers, which take neither), _WINDOWED (16-byte work items) or _WINDOWED_WIDE;
 * *n_windows (optional) = windows the hop's frontier is sorted into.  A pure query: nothing is launched. */
TG_API int tg_ns_homo_batched_form(const tg_graph *csc, int64_t n_batches, int64_t n_seeds, const int64_t *fanout,
                            int32_t n_hops, const tg_ns_config *cfg, const tg_ns_out *out, int64_t workspace_bytes,
                            int32_t mode, int32_t *form, int32_t *n_windows);

/* Which PIPELINE of the window-ordered form the call takes under the current tuning: *staged = 1 the staged one (gather
 * into stage slots, then emit; needs the larger workspace), 0 the push one or no window-ordered form at all.  A pure query. */
TG_API int tg_ns_homo_batched_pipeline(const tg_graph *csc, int64_t n_batches, int64_t n_seeds, const int64_t *fanout,
                                int32_t n_hops, const tg_ns_config *cfg, const tg_ns_out *out, int64_t workspace_bytes,
                                int32_t mode, int32_t *staged);

/* Tuning of the window-ordered form (process-wide; defaults come from TG_WIN_* environment variables read once).
 * Outputs never depend on it.  _set: a zero field (negative for the four flags) keeps the current value. */
typedef struct {
    int64_t window_bytes;    /* bytes of the gathered array per window (default 512 KiB) */
    int32_t gather_blocks;   /* workgroups of the persistent gather kernel (default 256) */
    int32_t gather_threads;  /* its workgroup size (default 512) */
    int32_t emit_threads;    /* workgroup size of the emit kernels (default 256) */
    int32_t direct_hop0;     /* hop 0 issues its gathers itself, unordered (default 1) */
    int32_t fuse_first_hops; /* seeds + hop 0 + hop 1's emit pass in one kernel (default 1) */
    int32_t fold_hist;       /* emit kernels persistent, window histogram of the items kept in LDS instead of a pass of its
                                own over the items (default 1; needs the two flags above and >= 2 hops) */
    int32_t emit_blocks;     /* workgroups of the persistent emit kernels (default 768, at most 1024) */
    int32_t staged;          /* gather first, emit afterwards through packed 64-byte stage slots: 0 = never,
                                1 = whenever it applies, 2 (default) = where it measures faster than the push form: launches
                                of >= 2 048 batches whose slots are one chunk; launches it does not fit -- ordered fan-outs
                                > 30, ids beyond 32 bits, a workspace without the slots -- take the push form anyway;
                                < 0 in _set keeps */
    int32_t stage_round_chunks;   /* staged emit kernel: 64-slot chunks per round (default 2: the smaller tile lets five workgroups share a CU, emit 3.15 -> 3.07 ms; at most 16) */
    int32_t stage_gather_threads; /* staged gather kernel: workgroup size (default 512) */
    int32_t stage_gather_blocks;  /* ... and workgroups (default 512) */
    int32_t stage_emit_threads;   /* staged emit kernel: workgroup size (default 256) */
    int32_t stage_parts;          /* staged form: the last hop runs in this many parts (batch ranges), part p's emit pass on
                                     a side stream beside part p + 1's sort and gather (default 1 = one stream: measured, the overlap
                                     buys nothing -- both kernels are bound by vector-ALU work -- and every part costs 0.2 ms) */
    int32_t stage_part_min_batches; /* ... as long as every part keeps at least this many batches (default 1024) */
    int32_t stage_sort_blocks;      /* staged form: workgroups of the item sort = rows of its histogram = persistent workgroups
                                       of the first kernel (default 768, at most 1024) */
    int32_t stage_fine;             /* staged form: second sort level (tiles of the coarse-sorted items ordered by vertex in LDS,
                                       so that a gather workgroup's slice lies in a few columns; default 1; < 0 in _set keeps) */
    int32_t stage_concurrent;       /* staged form, two hops, several parts: the parts' whole chains alternate between the caller's
                                       stream and the side stream (default 0; < 0 in _set keeps) */
    int32_t stage_split;            /* staged form: rows / cols / edge_index of an ordered hop are written by a pass of their own on
                                       the side stream, beside the hop's sort and gather; the emit pass writes `samples` only
                                       (< 0 in _set keeps) */
    int32_t stage_split_round_chunks; /* ... whose tiles hold this many 64-slot chunks (default 4; 0 in _set keeps) */
    int32_t store_align64;          /* the emit passes' store instructions start on 64-byte boundaries (the elements before the
                                       first boundary are stored alone): non-temporal stores of a chunk that two instructions
                                       share go out as two partial writes (default 1; 0: 16-byte boundaries; < 0 in _set keeps) */
    int32_t stage_fine_sub_bits;    /* staged form: the second sort level orders a window's vertices into 1 << this many sub-ranges
                                       (4 .. 7, default 7: the finer the order, the fewer columns a gather workgroup's slice touches --
                                       gather 2.39 / 2.26 / 2.11 / 2.01 ms at 4 / 5 / 6 / 7, the sort 0.46 / 0.47 / 0.52 / 0.56; 0 in _set keeps) */
    int32_t stage_fine_blocks;      /* staged form: workgroups of the second sort level's two passes (default 2048, rounded up to a
                                       multiple of 8; 0 in _set keeps) */
} tg_ns_win_tuning;
TG_API int tg_ns_win_tuning_get(tg_ns_win_tuning *t);
TG_API int tg_ns_win_tuning_set(const tg_ns_win_tuning *t);

/* Per-stage times of the window-ordered launch: tg_ns_win_stage_timing(1) makes every later launch record HIP events
 * between its kernels (on its stream); tg_ns_win_stage_times waits for the last launch and returns up to `cap` stage
 * durations in ms with their names (24 bytes each, "<stage>.h<hop>").  Measurement only. */
TG_API int tg_ns_win_stage_timing(int32_t enable);
TG_API int tg_ns_win_stage_times(float *ms, char *names, int32_t cap, int32_t *n);

/* One hop of the unweighted, unfiltered sampler over a flat frontier, spread over the whole device (the
 * per-vertex work of neighbor_sampling.rs:195-218 without the per-batch bookkeeping).  Used where the frontier
 * is not "one seed batch": the owner side of the range-partitioned sampler, relation-hops of the heterogeneous
 * sampler.  Vertex i of the frontier is sampled with draw id ids[i] (or id_base + i) and call id call_ids[i]
 * (or rng.call_id); vertices < 0 are empty slots.  Outputs are sized for the worst case m * fanout;
 * offsets[m] is the number of samples.  No synchronisation.  Fan-outs up to 4096: above 128 one wavefront per vertex
 * runs the ticket chain with its displaced entries in LDS (the batched kernel stops at 255). */
typedef struct {
    const int64_t *vertices; /* [m] */
    const int64_t *ids;      /* [m] or NULL */
    const int64_t *call_ids; /* [m] or NULL */
    int64_t m;
    int64_t id_base;
    int32_t fanout;
    int32_t sampler;  /* TG_SAMPLER_UNIFORM or TG_SAMPLER_UNIFORM_REPL */
    uint32_t rng_tag; /* 0 = TG_TAG_NS_HOMO */
    uint32_t _reserved;
} tg_hop_in;

typedef struct {
    int64_t *cnt;       /* [m] samples per frontier vertex */
    int64_t *offsets;   /* [m + 1] exclusive prefix of cnt */
    int64_t *neighbors; /* [m * fanout] indices[edge_ptr], grouped by frontier vertex in reservoir-slot order */
    int64_t *edge_ptrs; /* [m * fanout] */
    int64_t *parents;   /* [m * fanout] index of the frontier vertex */
} tg_hop_out;

TG_API int tg_ns_hop_workspace_bytes(int64_t m, int64_t *bytes);
TG_API int tg_ns_hop(const tg_graph *csc, const tg_hop_in *in, const tg_rng *rng, const tg_hop_out *out,
                     void *workspace, int64_t workspace_bytes, void *stream);

/* The same flat hop for the unweighted samplers UNDER A TEMPORAL FILTER (neighbor_sampling.rs:36-77): columns are
 * cut into groups of 512 edges that are counted all over the device, then every vertex draws its ranks (ticket
 * form) and fetches them.  `states` is the filter state of every frontier vertex, states_out [m * fanout] that of
 * every sample.  group_cap bounds the number of 512-edge groups the frontier's columns may have
 * (sum of ceil(deg / 512)); if it is reached status[0] (device int32, zeroed by the caller) becomes 1 and all
 * counts are 0 -- retry with a larger workspace. */
typedef struct {
    int32_t filter_mode; /* TG_FILTER_STATIC / RELATIVE / DYNAMIC */
    int32_t forward;
    int64_t win_lo, win_hi; /* inclusive window */
    const int64_t *states;  /* [m]  Fan-outs up to 1024 (above 64 the ticket chain keeps its displaced
 * entries in LDS). */
} tg_hop_filter;

TG_API int tg_ns_hop_scan_workspace_bytes(int64_t m, int32_t fanout, int64_t group_cap, int64_t *bytes);
TG_API int tg_ns_hop_scan(const tg_graph *csc, const tg_hop_in *in, const tg_hop_filter *filter, const tg_rng *rng,
                          const tg_hop_out *out, int64_t *states_out, int32_t *status, void *workspace,
                          int64_t workspace_bytes, int64_t group_cap, void *stream);

/* The flat hop for the WEIGHTED sampler (sampling.rs:28-55), with or without a temporal filter (filter = NULL or
 * filter_mode = TG_FILTER_NONE: none).  One wavefront per frontier vertex all over the device; a column's
 * left-to-right f64 running sum is kept exactly.  status[0] |= 2 where the reference panics (sum <= 0).
 * Workspace: tg_ns_hop_scan_workspace_bytes(m, fanout, 1). */
TG_API int tg_ns_hop_weighted(const tg_graph *csc, const tg_hop_in *in, const tg_hop_filter *filter, const tg_rng *rng,
                              const tg_hop_out *out, int64_t *states_out, int32_t *status, void *workspace,
                              int64_t workspace_bytes, void *stream);
/* The same hop in the GROUP FORM: the columns are cut into the 512-edge groups of the filtered hop above; chunk totals and
 * draws run flat over the groups of ALL frontier columns, the carries of the blocked running sum per column between
 * them -- a hub column no longer occupies one wavefront or one workgroup while the device idles (RMAT-24, 64 batches of
 * 1 024 seeds: 39.7 -> 15.3 ms).  Same draws, same sums, same result.  group_cap >= 1024 bounds the frontier's groups as in
 * tg_ns_hop_scan (reached: status[0] |= 1, all counts 0 -- retry with more); workspace:
 * tg_ns_hop_weighted_workspace_bytes(m, fanout, group_cap) (64 bytes more per group than the filtered hop's).
 * tg_ns_hop_segments with the weighted sampler takes this form under the same condition and needs the same workspace. */
TG_API int tg_ns_hop_weighted_workspace_bytes(int64_t m, int32_t fanout, int64_t group_cap, int64_t *bytes);
TG_API int tg_ns_hop_weighted_groups(const tg_graph *csc, const tg_hop_in *in, const tg_hop_filter *filter, const tg_rng *rng,
                                     const tg_hop_out *out, int64_t *states_out, int32_t *status, void *workspace,
                                     int64_t workspace_bytes, int64_t group_cap, void *stream);

/* neighbor_sampling_heterogenous (src/algo/neighbor_sampling.rs:233-356; binding python.rs:275-395) in ONE launch for
 * n_batches independent seed batches, unweighted samplers without filter (the surface's default).  Node types and
 * relations are indexed in the caller's `node_types` / `edge_types` order, which is also the order relations are
 * visited in every hop (the reference iterates a HashMap there).  Host arrays describe the problem; the pointers
 * inside are device pointers.  Batch b owns samples[t][b*cap_nodes[t] ..], rows/cols/edge_index[r][b*cap_edges[r] ..],
 * layer_offsets[((b*n_rels + r)*n_hops + h)*3 ..] = (len(samples[src]), len(edges r), len(samples[dst])) when hop h
 * reaches relation r (:314-315), counts[b*(n_types+n_rels) ..] = {len(samples[t])..., len(edges r)...}.
 * Draw address: tag TG_TAG_NS_HETERO | r << 8, id = slot of the frontier vertex in its type's list, call id
 * rng.call_id + b.  Weighted / filtered heterogeneous sampling is driven per (hop, relation) through
 * tg_ns_homo_batched / tg_ns_hop_scan / tg_ns_hop_weighted with the same addresses. */
#define TG_HET_MAX_TYPES 8
#define TG_HET_MAX_RELS 16
typedef struct {
    int32_t n_types, n_rels, n_hops;
    int32_t sampler;               /* TG_SAMPLER_UNIFORM or TG_SAMPLER_UNIFORM_REPL */
    const int32_t *rel_src;        /* [n_rels] node type sampled FROM (CSC rows) */
    const int32_t *rel_dst;        /* [n_rels] node type whose frontier is expanded (CSC columns) */
    const tg_graph *graphs;        /* [n_rels] CSC per relation */
    const int64_t *fanout;         /* [n_rels * n_hops]; 0 = relation not sampled in that hop */
    const int64_t *const *inputs;  /* [n_types] device [n_batches * n_inputs[t]], NULL where n_inputs[t] == 0 */
    const int64_t *n_inputs;       /* [n_types] seeds per batch */
} tg_het_problem;

typedef struct {
    int64_t *const *samples;    /* [n_types] device slabs [n_batches * cap_nodes[t]] */
    const int64_t *cap_nodes;   /* [n_types] >= tg_ns_hetero_capacity() */
    int64_t *const *rows;       /* [n_rels] device slabs [n_batches * cap_edges[r]] */
    int64_t *const *cols;
    int64_t *const *edge_index;
    const int64_t *cap_edges;   /* [n_rels] */
    int64_t *layer_offsets;     /* device [n_batches * n_rels * n_hops * 3] */
    int64_t *counts;            /* device [n_batches * (n_types + n_rels)] */
} tg_het_out;

TG_API int tg_ns_hetero_capacity(const tg_het_problem *problem, int64_t *cap_nodes, int64_t *cap_edges);
TG_API int tg_ns_hetero_batched(const tg_het_problem *problem, int64_t n_batches, const tg_rng *rng, const tg_het_out *out,
                                void *stream);

/* Device-side bookkeeping for neighbor_sampling_heterogenous (neighbor_sampling.rs:292-352) when its (hop, relation)
 * steps run as flat hops -- temporal filters, the weighted sampler, sizes beyond the fused launch: list lengths,
 * frontier slices and edge counts live in `meta` (device, tg_het_meta_words int64 words:
 * len[T] | fbeg[T] | fend[T] | ne[R] | layer_offsets[R][H][3] | scratch words; the caller initialises len = fend =
 * number of inputs, fbeg = 0, ne = 0), so a call needs no read-back between steps.  Per hop, per relation in
 * `edge_types` order: tg_het_step_begin (frontier = list[dst][fbeg, fend) into a buffer of cap_f slots padded with -1,
 * draw ids = slot in the list, layer_offsets[rel][hop]) -> tg_ns_hop / tg_ns_hop_scan / tg_ns_hop_weighted with m =
 * cap_f -> tg_het_step_end (append samples / states to list[src], (row, col, edge pointer) to the relation's lists,
 * advance the lengths; status |= 4 if a capacity were exceeded); after the relations tg_het_hop_end. */
TG_API int tg_het_meta_words(int32_t n_types, int32_t n_rels, int32_t n_hops, int64_t *words);
TG_API int tg_het_step_begin(const int64_t *list_dst, const int64_t *state_dst, int64_t *meta, int32_t n_types, int32_t n_rels,
                             int32_t n_hops, int32_t src, int32_t dst, int32_t rel, int32_t hop, int64_t cap_f,
                             int64_t *frontier, int64_t *fstate, int64_t *ids, void *stream);
TG_API int tg_het_step_end(const tg_hop_out *out, const int64_t *states_out, int64_t cap_f, int32_t fanout, int64_t *meta,
                           int32_t n_types, int32_t n_rels, int32_t n_hops, int32_t src, int32_t rel, int64_t *list_src,
                           int64_t *state_src, int64_t cap_list, int64_t *rows, int64_t *cols, int64_t *edge_index,
                           int64_t cap_edges, int32_t *status, void *stream);
TG_API int tg_het_hop_end(int64_t *meta, int32_t n_types, int32_t n_rels, int32_t n_hops, void *stream);

/* The same hop with ALL its relations in one set of launches (a per-call operator is bound by its number of launches).
 * A hop's frontier is fixed when the hop starts (neighbor_sampling.rs:345-348 advance the slices at its end), so the
 * relations' frontiers are concatenated (entry j owns slots [begin, begin + cap) of the concatenation, cap = its
 * worst-case frontier size), sampled by ONE tg_ns_hop_segments call, and their appends are replayed in relation order:
 *   tg_het_hop_begin_all  snapshots lengths / edge counts / frontier starts; builds the concatenated frontier (-1
 *                         padded), its draw ids and filter states.
 *   tg_ns_hop_segments    the flat filtered / weighted hop over a frontier made of up to TG_HOP_MAX_SEGMENTS segments,
 *                         each with its own graph, fan-out and draw tag (hop_in.fanout / rng_tag are ignored;
 *                         hop_in.sampler = TG_SAMPLER_WEIGHTED takes tg_ns_hop_weighted's algorithm, the unweighted
 *                         samplers tg_ns_hop_scan's and need a filter).  Outputs are compact over the concatenation, so a
 *                         segment's samples are contiguous; `parents` index the concatenation.  Workspace:
 *                         tg_ns_hop_scan_workspace_bytes(m, largest fan-out, group_cap).
 *   tg_het_hop_end_all    for the entries in order: layer_offsets[rel][hop], append samples / states / edges, advance
 *                         lengths and edge counts; `last` != 0 also sets the next hop's frontier slices.  out_cap: an
 *                         upper bound of the hop's samples (sizes the launch).
 * layout_dev (device, [segments + 1], NULL = none): the frontier WITHOUT its padding -- tg_het_hop_begin_all packs the
 * segments' real frontiers back to back and writes their starts and the total there; tg_ns_hop_segments and
 * tg_het_hop_end_all then work on the real length (the host-side begin / cap / m stay the worst case that sizes the
 * launches).  Needs m <= 2^17 and group_cap <= 2^20.
 * Entries list EVERY relation sampled in the hop, in `edge_types` order, also those whose frontier is necessarily empty
 * (segment = -1: only their layer offset is recorded).  More than TG_HET_HOP_MAX_ENTRIES relations or
 * TG_HOP_MAX_SEGMENTS segments: several begin / sample / end rounds, `last` on the final one. */
#define TG_HOP_MAX_SEGMENTS 8
#define TG_HET_HOP_MAX_ENTRIES 16
typedef struct {
    const tg_graph *graph;
    int64_t begin;    /* first frontier slot of the segment; ascending, segment 0 starts at 0 */
    int32_t fanout;
    uint32_t rng_tag; /* 0 = TG_TAG_NS_HOMO */
} tg_hop_segment;
typedef struct {
    int32_t rel, src, dst;
    int32_t segment;                      /* >= 0: the relation has a frontier segment in this round; -1: none */
    int64_t begin, cap;                   /* its slots in the concatenated frontier */
    const int64_t *list_dst, *state_dst;  /* sample list / filter states of the dst type (the frontier is read there) */
    int64_t *list_src, *state_src;        /* sample list / filter states of the src type (samples are appended) */
    int64_t cap_list_src;
    int64_t *rows, *cols, *edge_index;    /* the relation's edge lists */
    int64_t cap_edges;
} tg_het_entry;
TG_API int tg_ns_hop_segments(const tg_hop_segment *segments, int32_t n_segments, const tg_hop_in *in,
                              const int64_t *layout_dev, const tg_hop_filter *filter, const tg_rng *rng,
                              const tg_hop_out *out, int64_t *states_out, int32_t *status, void *workspace,
                              int64_t workspace_bytes, int64_t group_cap, void *stream);
TG_API int tg_het_hop_begin_all(const tg_het_entry *entries, int32_t n_entries, int64_t *meta, int32_t n_types, int32_t n_rels,
                                int32_t n_hops, int64_t m_total, int64_t *frontier, int64_t *fstate, int64_t *ids,
                                int64_t *layout_dev, void *stream);
TG_API int tg_het_hop_end_all(const tg_het_entry *entries, int32_t n_entries, const tg_hop_out *out, const int64_t *states_out,
                              int64_t m_total, int64_t out_cap, const int64_t *layout_dev, int64_t *meta, int32_t n_types,
                              int32_t n_rels, int32_t n_hops, int32_t hop, int32_t last, int32_t *status, void *stream);

/* neighbor_sampling_homogenous over a RANGE-PARTITIONED CSC (graphs beyond one GPU's HBM; host protocol in
 * tch_geometric/partitioned.py; SURVEY.md 8(e) mode 2).  The origin rank keeps ordinary
 * tg_ns_out slabs; per hop: tg_part_requests (origin) -> all-to-all -> tg_part_count + tg_part_sample (owner) ->
 * all-to-all -> tg_part_emit (origin).  Results equal tg_ns_homo_batched on the replicated graph bit for bit (requests
 * carry the requester's draw address).  EVERY size stays on the device (kernels read the number of requests from
 * device memory): with world == 1 nothing is read back; with world > 1 the host reads only the all-to-all split sizes.
 *  - request: 16 bytes {int64 vertex; uint32 batch; uint32 slot}; the requester's call id = its first call id + batch.
 *  - request_cap = n_batches * the widest frontier (< 2^32); workspace: tg_part_workspace_bytes, 256-byte aligned.
 *  - tg_part_begin: copies the seeds into the slabs, frontier = the seeds.
 *  - tg_part_requests: `requests` [<= request_cap] grouped by owner = min(vertex / shard_size, world-1);
 *    send_counts[world + 1] (device): requests per owner, then their total.
 *  - tg_part_count: owner of columns [v_lo, v_lo + shard.n_major).  m_dev: number of received requests (device);
 *    m_cap: host-known upper bound of it (sizes the launches and the prefix sum); seg_off[world + 1] (host): requests of
 *    requesting rank p are [seg_off[p], seg_off[p+1]); seg_call0[world] (host): that rank's first call id.
 *    -> cnt[m] u32 samples per request, off[m + 1] their exclusive prefix (off[m] = total), reply_counts[world + 1] (device): reply entries
 *    per requesting rank, then their total.  scan_tmp: tg_part_scan_workspace_bytes(m_cap).
 *  - tg_part_sample: reply [sum cnt] entries = (neighbour id, global edge pointer = local pointer + e_lo), compact, in
 *    request order; unweighted samplers, fanout <= TG_MAX_FANOUT.
 *  - reply_format (the same value on every rank; entry size in int64 words in brackets):
 *    TG_PART_REPLY_PAIRS [2] neighbour, edge pointer; TG_PART_REPLY_TRIPLES [3] + the sample's filter state;
 *    TG_PART_REPLY_PACKED [1] neighbour | edge pointer << 32 -- valid when the whole graph has < 2^32 vertices and
 *    < 2^32 edges, halves what the reply all-to-all moves; TG_PART_REPLY_PACKED_STATE [2] packed word, filter state.
 *  - tg_part_emit: cnt / reply as returned to the origin (request order); cnt_prefix: the exclusive prefix of cnt if
 *    the caller already holds it (world == 1: tg_part_count's `off`), else NULL; hop_cap = n_batches * this hop's
 *    widest frontier; compacts the replies into the slabs in slot order, advances the frontier, writes layer_offsets[hop],
 *    counts. */
#define TG_PART_REPLY_PACKED 1
#define TG_PART_REPLY_PAIRS 2
#define TG_PART_REPLY_TRIPLES 3
#define TG_PART_REPLY_PACKED_STATE 4
TG_API int tg_part_workspace_bytes(int64_t n_batches, int64_t request_cap, int32_t world, int64_t *bytes);
TG_API int tg_part_scan_workspace_bytes(int64_t n, int64_t *bytes);
TG_API int tg_part_begin(const int64_t *seeds, const int64_t *seeds_state, int64_t n_batches, int64_t n_seeds, int32_t n_hops,
                         const tg_ns_out *out, int64_t request_cap, int32_t world, void *workspace, void *stream);
TG_API int tg_part_requests(const tg_ns_out *out, int64_t n_batches, int64_t request_cap, int64_t shard_size, int32_t world,
                            void *workspace, void *requests, int64_t *request_states, int64_t *send_counts, void *stream);
TG_API int tg_part_count(const tg_graph *shard, int64_t v_lo, const void *requests, const int64_t *m_dev, int64_t m_cap,
                         int32_t world, const int64_t *seg_off, const uint64_t *seg_call0, int32_t fanout, int32_t sampler,
                         uint32_t *cnt, int64_t *off, int64_t *reply_counts, void *scan_tmp, int64_t scan_tmp_bytes,
                         void *stream);
TG_API int tg_part_sample(const tg_graph *shard, int64_t v_lo, int64_t e_lo, const void *requests, const int64_t *m_dev,
                          int64_t m_cap, int32_t world, const int64_t *seg_off, const uint64_t *seg_call0, int32_t fanout,
                          int32_t sampler, uint64_t seed, const uint32_t *cnt, const int64_t *off, int64_t *reply,
                          int32_t reply_format, void *stream);
/* Temporal filters and the weighted sampler (neighbor_sampling.rs:36-77, :131-158) on a partitioned graph: the origin
 * also sends every frontier vertex's filter state (seeds_state / request_states [<= request_cap], grouped like the
 * requests; NULL = no filter); the owner unpacks the requests into the flat-hop input arrays (tg_part_unpack: vertices
 * rebased to the shard, -1 for padding; draw ids = the requesters' slots; call ids = the requesters'), runs
 * tg_ns_hop_scan / tg_ns_hop_weighted over them (m = m_cap, shard.timestamps / shard.weights = the shard's slices) and
 * packs the hop's compact outputs into the reply (tg_part_pack: cnt u32, entries in reply_format, reply_counts).
 * tg_part_emit with a reply_format that carries the filter state also fills the `states` slab.  Same draws as the replicated sampler, so the same results. */
/* tg_part_sample with a workspace (tg_part_sample_workspace_bytes(m_cap); 256-byte aligned): hops with many requests
 * (>= 2^21) against a large shard are first counting-sorted by the WINDOW of their column (found from the vertex id through
 * a vertex -> window table) and sampled in that order, XCD by XCD, so that the `indices[edge_ptr]` gathers hit L2 instead
 * of costing a line request each; the replies are the same words in the same places.  workspace == NULL or a small hop:
 * tg_part_sample. */
TG_API int tg_part_sample_workspace_bytes(int64_t m_cap, int64_t *bytes);
/* when tg_part_sample_ws orders a hop (process-wide; defaults 2^21 requests, 2^24 shard edges; negative = keep).  Results
 * never depend on it; tests lower the thresholds to run the ordered path on small graphs. */
TG_API int tg_part_sample_order_thresholds(int64_t min_requests, int64_t min_edges);
TG_API int tg_part_sample_ws(const tg_graph *shard, int64_t v_lo, int64_t e_lo, const void *requests, const int64_t *m_dev,
                      int64_t m_cap, int32_t world, const int64_t *seg_off, const uint64_t *seg_call0, int32_t fanout,
                      int32_t sampler, uint64_t seed, const uint32_t *cnt, const int64_t *off, int64_t *reply,
                      int32_t reply_format, void *workspace, int64_t workspace_bytes, void *stream);
/* Slot replies (unweighted, unfiltered sampling; round 4).  Instead of tg_part_count + tg_part_sample's compact reply the
 * owner answers every request with ONE fixed-size packed slot -- {column start : 32, cnt : 8, fanout x {neighbour :
 * vertex_bits, position inside the column : position_bits}} in W = 16 or 32 32-bit words (tg_part_slot_words; 0 = no
 * format: fan-out > 16 or more than 1024 bits) -- written at the request's index: slots [m_cap][W], 64-byte aligned.
 * The reply all-to-all carries W words per request with the request exchange's split sizes mirrored, so the host reads
 * back ONE set of sizes per hop instead of two, and neither side runs a count / prefix pass.  vertex_bits = bits of the
 * largest vertex id of the WHOLE graph, position_bits = bits of (longest column of the whole graph - 1): every rank
 * passes the same values.  A shard holds < 2^32 edges.  workspace: tg_part_sample_workspace_bytes(m_cap) (hops with many
 * requests are sampled in window order, as tg_part_sample_ws) or NULL.  order_scale (>= 1) multiplies the request threshold
 * of that ordering for this hop: a call's FIRST hop asks about its seeds -- mostly short columns that are read whole, one
 * or two lines each -- and pays for the ordering only from about 4 x as many requests on (pass 4 there, 1 for later hops).
 * tg_part_emit_slots: slots as returned to the origin (its send-buffer order); e_lo_of: host [world], the global edge
 * offset of every rank's shard -- a sample's edge pointer = e_lo_of[owner] + column start + position.
 * Same draws as tg_part_sample, so the same results as the replicated sampler. */
TG_API int tg_part_slot_words(int32_t fanout, int32_t vertex_bits, int32_t position_bits, int32_t *words);
TG_API int tg_part_sample_slots(const tg_graph *shard, int64_t v_lo, const void *requests, const int64_t *m_dev, int64_t m_cap,
                                int32_t world, const int64_t *seg_off, const uint64_t *seg_call0, int32_t fanout,
                                int32_t sampler, uint64_t seed, int32_t vertex_bits, int32_t position_bits, int32_t order_scale,
                                void *slots, void *workspace, int64_t workspace_bytes, void *stream);
TG_API int tg_part_emit_slots(const tg_ns_out *out, int64_t n_batches, int64_t n_seeds, int64_t request_cap, int32_t world,
                              int32_t fanout, int32_t hop, int32_t n_hops, void *workspace, const void *slots,
                              int32_t vertex_bits, int32_t position_bits, const int64_t *e_lo_of, void *stream);
TG_API int tg_part_unpack(int64_t v_lo, int64_t n_major, const void *requests, const int64_t *m_dev, int64_t m_cap,
                          int32_t world, const int64_t *seg_off, const uint64_t *seg_call0, int64_t *vertices, int64_t *ids,
                          int64_t *call_ids, void *stream);
TG_API int tg_part_pack(const tg_hop_out *hop, const int64_t *states_out, const int64_t *m_dev, int64_t m_cap, int64_t e_lo,
                        int32_t world, const int64_t *seg_off, uint32_t *cnt, int64_t *reply, int32_t reply_format,
                        int64_t *reply_counts, void *stream);
TG_API int tg_part_emit(const tg_ns_out *out, int64_t n_batches, int64_t n_seeds, int64_t request_cap, int64_t hop_cap,
                        int32_t world, int32_t fanout, int32_t hop, int32_t n_hops, void *workspace, const uint32_t *cnt,
                        const int64_t *cnt_prefix, const int64_t *reply, int32_t reply_format, void *stream);

/* random_walk (src/algo/random_walk.rs:10-75; binding python.rs:584-608).
 * walks: [n, walk_length + 1] device int64, -1 padded after a dead end. */
TG_API int tg_random_walk(const tg_graph *csr, const int64_t *start, int64_t n, int64_t walk_length, float p, float q,
                   const tg_rng *rng, int64_t *walks, void *stream);

/* The EDGE SET of a CSR: SparseGraph::has_edge (src/data/graph.rs:80-83, a binary search of the row: ~log2(deg) dependent
 * random line requests) as one hash probe.  node2vec with p != q asks has_edge once per proposal (random_walk.rs:57-63);
 * with the set a 1 M-walker call on RMAT-24 takes a quarter of the time, with the same walks.  Optional: built once per
 * graph by the caller (8 bytes x the next power of two >= 2 x n_edges; vertex ids must be < 2^32 - 1), handed to
 * tg_random_walk_es; edge_set = NULL is tg_random_walk. */
TG_API int tg_edge_set_bytes(const tg_graph *csr, int64_t *bytes);
TG_API int tg_edge_set_build(const tg_graph *csr, void *edge_set, int64_t bytes, void *stream);
TG_API int tg_random_walk_es(const tg_graph *csr, const void *edge_set, int64_t edge_set_bytes, const int64_t *start, int64_t n,
                             int64_t walk_length, float p, float q, const tg_rng *rng, int64_t *walks, void *stream);

/* tempo_random_walk (random_walk.rs:80-158; binding python.rs:611-642).
 * walks, walks_ts: [n, walk_length] device int64. */
TG_API int tg_tempo_random_walk(const tg_graph *csr, const int64_t *node_ts, const int64_t *edge_ts, const int64_t *start,
                         const int64_t *start_ts, int64_t n, int64_t walk_length, int64_t win0, int64_t win1,
                         const tg_rng *rng, int64_t *walks, int64_t *walks_ts, void *stream);

/* biased_tempo_random_walk (random_walk.rs:160-288; binding python.rs:645-687): time-respecting walk whose next
 * vertex is drawn with BiasType weights (random_walk.rs:165-182) by the one-slot weighted reservoir (f32,
 * utils/sampling.rs:28-55); a walker without candidates restarts from its start vertex, at most retry_count times.
 * walks, walks_ts: [n, walk_length] device int64.  status (device int32, zeroed by the caller): bit 1 = the reference
 * would have panicked (empty float range, sampling.rs:49), bit 0 = a row longer than `max_degree` met the linear
 * bias.  `max_degree` (largest CSR row) sizes the sort slab the linear bias needs for rows above 1024 edges:
 * workspace = tg_biased_walk_workspace_bytes(n, max_degree, bias), 0 for the other biases. */
#define TG_BIAS_UNIFORM 0
#define TG_BIAS_LINEAR 1
#define TG_BIAS_EXPONENTIAL 2
TG_API int tg_biased_walk_workspace_bytes(int64_t n, int64_t max_degree, int32_t bias, int64_t *bytes);
TG_API int tg_biased_tempo_random_walk(const tg_graph *csr, const int64_t *node_ts, const int64_t *edge_ts,
                                       const int64_t *start, const int64_t *start_ts, int64_t n, int64_t walk_length,
                                       int32_t bias, int32_t forward, int64_t retry_count, int64_t max_degree,
                                       const tg_rng *rng, int64_t *walks, int64_t *walks_ts, int32_t *status,
                                       void *workspace, int64_t workspace_bytes, void *stream);

/* negative_sample_neighbors_homogenous / _heterogenous (src/algo/negative_sampling.rs:6-131; bindings
 * python.rs:690-783) as one problem description.  Host arrays are indexed by node type (order of the caller's
 * `node_types`) and relation (order of `edge_types`); the reference's HashMap visiting order is replaced by
 * those orders. */
typedef struct {
    int32_t n_types, n_rels;
    int32_t homogeneous; /* 1: the homogeneous operator (no relation draw; n_types = n_rels = 1) */
    int32_t inbound;     /* negative_sampling.rs:112-115 */
    const int32_t *rel_src;       /* host [n_rels] node-type index of the relation's source */
    const int32_t *rel_dst;       /* host [n_rels] */
    const tg_graph *graphs;       /* host [n_rels] CSR views (device pointers inside) */
    const int64_t *node_count;    /* host [n_rels]: graph_size.1 / sizes[rel].1 (negative_sampling.rs:28,105) */
    const int64_t *const *inputs; /* host [n_types] device pointers */
    const int64_t *n_inputs;      /* host [n_types]; < 0: the type has no entry in `inputs` */
    int64_t num_neg, try_count;
} tg_neg_problem;

typedef struct {
    int64_t *const *samples; /* host [n_types] device buffers, capacity max(n_inputs[t],0) + total items */
    int64_t *const *rows;    /* host [n_rels] device buffers, capacity n_inputs[src] * num_neg */
    int64_t *const *cols;    /* host [n_rels] */
    int64_t *n_samples;      /* device [n_types] */
    int64_t *n_edges;        /* device [n_rels] */
    int32_t *panic;          /* device [1]: 1 where the reference would panic (inbound row out of range) */
} tg_neg_out;

TG_API int tg_neg_workspace_bytes(const tg_neg_problem *problem, int64_t *bytes);
TG_API int tg_neg_sample(const tg_neg_problem *problem, const tg_rng *rng, const tg_neg_out *out, void *workspace,
                         void *stream);

/* hgt_sampling (src/algo/hgt_sampling.rs:138-278; binding python.rs:399-482).  Host arrays are indexed by node
 * type (`node_types` order) and relation (`edge_types` order); graphs are CSC, graphs[r].timestamps is the
 * relation's row_timestamps or NULL.  All state lives in the caller's workspace; nothing synchronises. */
typedef struct {
    int32_t n_types, n_rels, n_hops;
    int32_t has_timerange;          /* timerange = [tr_lo, tr_hi) (python.rs:450) */
    const int32_t *rel_src;         /* host [n_rels] */
    const int32_t *rel_dst;         /* host [n_rels] */
    const tg_graph *graphs;         /* host [n_rels] */
    const int64_t *const *inputs;   /* host [n_types] device pointers (NULL when absent) */
    const int64_t *const *input_ts; /* host [n_types] device pointers, or NULL: no input timestamps */
    const int64_t *n_inputs;        /* host [n_types]; < 0: the type has no entry in `inputs` */
    const int64_t *num_samples;     /* host [n_types * n_hops]; < 0: the type has no entry in `num_samples` */
    int64_t tr_lo, tr_hi;
} tg_hgt_problem;

typedef struct {
    int64_t *const *samples;    /* host [n_types] device buffers, capacity max(n_inputs,0) + sum of num_samples */
    int64_t *const *sample_ts;  /* host [n_types], same capacity */
    int64_t *const *rows;       /* host [n_rels] device buffers, capacity 50 * capacity(samples[dst]) */
    int64_t *const *cols;       /* host [n_rels] */
    int64_t *const *edge_index; /* host [n_rels] */
    int64_t *n_samples;         /* device [n_types] */
    int64_t *n_edges;           /* device [n_rels] */
    int32_t *panic;             /* device [1]: 1 where the reference would panic */
} tg_hgt_out;

TG_API int tg_hgt_workspace_bytes(const tg_hgt_problem *problem, int64_t *bytes);
TG_API int tg_hgt_sample(const tg_hgt_problem *problem, const tg_rng *rng, const tg_hgt_out *out, void *workspace,
                         int64_t workspace_bytes, void *stream);

/* budget_sampling (src/algo/budget_sampling.rs:63-265; binding python.rs:486-581), one (layer, node type) at a
 * time: Budget::update + Budget::sample for every frontier node of the type (SURVEY.md 8(f) "next" row).  The host
 * appends the selected candidates in (node, slot) order.  Selected slot (j, s) lives at [j * fanout + s];
 * sel_rel < 0 marks an empty slot. */
typedef struct {
    const tg_graph *graphs;  /* host [n_rels]: CSC of the relations INTO the node type (timestamps optional) */
    const int32_t *rel_ids;  /* host [n_rels]: their index in the caller's relation list */
    int32_t n_rels;
    int32_t node_type;       /* index of the type being expanded (rng tag) */
    int32_t fanout;          /* num_neighbors[type][layer], <= 64 */
    int32_t filter_on, forward, relative;
    int64_t win_lo, win_hi;  /* half-open window */
    const int64_t *nodes;    /* [n_front] frontier nodes of the type */
    const int64_t *nodes_ts; /* [n_front] */
    int64_t n_front;
    int64_t id_base;         /* slot of nodes[0] in the type's sample list (draw id = id_base + j) */
} tg_budget_layer_in;

typedef struct {
    int64_t *sel_v, *sel_ts, *sel_rel, *sel_i; /* each [n_front * fanout] */
} tg_budget_layer_out;

TG_API int tg_budget_layer(const tg_budget_layer_in *in, const tg_rng *rng, const tg_budget_layer_out *out, void *stream);

/* budget_sampling as ONE stream-ordered call without host bookkeeping: sample lists, their lengths and the frontier
 * bounds live on the device (workspace); per (layer, node type) step the per-node selection is followed by a stable
 * multi-way partition that appends every chosen candidate to its source type's list and its relation's edge lists in
 * (node, slot) order (budget_sampling.rs:223-257).  Node types / relations are indexed in the caller's order.
 * Up to 8 node types, 16 relations, num_neighbors <= 64 (tg_budget_layer is the per-step form without these
 * limits on types / relations).  counts [n_types + n_rels] = list lengths.  Quirk kept: edge_index holds the
 * neighbour's index inside its column (budget_sampling.rs:116). */
typedef struct {
    int32_t n_types, n_rels, n_hops;
    int32_t filter_on, forward, relative;
    int64_t win_lo, win_hi;          /* half-open window (python.rs:541-548) */
    const int32_t *rel_src, *rel_dst; /* host [n_rels] */
    const tg_graph *graphs;           /* host [n_rels] CSC, timestamps optional */
    const int64_t *num_neighbors;     /* host [n_types * n_hops] */
    const int64_t *const *inputs;     /* host [n_types] device pointers (NULL where n_inputs[t] == 0) */
    const int64_t *const *input_ts;   /* host [n_types] device pointers or NULL (timestamp -1) */
    const int64_t *n_inputs;          /* host [n_types] */
} tg_budget_problem;

typedef struct {
    int64_t *const *samples;    /* [n_types] device slabs [cap_nodes[t]] */
    int64_t *const *sample_ts;
    const int64_t *cap_nodes;
    int64_t *const *rows;       /* [n_rels] device slabs [cap_edges[r]] */
    int64_t *const *cols;
    int64_t *const *edge_index;
    const int64_t *cap_edges;
    int64_t *counts;            /* device [n_types + n_rels] */
} tg_budget_out;

TG_API int tg_budget_capacity(const tg_budget_problem *problem, int64_t *cap_nodes, int64_t *cap_edges);
TG_API int tg_budget_workspace_bytes(const tg_budget_problem *problem, int64_t *bytes);
TG_API int tg_budget_sample(const tg_budget_problem *problem, const tg_rng *rng, const tg_budget_out *out, void *workspace,
                            int64_t workspace_bytes, void *stream);

/* ---- synthetic inputs of the measurement harness ---- */

/* R-MAT edge list: n_edges edges over 2^scale vertices, (a,b,c,d) =
 * (0.57,0.19,0.19,0.05), one Philox word per bit; duplicates and self loops
 * kept.  row, col: [n_edges] device int64. */
TG_API int tg_rmat_edges(int32_t scale, int64_t n_edges, uint64_t seed, int64_t *row, int64_t *col, void *stream);
/* rectangular form for bipartite relations: 2^row_scale x 2^col_scale, row / column bits drawn to their own depths */
TG_API int tg_rmat_edges_rect(int32_t row_scale, int32_t col_scale, int64_t n_edges, uint64_t seed, int64_t *row,
                              int64_t *col, void *stream);

/* Seed batches: out[b*n_seeds + i] = Philox(seed; first_batch + b, i) mod n_nodes. */
TG_API int tg_seed_batches(uint64_t seed, int64_t first_batch, int64_t n_batches, int64_t n_seeds, int64_t n_nodes,
                    int64_t *out, void *stream);

/* to_csc / to_csr (src/data/storage.rs:103-126; bindings python.rs:27-53): COO (row, col) of a size0 x size1 graph
 * -> ptrs [size1 + 1 | size0 + 1], indices [nnz], perm [nnz] (stable sort by the reference's key
 * col*size0+row | row*size1+col).  Negative ids are not checked. */
TG_API int tg_coo_to_csx_workspace_bytes(int64_t nnz, int64_t size0, int64_t size1, int64_t *bytes);
TG_API int tg_coo_to_csx(const int64_t *row, const int64_t *col, int64_t nnz, int64_t size0, int64_t size1, int32_t csc,
                         int64_t *ptrs, int64_t *indices, int64_t *perm, void *workspace, int64_t workspace_bytes,
                         void *stream);

/* Input validation for hosts: sets flag[0] |= 1 (device int32, zeroed by the caller) when any of values[0..n) lies
 * outside [lo, hi).  The samplers index `ptrs` with their inputs and do NOT check them (the reference panics on an
 * out-of-range node id; a device kernel would fault). */
TG_API int tg_check_range(const int64_t *values, int64_t n, int64_t lo, int64_t hi, int32_t *flag, void *stream);
/* The same check without a read-back before the sampler runs: out[i] = values[i] if it lies in [lo, hi) else lo (a valid
 * id, so the sampler cannot fault), flag[0] |= 1 (device int64 word, zeroed by the caller -- e.g. the last word of the
 * array the host reads back anyway when the call ends) if any did not.  The host raises after that one read-back. */
TG_API int tg_sanitize_range(const int64_t *values, int64_t n, int64_t lo, int64_t hi, int64_t *out, int64_t *flag,
                             void *stream);

/* ind2ptr (src/data/storage.rs:67-101) on the device: sorted `ind` [numel] -> out [m+1]. */
TG_API int tg_ind2ptr(const int64_t *ind, int64_t numel, int64_t m, int64_t *out, void *stream);

/* Mini-batch materialisation, the step AFTER the path: dst[i, :] = src[index[i], :] for
 * i < n, rows of `row_bytes` bytes, source rows `src_stride_bytes` apart (>= row_bytes), dst rows packed.  This is the
 * `x[samples]` / `edge_attr[perm[edge_index]]` gather the reference's examples leave to PyG's filter_data
 * (examples/neighbor_sampling.py:24,36,48; examples/neighbor_sampling_typed.py:27).  dtype-agnostic (bytes); uses
 * 16-byte vectors when pointers, stride and row length are 16-byte multiples.  An index outside [0, n_src_rows)
 * writes a zero row and sets status[0] |= 1 (device int32, zeroed by the caller; may be NULL). */
TG_API int tg_gather_rows(const void *src, int64_t n_src_rows, int64_t row_bytes, int64_t src_stride_bytes,
                          const int64_t *index, int64_t n, void *dst, int32_t *status, void *stream);

/* Per-batch slabs of tg_ns_homo_batched -> flat batch-major arrays: batch b's samples to flat_samples[node_off[b] ..],
 * its rows / cols / edge pointers to flat_*[edge_off[b] ..] (offsets: device arrays, exclusive prefixes of the batch
 * counts).  What a loader hands on to tg_gather_rows (tch_geometric/loader.py). */
TG_API int tg_ns_homo_compact(const tg_ns_out *out, int64_t n_batches, const int64_t *node_off, const int64_t *edge_off,
                              int64_t *flat_samples, int64_t *flat_rows, int64_t *flat_cols, int64_t *flat_edge_index,
                              void *stream);

/* Ragged rows of an int64 slab -> one flat array: dst[offsets[r] + i] = src[r * pitch + i] for
 * i < lens[r * lens_stride] (lens, offsets: device arrays).  The per-type / per-relation slabs of
 * tg_ns_hetero_batched are flattened with it (tch_geometric/loader.py). */
TG_API int tg_compact_rows(const int64_t *src, int64_t pitch, const int64_t *lens, int64_t lens_stride, const int64_t *offsets,
                           int64_t n_rows, int64_t *dst, void *stream);

/* Harness calibration (not part of the sampling path): n_threads lanes each issue per_thread
 * independent random 8-byte loads from table[0..n_table); sink: [n_threads]. */
TG_API int tg_probe_random_gather(const int64_t *table, int64_t n_table, int64_t n_threads, int64_t per_thread,
                                  uint64_t seed, int64_t *sink, void *stream);

/* Debug build only (`make dbg`, -DTG_DEBUG_BOUNDS -> lib/libtchgeo_hip_dbg.so): registers a device word; the multi-hop
 * neighbor-sampling kernels then compare every frontier id with n_major before it indexes `ptrs`, raise bit 0 of the word
 * for an offender and sample vertex 0 instead of reading out of bounds.  For experiments that drop or alter a hop's work.
 * The regular build trusts the ids (contract: ids in `indices` are < n_major) and returns TG_ERR_UNSUPPORTED here. */
TG_API int tg_debug_bounds_set_flag(uint32_t *device_word);

/* Harness calibration: the speed of light of neighbor_sampling_homogenous's output contract.  Moves the ALGORITHMIC bytes
 * of a finished launch `src` (per seed 8 B read + 8 B written, per expanded frontier slot 24 B read, per sampled edge 8 B
 * read + 32 B written) as pure streams into the slabs `dst` (same pitch, other memory) -- precomputed contents,
 * no draws, no random access, no ordering.  sink: [n_batches]. */
TG_API int tg_probe_ns_sol(const tg_ns_out *src, const tg_ns_out *dst, const int64_t *seeds, int64_t n_batches,
                           int64_t n_seeds, int32_t n_hops, int64_t *sink, void *stream);

#ifdef __cplusplus
}
#endif
#endif
