// negative_sample_neighbors_{homogenous,heterogenous} on gfx950 -- replaces
// src/algo/negative_sampling.rs:6-131 (reference).
//
// The reference walks items (input i, negative jn) sequentially: up to
// try_count uniform candidates w, the first with !has_edge(v,w) && v != w
// wins, w gets a local id from a HashMap in FIRST-SEEN order, and the edge
// (i, id) is pushed.  With counter-addressed draws every item is independent,
// so the search (the expensive part: one binary search per try) runs one lane
// per item across the whole chip.  The sequential parts are then restated as
// order-preserving parallel primitives:
//   * local ids: a device hash map keyed by node id that keeps, per node, the
//     MIN item position that produced it; an item is the node's first sight
//     iff it holds that minimum, and the rank of that item among first
//     sights (a prefix sum in item order) is the node's slot after the inputs;
//   * inputs map to the LAST input slot holding that value (HashMap::extend
//     overwrites, negative_sampling.rs:26) -> atomicMax on the slot index;
//   * edges: a prefix sum over accepted items in item order.
// Heterogeneous graphs reuse the same kernels with a (dst type / relation)
// mask per pass; item order is node-type-major in `node_types` order.
#include <vector>

#include <cstring>

#include <rocprim/device/device_scan.hpp>

#include "tg_device.h"
#include "tg_host.h"
#include "tg_map.h"
#include "tg_scan.h"

namespace tg {

constexpr int NEG_MAX_RELS = 32;

struct NegRel {
    const int64_t *ptrs;
    const int64_t *indices;
    int64_t node_count; // sizes[r].1
    int64_t row_count;  // rows of the CSR (bound for the inbound lookup)
    int32_t dst_type;
    int32_t _pad;
};
struct NegRelTable {
    NegRel r[NEG_MAX_RELS];
};
struct NegSrcRels { // relations whose source type is the type being processed, in edge_types order
    int32_t n;
    int32_t rel[NEG_MAX_RELS];
};

__device__ __forceinline__ bool neg_has_edge(const int64_t *__restrict__ ptrs, const int64_t *__restrict__ indices,
                                             int64_t x, int64_t y) { // graph.rs:80-83
    int64_t lo = ptrs[x], hi = ptrs[x + 1];
    while (lo < hi) {
        const int64_t mid = lo + ((hi - lo) >> 1);
        const int64_t v = indices[mid];
        if (v == y) return true;
        if (v < y)
            lo = mid + 1;
        else
            hi = mid;
    }
    return false;
}

// one lane per item (i, jn) of one source type.  cand[p] = accepted node or -1; rel_of[p] = relation.
__global__ void neg_candidates_kernel(NegRelTable tab, NegSrcRels src, const int64_t *__restrict__ inputs,
                                      int64_t n_inputs, int64_t num_neg, int64_t try_count, int inbound,
                                      int hetero, uint64_t seed, uint64_t call_id, uint32_t tag, int64_t item_base,
                                      int64_t *cand, int32_t *rel_of, int *panic) {
    const int64_t m = n_inputs * num_neg;
    const CallKey ck = call_key(seed, call_id, tag);
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < m; q += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = q / num_neg, jn = q - i * num_neg;
        const int64_t v = inputs[i];
        int r = src.rel[0];
        if (hetero) { // negative_sampling.rs:104
            const Draw d = draw(ck, (uint64_t)i, (uint32_t)jn, 0xFFFFFFFFu);
            r = src.rel[bounded64(d.a(), (uint64_t)src.n)];
        }
        const NegRel R = tab.r[r];
        int64_t found = -1;
        for (int64_t t = 0; t < try_count; ++t) { // :33 / :110
            const Draw d = draw(ck, (uint64_t)i, (uint32_t)jn, (uint32_t)t);
            const int64_t w = (int64_t)bounded64(d.a(), (uint64_t)R.node_count);
            bool he;
            if (inbound) { // :113 has_edge(w, v) on the src->dst CSR: the reference panics when w is not a row
                if (w >= R.row_count) {
                    *panic = 1;
                    break;
                }
                he = neg_has_edge(R.ptrs, R.indices, w, v);
            } else {
                he = neg_has_edge(R.ptrs, R.indices, v, w);
            }
            if (!he && v != w) { // :35 / :117
                found = w;
                break;
            }
        }
        cand[item_base + q] = found;
        rel_of[item_base + q] = r;
    }
}

// inputs of the dst type: value -> LAST slot holding it; also copies them to the head of `samples`
__global__ void neg_insert_inputs_kernel(const int64_t *__restrict__ inputs, int64_t n, int64_t *keys, int64_t *vals,
                                         int64_t mask, int64_t *samples) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t v = inputs[i];
        samples[i] = v;
        const int64_t s = map_slot_insert(keys, mask, v);
        atomicMax(reinterpret_cast<long long *>(&vals[s]), (long long)i);
    }
}
// items that target `dst_type`: known input -> its slot; otherwise remember the smallest item position per node
__global__ void neg_insert_items_kernel(NegRelTable tab, int dst_type, const int64_t *__restrict__ cand,
                                        const int32_t *__restrict__ rel_of, int64_t m, const int64_t *in_keys,
                                        const int64_t *in_vals, int64_t in_mask, int64_t *new_keys, int64_t *new_vals,
                                        int64_t new_mask, int64_t *ids) {
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < m; p += (int64_t)gridDim.x * blockDim.x) {
        const int64_t w = cand[p];
        if (w < 0 || tab.r[rel_of[p]].dst_type != dst_type) continue;
        const int64_t s = map_slot_find(in_keys, in_mask, w);
        if (s >= 0) {
            ids[p] = in_vals[s];
        } else {
            ids[p] = -1; // resolved after the ranks are known
            const int64_t t = map_slot_insert(new_keys, new_mask, w);
            atomicMin(reinterpret_cast<long long *>(&new_vals[t]), (long long)p);
        }
    }
}

// Single-workgroup exclusive scan over item flags, in item order.
//  mode 0: flag = item targets dst_type, is not an input, and holds its node's minimum position (first sight)
//  mode 1: flag = item accepted and of relation `sel`
// rank[p] is written for flagged items; total[0] receives the flag count.
__device__ __forceinline__ int64_t neg_flag_of(const NegRelTable &tab, int mode, int sel, const int64_t *__restrict__ cand,
                                               const int32_t *__restrict__ rel_of, const int64_t *__restrict__ ids,
                                               const int64_t *new_keys, const int64_t *new_vals, int64_t new_mask, int64_t p) {
    int64_t f = 0;
    const int64_t w = cand[p];
    if (w >= 0) {
        if (mode == 1) {
            f = rel_of[p] == sel;
        } else if (tab.r[rel_of[p]].dst_type == sel && ids[p] < 0) {
            const int64_t t = map_slot_find(new_keys, new_mask, w);
            f = (t >= 0 && new_vals[t] == p);
        }
    }
    return f;
}
__global__ void neg_flag_kernel(NegRelTable tab, int mode, int sel, const int64_t *__restrict__ cand,
                                const int32_t *__restrict__ rel_of, const int64_t *__restrict__ ids, int64_t begin,
                                int64_t end, const int64_t *new_keys, const int64_t *new_vals, int64_t new_mask,
                                int64_t *flag) {
    for (int64_t p = begin + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < end; p += (int64_t)gridDim.x * blockDim.x)
        flag[p] = neg_flag_of(tab, mode, sel, cand, rel_of, ids, new_keys, new_vals, new_mask, p);
}
// the same for a short item range in ONE launch of one workgroup: flags, their ranks (rank[begin .. end], one word past
// the range is written too) and the total, also stored where the caller reports it
__global__ void __launch_bounds__(SCAN1_THREADS)
    neg_rank1_kernel(NegRelTable tab, int mode, int sel, const int64_t *__restrict__ cand, const int32_t *__restrict__ rel_of,
                     const int64_t *__restrict__ ids, int64_t begin, int64_t end, const int64_t *new_keys,
                     const int64_t *new_vals, int64_t new_mask, int64_t *rank, int64_t *total, int64_t total_add,
                     int64_t *report) {
    block_scan_exclusive_plus1(
        end - begin,
        [&](int64_t i) { return neg_flag_of(tab, mode, sel, cand, rel_of, ids, new_keys, new_vals, new_mask, begin + i); },
        rank + begin);
    __syncthreads();
    if (threadIdx.x == 0) {
        const int64_t t = rank[end] + total_add;
        total[0] = t;
        report[0] = t;
    }
}
// total[0] = number of flagged items in [begin, end) + total_add, from the exclusive scan `rank` of `flag`
__global__ void neg_total_kernel(const int64_t *__restrict__ flag, const int64_t *__restrict__ rank, int64_t begin,
                                 int64_t end, int64_t *total, int64_t total_add) {
    total[0] = (end > begin ? rank[end - 1] + flag[end - 1] : 0) + total_add;
}

// after mode-0 ranks: first sights append their node to `samples`; every non-input item learns its id
__global__ void neg_assign_ids_kernel(NegRelTable tab, int dst_type, const int64_t *__restrict__ cand,
                                      const int32_t *__restrict__ rel_of, int64_t m, int64_t n_inputs,
                                      const int64_t *new_keys, const int64_t *new_vals, int64_t new_mask,
                                      const int64_t *__restrict__ rank, int64_t *ids, int64_t *samples) {
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < m; p += (int64_t)gridDim.x * blockDim.x) {
        const int64_t w = cand[p];
        if (w < 0 || tab.r[rel_of[p]].dst_type != dst_type || ids[p] >= 0) continue;
        const int64_t t = map_slot_find(new_keys, new_mask, w);
        const int64_t first = new_vals[t];
        const int64_t id = n_inputs + rank[first];
        if (first == p) samples[id] = w; // :37-38
        ids[p] = id;
    }
}
// NOTE: ids[p] is overwritten with a non-negative id here; the `ids[p] >= 0` early-out above only skips items
// that resolved to an input slot in neg_insert_items_kernel (their id was set there).

__global__ void neg_emit_edges_kernel(int rel, const int64_t *__restrict__ cand, const int32_t *__restrict__ rel_of,
                                      const int64_t *__restrict__ ids, const int64_t *__restrict__ erank,
                                      int64_t begin, int64_t end, int64_t num_neg, int64_t *rows, int64_t *cols) {
    for (int64_t p = begin + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < end;
         p += (int64_t)gridDim.x * blockDim.x) {
        if (cand[p] < 0 || rel_of[p] != rel) continue;
        const int64_t e = erank[p];
        rows[e] = (p - begin) / num_neg; // :40 / :122 i
        cols[e] = ids[p];                //            j
    }
}

} // namespace tg

extern "C" int tg_neg_workspace_bytes(const tg_neg_problem *pb, int64_t *bytes) {
    TG_REQUIRE(pb && bytes, "tg_neg_workspace_bytes: null argument");
    int64_t m = 0, max_in = 0;
    for (int t = 0; t < pb->n_types; ++t) {
        const int64_t n = pb->n_inputs[t] > 0 ? pb->n_inputs[t] : 0;
        m += n * pb->num_neg;
        if (n > max_in) max_in = n;
    }
    const int64_t in_cap = tg::pow2_at_least(2 * max_in + 2), new_cap = tg::pow2_at_least(2 * m + 2);
    // cand, ids, rank, erank, flag: m i64 each; rel_of: m i32 (padded); maps: 2*(in_cap + new_cap) i64; totals; panic
    // flag; temporary storage of the device-wide scans
    size_t scan_temp = 0;
    hipError_t e = rocprim::exclusive_scan(nullptr, scan_temp, (int64_t *)nullptr, (int64_t *)nullptr, (int64_t)0,
                                           (size_t)(m > 0 ? m : 1), rocprim::plus<int64_t>(), (hipStream_t)0, false);
    if (e != hipSuccess) return tg::fail(TG_ERR_HIP, "rocprim::exclusive_scan size query failed: %s", hipGetErrorString(e));
    *bytes = 8 * (5 * m + 2 + 2 * in_cap + 2 * new_cap + 8) + ((4 * m + 15) & ~(int64_t)15) + 64 + (int64_t)((scan_temp + 15) & ~(size_t)15) + 512;
    return TG_OK;
}

extern "C" int tg_neg_sample(const tg_neg_problem *pb, const tg_rng *rng, const tg_neg_out *out, void *workspace,
                             void *stream_) {
    using namespace tg;
    TG_REQUIRE(pb && rng && out && workspace, "tg_neg_sample: null argument");
    TG_REQUIRE(pb->n_types >= 1 && pb->n_rels >= 1 && pb->n_rels <= NEG_MAX_RELS,
               "tg_neg_sample: %d relations outside [1, %d]", pb->n_rels, NEG_MAX_RELS);
    TG_REQUIRE(pb->num_neg >= 0 && pb->try_count >= 0, "tg_neg_sample: negative counts");
    hipStream_t stream = (hipStream_t)stream_;
    NegRelTable tab;
    for (int r = 0; r < pb->n_rels; ++r) {
        TG_REQUIRE(pb->graphs[r].ptrs, "tg_neg_sample: relation %d has no CSR", r);
        TG_REQUIRE(pb->node_count[r] >= 1, "tg_neg_sample: relation %d has an empty node range", r);
        tab.r[r] = NegRel{pb->graphs[r].ptrs, pb->graphs[r].indices, pb->node_count[r], pb->graphs[r].n_major,
                          pb->rel_dst[r], 0};
    }
    int64_t m = 0, max_in = 0;
    for (int t = 0; t < pb->n_types; ++t) {
        const int64_t n = pb->n_inputs[t] > 0 ? pb->n_inputs[t] : 0;
        m += n * pb->num_neg;
        if (n > max_in) max_in = n;
    }
    const int64_t in_cap = pow2_at_least(2 * max_in + 2), new_cap = pow2_at_least(2 * m + 2);
    int64_t *ws = reinterpret_cast<int64_t *>(workspace);
    int64_t *cand = ws, *ids = cand + m, *rank = ids + m, *erank = rank + m + 1; // one word of slack behind either rank array
    int64_t *in_keys = erank + m + 1, *in_vals = in_keys + in_cap, *new_keys = in_vals + in_cap, *new_vals = new_keys + new_cap;
    int64_t *totals = new_vals + new_cap; // [0] scratch total
    int *panic = reinterpret_cast<int *>(totals + 4);
    int32_t *rel_of = reinterpret_cast<int32_t *>(totals + 8);
    int64_t *flag = reinterpret_cast<int64_t *>(reinterpret_cast<unsigned char *>(rel_of) + ((4 * m + 15) & ~(int64_t)15));
    void *scan_temp = reinterpret_cast<void *>(((uintptr_t)(flag + m) + 255) & ~(uintptr_t)255);
    size_t scan_temp_bytes = 0;
    TG_HIP(rocprim::exclusive_scan(nullptr, scan_temp_bytes, (int64_t *)nullptr, (int64_t *)nullptr, (int64_t)0,
                                   (size_t)(m > 0 ? m : 1), rocprim::plus<int64_t>(), stream, false));
    // ranks of the flagged items of [b, e) (device-wide: flag kernel + rocPRIM scan), their count (+ add) into total
    auto ranks_of = [&](int mode, int sel, int64_t b, int64_t e, int64_t *rank_out, int64_t *total, int64_t add,
                        int64_t *report) -> int {
        if (e - b <= 16384) { // one workgroup beats flag kernel + the library's scan + the total up to about here
            hipLaunchKernelGGL(neg_rank1_kernel, dim3(1), dim3(SCAN1_THREADS), 0, stream, tab, mode, sel, cand, rel_of, ids, b, e,
                               new_keys, new_vals, new_cap - 1, rank_out, total, add, report);
            return TG_OK;
        }
        hipLaunchKernelGGL(neg_flag_kernel, dim3(grid_1d(e - b)), dim3(256), 0, stream, tab, mode, sel, cand, rel_of, ids, b,
                           e, new_keys, new_vals, new_cap - 1, flag);
        size_t stb = scan_temp_bytes;
        TG_HIP(rocprim::exclusive_scan(scan_temp, stb, flag + b, rank_out + b, (int64_t)0, (size_t)(e - b),
                                       rocprim::plus<int64_t>(), stream, false));
        hipLaunchKernelGGL(neg_total_kernel, dim3(1), dim3(1), 0, stream, flag, rank_out, b, e, total, add);
        TG_HIP(hipMemcpyAsync(report, total, sizeof(int64_t), hipMemcpyDeviceToDevice, stream));
        return TG_OK;
    };
    TG_HIP(hipMemsetAsync(panic, 0, sizeof(int), stream));

    // ---- 1. candidates, one source type at a time (item order = node_types order, then i, then jn)
    std::vector<int64_t> item_begin((size_t)pb->n_types + 1, 0);
    for (int t = 0; t < pb->n_types; ++t) {
        const int64_t n = pb->n_inputs[t] > 0 ? pb->n_inputs[t] : 0;
        item_begin[t + 1] = item_begin[t] + n * pb->num_neg;
        if (n * pb->num_neg == 0) continue;
        NegSrcRels src;
        src.n = 0;
        for (int r = 0; r < pb->n_rels; ++r)
            if (pb->rel_src[r] == t) src.rel[src.n++] = r; // :65-71
        TG_REQUIRE(src.n > 0, "tg_neg_sample: node type %d has inputs but no outgoing relation (the reference panics)", t);
        const uint32_t tag = pb->homogeneous ? TAG_NEG_HOMO : (TAG_NEG_HETERO | ((uint32_t)t << 8));
        hipLaunchKernelGGL(neg_candidates_kernel, dim3(grid_1d(n * pb->num_neg)), dim3(256), 0, stream, tab, src,
                           pb->inputs[t], n, pb->num_neg, pb->try_count, pb->inbound, pb->homogeneous ? 0 : 1,
                           rng->seed, rng->call_id, tag, item_begin[t], cand, rel_of, panic);
        TG_LAUNCH_CHECK();
    }
    // ---- 2. local ids per destination type
    for (int dt = 0; dt < pb->n_types; ++dt) {
        const int64_t n_in = pb->n_inputs[dt] > 0 ? pb->n_inputs[dt] : 0;
        hipLaunchKernelGGL(fill_i64_kernel, dim3(grid_1d(in_cap)), dim3(256), 0, stream, in_keys, in_cap, MAP_EMPTY);
        hipLaunchKernelGGL(fill_i64_kernel, dim3(grid_1d(in_cap)), dim3(256), 0, stream, in_vals, in_cap, (int64_t)-1);
        hipLaunchKernelGGL(fill_i64_kernel, dim3(grid_1d(new_cap)), dim3(256), 0, stream, new_keys, new_cap, MAP_EMPTY);
        hipLaunchKernelGGL(fill_i64_kernel, dim3(grid_1d(new_cap)), dim3(256), 0, stream, new_vals, new_cap,
                           (int64_t)INT64_MAX);
        if (n_in > 0)
            hipLaunchKernelGGL(neg_insert_inputs_kernel, dim3(grid_1d(n_in)), dim3(256), 0, stream, pb->inputs[dt],
                               n_in, in_keys, in_vals, in_cap - 1, out->samples[dt]);
        if (m > 0) {
            hipLaunchKernelGGL(neg_insert_items_kernel, dim3(grid_1d(m)), dim3(256), 0, stream, tab, dt, cand, rel_of,
                               m, in_keys, in_vals, in_cap - 1, new_keys, new_vals, new_cap - 1, ids);
            if (int rcs = ranks_of(0, dt, 0, m, rank, totals, n_in, out->n_samples + dt)) return rcs;
            hipLaunchKernelGGL(neg_assign_ids_kernel, dim3(grid_1d(m)), dim3(256), 0, stream, tab, dt, cand, rel_of, m,
                               n_in, new_keys, new_vals, new_cap - 1, rank, ids, out->samples[dt]);
        } else {
            hipLaunchKernelGGL(fill_i64_kernel, dim3(1), dim3(64), 0, stream, out->n_samples + dt, (int64_t)1, n_in);
        }
        // n_samples[dt] = n_in + number of first sights (reported by ranks_of)
        TG_LAUNCH_CHECK();
    }
    // ---- 3. edges per relation, in item order
    for (int r = 0; r < pb->n_rels; ++r) {
        const int t = pb->rel_src[r];
        const int64_t b = item_begin[t], e = item_begin[t + 1];
        if (e > b) {
            if (int rcs = ranks_of(1, r, b, e, erank, totals + 1, 0, out->n_edges + r)) return rcs;
            hipLaunchKernelGGL(neg_emit_edges_kernel, dim3(grid_1d(e - b)), dim3(256), 0, stream, r, cand, rel_of, ids,
                               erank, b, e, pb->num_neg, out->rows[r], out->cols[r]);
        } else {
            TG_HIP(hipMemsetAsync(out->n_edges + r, 0, sizeof(int64_t), stream));
        }
        TG_LAUNCH_CHECK();
    }
    TG_HIP(hipMemcpyAsync(out->panic, panic, sizeof(int), hipMemcpyDeviceToDevice, stream));
    return TG_OK;
}
