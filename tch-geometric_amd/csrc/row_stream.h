// Streaming a CSR row through one wavefront for the temporal walks: (neighbour id, effective timestamp) of every
// edge, 64 edges per call of the visitor, in edge order.  Effective timestamp = the edge's, or the neighbour's node
// timestamp when the edge has none (random_walk.rs:121-125 / :231-237).
//
// Written for the memory pipeline: P chunks (P x 64 edges, two 8-byte loads per lane and chunk) are issued together
// and the next P are in flight while the current ones are visited; every load is unconditional -- lanes past the
// row's end re-read its first edge and are reported invalid -- because a branch around a load makes the compiler
// wait for all loads in flight.  Only the node-timestamp fallback is a conditional gather (taken when some edge of
// the chunk has no timestamp).
#pragma once
#include "tg_device.h"

namespace tg {

template <int P, typename Visit>
__device__ __forceinline__ void stream_row(const int64_t *__restrict__ indices, const int64_t *__restrict__ edge_ts,
                                           const int64_t *__restrict__ node_ts, int64_t b, int64_t e, int lane,
                                           Visit &&visit) {
    if (b >= e) return;
    struct Round {
        int64_t v[P], ts[P];
    };
    auto issue = [&](Round &r, int64_t base) {
#pragma unroll
        for (int u = 0; u < P; ++u) {
            const int64_t ee = base + u * 64 + lane;
            const int64_t c = ee < e ? ee : b;
            r.v[u] = indices[c];
            r.ts[u] = edge_ts[c];
        }
    };
    auto consume = [&](const Round &r, int64_t base) {
#pragma unroll
        for (int u = 0; u < P; ++u) {
            if (base + u * 64 >= e) break; // uniform
            const int64_t ee = base + u * 64 + lane;
            const bool valid = ee < e;
            int64_t ts = r.ts[u];
            const bool need = valid && ts == -1;
            if (__ballot(need) != 0ull) {
                if (need) ts = node_ts[r.v[u]];
            }
            visit(ee, valid, r.v[u], ts);
        }
    };
    constexpr int64_t R = (int64_t)64 * P;
    Round ra, rb;
    issue(ra, b);
    for (int64_t base = b; base < e; base += 2 * R) {
        issue(rb, base + R);
        consume(ra, base);
        issue(ra, base + 2 * R);
        consume(rb, base + R);
    }
}

// The same walk over a row WITHOUT its neighbour ids: the walks only need the id of the ONE edge they pick (read afterwards
// from its position) -- and of an edge without a timestamp, for the node-timestamp fallback -- so a pass moves 8 bytes per
// inspected edge instead of 16.  visit(edge position, valid, effective timestamp).
template <int P, typename Visit>
__device__ __forceinline__ void stream_row_ts(const int64_t *__restrict__ indices, const int64_t *__restrict__ edge_ts,
                                              const int64_t *__restrict__ node_ts, int64_t b, int64_t e, int lane,
                                              Visit &&visit) {
    if (b >= e) return;
    struct Round {
        int64_t ts[P];
    };
    auto issue = [&](Round &r, int64_t base) {
#pragma unroll
        for (int u = 0; u < P; ++u) {
            const int64_t ee = base + u * 64 + lane;
            r.ts[u] = edge_ts[ee < e ? ee : b];
        }
    };
    auto consume = [&](const Round &r, int64_t base) {
#pragma unroll
        for (int u = 0; u < P; ++u) {
            if (base + u * 64 >= e) break; // uniform
            const int64_t ee = base + u * 64 + lane;
            const bool valid = ee < e;
            int64_t ts = r.ts[u];
            const bool need = valid && ts == -1;
            if (__ballot(need) != 0ull) {
                if (need) ts = node_ts[indices[ee]];
            }
            visit(ee, valid, ts);
        }
    };
    constexpr int64_t R = (int64_t)64 * P;
    Round ra, rb;
    issue(ra, b);
    for (int64_t base = b; base < e; base += 2 * R) {
        issue(rb, base + R);
        consume(ra, base);
        issue(ra, base + 2 * R);
        consume(rb, base + R);
    }
}

} // namespace tg
