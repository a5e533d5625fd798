// Staged form of the window-ordered launch (included by ns_homo_win.hip; same namespace, same parameter block).
//
// Where the push form (K1 emit -> sort -> K4 gather) spends its time, measured on MI355X (profiles/r03):
//   * K4 writes every frontier vertex's <= k gathered neighbours as an 80-byte run at an 8-byte-aligned place of the
//     `samples` slab: 43.8 M runs per 16 384-batch launch at 13.3 G runs/s = 3.3 ms -- exactly what a bare scatter of such
//     runs costs (tools/probe_permute.hip: a 64-byte chunk written partly costs 2.2x a whole one; HBM writes whole 64-byte
//     words).  Whole aligned 64-byte chunks scatter at 50 G/s.
//   * K1 looks the column bounds of 45 M frontier vertices up at random (hidden under its streams) and both kernels run
//     the ticket draws.
// Here the hop is turned round -- gather first, emit afterwards:
//   E0 / E(h-1) hand the next frontier over as 8-byte ITEMS (vertex, batch | index) while they hold the gathered vertex in
//          a register (no column look-up yet);
//   sort   counting sort of the 8-byte items by the WINDOW of their column start, found from the vertex id through a
//          vertex -> window table (the offsets are monotone), scans shared with the push form;
//   G      window-ordered: column bounds (now L2 hits: window order is vertex order), draws, gathers (L2 hits), and the
//          item's SLOT of the stage array -- {column start, degree, <= W-2 neighbours} = one or two whole 64-byte chunks at
//          the item's ORIGINAL index, i.e. a scatter of whole chunks;
//   E(h)   per batch, in slot order: reads its frontier's stage slots as one coalesced stream, counts / scans / draws and
//          writes ALL FOUR output streams coalesced (and the next hop's items).
// Outputs are the push form's and the fused kernel's bit for bit (neighbor_sampling.rs:188-223: positions are fixed by
// per-batch prefix sums, draws are addressed by (call, slot)).

struct WinItem8 {
    uint32_t v, bs; // vertex; batch << idx_bits | index inside the hop's frontier of that batch
};
static_assert(sizeof(WinItem8) == 8, "item layout");

// window of a vertex's column start from the vertex id alone: vtab[i] = first vertex whose column starts at or beyond
// window i (ptrs is monotone), so window(v) = largest i with vtab[i] <= v; then the XCD-major bucket as win_bucket
__device__ __forceinline__ uint32_t win_vertex_bucket(const uint32_t *vtab, int n_windows, int n_buckets, uint32_t v) {
    int lo = 0, hi = n_windows;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (vtab[mid] <= v)
            lo = mid;
        else
            hi = mid;
    }
    return ((uint32_t)lo & 7u) * (uint32_t)(n_buckets >> 3) + ((uint32_t)lo >> 3);
}

__global__ void win_vtab_kernel(const WinParams p, int64_t n_major) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.n_windows) return;
    const uint64_t target = (uint64_t)i << p.shift;
    int64_t lo = 0, hi = n_major; // first v in [0, n_major] with ptrs[v] >= target (ptrs[n_major] = n_edges >= target)
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        const uint64_t e = p.ptrs32 ? (uint64_t)p.ptrs32[mid] : (uint64_t)p.ptrs[mid];
        if (e >= target)
            hi = mid;
        else
            lo = mid + 1;
    }
    p.vtab[i] = (uint32_t)lo;
}

// ---------------------------------------------------------------- E0: seeds, hop 0 (direct) + the items of hop 1
template <int KMAX, bool REPLACE>
__global__ void win_stage_first_kernel(const WinParams p, const int k0) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int64_t b = p.b0 + blockIdx.x;
    int64_t *samples = p.samples + b * p.cap_nodes;
    for (int64_t i = threadIdx.x; i < p.n_seeds; i += blockDim.x) samples[i] = p.seeds[b * p.n_seeds + i]; // :184
    const CallKey ck = call_key(p.seed, p.call_id + (uint64_t)b, p.tag);
    if (threadIdx.x == 0) p.call_keys[b] = ck;
    __syncthreads();
    const WinState st =
        win_emit_hop<WinItemN, KMAX, REPLACE, true, false, true>(p, smem, b, 0, k0, WinState{0, p.n_seeds, 0, 0}, ck);
    win_store_state(p, b, 0, st);
}

// ---------------------------------------------------------------- sort: window histogram of the 8-byte items
// workgroup r counts the items of batches r, r + gridDim.x, ... (the batches win_scatter8_kernel's workgroup r moves)
__global__ void __launch_bounds__(WIN_PART_THREADS) win_hist8_kernel(const WinParams p) {
    extern __shared__ __align__(16) unsigned char smem[];
    uint32_t *h = reinterpret_cast<uint32_t *>(smem);
    uint32_t *lvtab = h + p.n_buckets;
    const int nb = p.n_buckets;
    const WinItem8 *items = static_cast<const WinItem8 *>(p.items_in);
    for (int i = threadIdx.x; i < nb; i += blockDim.x) h[i] = 0;
    for (int i = threadIdx.x; i < p.n_windows; i += blockDim.x) lvtab[i] = p.vtab[i];
    __syncthreads();
    constexpr int U = 4;
    for (int64_t b = p.b0 + blockIdx.x; b < p.b0 + p.n_batches; b += gridDim.x) {
        const WinState st = p.state[b];
        const int64_t n = st.end - st.begin;
        const WinItem8 *src = items + b * p.item_pitch;
        for (int64_t t0 = 0; t0 < n; t0 += (int64_t)U * WIN_PART_THREADS) {
            uint32_t v[U];
            bool ok[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t j = t0 + (int64_t)u * WIN_PART_THREADS + threadIdx.x;
                ok[u] = j < n;
                if (ok[u]) v[u] = src[j].v;
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (ok[u]) atomicAdd(&h[win_vertex_bucket(lvtab, p.n_windows, nb, v[u])], 1u);
        }
    }
    __syncthreads();
    uint32_t *row = p.hist + (size_t)blockIdx.x * nb;
    for (int i = threadIdx.x; i < nb; i += blockDim.x) row[i] = h[i];
}

// ---------------------------------------------------------------- sort: scatter of the 8-byte items (rows = emit workgroups)
__global__ void __launch_bounds__(WIN_PART_THREADS) win_scatter8_kernel(const WinParams p) {
    extern __shared__ __align__(16) unsigned char smem[];
    uint32_t *cur = reinterpret_cast<uint32_t *>(smem);
    uint32_t *lvtab = cur + p.n_buckets;
    const int nb = p.n_buckets;
    const WinItem8 *items = static_cast<const WinItem8 *>(p.items_in);
    WinItem8 *sorted = static_cast<WinItem8 *>(p.items_sorted);
    const uint32_t *row = p.hist + (size_t)blockIdx.x * nb;
    for (int i = threadIdx.x; i < nb; i += blockDim.x) cur[i] = p.base[i] + row[i];
    for (int i = threadIdx.x; i < p.n_windows; i += blockDim.x) lvtab[i] = p.vtab[i];
    __syncthreads();
    constexpr int U = 4;
    for (int64_t b = p.b0 + blockIdx.x; b < p.b0 + p.n_batches; b += gridDim.x) {
        const WinState st = p.state[b];
        const int64_t n = st.end - st.begin; // the frontier of the hop about to be gathered
        const WinItem8 *src = items + b * p.item_pitch;
        for (int64_t t0 = 0; t0 < n; t0 += (int64_t)U * WIN_PART_THREADS) {
            WinItem8 it[U];
            bool ok[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t j = t0 + (int64_t)u * WIN_PART_THREADS + threadIdx.x;
                ok[u] = j < n;
                if (ok[u]) it[u] = src[j];
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (ok[u]) sorted[atomicAdd(&cur[win_vertex_bucket(lvtab, p.n_windows, nb, it[u].v)], 1u)] = it[u];
        }
    }
}

// the ticket sampler with its positions left in registers (G: lane = item, nothing is staged)
template <int KMAX>
__device__ __forceinline__ void sample_tickets_reg(CallKey ck, uint64_t id, uint32_t n, int k, uint32_t (&pos)[KMAX]) {
    uint32_t keys[KMAX], vals[KMAX];
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
            pos[s] = (tr < n - (uint32_t)k) ? (uint32_t)k + tr : (uint32_t)s;
        }
    }
}

// ---------------------------------------------------------------- G: window-ordered gather into the stage slots
// W = words per stage slot (16: one 64-byte chunk, fan-outs <= 14; 32: two chunks, fan-outs <= 30); KMAX = W - 2.
template <int W, bool REPLACE>
__global__ void win_stage_gather_kernel(const WinParams p) {
    constexpr int KMAX = W - 2;
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ unsigned long long slice_lo[2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // per wave: tile [64][W + 1] u32 (the odd pitch keeps lane-per-row accesses off each other's banks), slot index [64]
    uint32_t *tile = reinterpret_cast<uint32_t *>(smem) + (size_t)wave * (64 * (W + 1) + 64);
    uint32_t *jrow = tile + 64 * (W + 1);
    const WinItem8 *items = static_cast<const WinItem8 *>(p.items_sorted);
    const int k = p.k;
    WinQueues::Q *Q = &p.queues->q[blockIdx.x & 7];
    const unsigned long long qend = Q->end;
    const unsigned long long slice = blockDim.x;
    const uint32_t idx_mask = (1u << p.idx_bits) - 1u;

    if (tid == 0) slice_lo[0] = atomicAdd(&Q->head, slice);
    __syncthreads();
    for (int buf = 0;; buf ^= 1) {
        const unsigned long long lo = slice_lo[buf];
        if (lo >= qend) break;
        unsigned long long nxt = 0;
        if (tid == 0) nxt = atomicAdd(&Q->head, slice);
        const unsigned long long j = lo + (unsigned long long)wave * 64 + lane;
        const bool live = j < qend;
        uint32_t row[W];
#pragma unroll
        for (int w = 0; w < W; ++w) row[w] = 0u;
        uint32_t slot_index = 0xffffffffu;
        if (live) {
            WinItem8 it = items[j];
            TG_CHECK_VERTEX(p, it.v);
            const uint32_t b = it.bs >> p.idx_bits, idx = it.bs & idx_mask;
            uint64_t e0, e1;
            if (p.ptrs32) {
                e0 = p.ptrs32[it.v];
                e1 = p.ptrs32[it.v + 1];
            } else {
                e0 = (uint64_t)p.ptrs[it.v];
                e1 = (uint64_t)p.ptrs[(int64_t)it.v + 1];
            }
            const uint32_t n = (uint32_t)(e1 - e0);
            const uint32_t cnt = (n == 0) ? 0u : (REPLACE ? (uint32_t)k : min(n, (uint32_t)k));
            slot_index = (uint32_t)((int64_t)b * p.item_pitch + idx);
            row[0] = (uint32_t)e0;
            row[1] = n;
            if (cnt > 0) {
                uint32_t pos[KMAX];
                if (REPLACE || n > (uint32_t)k) {
                    const CallKey ck = p.call_keys[b];
                    const uint64_t did = (uint64_t)(p.id_base + p.state[b].begin + (int64_t)idx);
                    if (REPLACE) { // sampling.rs:57-69
                        Draw d;
#pragma unroll
                        for (int s = 0; s < KMAX; ++s)
                            if (s < k) {
                                if ((s & 1) == 0) d = draw(ck, did, (uint32_t)(s >> 1), D1_REPLACE);
                                pos[s] = bounded32(d.half(s & 1), n);
                            }
                    } else
                        sample_tickets_reg<KMAX>(ck, did, n, k, pos);
                } else {
#pragma unroll
                    for (int s = 0; s < KMAX; ++s) pos[s] = (uint32_t)s; // sampling.rs:12-15: the reservoir is just filled
                }
                if (p.indices32) {
#pragma unroll
                    for (int s = 0; s < KMAX; ++s)
                        if ((uint32_t)s < cnt) row[2 + s] = p.indices32[e0 + pos[s]];
                } else {
#pragma unroll
                    for (int s = 0; s < KMAX; ++s)
                        if ((uint32_t)s < cnt) row[2 + s] = (uint32_t)p.indices[e0 + pos[s]];
                }
            }
        }
#pragma unroll
        for (int w = 0; w < W; ++w) tile[lane * (W + 1) + w] = row[w];
        jrow[lane] = slot_index;
        wave_lds_handoff();
        // W lanes write one item's slot: whole aligned 64-byte chunks, 64 / W items per store instruction
        constexpr int PER = 64 / W;
#pragma unroll
        for (int r = 0; r < W; ++r) {
            const int item_l = r * PER + lane / W, word = lane % W;
            const uint32_t sj = jrow[item_l];
            if (sj != 0xffffffffu) p.stage[(size_t)sj * W + word] = tile[item_l * (W + 1) + word];
        }
        if (tid == 0) slice_lo[buf ^ 1] = nxt;
        __syncthreads();
    }
}

// ---------------------------------------------------------------- E(h): per batch, slot order: stage -> the four streams
constexpr int WIN_STAGE_ROUND_CHUNKS_MAX = 16;

// LDS of E: chunk offsets | tile [round chunks * 64][W + 1] u32 | per wave: positions [64*k] u32, lanes [64*k] u8,
// first output of each lane [64] u32
__host__ __device__ inline size_t win_stage_emit_wave_bytes(int k) {
    return (size_t)64 * k * sizeof(uint32_t) + (((size_t)64 * k + 15) & ~(size_t)15) + 64 * sizeof(uint32_t);
}
__host__ __device__ inline size_t win_stage_emit_lds_bytes(int W, int k, int n_waves, int round_chunks) {
    return (((size_t)(round_chunks + 1) * sizeof(uint32_t) + 15) & ~(size_t)15) +
           (size_t)round_chunks * 64 * (W + 1) * sizeof(uint32_t) + (size_t)n_waves * win_stage_emit_wave_bytes(k);
}

template <int W, int KMAX, bool REPLACE, bool NEXT>
__device__ __forceinline__ void win_stage_emit_batch(const WinParams &p, unsigned char *smem, const int64_t b,
                                                     const int round_chunks) {
    typedef long long i64x2 __attribute__((ext_vector_type(2)));
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n_waves = blockDim.x >> 6;
    const int hop = p.hop, k = p.k;
    uint32_t *chunk_off = reinterpret_cast<uint32_t *>(smem);
    const size_t off_bytes = (((size_t)(round_chunks + 1) * sizeof(uint32_t)) + 15) & ~(size_t)15;
    uint32_t *tile = reinterpret_cast<uint32_t *>(smem + off_bytes);
    unsigned char *wbase = smem + off_bytes + (size_t)round_chunks * 64 * (W + 1) * sizeof(uint32_t) +
                           (size_t)wave * win_stage_emit_wave_bytes(k);
    uint32_t *spos = reinterpret_cast<uint32_t *>(wbase);
    uint8_t *slane = wbase + (size_t)64 * k * sizeof(uint32_t);
    uint32_t *lfirst = reinterpret_cast<uint32_t *>(wbase + (size_t)64 * k * sizeof(uint32_t) + (((size_t)64 * k + 15) & ~(size_t)15));

    int64_t *samples = p.samples + b * p.cap_nodes;
    int64_t *rows = p.rows + b * p.cap_edges;
    int64_t *cols = p.cols + b * p.cap_edges;
    int64_t *eidx = p.edge_index + b * p.cap_edges;
    const int64_t n_seeds = p.n_seeds;
    const WinState st = p.state[b];
    const int64_t begin = st.begin, end = st.end;
    int64_t ne = st.ne;
    const CallKey ck = p.call_keys[b];
    const uint32_t *stage = p.stage + (size_t)(b * p.item_pitch) * W;
    WinItem8 *next_items = static_cast<WinItem8 *>(p.items_in) + b * p.next_pitch;

    if (tid == 0) {
        int64_t *lo = p.layer_offsets + (b * p.n_hops + hop) * 3; // :193
        lo[0] = n_seeds + ne;
        lo[1] = ne;
        lo[2] = n_seeds + ne;
    }
    const int64_t round_slots = (int64_t)round_chunks * 64;
    for (int64_t round_begin = begin; round_begin < end; round_begin += round_slots) {
        const int64_t round_end = min(end, round_begin + round_slots);
        const int nc = (int)((round_end - round_begin + 63) >> 6);
        for (int c = wave; c < nc; c += n_waves) { // pass A: the chunk's stage slots -> LDS (one coalesced stream), counts
            const int64_t idx0 = round_begin - begin + (int64_t)c * 64; // index of the chunk's first slot in the frontier
            const int64_t live = min((int64_t)64, round_end - (round_begin + (int64_t)c * 64));
            uint32_t *t = tile + (size_t)c * 64 * (W + 1);
            constexpr int LOADS = W / 4; // 16-byte loads per lane to cover 64 slots of W words
            u32x4 x[LOADS];
#pragma unroll
            for (int r = 0; r < LOADS; ++r) {
                const int word_at = (r * 64 + lane) * 4; // word index inside the chunk's 64 * W words
                const int rowi = word_at / W;
                x[r] = u32x4{0u, 0u, 0u, 0u};
                if (rowi < live)
                    x[r] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(stage + (size_t)idx0 * W + word_at));
            }
#pragma unroll
            for (int r = 0; r < LOADS; ++r) {
                const int word_at = (r * 64 + lane) * 4;
                const int rowi = word_at / W, w0 = word_at % W;
                uint32_t *dst = t + rowi * (W + 1) + w0;
                dst[0] = x[r].x;
                dst[1] = x[r].y;
                dst[2] = x[r].z;
                dst[3] = x[r].w;
            }
            wave_lds_handoff();
            const uint32_t n = t[lane * (W + 1) + 1];
            const uint32_t cnt = (n == 0) ? 0u : (REPLACE ? (uint32_t)k : min(n, (uint32_t)k));
            const uint32_t tot = wave_sum(cnt);
            if (lane == 0) chunk_off[c] = tot;
        }
        __syncthreads();
        if (wave == 0) { // scan of chunk totals (nc <= 64)
            const uint32_t v = lane < nc ? chunk_off[lane] : 0u;
            const uint32_t incl = wave_inclusive_scan(v);
            if (lane < nc) chunk_off[lane] = incl - v;
            if (lane == 63) chunk_off[nc] = incl;
        }
        __syncthreads();
        for (int c = wave; c < nc; c += n_waves) { // pass B: draws, the four streams (and the next hop's items)
            const uint32_t *t = tile + (size_t)c * 64 * (W + 1);
            const int64_t i0 = round_begin + (int64_t)c * 64;
            const int64_t i = i0 + lane;
            const uint32_t n = t[lane * (W + 1) + 1];
            const uint32_t cnt = (n == 0) ? 0u : (REPLACE ? (uint32_t)k : min(n, (uint32_t)k));
            const uint64_t did = (uint64_t)(p.id_base + i);
            const uint32_t incl = wave_inclusive_scan(cnt);
            const uint32_t excl = incl - cnt;
            const uint32_t total = __shfl(incl, 63, 64);
            const int64_t ea = ne + (int64_t)chunk_off[c];
            lfirst[lane] = excl;
            if (cnt > 0) {
                if (REPLACE) { // sampling.rs:57-69
                    Draw d;
                    for (int s = 0; s < k; ++s) {
                        if ((s & 1) == 0) d = draw(ck, did, (uint32_t)(s >> 1), D1_REPLACE);
                        spos[excl + s] = bounded32(d.half(s & 1), n);
                        slane[excl + s] = (uint8_t)lane;
                    }
                } else if (n <= (uint32_t)k) { // sampling.rs:12-15
                    for (uint32_t s = 0; s < cnt; ++s) {
                        spos[excl + s] = s;
                        slane[excl + s] = (uint8_t)lane;
                    }
                } else {
                    sample_tickets<KMAX>(ck, did, n, k, spos, slane, excl, lane);
                }
            }
            wave_lds_handoff();
            // the gathered neighbour of output q of this chunk: slot `l`'s (q - first output of l)-th value
            auto value_of = [&](uint32_t q) -> int64_t {
                const int l = slane[q];
                return (int64_t)t[l * (W + 1) + 2 + (q - lfirst[l])];
            };
            // rows / cols / edge_index share the parity of their addresses (equal pitch, 16-byte aligned bases); samples
            // (offset by n_seeds, pitch cap_nodes) has its own.  An element at an address that is not 16-byte aligned is
            // stored alone, the rest as 16-byte pairs, one stream after the other (the emit kernel's store shape).
            const uint32_t head = (uint32_t)(((uintptr_t)(rows + ea) >> 3) & 1);
            const uint32_t head_s = (uint32_t)(((uintptr_t)(samples + n_seeds + ea) >> 3) & 1);
            if (lane == 0 && total > 0) {
                if (head) {
                    __builtin_nontemporal_store(n_seeds + ea, &rows[ea]);
                    __builtin_nontemporal_store(i0 + (int64_t)slane[0], &cols[ea]);
                    __builtin_nontemporal_store((int64_t)t[slane[0] * (W + 1)] + (int64_t)spos[0], &eidx[ea]);
                }
                if (head_s) __builtin_nontemporal_store(value_of(0u), &samples[n_seeds + ea]);
            }
            for (uint32_t q = head_s + 2u * lane; q < total; q += 128) { // :215
                const int64_t e = ea + q;
                if (q + 1 < total) {
                    i64x2 s2 = {value_of(q), value_of(q + 1)};
                    __builtin_nontemporal_store(s2, reinterpret_cast<i64x2 *>(&samples[n_seeds + e]));
                } else
                    __builtin_nontemporal_store(value_of(q), &samples[n_seeds + e]);
            }
            for (uint32_t q = head + 2u * lane; q < total; q += 128) { // :217
                const int64_t e = ea + q;
                if (q + 1 < total) {
                    i64x2 r = {n_seeds + e, n_seeds + e + 1};
                    __builtin_nontemporal_store(r, reinterpret_cast<i64x2 *>(&rows[e]));
                } else
                    __builtin_nontemporal_store(n_seeds + e, &rows[e]);
            }
            for (uint32_t q = head + 2u * lane; q < total; q += 128) {
                const int64_t e = ea + q;
                if (q + 1 < total) {
                    i64x2 cc = {i0 + (int64_t)slane[q], i0 + (int64_t)slane[q + 1]};
                    __builtin_nontemporal_store(cc, reinterpret_cast<i64x2 *>(&cols[e]));
                } else
                    __builtin_nontemporal_store(i0 + (int64_t)slane[q], &cols[e]);
            }
            for (uint32_t q = head + 2u * lane; q < total; q += 128) {
                const int64_t e = ea + q;
                const int l0 = slane[q];
                if (q + 1 < total) {
                    const int l1 = slane[q + 1];
                    i64x2 x = {(int64_t)t[l0 * (W + 1)] + (int64_t)spos[q], (int64_t)t[l1 * (W + 1)] + (int64_t)spos[q + 1]};
                    __builtin_nontemporal_store(x, reinterpret_cast<i64x2 *>(&eidx[e]));
                } else
                    __builtin_nontemporal_store((int64_t)t[l0 * (W + 1)] + (int64_t)spos[q], &eidx[e]);
            }
            if (NEXT) { // the new samples are the next hop's frontier: hand them over as items while they are in LDS
                for (uint32_t q = lane; q < total; q += 64) {
                    const uint32_t v = (uint32_t)value_of(q);
                    const uint32_t rel = (uint32_t)(n_seeds + ea + q - end); // index in the next frontier (it begins at `end`)
                    next_items[rel] = WinItem8{v, ((uint32_t)b << p.next_idx_bits) | rel};
                }
            }
            wave_lds_handoff();
        }
        __syncthreads();
        ne += chunk_off[nc];
        __syncthreads();
    }
    win_store_state(p, b, hop, WinState{end, n_seeds + ne, ne, begin}); // :221-222
}

template <int W, int KMAX, bool REPLACE, bool NEXT>
__global__ void win_stage_emit_kernel(const WinParams p, const int round_chunks) {
    extern __shared__ __align__(16) unsigned char smem[];
    win_stage_emit_batch<W, KMAX, REPLACE, NEXT>(p, smem, p.b0 + blockIdx.x, round_chunks);
}
