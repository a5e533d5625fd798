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

// ---------------------------------------------------------------- the sort key of an item (round 4: two levels)
// vtab[i] = first vertex whose column starts at or beyond window i (ptrs is monotone; vtab[n_windows] = n_major), so the
// window of a vertex's column start is the largest i with vtab[i] <= v -- a binary search over a table the workgroup holds
// in LDS, no column look-up.  Windows map to WINDOW BUCKETS XCD-major (as win_bucket); the sort key has two levels:
//   coarse = window bucket / 8      (<= 1 024 coarse buckets: one counting-sort pass whose open destination lines -- rows x
//            coarse buckets -- stay in the L2s, so the 8-byte items leave as whole chunks; the round-3 sort scattered them
//            over rows x 2 048 buckets = 134 MB of open lines and paid a partial-chunk write per item)
//   fine   = (window bucket % 8) * S + sub-bucket of the vertex inside its window's vertex range (S = 16 .. 64 by a shift:
//            tg_ns_win_tuning.stage_fine_sub_bits)
//            (a pass of its own INSIDE each coarse bucket, whose ~1 MB segment is L2-resident).
// Items of one vertex end up adjacent and the vertices of a window in ascending order: a gather workgroup's slice then
// lies in one or two columns, which its L1 holds (760 G gathers/s against 260 from L2: profiles/r04/probe_gather_small.json).
struct StageKey {
    uint32_t coarse, fine;
};
constexpr int WIN_FINE_SUB_BITS_MAX = 7;                              // sub-ranges of a window's vertex range: 16 .. 128 (default)
constexpr int WIN_FINE_PER_COARSE_MAX = 8 << WIN_FINE_SUB_BITS_MAX; // fine keys per coarse bucket, at most
__device__ __forceinline__ StageKey win_stage_key(const uint32_t *vtab, int n_windows, int n_wbuckets, uint32_t v, int sub_bits) {
    int lo = 0, hi = n_windows;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (vtab[mid] <= v)
            lo = mid;
        else
            hi = mid;
    }
    const uint32_t wb = ((uint32_t)lo & 7u) * (uint32_t)(n_wbuckets >> 3) + ((uint32_t)lo >> 3);
    const uint32_t v0 = vtab[lo], range = vtab[lo + 1] - v0; // >= 1 for a vertex of the graph
    const int sh = max(0, 32 - sub_bits - (int)__clz((int)((range - 1u) | 1u))); // (range - 1) >> sh < 1 << sub_bits
    const uint32_t sub = min((1u << sub_bits) - 1u, (v - v0) >> sh);
    return StageKey{wb >> 3, ((wb & 7u) << sub_bits) | sub};
}

__global__ void win_vtab_kernel(const WinParams p, int64_t n_major) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > p.n_windows) return;
    if (i == p.n_windows) {
        p.vtab[i] = (uint32_t)n_major;
        return;
    }
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

// LDS tables of the sort kernels and of the folded first kernel: counters [n_buckets] | vertex table [n_windows + 1]
__host__ __device__ inline size_t win_stage_tables_bytes(int n_buckets, int n_windows) {
    return ((size_t)n_buckets + (size_t)n_windows + 1) * sizeof(uint32_t);
}

// ---------------------------------------------------------------- E0: seeds, hop 0 (direct) + the items of hop 1
// Persistent: workgroup r walks batches b0 + r, b0 + r + gridDim.x, ... and keeps the COARSE histogram of the items it hands
// over in LDS (row r of `hist`): the counting sort needs no pass of its own over the items to count them (hop 0 is bound
// by its random line requests; the binary search per new item hides beneath them).
template <int KMAX, bool REPLACE>
__global__ void __launch_bounds__(WIN_EMIT_MAX_THREADS) win_stage_first_kernel(const WinParams p, const int k0) {
    extern __shared__ __align__(16) unsigned char smem[];
    uint32_t *lhist = reinterpret_cast<uint32_t *>(smem + win_emit_lds_bytes(p.kmax, blockDim.x >> 6, WIN_STAGE_FIRST_RC));
    uint32_t *lvtab = lhist + p.n_buckets;
    for (int i = threadIdx.x; i < p.n_buckets; i += blockDim.x) lhist[i] = 0;
    for (int i = threadIdx.x; i <= p.n_windows; i += blockDim.x) lvtab[i] = p.vtab[i];
    for (int64_t b = p.b0 + blockIdx.x; b < p.b0 + p.n_batches; b += gridDim.x) {
        int64_t *samples = p.samples + b * p.cap_nodes;
        for (int64_t i = threadIdx.x; i < p.n_seeds; i += blockDim.x) samples[i] = p.seeds[b * p.n_seeds + i]; // :184
        const CallKey ck = call_key(p.seed, p.call_id + (uint64_t)b, p.tag);
        if (threadIdx.x == 0) p.call_keys[b] = ck;
        __syncthreads();
        const WinState st = win_emit_hop<WinItemN, KMAX, REPLACE, true, false, true, true, WIN_STAGE_FIRST_RC>(
            p, smem, b, 0, k0, WinState{0, p.n_seeds, 0, 0}, ck, lhist);
        win_store_state(p, b, 0, st);
        __syncthreads(); // the LDS staging of this batch is done before the next batch reuses it
    }
    uint32_t *row = p.hist + (size_t)blockIdx.x * p.n_buckets;
    for (int i = threadIdx.x; i < p.n_buckets; i += blockDim.x) row[i] = lhist[i];
}

// ---------------------------------------------------------------- S(h): the three streams that need no gather, on the side stream
// rows / cols / edge_index of an ordered hop depend on the frontier's column bounds and draws only, not on the gathered
// neighbours: this pass (the push form's emit pass without its items; workgroup per batch) writes them BESIDE the hop's
// sort and gather -- those hang on LDS / L2 latency and leave the HBM write path idle -- and the emit pass behind the
// gather is left with `samples` alone (tg_ns_win_tuning.stage_split).  It reads the per-batch state and writes none.
template <int KMAX, bool REPLACE>
__global__ void __launch_bounds__(WIN_EMIT_MAX_THREADS) win_stage_side_kernel(const WinParams p) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int64_t b = p.b0 + blockIdx.x;
    (void)win_emit_hop<WinItemN, KMAX, REPLACE, false, false, false, false, WIN_STAGE_SIDE_RC>(p, smem, b, p.hop, p.k, p.state[b],
                                                                                             p.call_keys[b]);
}

// ---------------------------------------------------------------- sort, level 1: scatter by coarse bucket (rows = first-kernel workgroups)
__global__ void __launch_bounds__(WIN_PART_THREADS) win_scatter8_kernel(const WinParams p) {
    extern __shared__ __align__(16) unsigned char smem[];
    uint32_t *cur = reinterpret_cast<uint32_t *>(smem);
    uint32_t *lvtab = cur + p.n_buckets;
    const int nb = p.n_buckets;
    const WinItem8 *items = static_cast<const WinItem8 *>(p.items_in);
    WinItem8 *sorted = static_cast<WinItem8 *>(p.items_sorted);
    const uint32_t *row = p.hist + (size_t)blockIdx.x * nb;
    for (int i = threadIdx.x; i < nb; i += blockDim.x) cur[i] = p.base[i] + row[i];
    for (int i = threadIdx.x; i <= p.n_windows; i += blockDim.x) lvtab[i] = p.vtab[i];
    __syncthreads();
    constexpr int U = 4;
    for (int64_t b = p.b0 + blockIdx.x; b < p.b0 + p.n_batches; b += gridDim.x) {
        const WinState st = p.state[b];
        const int64_t n = st.end - st.begin; // the frontier of the hop about to be gathered
        const WinItem8 *src = items + b * p.item_pitch;
        for (int64_t t0 = 0; t0 < n; t0 += (int64_t)U * blockDim.x) {
            WinItem8 it[U];
            bool ok[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t j = t0 + (int64_t)u * blockDim.x + threadIdx.x;
                ok[u] = j < n;
                if (ok[u]) it[u] = src[j];
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (ok[u]) sorted[atomicAdd(&cur[p.item_keys[b * p.item_pitch + t0 + (int64_t)u * blockDim.x + threadIdx.x]], 1u)] = it[u];
        }
    }
}

// The same pass through LDS tiles: a tile (one batch's items, at most WIN_SC_TILE) is ranked per coarse bucket with LDS
// atomics, laid out bucket by bucket in LDS and written out linearly -- a bucket's items of the tile leave as ONE contiguous
// run (~85 bytes at 2.7 K items over 256 buckets) instead of 8 bytes at a time from whichever lane drew the slot: the direct
// form sent 26.7 M write requests to memory for 5.6 M chunks of items (profiles/r04/pmc_traffic_staged_bpl16384.json).
constexpr int WIN_SC_TILE = 4096;
__host__ __device__ inline size_t win_scatter8_tiled_lds(int n_buckets, int n_windows) {
    const size_t tables = ((size_t)4 * n_buckets + (size_t)n_windows + 1) * sizeof(uint32_t) + (size_t)WIN_SC_TILE * sizeof(uint16_t);
    return ((tables + 15) & ~(size_t)15) + (size_t)WIN_SC_TILE * sizeof(WinItem8);
}
__global__ void __launch_bounds__(WIN_PART_THREADS) win_scatter8_tiled_kernel(const WinParams p) {
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ uint32_t wtot[WIN_PART_THREADS / 64];
    const int nb = p.n_buckets, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t *cur = reinterpret_cast<uint32_t *>(smem); // this workgroup's cursor inside every bucket (global positions)
    uint32_t *lvtab = cur + nb;
    uint32_t *cnt = lvtab + p.n_windows + 1; // the tile's items per bucket
    uint32_t *off = cnt + nb;                // their exclusive prefix = where the bucket's run starts in the LDS tile
    uint32_t *gb = off + nb;                 // the cursor before the tile
    uint16_t *skey = reinterpret_cast<uint16_t *>(gb + nb);
    WinItem8 *sitem = reinterpret_cast<WinItem8 *>(
        smem + (((((size_t)4 * nb + (size_t)p.n_windows + 1) * sizeof(uint32_t) + (size_t)WIN_SC_TILE * sizeof(uint16_t)) + 15) & ~(size_t)15));
    const WinItem8 *items = static_cast<const WinItem8 *>(p.items_in);
    WinItem8 *sorted = static_cast<WinItem8 *>(p.items_sorted);
    const uint32_t *row = p.hist + (size_t)blockIdx.x * nb;
    for (int i = tid; i < nb; i += blockDim.x) cur[i] = p.base[i] + row[i];
    for (int i = tid; i <= p.n_windows; i += blockDim.x) lvtab[i] = p.vtab[i];
    constexpr int U = WIN_SC_TILE / WIN_PART_THREADS;
    for (int64_t b = p.b0 + blockIdx.x; b < p.b0 + p.n_batches; b += gridDim.x) {
        const WinState st = p.state[b];
        const int64_t n = st.end - st.begin; // the frontier of the hop about to be gathered
        const WinItem8 *src = items + b * p.item_pitch;
        for (int64_t t0 = 0; t0 < n; t0 += WIN_SC_TILE) {
            const int nt = (int)min((int64_t)WIN_SC_TILE, n - t0);
            for (int i = tid; i < nb; i += blockDim.x) cnt[i] = 0;
            __syncthreads();
            WinItem8 it[U];
            uint32_t key[U], rk[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int j = u * WIN_PART_THREADS + tid;
                key[u] = 0xffffffffu;
                if (j < nt) {
                    it[u] = src[t0 + j];
                    key[u] = p.item_keys[b * p.item_pitch + t0 + j]; // found once, by whoever counted the item
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (key[u] != 0xffffffffu) rk[u] = atomicAdd(&cnt[key[u]], 1u);
            __syncthreads();
            { // exclusive prefix of the counts (nb <= the workgroup's threads), the cursors move on
                const uint32_t c = tid < nb ? cnt[tid] : 0u;
                const uint32_t incl = wave_inclusive_scan(c);
                if (lane == 63) wtot[wave] = incl;
                __syncthreads();
                uint32_t before = 0;
                for (int w = 0; w < wave; ++w) before += wtot[w];
                if (tid < nb) {
                    off[tid] = before + incl - c;
                    gb[tid] = cur[tid];
                    cur[tid] += c;
                }
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (key[u] != 0xffffffffu) {
                    const uint32_t at = off[key[u]] + rk[u];
                    sitem[at] = it[u];
                    skey[at] = (uint16_t)key[u];
                }
            __syncthreads();
            for (int t = tid; t < nt; t += WIN_PART_THREADS) {
                const uint32_t k = skey[t];
                sorted[gb[k] + ((uint32_t)t - off[k])] = sitem[t];
            }
            __syncthreads();
        }
    }
}

// the histogram as a pass of its own (hops beyond the first: their items come from the emit kernel, which is not persistent)
__global__ void __launch_bounds__(WIN_PART_THREADS) win_hist8_kernel(const WinParams p) {
    extern __shared__ __align__(16) unsigned char smem[];
    uint32_t *h = reinterpret_cast<uint32_t *>(smem);
    uint32_t *lvtab = h + p.n_buckets;
    const int nb = p.n_buckets;
    const WinItem8 *items = static_cast<const WinItem8 *>(p.items_in);
    for (int i = threadIdx.x; i < nb; i += blockDim.x) h[i] = 0;
    for (int i = threadIdx.x; i <= p.n_windows; i += blockDim.x) lvtab[i] = p.vtab[i];
    __syncthreads();
    constexpr int U = 4;
    for (int64_t b = p.b0 + blockIdx.x; b < p.b0 + p.n_batches; b += gridDim.x) {
        const WinState st = p.state[b];
        const int64_t n = st.end - st.begin;
        const WinItem8 *src = items + b * p.item_pitch;
        for (int64_t t0 = 0; t0 < n; t0 += (int64_t)U * blockDim.x) {
            uint32_t v[U];
            bool ok[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t j = t0 + (int64_t)u * blockDim.x + threadIdx.x;
                ok[u] = j < n;
                if (ok[u]) v[u] = src[j].v;
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (ok[u]) {
                    const uint32_t c = win_stage_key(lvtab, p.n_windows, p.n_wbuckets, v[u], 4).coarse;
                    atomicAdd(&h[c], 1u);
                    p.item_keys[b * p.item_pitch + t0 + (int64_t)u * blockDim.x + threadIdx.x] = (uint16_t)c;
                }
        }
    }
    __syncthreads();
    uint32_t *row = p.hist + (size_t)blockIdx.x * nb;
    for (int i = threadIdx.x; i < nb; i += blockDim.x) row[i] = h[i];
}

// ---------------------------------------------------------------- sort, level 2: inside each coarse bucket, by fine key
// The level-1 output is cut into TILES of WIN_FINE_TILE consecutive items (a tile lies in one or two coarse buckets: they
// hold ~170 K items each).  COUNT: a tile counts its items per (coarse bucket, fine key) in LDS and reserves, with one
// returning atomic per non-empty bin, its own range inside that key's run (`fine_tot`, one counter per absolute key).
// STARTS: the runs' starts = the coarse bucket's start + the prefix of its keys' totals.  PLACE: the tile reads its items
// again (L2), ranks them per bin in LDS and writes each to start + reserved offset + rank -- scattered 8-byte writes, but
// inside the few coarse segments (~1 MB each) the resident tiles work on, so they merge in the L2s.  The order inside a
// key's run depends on which tile's atomic came first; it does not matter (outputs are addressed by the item's own slot).
// Bins beyond the tile's first two coarse buckets are clamped into the last bin of the second (possible only when coarse
// buckets hold fewer items than a tile): the order only matters for speed, and COUNT and PLACE clamp alike.
constexpr int WIN_FINE_TILE = 8192, WIN_FINE_THREADS = 1024, WIN_FINE_BINS_MAX = 2 * WIN_FINE_PER_COARSE_MAX;
template <bool PLACE>
__global__ void __launch_bounds__(WIN_FINE_THREADS) win_sort_fine_kernel(const WinParams p) {
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ uint32_t h[WIN_FINE_BINS_MAX];
    __shared__ uint32_t c0_s;
    uint32_t *lvtab = reinterpret_cast<uint32_t *>(smem);
    const WinItem8 *src = static_cast<const WinItem8 *>(p.items_sorted);
    WinItem8 *dst = static_cast<WinItem8 *>(p.items_fine);
    const int tid = threadIdx.x;
    const int sub_bits = p.fine_sub_bits, per_coarse = 8 << sub_bits, bins = 2 * per_coarse;
    uint32_t *lbase = lvtab + p.n_windows + 1; // the coarse buckets' starts in the level-1 output
    for (int i = tid; i <= p.n_windows; i += blockDim.x) lvtab[i] = p.vtab[i];
    for (int i = tid; i <= p.n_buckets; i += blockDim.x) lbase[i] = p.base[i];
    __syncthreads(); // the tables are read by every thread from the first tile on
    const uint32_t n = p.base[p.n_buckets];
    constexpr int U = WIN_FINE_TILE / WIN_FINE_THREADS;
    for (uint32_t tile = blockIdx.x; (uint64_t)tile * WIN_FINE_TILE < n; tile += gridDim.x) {
        const uint32_t t0 = tile * WIN_FINE_TILE;
        uint32_t *tile_off = p.fine_tile_off + (size_t)tile * bins;
        WinItem8 it[U];
        uint32_t key[U], rank[U];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t j = t0 + (uint32_t)u * WIN_FINE_THREADS + tid;
            ok[u] = j < n;
            if (ok[u]) it[u] = src[j];
        }
        for (int i = tid; i < bins; i += WIN_FINE_THREADS) h[i] = PLACE ? tile_off[i] : 0u; // PLACE: the tile's reserved offsets, then cursors
        // No search of the vertex table here: the level-1 output is ordered by coarse bucket, so an item's coarse bucket
        // follows from its POSITION (the bucket starts `base`), and inside a coarse bucket the window is one of 8 known
        // ones -- 7 comparisons against splitters that are the same for the whole tile.
        for (int c = tid; c < p.n_buckets; c += WIN_FINE_THREADS)
            if (lbase[c] <= t0 && t0 < lbase[c + 1]) c0_s = (uint32_t)c; // exactly one: the tile's first item exists
        __syncthreads();
        const uint32_t c0 = min(c0_s, (uint32_t)p.n_buckets - 1u);
        // splitters: the first vertices of the 8 windows of coarse bucket c0 (s0) and c0 + 1 (s1), uniform over the tile;
        // window r of coarse bucket c = window lo0(c) + 8 r (window buckets are XCD-major: bucket wb holds window (wb % wq) * 8 + wb / wq)
        const uint32_t wq = (uint32_t)(p.n_wbuckets >> 3);
        const uint32_t lo0 = ((c0 * 8u) % wq) * 8u + (c0 * 8u) / wq, lo1 = (((c0 + 1u) * 8u) % wq) * 8u + ((c0 + 1u) * 8u) / wq;
        uint32_t s0[8], s1[8];
#pragma unroll
        for (int q = 1; q < 8; ++q) {
            s0[q] = lo0 + 8u * q < (uint32_t)p.n_windows ? lvtab[lo0 + 8u * q] : 0xffffffffu;
            s1[q] = (c0 + 1u < (uint32_t)p.n_buckets && lo1 + 8u * q < (uint32_t)p.n_windows) ? lvtab[lo1 + 8u * q] : 0xffffffffu;
        }
        const uint32_t bnd1 = lbase[min(c0 + 1u, (uint32_t)p.n_buckets)], bnd2 = lbase[min(c0 + 2u, (uint32_t)p.n_buckets)];
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (ok[u]) {
                const uint32_t j = t0 + (uint32_t)u * WIN_FINE_THREADS + tid;
                const bool second = j >= bnd1;
                key[u] = (uint32_t)bins - 1u; // beyond the tile's first two coarse buckets: clamped, COUNT and PLACE alike
                if (j < bnd2 || !second) {
                    const uint32_t v = it[u].v;
                    uint32_t r = 0;
#pragma unroll
                    for (int q = 1; q < 8; ++q) r += (uint32_t)(v >= (second ? s1[q] : s0[q]));
                    const uint32_t lo = (second ? lo1 : lo0) + 8u * r;
                    const uint32_t v0 = lvtab[lo], range = lvtab[lo + 1] - v0;
                    const int sh = max(0, 32 - sub_bits - (int)__clz((int)((range - 1u) | 1u)));
                    const uint32_t sub = min((1u << sub_bits) - 1u, (v - v0) >> sh);
                    key[u] = min((uint32_t)bins - 1u, (second ? (uint32_t)per_coarse : 0u) + ((r << sub_bits) | sub));
                }
                rank[u] = atomicAdd(&h[key[u]], 1u);
            }
        if (PLACE) {
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (ok[u]) dst[p.fine_start[(size_t)c0 * per_coarse + key[u]] + rank[u]] = it[u];
        } else {
            __syncthreads();
            for (int i = tid; i < bins; i += WIN_FINE_THREADS) {
                const uint32_t cnt = h[i];
                tile_off[i] = cnt ? atomicAdd(&p.fine_tot[(size_t)c0 * per_coarse + i], cnt) : 0u;
            }
        }
        __syncthreads();
    }
}

// fine_start = the exclusive prefix of fine_tot over ALL absolute keys in key order ((n_buckets + 1) * 128 of them: the last
// row takes bins clamped out of the last bucket).  A global prefix, not "coarse start + prefix inside the bucket": clamped
// bins count under a neighbouring key, and only a prefix over everything keeps the runs disjoint whatever was counted
// where.  One workgroup: a wavefront sums a bucket's 128 keys, the bucket totals are scanned in LDS, then the same
// wavefronts write their bucket's starts.
// Two launches: the keys' totals per coarse bucket (a wavefront per bucket), then every workgroup scans the bucket totals for
// itself (a thousand values) and its 16 wavefronts write their buckets' starts.  (One workgroup doing all of it took 124 us
// at 512 keys per bucket.)
__global__ void __launch_bounds__(1024) win_fine_rowsum_kernel(const WinParams p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rows = p.n_buckets + 1, c = blockIdx.x * 16 + wave;
    if (c >= rows) return;
    const int per_coarse = 8 << p.fine_sub_bits, per_lane = per_coarse >> 6;
    uint32_t mine = 0;
    for (int i = 0; i < per_lane; ++i) mine += p.fine_tot[(size_t)c * per_coarse + per_lane * lane + i];
    const uint32_t t = wave_sum(mine);
    if (lane == 0) p.fine_rowtot[c] = t;
}
__global__ void __launch_bounds__(1024) win_fine_starts_kernel(const WinParams p) {
    __shared__ uint32_t tot[WIN_MAX_BUCKETS / 8 + 8], wave_tot[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rows = p.n_buckets + 1;
    const int per_coarse = 8 << p.fine_sub_bits, per_lane = per_coarse >> 6; // 2 .. 32 consecutive keys per lane
    uint32_t carry = 0;
    for (int c0 = 0; c0 < rows; c0 += 1024) { // exclusive scan of the bucket totals, 1 024 at a time
        const int c = c0 + tid;
        const uint32_t v = c < rows ? p.fine_rowtot[c] : 0u;
        const uint32_t incl = wave_inclusive_scan(v);
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        uint32_t off = carry;
        for (int w = 0; w < wave; ++w) off += wave_tot[w];
        if (c < rows) tot[c] = off + incl - v;
        uint32_t all = 0;
        for (int w = 0; w < 16; ++w) all += wave_tot[w];
        carry += all;
        __syncthreads();
    }
    const int c = blockIdx.x * 16 + wave;
    if (c < rows) {
        uint32_t mine = 0;
        for (int i = 0; i < per_lane; ++i) mine += p.fine_tot[(size_t)c * per_coarse + per_lane * lane + i];
        const uint32_t incl = wave_inclusive_scan(mine);
        uint32_t at = tot[c] + incl - mine;
        for (int i = 0; i < per_lane; ++i) {
            p.fine_start[(size_t)c * per_coarse + per_lane * lane + i] = at;
            at += p.fine_tot[(size_t)c * per_coarse + per_lane * lane + i];
        }
    }
}

// ---------------------------------------------------------------- the stage slot (round 4: positions travel with the neighbours)
// One slot per frontier vertex, W = 16 or 32 words (one or two whole 64-byte chunks), a bit stream:
//   bits  0..31  column start (edge pointer of the column's first edge; the narrow form has 32-bit edge pointers)
//   bits 32..39  number of samples cnt (0 .. fan-out <= 255)
//   then cnt pairs {neighbour id: bv bits, position inside the column: bp bits}, slot order (s = 0 .. cnt - 1)
// bv = bits of the largest vertex id, bp = bits of the largest column length (tg_graph.max_degree; the edge count when the
// caller gave none) -- per LAUNCH, uniform, passed as kernel arguments; the host picks W so that 40 + k (bv + bp) bits fit.
// RMAT-24, k = 10: 40 + 10 * (24 + 19) = 470 bits: one chunk.  Because the chosen POSITIONS travel with the gathered
// neighbours, the emit pass computes no draws: it is a pure stream (the round-3 form recomputed them, and the two passes
// then both hung on the vector ALUs).  StageBits / BitWriter / BitReader: stage_bits.h.

// ---------------------------------------------------------------- G: window-ordered gather into the stage slots
// W = words per stage slot; KMAX = unroll bound of the slot loops (>= the hop's fan-out).
template <int W, int KMAX, bool REPLACE>
__global__ void win_stage_gather_kernel(const WinParams p, const StageBits sb) {
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ unsigned long long slice_lo[2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // per wave: tile [64][W + 1] u32 (the odd pitch keeps lane-per-row accesses off each other's banks), slot index [64]
    uint32_t *tile = reinterpret_cast<uint32_t *>(smem) + (size_t)wave * (64 * (W + 1) + 64);
    uint32_t *jrow = tile + 64 * (W + 1);
    const WinItem8 *items = static_cast<const WinItem8 *>(p.items_fine);
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
        uint32_t slot_index = 0xffffffffu, e0w = 0, cnt = 0;
        uint32_t pos[KMAX], nbr[KMAX];
#pragma unroll
        for (int s = 0; s < KMAX; ++s) pos[s] = nbr[s] = 0u;
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
            cnt = (n == 0) ? 0u : (REPLACE ? (uint32_t)k : min(n, (uint32_t)k));
            slot_index = (uint32_t)((int64_t)b * p.item_pitch + idx);
            e0w = (uint32_t)e0;
            if (cnt > 0) {
                if (REPLACE || n > (uint32_t)k) {
                    const CallKey ck = p.call_keys[b];
                    // the hop's frontier begins at the same slot in every batch when it is hop 1 (right behind the seeds):
                    // no look-up of the batch's state then
                    const int64_t fbegin = p.hop == 1 ? p.n_seeds : p.state[b].begin;
                    const uint64_t did = (uint64_t)(p.id_base + fbegin + (int64_t)idx);
                    if (REPLACE) // sampling.rs:57-69
                        slot_draws<KMAX, true>(ck, did, n, k, pos);
                    else
                        sample_tickets_reg<KMAX>(ck, did, n, k, pos);
                } else {
#pragma unroll
                    for (int s = 0; s < KMAX; ++s) pos[s] = (uint32_t)s; // sampling.rs:12-15: the reservoir is just filled
                }
                if (p.indices32) {
#pragma unroll
                    for (int s = 0; s < KMAX; ++s)
                        if ((uint32_t)s < cnt) nbr[s] = p.indices32[e0 + pos[s]];
                } else {
#pragma unroll
                    for (int s = 0; s < KMAX; ++s)
                        if ((uint32_t)s < cnt) nbr[s] = (uint32_t)p.indices[e0 + pos[s]];
                }
            }
        }
        { // the slot as a bit stream into this lane's row of the tile
            BitWriter bw{tile + lane * (W + 1), (uint64_t)e0w | ((uint64_t)cnt << 32), 8, 1};
            tile[lane * (W + 1)] = e0w;
            bw.acc >>= 32;
#pragma unroll
            for (int s = 0; s < KMAX; ++s) {
                if (s < k) {
                    const bool on = (uint32_t)s < cnt;
                    bw.push(on ? nbr[s] : 0u, sb.bv);
                    bw.push(on ? pos[s] : 0u, sb.bp);
                }
            }
            bw.finish(W);
        }
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

// LDS of E: chunk offsets | tile [round chunks * 64][W + 1] u32 | per wave: neighbours [64*k] u32, edge pointers [64*k] u32,
// lanes [64*k] u8
// (split: the neighbours alone)
__host__ __device__ inline size_t win_stage_emit_wave_bytes(int k, bool split) {
    return split ? (size_t)64 * k * sizeof(uint32_t)
                 : (size_t)2 * 64 * k * sizeof(uint32_t) + (((size_t)64 * k + 15) & ~(size_t)15);
}
__host__ __device__ inline size_t win_stage_emit_lds_bytes(int W, int k, int n_waves, int round_chunks, bool split) {
    return (((size_t)(round_chunks + 1) * sizeof(uint32_t) + 15) & ~(size_t)15) +
           (size_t)round_chunks * 64 * (W + 1) * sizeof(uint32_t) + (size_t)n_waves * win_stage_emit_wave_bytes(k, split);
}

// SPLIT: rows / cols / edge_index were written by the side pass; only `samples` (and the next hop's items) leave here.
template <int W, int KMAX, bool NEXT, bool SPLIT>
__device__ __forceinline__ void win_stage_emit_batch(const WinParams &p, const StageBits sb, unsigned char *smem,
                                                     const int64_t b, const int round_chunks) {
    typedef long long i64x2 __attribute__((ext_vector_type(2)));
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n_waves = blockDim.x >> 6;
    const int hop = p.hop, k = p.k;
    uint32_t *chunk_off = reinterpret_cast<uint32_t *>(smem);
    const size_t off_bytes = (((size_t)(round_chunks + 1) * sizeof(uint32_t)) + 15) & ~(size_t)15;
    uint32_t *tile = reinterpret_cast<uint32_t *>(smem + off_bytes);
    unsigned char *wbase = smem + off_bytes + (size_t)round_chunks * 64 * (W + 1) * sizeof(uint32_t) +
                           (size_t)wave * win_stage_emit_wave_bytes(k, SPLIT);
    uint32_t *sval = reinterpret_cast<uint32_t *>(wbase);
    uint32_t *sptr = sval + (size_t)64 * k;
    uint8_t *slane = reinterpret_cast<uint8_t *>(sptr + (size_t)64 * k);

    int64_t *samples = p.samples + b * p.cap_nodes;
    int64_t *rows = p.rows + b * p.cap_edges;
    int64_t *cols = p.cols + b * p.cap_edges;
    int64_t *eidx = p.edge_index + b * p.cap_edges;
    const int64_t n_seeds = p.n_seeds;
    const WinState st = p.state[b];
    const int64_t begin = st.begin, end = st.end;
    int64_t ne = st.ne;
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
            const uint32_t cnt = t[lane * (W + 1) + 1] & 0xffu; // rows past `live` were zero-filled
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
        for (int c = wave; c < nc; c += n_waves) { // pass B: unpack, the four streams (and the next hop's items)
            const uint32_t *t = tile + (size_t)c * 64 * (W + 1);
            const int64_t i0 = round_begin + (int64_t)c * 64;
            const uint32_t e0 = t[lane * (W + 1)];
            const uint32_t hdr = t[lane * (W + 1) + 1];
            const uint32_t cnt = hdr & 0xffu;
            const uint32_t incl = wave_inclusive_scan(cnt);
            const uint32_t excl = incl - cnt;
            const uint32_t total = __shfl(incl, 63, 64);
            const int64_t ea = ne + (int64_t)chunk_off[c];
            { // lane = slot: its pairs out of the bit stream into output order
                BitReader br{t + lane * (W + 1), (uint64_t)(hdr >> 8), 24, 2};
#pragma unroll
                for (int s = 0; s < KMAX; ++s) {
                    if (s < k) {
                        const uint32_t v = br.pop(sb.bv);
                        const uint32_t ps = br.pop(sb.bp);
                        if ((uint32_t)s < cnt) {
                            sval[excl + s] = v;
                            if (!SPLIT) {
                                sptr[excl + s] = e0 + ps; // the CSC edge pointer (narrow form: < 2^32)
                                slane[excl + s] = (uint8_t)lane;
                            }
                        }
                    }
                }
            }
            wave_lds_handoff();
            // rows / cols / edge_index share the alignment of their addresses (equal pitch, 16-byte aligned bases); samples
            // (offset by n_seeds, pitch cap_nodes) has its own.  The elements up to the next 64-byte boundary (head: 0 .. 7)
            // are stored alone, the rest as 16-byte pairs, one stream after the other: every store instruction of the
            // wavefront then covers whole aligned 64-byte chunks.  The stores are non-temporal, and a chunk that two
            // instructions share went out to memory as two partial writes: 14 % more write requests than the streams hold
            // chunks (profiles/r04/pmc_traffic_staged_bpl16384.json); aligned: emit pass 3.03 -> 2.93 ms
            // (tg_ns_win_tuning.store_align64, profiles/r04/ab_store_align64.jsonl)
            const uint32_t am = p.store_align;
            const uint32_t head = (uint32_t)((am + 1u - (uint32_t)(((uintptr_t)(rows + ea) >> 3) & am)) & am);
            const uint32_t head_s = (uint32_t)((am + 1u - (uint32_t)(((uintptr_t)(samples + n_seeds + ea) >> 3) & am)) & am);
            if ((uint32_t)lane < head && (uint32_t)lane < total && !SPLIT) {
                __builtin_nontemporal_store(n_seeds + ea + (int64_t)lane, &rows[ea + lane]);
                __builtin_nontemporal_store(i0 + (int64_t)slane[lane], &cols[ea + lane]);
                __builtin_nontemporal_store((int64_t)sptr[lane], &eidx[ea + lane]);
            }
            if ((uint32_t)lane < head_s && (uint32_t)lane < total)
                __builtin_nontemporal_store((int64_t)sval[lane], &samples[n_seeds + ea + lane]);
            for (uint32_t q = head_s + 2u * lane; q < total; q += 128) { // :215
                const int64_t e = ea + q;
                if (q + 1 < total) {
                    i64x2 s2 = {(int64_t)sval[q], (int64_t)sval[q + 1]};
                    __builtin_nontemporal_store(s2, reinterpret_cast<i64x2 *>(&samples[n_seeds + e]));
                } else
                    __builtin_nontemporal_store((int64_t)sval[q], &samples[n_seeds + e]);
            }
            if (!SPLIT)
            for (uint32_t q = head + 2u * lane; q < total; q += 128) { // :217
                const int64_t e = ea + q;
                if (q + 1 < total) {
                    i64x2 r = {n_seeds + e, n_seeds + e + 1};
                    __builtin_nontemporal_store(r, reinterpret_cast<i64x2 *>(&rows[e]));
                } else
                    __builtin_nontemporal_store(n_seeds + e, &rows[e]);
            }
            if (!SPLIT)
            for (uint32_t q = head + 2u * lane; q < total; q += 128) {
                const int64_t e = ea + q;
                if (q + 1 < total) {
                    i64x2 cc = {i0 + (int64_t)slane[q], i0 + (int64_t)slane[q + 1]};
                    __builtin_nontemporal_store(cc, reinterpret_cast<i64x2 *>(&cols[e]));
                } else
                    __builtin_nontemporal_store(i0 + (int64_t)slane[q], &cols[e]);
            }
            if (!SPLIT)
            for (uint32_t q = head + 2u * lane; q < total; q += 128) {
                const int64_t e = ea + q;
                if (q + 1 < total) {
                    i64x2 x = {(int64_t)sptr[q], (int64_t)sptr[q + 1]};
                    __builtin_nontemporal_store(x, reinterpret_cast<i64x2 *>(&eidx[e]));
                } else
                    __builtin_nontemporal_store((int64_t)sptr[q], &eidx[e]);
            }
            if (NEXT) { // the new samples are the next hop's frontier: hand them over as items while they are in LDS
                for (uint32_t q = lane; q < total; q += 64) {
                    const uint32_t rel = (uint32_t)(n_seeds + ea + q - end); // index in the next frontier (it begins at `end`)
                    next_items[rel] = WinItem8{sval[q], ((uint32_t)b << p.next_idx_bits) | rel};
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

template <int W, int KMAX, bool NEXT, bool SPLIT>
__global__ void win_stage_emit_kernel(const WinParams p, const StageBits sb, const int round_chunks) {
    extern __shared__ __align__(16) unsigned char smem[];
    win_stage_emit_batch<W, KMAX, NEXT, SPLIT>(p, sb, smem, p.b0 + blockIdx.x, round_chunks);
}
