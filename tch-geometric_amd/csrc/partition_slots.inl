// Slot replies of the partitioned sampler (included by partition.hip, inside namespace tg).
//
// The compact reply of tg_part_count / tg_part_sample costs the owner a count pass, a device-length prefix sum and
// scattered 8-byte stores, the origin a second prefix sum, and the host a SECOND size read-back per hop (how many reply
// entries each peer gets).  Here the owner answers every request with one fixed-size packed slot -- the stage slot of
// tg_ns_homo_batched's staged pipeline (stage_bits.h: column start, count, fan-out x {neighbour, position}), 64 or 128
// bytes, written as whole chunks at the REQUEST's index.  The reply all-to-all then has the request exchange's split sizes
// mirrored (W words per request): no reply sizes to learn, one read-back per hop instead of two, and no count / prefix
// kernels on either side.  On RMAT-24 with fan-out 10 a slot is 470 bits (64 B) against 4 + 8 * cnt bytes of the compact
// reply: fewer bytes for a column of >= 8 samples, more for a short one (the first hop's seeds) -- about equal over a call.
//
// The global edge pointer of a sample = e_lo of the OWNER's shard + column start + position; the origin knows the owner of
// a request from where it sits in its send buffer (grouped by owner) and every shard's e_lo from a table made once.
// Same draws as everywhere: (seed, requester's first call id + batch, NS_HOMO, slot) -- results equal the replicated sampler's.

// ---------------------------------------------------------------- owner: requests (in window order) -> slots
// call keys of (requesting rank, batch) for batch < PART_KEY_CAP: one Philox block each, made once per hop instead of once
// per request (a request of a later batch derives its key itself)
constexpr int PART_KEY_CAP = 4096;
__global__ void part_call_keys_kernel(const PartOwnerParams p, CallKey *keys) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < p.world * PART_KEY_CAP)
        keys[i] = call_key(p.seed, p.seg_call0[i / PART_KEY_CAP] + (uint64_t)(i % PART_KEY_CAP), TAG_NS_HOMO);
}

template <int W, int KMAX, bool REPLACE>
__global__ void __launch_bounds__(512) part_slot_sample_kernel(const PartOwnerParams p, const PartSorted *__restrict__ sorted,
                                                               const CallKey *__restrict__ keys, const StageBits sb,
                                                               uint32_t *__restrict__ slots, const int chunks_per_wave) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
    // per wave: tile [64][W + 1] u32 (odd pitch: lane-per-row accesses stay off each other's banks), request index [64]
    uint32_t *tile = reinterpret_cast<uint32_t *>(smem) + (size_t)wave * (64 * (W + 1) + 64);
    uint32_t *jrow = tile + 64 * (W + 1);
    const int k = p.k;
    const int64_t m = *p.m_dev;
    const int64_t n_chunks = (m + 63) >> 6;
    // blocks with equal blockIdx % 8 share an XCD (a speed matter only): group x sweeps the x-th eighth of the order.  A
    // workgroup takes a FEW consecutive chunks and ends (the grid covers the host's bound of the request count; the blocks
    // beyond the real count end at once): short-lived workgroups keep compute units turning over, so the collective
    // kernels of another super-batch in flight (partitioned.interleave) get theirs promptly -- persistent workgroups
    // here starved them (2 lanes over RCCL: 11 ms per call instead of 4.4)
    const int x = blockIdx.x & 7, local = blockIdx.x >> 3;
    const int64_t c_lo = n_chunks * x / 8, c_hi = n_chunks * (x + 1) / 8;
    const int64_t c_first = c_lo + ((int64_t)local * n_waves + wave) * chunks_per_wave;
    for (int64_t c = c_first; c < min(c_hi, c_first + chunks_per_wave); ++c) {
        const int64_t i = (c << 6) + lane;
        uint32_t slot_index = 0xffffffffu, e0w = 0, cnt = 0;
        uint32_t pos[KMAX], nbr[KMAX];
#pragma unroll
        for (int s = 0; s < KMAX; ++s) pos[s] = nbr[s] = 0u;
        if (i < m) {
            PartSorted r; // the request: moved into window order by the sort, or read where it arrived
            if (sorted) {
                r = sorted[i];
            } else {
                const PartRequest q = p.req[i];
                const int64_t w = q.vertex - p.v_lo;
                r = PartSorted{(w >= 0 && w < p.n_major) ? (uint32_t)w : 0xffffffffu, (uint32_t)i, q.batch, q.slot};
            }
            const int64_t j = r.j;
            slot_index = r.j;
            if (r.v != 0xffffffffu) { // (a vertex that is not mine gets an empty slot, as tg_part_count counts 0)
                int64_t e0;
                uint32_t n;
                if (p.ptrs32) {
                    const uint32_t a = p.ptrs32[r.v];
                    e0 = a;
                    n = p.ptrs32[r.v + 1] - a;
                } else {
                    e0 = p.ptrs[r.v];
                    n = (uint32_t)(p.ptrs[(int64_t)r.v + 1] - e0);
                }
                cnt = (n == 0) ? 0u : (REPLACE ? (uint32_t)k : min(n, (uint32_t)k));
                e0w = (uint32_t)e0;
                if (cnt > 0) {
                    if (REPLACE || n > (uint32_t)k) {
                        int src = 0; // requesting rank of request j: the segment that holds j
                        while (src + 1 < p.world && p.seg_off[src + 1] <= j) ++src;
                        const CallKey ck = (keys && r.batch < (uint32_t)PART_KEY_CAP)
                                               ? keys[src * PART_KEY_CAP + (int)r.batch]
                                               : call_key(p.seed, p.seg_call0[src] + (uint64_t)r.batch, TAG_NS_HOMO);
                        if (REPLACE) // sampling.rs:57-69
                            slot_draws<KMAX, true>(ck, (uint64_t)r.slot, n, k, pos);
                        else
                            sample_tickets_reg<KMAX>(ck, (uint64_t)r.slot, n, k, pos);
                    } else {
#pragma unroll
                        for (int s = 0; s < KMAX; ++s) pos[s] = (uint32_t)s; // sampling.rs:12-15
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
        // W lanes write one request's slot: whole aligned 64-byte chunks, 64 / W requests per store instruction
        constexpr int PER = 64 / W;
#pragma unroll
        for (int r = 0; r < W; ++r) {
            const int item_l = r * PER + lane / W, word = lane % W;
            const uint32_t sj = jrow[item_l];
            if (sj != 0xffffffffu) slots[(size_t)sj * W + word] = tile[item_l * (W + 1) + word];
        }
        wave_lds_handoff();
    }
}

// ---------------------------------------------------------------- origin: slots (request order) -> the slabs, slot order
struct PartSlotEmitParams {
    int64_t n_seeds, cap_nodes, cap_edges;
    int64_t *samples, *rows, *cols, *edge_index, *layer_offsets, *counts;
    PartState *state;
    const uint32_t *req_pos; // where each frontier slot's request went in the send buffer = where its reply came back
    const uint32_t *slots;   // [requests][W] as returned
    const int64_t *base;     // [world + 1] start of each owner's requests in the send buffer (world > 1)
    int64_t e_lo_of[PART_MAX_WORLD];
    int32_t k, hop, n_hops, world;
};

__host__ __device__ inline size_t part_slot_emit_wave_bytes(int W, int k) { // tile | request positions | staging
    return (size_t)64 * (W + 1) * 4 + 64 * 4 + 64 * sizeof(int64_t) + (size_t)2 * 64 * k * 4 + (((size_t)64 * k + 15) & ~(size_t)15);
}

template <int W, int KMAX, int PART_SLOT_EMIT_WAVES>
__global__ void __launch_bounds__(64 * PART_SLOT_EMIT_WAVES) part_slot_emit_kernel(const PartSlotEmitParams p, const StageBits sb) {
    typedef long long i64x2 __attribute__((ext_vector_type(2)));
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ uint32_t wave_tot[PART_SLOT_EMIT_WAVES];
    __shared__ int64_t lbase[PART_MAX_WORLD + 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t b = blockIdx.x;
    const int k = p.k;
    unsigned char *wbase = smem + (size_t)wave * part_slot_emit_wave_bytes(W, k);
    uint32_t *tile = reinterpret_cast<uint32_t *>(wbase);
    uint32_t *rpl = tile + 64 * (W + 1);
    int64_t *ebase = reinterpret_cast<int64_t *>(rpl + 64);
    uint32_t *sval = reinterpret_cast<uint32_t *>(ebase + 64);
    uint32_t *sptr = sval + 64 * k;
    uint8_t *slane = reinterpret_cast<uint8_t *>(sptr + 64 * k);
    int64_t *samples = p.samples + b * p.cap_nodes;
    int64_t *rows = p.rows + b * p.cap_edges, *cols = p.cols + b * p.cap_edges, *eidx = p.edge_index + b * p.cap_edges;
    const PartState st = p.state[b];
    const int64_t begin = st.begin, end = st.end, fbase = st.fbase;
    int64_t ne = st.ne;
    const int64_t n_seeds = p.n_seeds;
    if (p.world > 1 && tid <= p.world) lbase[tid] = p.base[tid];
    if (tid == 0) { // neighbor_sampling.rs:193
        int64_t *lo = p.layer_offsets + (b * p.n_hops + p.hop) * 3;
        lo[0] = n_seeds + ne;
        lo[1] = ne;
        lo[2] = n_seeds + ne;
    }
    __syncthreads();
    for (int64_t round_begin = begin; round_begin < end; round_begin += (int64_t)PART_SLOT_EMIT_WAVES * 64) {
        const int64_t i0 = round_begin + (int64_t)wave * 64, i = i0 + lane;
        const bool live = i < end;
        uint32_t rp = 0xffffffffu;
        int64_t e_lo = p.e_lo_of[0];
        if (live) {
            rp = p.req_pos[fbase + (i - begin)];
            if (p.world > 1) {
                int o = 0;
                while (o + 1 < p.world && lbase[o + 1] <= (int64_t)rp) ++o;
                e_lo = p.e_lo_of[o];
            }
        }
        rpl[lane] = rp;
        ebase[lane] = e_lo;
        wave_lds_handoff();
        { // W / 4 lanes fetch one request's slot, 16 bytes each: whole 64-byte chunks
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            constexpr int LOADS = W / 4;
            u32x4 x[LOADS];
#pragma unroll
            for (int r = 0; r < LOADS; ++r) {
                const int word_at = (r * 64 + lane) * 4; // word index inside the chunk's 64 * W words
                const uint32_t q = rpl[word_at / W];
                x[r] = u32x4{0u, 0u, 0u, 0u};
                if (q != 0xffffffffu)
                    x[r] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p.slots + (size_t)q * W + word_at % W));
            }
#pragma unroll
            for (int r = 0; r < LOADS; ++r) {
                const int word_at = (r * 64 + lane) * 4;
                uint32_t *dst = tile + (word_at / W) * (W + 1) + word_at % W;
                dst[0] = x[r].x;
                dst[1] = x[r].y;
                dst[2] = x[r].z;
                dst[3] = x[r].w;
            }
        }
        wave_lds_handoff();
        const uint32_t e0 = tile[lane * (W + 1)];
        const uint32_t hdr = tile[lane * (W + 1) + 1];
        const uint32_t cnt = hdr & 0xffu; // a row past the frontier's end was zero-filled
        const uint32_t incl = wave_inclusive_scan(cnt);
        const uint32_t excl = incl - cnt;
        const uint32_t total = __shfl(incl, 63, 64);
        if (lane == 0) wave_tot[wave] = total;
        { // lane = frontier slot: its pairs out of the bit stream into output order
            BitReader br{tile + lane * (W + 1), (uint64_t)(hdr >> 8), 24, 2};
#pragma unroll
            for (int s = 0; s < KMAX; ++s) {
                if (s < k) {
                    const uint32_t v = br.pop(sb.bv);
                    const uint32_t ps = br.pop(sb.bp);
                    if ((uint32_t)s < cnt) {
                        sval[excl + s] = v;
                        sptr[excl + s] = e0 + ps; // local edge pointer (a shard holds < 2^32 edges)
                        slane[excl + s] = (uint8_t)lane;
                    }
                }
            }
        }
        __syncthreads();
        int64_t ea = ne;
        uint32_t round_total = 0;
#pragma unroll
        for (int u = 0; u < PART_SLOT_EMIT_WAVES; ++u) {
            if (u < wave) ea += wave_tot[u];
            round_total += wave_tot[u];
        }
        // the four streams; the elements up to the next 64-byte boundary are stored alone, the rest as 16-byte pairs: every
        // store instruction of the wavefront then covers whole aligned 64-byte chunks (ns_homo_stage.inl's emit pass)
        const uint32_t head = (uint32_t)((8u - (uint32_t)(((uintptr_t)(rows + ea) >> 3) & 7u)) & 7u);
        const uint32_t head_s = (uint32_t)((8u - (uint32_t)(((uintptr_t)(samples + n_seeds + ea) >> 3) & 7u)) & 7u);
        if ((uint32_t)lane < head && (uint32_t)lane < total) {
            __builtin_nontemporal_store(n_seeds + ea + (int64_t)lane, &rows[ea + lane]);                       // :217
            __builtin_nontemporal_store(i0 + (int64_t)slane[lane], &cols[ea + lane]);
            __builtin_nontemporal_store(ebase[slane[lane]] + (int64_t)sptr[lane], &eidx[ea + lane]);
        }
        if ((uint32_t)lane < head_s && (uint32_t)lane < total)
            samples[n_seeds + ea + lane] = (int64_t)sval[lane];                                               // :215 (the next hop's frontier)
        for (uint32_t q = head_s + 2u * lane; q < total; q += 128) {
            const int64_t e = ea + q;
            if (q + 1 < total) {
                i64x2 s2 = {(int64_t)sval[q], (int64_t)sval[q + 1]};
                *reinterpret_cast<i64x2 *>(&samples[n_seeds + e]) = s2;
            } else
                samples[n_seeds + e] = (int64_t)sval[q];
        }
        for (uint32_t q = head + 2u * lane; q < total; q += 128) {
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
            if (q + 1 < total) {
                i64x2 x = {ebase[slane[q]] + (int64_t)sptr[q], ebase[slane[q + 1]] + (int64_t)sptr[q + 1]};
                __builtin_nontemporal_store(x, reinterpret_cast<i64x2 *>(&eidx[e]));
            } else
                __builtin_nontemporal_store(ebase[slane[q]] + (int64_t)sptr[q], &eidx[e]);
        }
        ne += round_total;
        __syncthreads(); // wave_tot and the staging are rewritten next round
    }
    if (tid == 0) { // :221-222
        p.state[b] = PartState{end, n_seeds + ne, ne, fbase};
        p.counts[b * 2 + 0] = n_seeds + ne;
        p.counts[b * 2 + 1] = ne;
    }
}
