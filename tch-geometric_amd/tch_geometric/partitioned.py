"""neighbor_sampling_homogenous over a RANGE-PARTITIONED CSC (mode 2; BASELINE cfg5).

For graphs that do not fit one GPU's HBM: rank r owns the columns of the contiguous vertex range
[r*S, (r+1)*S) (`ptrs` rebased to 0, `indices` holding global ids) and the seed batches it was given.  Per
hop, every rank
  1. buckets its frontier by owning rank (stable, so each bucket stays in frontier order),
  2. exchanges bucket sizes, then the requests (vertex, call id, slot) -- two all-to-alls,
  3. samples the requests it received with the SAME draws the requester would have used (the kernel's
     remote-frontier mode: draw id = requester's slot, call id = requester's batch), so results equal the
     replicated-graph sampler bit for bit,
  4. returns per-request counts and (neighbour id, global edge pointer) pairs -- two more all-to-alls,
  5. reassembles them in slot order, which is the reference's output order
     (src/algo/neighbor_sampling.rs:195-218).
xGMI is point-to-point and these messages are small (<= 16 B per request, 16 B per sample), so the exchange is
latency-bound: all batches of a call travel in ONE set of collectives per hop.  With the "nccl" backend (RCCL)
device tensors go straight into all_to_all_single; with "gloo" (tests) they are staged through the host.

`ns_homo_partitioned_device` is the device form (csrc/partition.hip kernels, fixed-stride replies, ordinary per-batch
output slabs); `ns_homo_partitioned` below is the same protocol spelled in torch operations, kept because it also
runs on CPU tensors with any owner-side sampler (the gloo tests use the oracle there).
Only the unweighted, unfiltered samplers are partitioned in this round.
"""
import torch
import torch.distributed as dist

from . import _cabi

SAMPLER_UNIFORM, SAMPLER_UNIFORM_REPL = _cabi.SAMPLER_UNIFORM, _cabi.SAMPLER_UNIFORM_REPL


class CscShard:
    """Columns [v_lo, v_hi) of a CSC: ptrs rebased to 0, indices = global row ids, e_lo = global edge offset."""

    def __init__(self, ptrs, indices, v_lo, v_hi, e_lo, n_nodes, shard_size):
        self.ptrs, self.indices = ptrs, indices
        self.v_lo, self.v_hi, self.e_lo = int(v_lo), int(v_hi), int(e_lo)
        self.n_nodes, self.shard_size = int(n_nodes), int(shard_size)
        self._view = None

    def graph_view(self):
        if self._view is None:
            self._view = _cabi.graph_view(self.ptrs, self.indices)
        return self._view

    @staticmethod
    def shard_size_for(n_nodes, world):
        return (n_nodes + world - 1) // world

    @classmethod
    def from_full(cls, ptrs, indices, rank, world):
        """Cut rank's shard out of a replicated CSC (tests / benchmarks; a real loader reads only its shard)."""
        n = ptrs.numel() - 1
        size = cls.shard_size_for(n, world)
        lo, hi = min(rank * size, n), min((rank + 1) * size, n)
        e_lo, e_hi = int(ptrs[lo]), int(ptrs[hi])
        return cls((ptrs[lo:hi + 1] - e_lo).contiguous(), indices[e_lo:e_hi].contiguous(), lo, hi, e_lo, n, size)


def _world(group):
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group), dist.get_rank(group)
    return 1, 0


def _all_to_all_rows(send, send_counts, recv_counts, group):
    """all_to_all_v over dim 0 of `send` ([M] or [M, C] int64); counts are python lists of rows per peer."""
    world, _ = _world(group)
    tail = tuple(send.shape[1:])
    n_recv = int(sum(recv_counts))
    if world == 1:
        return send.clone()
    if dist.get_backend(group) == "gloo":  # host staging (CPU tests); RCCL takes the device tensors directly
        s = send.cpu().contiguous()
        r = torch.empty((n_recv,) + tail, dtype=send.dtype)
        dist.all_to_all_single(r, s, output_split_sizes=list(recv_counts), input_split_sizes=list(send_counts),
                               group=group)
        return r.to(send.device)
    r = torch.empty((n_recv,) + tail, dtype=send.dtype, device=send.device)
    dist.all_to_all_single(r, send.contiguous(), output_split_sizes=list(recv_counts),
                           input_split_sizes=list(send_counts), group=group)
    return r


def _exchange_counts(counts, group):
    world, _ = _world(group)
    c = torch.as_tensor(counts, dtype=torch.int64)
    return [int(x) for x in _all_to_all_rows(c.reshape(world, 1), [1] * world, [1] * world, group).reshape(-1)] \
        if world > 1 else list(counts)


def _hip_hop(shard, local_vertices, call_ids, slot_ids, k, seed, sampler):
    """Owner side of one hop on the GPU (tg_ns_hop, whole-device flat hop):
    -> (cnt per request, neighbour ids, LOCAL edge pointers), in request order."""
    m = local_vertices.numel()
    if m == 0:
        z = torch.zeros(0, dtype=torch.int64, device=local_vertices.device)
        return z, z, z
    cnt, offsets, nbr, ep, _ = _cabi.ns_hop(shard.graph_view(), local_vertices.contiguous(), k, seed, sampler=sampler,
                                            ids=slot_ids.contiguous(), call_ids=call_ids.contiguous())
    total = int(offsets[m])  # size read-back
    return cnt, nbr[:total], ep[:total]


def _ragged_gather_index(src_start, cnt):
    """index of every element of ragged rows given each row's start in the source and its length"""
    total = int(cnt.sum())
    if total == 0:
        return torch.zeros(0, dtype=torch.int64, device=cnt.device)
    row = torch.repeat_interleave(torch.arange(cnt.numel(), device=cnt.device), cnt)
    dst_start = torch.cumsum(cnt, 0) - cnt
    return src_start[row] + (torch.arange(total, device=cnt.device) - dst_start[row])


def _a2a_flat(send, send_rows, recv_rows, row_len, group):
    """all-to-all of a flat int64 buffer made of rows of `row_len` words; *_rows: rows per peer (python lists)"""
    return _all_to_all_rows(send.reshape(-1, row_len), send_rows, recv_rows, group).reshape(-1)


def ns_homo_partitioned_device(shard, seeds, fanout, seed, first_call_id, sampler=SAMPLER_UNIFORM, group=None):
    """The device form of the exchange (csrc/partition.hip): the origin keeps the ordinary per-batch slabs of
    tg_ns_homo_batched; per hop  tg_part_requests -> all-to-all -> tg_part_sample (owner, fixed-stride replies, so no
    reply sizes are exchanged) -> all-to-all -> tg_part_emit.  The host reads only the bucket sizes (world integers)
    per hop.  Returns an `_cabi.NsBatchedOut` whose contents equal the replicated-graph sampler's bit for bit."""
    import ctypes as C
    lib, ptr = _cabi.lib, _cabi.ptr
    world, rank = _world(group)
    dev = seeds.device
    nb, B = seeds.shape
    H = len(fanout)
    out = _cabi.NsBatchedOut(nb, B, fanout, dev)
    so = out.struct()
    stream = _cabi.stream_ptr(dev)
    nbytes = C.c_int64(0)
    _cabi.check(lib.tg_part_workspace_bytes(C.c_int64(nb), C.c_int32(world), C.byref(nbytes)))
    ws = torch.zeros(nbytes.value // 8, dtype=torch.int64, device=dev)
    seeds = seeds.contiguous()
    _cabi.check(lib.tg_part_begin(ptr(seeds), C.c_int64(nb), C.c_int64(B), C.byref(so), ptr(ws), stream))
    hist_at = 4 * nb + nb + 1                          # bucket_sizes[world] inside the workspace (include/tchgeo.h)
    graph = shard.graph_view()
    cap = nb * B                                       # worst-case frontier of the hop
    for h, k in enumerate(fanout):
        req = torch.empty(cap * 3, dtype=torch.int64, device=dev)
        req_pos = torch.empty(cap, dtype=torch.int64, device=dev)
        _cabi.check(lib.tg_part_requests(C.byref(so), C.c_int64(nb), C.c_int64(cap), C.c_int64(shard.shard_size),
                                         C.c_int32(world), C.c_uint64(first_call_id), ptr(ws), ptr(req), ptr(req_pos), stream))
        send = ws[hist_at:hist_at + world].tolist()    # the hop's only read-back: requests per owner
        m_send = int(sum(send))
        if world > 1:
            recv = _exchange_counts(send, group)
            got = _a2a_flat(req[:m_send * 3], send, recv, 3, group)
        else:
            recv, got = send, req
        m_recv = int(sum(recv))
        reply = torch.empty(max(m_recv, 1) * k * 2, dtype=torch.int64, device=dev)
        _cabi.check(lib.tg_part_sample(C.byref(graph), C.c_int64(shard.v_lo), C.c_int64(shard.e_lo), ptr(got),
                                       C.c_int64(m_recv), C.c_int32(int(k)), C.c_int32(sampler), C.c_uint64(seed),
                                       ptr(reply), stream))
        back = _a2a_flat(reply[:m_recv * k * 2], recv, send, 2 * k, group) if world > 1 else reply
        if back.numel() == 0:
            back = torch.empty(2, dtype=torch.int64, device=dev)
        _cabi.check(lib.tg_part_emit(C.byref(so), C.c_int64(nb), C.c_int64(B), C.c_int32(int(k)), C.c_int32(h), C.c_int32(H),
                                     ptr(ws), ptr(req_pos), ptr(back), stream))
        cap *= k
    if H == 0:
        out.counts[:, 0] = B
    return out


def ns_homo_partitioned(shard, seeds, fanout, seed, first_call_id, sampler=SAMPLER_UNIFORM, group=None,
                        _hop_fn=None):
    """seeds: [n_batches, B] int64 on the shard's device; batch j of this rank has call id first_call_id + j.

    Returns a list of per-batch (samples, rows, cols, edge_index, layer_offsets), equal to what the
    replicated-graph sampler returns for the same (seed, call id).  Everything stays flat (batch-major) on the
    device until one final scatter; per hop the host only learns the all-to-all split sizes."""
    hop_fn = _hop_fn or _hip_hop
    world, rank = _world(group)
    dev = seeds.device
    nb, B = seeds.shape
    i64 = dict(dtype=torch.int64, device=dev)
    # frontier, batch-major: vertex, batch, slot (= index in the batch's sample list = draw id)
    f_vertex = seeds.reshape(-1)
    f_batch = torch.repeat_interleave(torch.arange(nb, **i64), B)
    f_slot = torch.arange(B, **i64).repeat(nb)
    n_edges = torch.zeros(nb, **i64)
    ne_at_hop, hops = [], []          # per hop: edges before it (per batch); (batch, slot, neighbour, parent slot, edge ptr)
    for k in fanout:
        ne_at_hop.append(n_edges)
        m = f_vertex.numel()
        if world > 1:
            # ---- 1. bucket by owner (stable: buckets keep frontier order)
            owner = torch.clamp(f_vertex // shard.shard_size, max=world - 1)
            perm = torch.argsort(owner, stable=True)
            send_counts = torch.bincount(owner, minlength=world).tolist()
            req = torch.stack([f_vertex[perm], first_call_id + f_batch[perm], f_slot[perm]], dim=1)
            # ---- 2. sizes, then requests
            recv_counts = _exchange_counts(send_counts, group)
            got = _all_to_all_rows(req, send_counts, recv_counts, group)
            r_vertex, r_call, r_slot = got[:, 0], got[:, 1], got[:, 2]
        else:
            r_vertex, r_call, r_slot = f_vertex, first_call_id + f_batch, f_slot
        # ---- 3. sample what this rank owns, with the requester's draws
        cnt_r, nbr_r, ep_r = hop_fn(shard, r_vertex - shard.v_lo, r_call, r_slot, int(k), seed, sampler)
        ep_r = ep_r + shard.e_lo
        if world > 1:
            # ---- 4. replies: per-request counts, then (neighbour, global edge pointer) rows
            cnt_sorted = _all_to_all_rows(cnt_r, recv_counts, send_counts, group)
            peer_of_req = torch.repeat_interleave(torch.arange(world, **i64), torch.as_tensor(recv_counts, **i64))
            rep_send = torch.zeros(world, **i64).index_add_(0, peer_of_req, cnt_r).tolist()
            rep_recv = _exchange_counts(rep_send, group)
            data_sorted = _all_to_all_rows(torch.stack([nbr_r, ep_r], dim=1), rep_send, rep_recv, group)
            # ---- 5. back to frontier (slot) order
            cnt_f = torch.empty_like(cnt_sorted)
            cnt_f[perm] = cnt_sorted
            start_sorted = torch.cumsum(cnt_sorted, 0) - cnt_sorted
            start_f = torch.empty_like(start_sorted)
            start_f[perm] = start_sorted
            gidx = _ragged_gather_index(start_f, cnt_f)
            new_nbr, new_ep = data_sorted[gidx, 0], data_sorted[gidx, 1]
        else:
            cnt_f, new_nbr, new_ep = cnt_r, nbr_r, ep_r
        parent_slot = torch.repeat_interleave(f_slot, cnt_f, output_size=new_nbr.numel())
        new_batch = torch.repeat_interleave(f_batch, cnt_f, output_size=new_nbr.numel())
        per_batch = torch.zeros(nb, **i64).index_add_(0, f_batch, cnt_f)
        # new samples of batch j occupy slots B + n_edges[j] ... in emission order (batches are contiguous blocks)
        first_new = torch.cumsum(per_batch, 0) - per_batch
        new_slot = B + n_edges[new_batch] + (torch.arange(new_batch.numel(), **i64) - first_new[new_batch])
        hops.append((new_batch, new_slot, new_nbr, parent_slot, new_ep))
        n_edges = n_edges + per_batch
        f_vertex, f_batch, f_slot = new_nbr, new_batch, new_slot
        del m
    # ---- one scatter into flat, batch-major outputs
    node_base = torch.cumsum(B + n_edges, 0) - (B + n_edges)
    edge_base = torch.cumsum(n_edges, 0) - n_edges
    total_nodes, total_edges = int((B + n_edges).sum()), int(n_edges.sum())
    samples = torch.empty(total_nodes, **i64)
    cols, eidx = torch.empty(total_edges, **i64), torch.empty(total_edges, **i64)
    rows = torch.empty(total_edges, **i64)
    samples[(node_base[:, None] + torch.arange(B, **i64)[None, :]).reshape(-1)] = seeds.reshape(-1)
    for new_batch, new_slot, new_nbr, parent_slot, new_ep in hops:
        samples[node_base[new_batch] + new_slot] = new_nbr
        e = edge_base[new_batch] + (new_slot - B)
        rows[e] = new_slot                                  # rows[e] = n_seeds + e (neighbor_sampling.rs:212-217)
        cols[e] = parent_slot
        eidx[e] = new_ep
    ns_host, ne_host = (B + n_edges).tolist(), n_edges.tolist()
    hop_host = torch.stack(ne_at_hop).tolist() if ne_at_hop else []
    s_parts, r_parts = torch.split(samples, ns_host), torch.split(rows, ne_host)
    c_parts, e_parts = torch.split(cols, ne_host), torch.split(eidx, ne_host)
    out = []
    for j in range(nb):
        lo = [(B + h[j], h[j], B + h[j]) for h in hop_host]           # neighbor_sampling.rs:193
        out.append((s_parts[j], r_parts[j], c_parts[j], e_parts[j], lo))
    return out
