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

`PartitionedSampler` is the device form (csrc/partition.hip kernels, compact replies, ordinary per-batch output slabs,
buffers allocated once); `ns_homo_partitioned` below is the same protocol spelled in torch operations, kept because it
also runs on CPU tensors with any owner-side sampler (the gloo tests use the oracle there).
"""
import torch
import torch.distributed as dist

from . import _cabi

SAMPLER_UNIFORM, SAMPLER_UNIFORM_REPL = _cabi.SAMPLER_UNIFORM, _cabi.SAMPLER_UNIFORM_REPL


class CscShard:
    """Columns [v_lo, v_hi) of a CSC: ptrs rebased to 0, indices = global row ids, e_lo = global edge offset."""

    def __init__(self, ptrs, indices, v_lo, v_hi, e_lo, n_nodes, shard_size, weights=None, timestamps=None,
                 n_edges_global=None):
        self.ptrs, self.indices = ptrs, indices
        self.n_edges_global = None if n_edges_global is None else int(n_edges_global)   # the whole graph's edge count
        self.weights, self.timestamps = weights, timestamps      # this shard's slices of the per-edge attributes
        self.v_lo, self.v_hi, self.e_lo = int(v_lo), int(v_hi), int(e_lo)
        self.n_nodes, self.shard_size = int(n_nodes), int(shard_size)
        self._view = None

    def graph_view(self):
        if self._view is None:   # u32 shadow of the neighbour ids: half the bytes per gathered line (ids < 2^32)
            i32 = self.indices.to(torch.int32) if self.n_nodes < 2 ** 32 and self.indices.is_cuda else None
            # ... and of the column starts (a shard of < 2^32 edges; the kernels read the words as unsigned)
            p32 = self.ptrs.to(torch.int32) if self.indices.numel() < 2 ** 32 and self.ptrs.is_cuda else None
            self._view = _cabi.graph_view(self.ptrs, self.indices, weights=self.weights, timestamps=self.timestamps,
                                          indices32=i32, ptrs32=p32)
        return self._view

    @staticmethod
    def shard_size_for(n_nodes, world):
        return (n_nodes + world - 1) // world

    @classmethod
    def from_full(cls, ptrs, indices, rank, world, weights=None, timestamps=None):
        """Cut rank's shard out of a replicated CSC (tests / benchmarks; a real loader reads only its shard)."""
        n = ptrs.numel() - 1
        size = cls.shard_size_for(n, world)
        lo, hi = min(rank * size, n), min((rank + 1) * size, n)
        e_lo, e_hi = int(ptrs[lo]), int(ptrs[hi])
        cut = lambda a: a[e_lo:e_hi].contiguous() if a is not None else None
        return cls((ptrs[lo:hi + 1] - e_lo).contiguous(), indices[e_lo:e_hi].contiguous(), lo, hi, e_lo, n, size,
                   weights=cut(weights), timestamps=cut(timestamps), n_edges_global=int(ptrs[n]))


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


def _exchange_counts(counts, group, device=None):
    """all-to-all of one integer per peer.  `counts`: python list or a [world] int64 tensor.  RCCL ("nccl") only takes
    device tensors, so the counts travel on `device` there; gloo stages them on the host.  Returns a python list
    (one read-back)."""
    world, _ = _world(group)
    if world == 1:
        return [int(x) for x in (counts.tolist() if torch.is_tensor(counts) else counts)]
    c = counts if torch.is_tensor(counts) else torch.as_tensor(counts, dtype=torch.int64)
    if dist.get_backend(group) != "gloo":
        if device is None or torch.device(device).type == "cpu":
            raise ValueError("the %s backend needs the compute device for the size exchange" % dist.get_backend(group))
        c = c.to(device)
    return [int(x) for x in _all_to_all_rows(c.reshape(world, 1), [1] * world, [1] * world, group).reshape(-1).tolist()]


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


class PartitionedSampler:
    """The device form of the exchange (csrc/partition.hip) with every buffer allocated once and reused across calls.

    Per hop, plain samplers (slot replies): tg_part_requests -> [sizes, requests all-to-all] -> tg_part_sample_slots (owner:
    one fixed-size packed slot per request) -> [slots all-to-all, the request sizes mirrored] -> tg_part_emit_slots.
    Per hop, filters / weights / wide fan-outs (compact replies): tg_part_requests -> [sizes, requests all-to-all] ->
    tg_part_count / tg_part_sample (owner: per-request counts + (neighbour, global edge pointer) entries) -> [counts,
    reply sizes, replies all-to-all] -> tg_part_emit.  Every size lives on the device; with world == 1 the plain samplers read nothing back at all (filters /
    weights: one status word when the call ends), with world > 1 the host reads only the all-to-all split sizes (one small
    read-back per hop with slot replies, two with compact ones).  The returned `_cabi.NsBatchedOut` equals
    the replicated-graph sampler's bit for bit."""

    def __init__(self, shard, n_batches, n_seeds, fanout, sampler=SAMPLER_UNIFORM, group=None,
                 filter_mode=_cabi.FILTER_NONE, forward=False, window=(0, 0), force_exchange=False,
                 packed_replies=None, slot_replies=None):
        """sampler: uniform / with replacement / weighted (shard.weights); filter_mode: a TemporalFilter mode over
        shard.timestamps (`sample()` then takes the seeds' filter states).  Filters and weights take the general
        owner path: tg_part_unpack -> tg_ns_hop_scan / tg_ns_hop_weighted -> tg_part_pack.
        force_exchange: with ONE rank, still run every collective of the multi-rank protocol (sizes, requests, counts,
        replies travel through all_to_all_single to the rank itself) -- this is how the RCCL transport is exercised on
        a one-GPU box.
        packed_replies: a reply entry is one word (neighbour | global edge pointer << 32) instead of two -- half the bytes
        of the largest all-to-all.  None = whenever the graph allows it (shard.n_nodes and shard.n_edges_global < 2^32);
        every rank must pass the same value.
        slot_replies: the owner answers every request with one fixed-size packed slot (csrc/partition_slots.inl) instead
        of a compact reply: no count / prefix passes, whole-chunk writes, and ONE size read-back per hop instead of two
        (the reply exchange mirrors the request exchange's split sizes).  None = whenever it applies -- unweighted,
        unfiltered sampling, fan-outs <= 16, every shard below 2^32 edges, a slot of <= 128 bytes; the ranks agree on
        that (and on the slot's bit widths) in one small all-reduce at construction.  True raises where it does not
        apply; every rank must pass the same value."""
        import ctypes as C
        self.C, self.shard, self.group, self.sampler = C, shard, group, sampler
        self.filter_mode, self.forward, self.window = filter_mode, bool(forward), tuple(window)
        self.filtered = filter_mode != _cabi.FILTER_NONE
        self.general = self.filtered or sampler == _cabi.SAMPLER_WEIGHTED
        if sampler == _cabi.SAMPLER_WEIGHTED and shard.weights is None:
            raise ValueError("the weighted sampler needs shard.weights")
        if self.filtered and shard.timestamps is None:
            raise ValueError("a temporal filter needs shard.timestamps")
        self.world, self.rank = _world(group)
        self.nb, self.B, self.fanout = int(n_batches), int(n_seeds), [int(k) for k in fanout]
        self.dev = shard.ptrs.device
        self.hop_cap, cap = [], self.nb * self.B
        for k in self.fanout:
            self.hop_cap.append(cap)
            cap *= k
        self.request_cap = max(self.hop_cap + [1])
        self.out = _cabi.NsBatchedOut(self.nb, self.B, self.fanout, self.dev, with_states=self.filtered)
        nbytes = C.c_int64(0)
        _cabi.check(_cabi.lib.tg_part_workspace_bytes(C.c_int64(self.nb), C.c_int64(self.request_cap),
                                                      C.c_int32(self.world), C.byref(nbytes)))
        i64 = dict(dtype=torch.int64, device=self.dev)
        self.ws = torch.empty(nbytes.value // 8 + 1, **i64)
        self.requests = torch.empty((self.request_cap, 2), **i64)            # 16-byte requests
        self.request_states = torch.empty(self.request_cap, **i64) if self.filtered else None
        self.send_counts = torch.zeros(self.world + 1, **i64)
        self.reply_counts = torch.zeros(self.world + 1, **i64)
        self._bufs = {}
        self._status_acc = torch.zeros(1, dtype=torch.int32, device=self.dev)   # OR of every hop's status word of a call
        self._group_mult = 1                                                     # column-group workspace guess, x8 on overflow
        self.exchange = self.world > 1 or bool(force_exchange)
        if force_exchange and not (dist.is_available() and dist.is_initialized()):
            raise ValueError("force_exchange needs an initialised process group")
        self.gloo = self.exchange and dist.get_backend(group) == "gloo"
        fits = shard.n_nodes < 2 ** 32 and shard.n_edges_global is not None and shard.n_edges_global < 2 ** 32
        if packed_replies and not fits:
            raise ValueError("packed replies need n_nodes and the graph's n_edges_global below 2^32")
        packed = fits if packed_replies is None else bool(packed_replies)
        self.reply_format = ((_cabi.PART_REPLY_PACKED_STATE if packed else _cabi.PART_REPLY_TRIPLES) if self.filtered
                             else (_cabi.PART_REPLY_PACKED if packed else _cabi.PART_REPLY_PAIRS))
        self.reply_words = 2 if self.reply_format == _cabi.PART_REPLY_PACKED_STATE else self.reply_format
        self._init_slots(slot_replies)

    def _init_slots(self, wanted):
        """slot replies: the format every rank agrees on -- vertex bits of the whole graph, position bits of its longest
        column, words per slot of every hop -- and the shards' global edge offsets"""
        C = self.C
        self.slots, self.slot_words = False, []
        if wanted is False:
            return
        shard = self.shard
        mine_ok = (not self.general and max(self.fanout + [0]) <= 16 and shard.indices.numel() < 2 ** 32
                   and shard.n_nodes <= 2 ** 32 and self.dev.type == "cuda")
        longest = _cabi.graph_max_degree(shard.graph_view(), self.dev) if mine_ok and shard.ptrs.numel() > 1 else 0
        word = torch.tensor([longest, 0 if mine_ok else 1, shard.e_lo], dtype=torch.int64)
        e_lo_of = [shard.e_lo]
        if self.exchange and self.world > 1:
            on = word if self.gloo else word.to(self.dev)
            parts = [torch.empty_like(on) for _ in range(self.world)]
            dist.all_gather(parts, on, group=self.group)
            allw = torch.stack(parts).cpu()
            word = torch.tensor([int(allw[:, 0].max()), int(allw[:, 1].max()), 0])
            e_lo_of = [int(x) for x in allw[:, 2].tolist()]
        ok = int(word[1]) == 0
        bits = lambda hi: max(1, int(hi).bit_length())        # bits that hold 0 .. hi
        self.slot_bv, self.slot_bp = bits(max(shard.n_nodes, 1) - 1), bits(max(int(word[0]), 1) - 1)
        words = []
        for k in self.fanout:
            w = C.c_int32(0)
            if ok:
                _cabi.check(_cabi.lib.tg_part_slot_words(C.c_int32(k), C.c_int32(self.slot_bv), C.c_int32(self.slot_bp),
                                                         C.byref(w)))
            words.append(w.value)
        ok = ok and all(w > 0 for w in words)
        if wanted and not ok:
            raise ValueError("slot replies do not apply here (filters / weights, a fan-out > 16, a shard of >= 2^32 edges, "
                             "or more than 1024 bits per slot)")
        self.slots, self.slot_words = ok, words
        self.e_lo_of = (C.c_int64 * 64)(*e_lo_of)

    def _buf(self, name, n, dtype, cols=None):
        """persistent scratch that only ever grows"""
        t = self._bufs.get(name)
        need = max(int(n), 1)
        if t is None or t.shape[0] < need:
            t = torch.empty((need,) if cols is None else (need, cols), dtype=dtype, device=self.dev)
            self._bufs[name] = t
        return t

    def _a2a(self, send, send_rows, recv_rows, name):
        """all_to_all_v over dim 0 into a persistent buffer (RCCL: device tensors; gloo: staged through the host)"""
        n_recv = int(sum(recv_rows))
        recv = self._buf(name, n_recv, send.dtype, send.shape[1] if send.dim() > 1 else None)[:n_recv]
        if self.gloo:
            r = torch.empty(recv.shape, dtype=send.dtype)
            dist.all_to_all_single(r, send.cpu().contiguous(), output_split_sizes=list(recv_rows),
                                   input_split_sizes=list(send_rows), group=self.group)
            recv.copy_(r)
        else:
            dist.all_to_all_single(recv, send.contiguous(), output_split_sizes=list(recv_rows),
                                   input_split_sizes=list(send_rows), group=self.group)
        return recv

    def _sizes(self, mine):
        """mine: [world] int64 device tensor -> (mine as a list, the peers' as a list): one read-back"""
        return self._sizes_end(self._sizes_begin(mine, 0))

    def _sizes_begin(self, mine, which):
        """Starts the size exchange: the collective and the copy of both rows to pinned host memory are ENQUEUED, nothing is
        waited for -- `steps` yields between _begin and _end, so the wait of one super-batch passes while the scheduler
        enqueues another's work.  (gloo stages through the host and is blocking by nature.)"""
        if self.gloo:
            theirs = torch.empty(self.world, dtype=torch.int64)
            m_host = mine.cpu()
            dist.all_to_all_single(theirs, m_host, group=self.group)
            return (m_host.tolist(), theirs.tolist())
        theirs = torch.empty_like(mine)
        dist.all_to_all_single(theirs, mine.contiguous(), group=self.group)
        key = "pinned_sizes_%d" % which
        pin = self._bufs.get(key)
        if pin is None:
            pin = self._bufs[key] = (torch.empty((2, self.world), dtype=torch.int64).pin_memory(), torch.cuda.Event())
        host, ev = pin
        host.copy_(torch.stack([mine, theirs]), non_blocking=True)
        ev.record(torch.cuda.current_stream(self.dev))
        return (host, ev)

    def _sizes_end(self, handle):
        if isinstance(handle[0], list):
            return handle
        host, ev = handle
        ev.synchronize()
        both = host.tolist()
        return both[0], both[1]

    def _owner_general(self, got, got_states, m_dev, m_cap, seg, call0, k, seed, stream):
        """owner side under a filter / with weights: requests -> flat-hop arrays -> tg_ns_hop_scan / _weighted -> reply"""
        C, lib, ptr = self.C, _cabi.lib, _cabi.ptr
        shard, graph = self.shard, self.shard.graph_view()
        i64 = torch.int64
        vert, ids, calls = (self._buf(n, m_cap, i64) for n in ("g_vert", "g_ids", "g_calls"))
        _cabi.check(lib.tg_part_unpack(C.c_int64(shard.v_lo), C.c_int64(shard.v_hi - shard.v_lo), ptr(got), ptr(m_dev),
                                       C.c_int64(m_cap), C.c_int32(self.world), seg, call0, ptr(vert), ptr(ids), ptr(calls),
                                       stream))
        hcnt, hoff = self._buf("g_cnt", m_cap, i64), self._buf("g_off", m_cap + 1, i64)
        nbr, ep, par, st_out = (self._buf(n, m_cap * k, i64) for n in ("g_nbr", "g_ep", "g_par", "g_st"))
        status = self._buf("g_status", 1, torch.int32)
        hin, hout, flt = _cabi.TgHopIn(), _cabi.TgHopOut(), _cabi.TgHopFilter()
        hin.vertices, hin.ids, hin.call_ids = vert.data_ptr(), ids.data_ptr(), calls.data_ptr()
        hin.m, hin.id_base, hin.fanout, hin.sampler, hin.rng_tag = m_cap, 0, k, self.sampler, 0
        hout.cnt, hout.offsets = hcnt.data_ptr(), hoff.data_ptr()
        hout.neighbors, hout.edge_ptrs, hout.parents = nbr.data_ptr(), ep.data_ptr(), par.data_ptr()
        flt.filter_mode, flt.forward = self.filter_mode, int(self.forward)
        flt.win_lo, flt.win_hi = self.window
        flt.states = got_states.data_ptr() if self.filtered else None
        rng = _cabi.TgRng(seed, 0)
        nbytes = C.c_int64(0)
        weighted = self.sampler == _cabi.SAMPLER_WEIGHTED
        # the column-group workspace is a guess (a vertex requested many times counts its groups every time); a hop whose
        # guess was too low raises status bit 1 and samples nothing -- sample() sees it in the call's ONE read-back,
        # enlarges the guess (kept for later calls) and runs the call again: no read-back per hop
        group_cap = max(1, int(self._group_mult * max(1024, graph.n_edges // 512 + 2 * m_cap + 2)))
        size_of = lib.tg_ns_hop_weighted_workspace_bytes if weighted else lib.tg_ns_hop_scan_workspace_bytes
        _cabi.check(size_of(C.c_int64(m_cap), C.c_int32(k), C.c_int64(group_cap),
                                                       C.byref(nbytes)))
        ws = self._buf("g_ws", nbytes.value // 8 + 1, i64)
        status.zero_()
        if weighted:
            _cabi.check(lib.tg_ns_hop_weighted_groups(C.byref(graph), C.byref(hin), C.byref(flt), C.byref(rng), C.byref(hout),
                                                      ptr(st_out), ptr(status), ptr(ws), C.c_int64(nbytes.value),
                                                      C.c_int64(group_cap), stream))
        else:
            _cabi.check(lib.tg_ns_hop_scan(C.byref(graph), C.byref(hin), C.byref(flt), C.byref(rng), C.byref(hout),
                                           ptr(st_out), ptr(status), ptr(ws), C.c_int64(nbytes.value), C.c_int64(group_cap),
                                           stream))
        self._status_acc.bitwise_or_(status)          # on the device: every hop's word survives to the end of the call
        cnt = self._buf("cnt", m_cap, torch.int32)
        reply = self._buf("reply", m_cap * k, i64, self.reply_words)
        _cabi.check(lib.tg_part_pack(C.byref(hout), ptr(st_out) if self.filtered else None, ptr(m_dev), C.c_int64(m_cap),
                                     C.c_int64(shard.e_lo), C.c_int32(self.world), seg, ptr(cnt), ptr(reply),
                                     C.c_int32(self.reply_format), ptr(self.reply_counts), stream))
        return cnt, hoff, reply

    def sample(self, seeds, seed, first_call_id, first_call_ids=None, seeds_state=None):
        """seeds: [n_batches, n_seeds] int64 on the shard's device; batch j draws with call id first_call_id + j.
        first_call_ids: every rank's first call id (list), if the caller knows them; else they are all-gathered.
        seeds_state: [n_batches, n_seeds] filter states of the seeds (with a temporal filter).

        Unweighted, unfiltered sampling reads nothing back (world == 1) -- it has no way to fail.  Under a filter / with
        weights the owner-side hops report through a status word that is OR-ed on the device over ALL hops of the call and
        read back once when the call ends (all ranks agree on it first: every rank raises or repeats together):
        bit 2 = a non-positive running weight sum in ANY hop -> RuntimeError (the reference panics, sampling.rs:49);
        bit 1 = a hop's column-group workspace was too small -> the guess grows 8x and the call runs again."""
        if not self.general:
            return self._sample_once(seeds, seed, first_call_id, first_call_ids, seeds_state)
        while True:
            self._status_acc.zero_()
            out = self._sample_once(seeds, seed, first_call_id, first_call_ids, seeds_state)
            word = self._agreed_status()
            if word & 2:
                raise RuntimeError("weighted sampling met a non-positive running weight sum (the reference panics here)")
            if not word & 1:
                return out
            if self._group_mult >= 4096:
                raise RuntimeError("tg_ns_hop_scan: column-group workspace overflow that enlarging does not cure")
            self._group_mult *= 8

    def _agreed_status(self):
        """the call's status word, OR-ed over the ranks (as a MAX per bit: RCCL has no bitwise reductions); one read-back"""
        bits = torch.stack([self._status_acc[0] & 1, (self._status_acc[0] >> 1) & 1]).to(torch.int64)
        if self.exchange and self.world > 1:
            if self.gloo:
                host = bits.cpu()
                dist.all_reduce(host, op=dist.ReduceOp.MAX, group=self.group)
                bits = host
            else:
                dist.all_reduce(bits, op=dist.ReduceOp.MAX, group=self.group)
        b = bits.tolist()
        return int(b[0]) | (int(b[1]) << 1)

    def _sample_once(self, seeds, seed, first_call_id, first_call_ids=None, seeds_state=None):
        for _ in self.steps(seeds, seed, first_call_id, first_call_ids, seeds_state):
            pass
        return self.out

    def steps(self, seeds, seed, first_call_id, first_call_ids=None, seeds_state=None):
        """One call as a GENERATOR: it yields right before every blocking size read-back (with an exchange, per hop: the
        request split sizes, and -- compact replies only -- the reply split sizes; slot replies mirror the request sizes),
        having enqueued everything up to there.  A scheduler that drives
        several samplers' generators in a fixed order from one host thread (`interleave`) keeps the GPU busy with one
        super-batch while the host waits for another's sizes.  Run to exhaustion it is `_sample_once`."""
        C, lib, ptr = self.C, _cabi.lib, _cabi.ptr
        world, nb, B, H = self.world, self.nb, self.B, len(self.fanout)
        assert tuple(seeds.shape) == (nb, B) and seeds.device == self.dev
        stream = _cabi.stream_ptr(self.dev)   # the stream current when the generator STARTS: a scheduler keeps it per lane
        so = self.out.struct()
        seeds = seeds.contiguous()
        if self.exchange and first_call_ids is None:
            mine = torch.tensor([first_call_id], dtype=torch.int64, device="cpu" if self.gloo else self.dev)
            parts = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(parts, mine, group=self.group)
            first_call_ids = [int(x) for x in torch.cat(parts).tolist()]
        call0 = (C.c_uint64 * 64)(*(first_call_ids if self.exchange else [first_call_id]))
        if self.filtered:
            assert seeds_state is not None and tuple(seeds_state.shape) == (nb, B)
            seeds_state = seeds_state.contiguous()
        _cabi.check(lib.tg_part_begin(ptr(seeds), ptr(seeds_state) if self.filtered else None, C.c_int64(nb), C.c_int64(B),
                                      C.c_int32(H), C.byref(so), C.c_int64(self.request_cap), C.c_int32(world),
                                      ptr(self.ws), stream))
        graph = self.shard.graph_view()
        shard = self.shard
        for h, k in enumerate(self.fanout):
            cap = self.hop_cap[h]
            _cabi.check(lib.tg_part_requests(C.byref(so), C.c_int64(nb), C.c_int64(self.request_cap),
                                             C.c_int64(shard.shard_size), C.c_int32(world), ptr(self.ws),
                                             ptr(self.requests), ptr(self.request_states) if self.filtered else None,
                                             ptr(self.send_counts), stream))
            got_states = self.request_states
            if not self.exchange:   # nothing travels and nothing is read back: sizes stay on the device
                got, m_cap, m_dev = self.requests, cap, self.send_counts[1:]
                seg = (C.c_int64 * 65)(0, cap)
            else:
                pending = self._sizes_begin(self.send_counts[:world], 0)
                yield ("request sizes", h)
                send, recv = self._sizes_end(pending)
                got = self._a2a(self.requests[:int(sum(send))], send, recv, "req_recv")
                if self.filtered:
                    got_states = self._a2a(self.request_states[:int(sum(send))], send, recv, "st_recv")
                m_cap = int(sum(recv))
                m_dev = self._buf("m_dev", 1, torch.int64)
                m_dev.fill_(m_cap)
                off_l, acc = [], 0
                for r in recv:
                    off_l.append(acc)
                    acc += r
                seg = (C.c_int64 * 65)(*(off_l + [acc]))
            if self.slots:
                # one fixed-size slot per request, written at the request's index: the reply exchange mirrors the request
                # exchange's split sizes -- nothing to read back
                W = self.slot_words[h]
                slots = self._buf("slots%d" % W, m_cap, torch.int64, W // 2)
                ws_bytes = C.c_int64(0)
                _cabi.check(lib.tg_part_sample_workspace_bytes(C.c_int64(m_cap), C.byref(ws_bytes)))
                sws = self._buf("sample_ws", ws_bytes.value // 8 + 64, torch.int64)
                _cabi.check(lib.tg_part_sample_slots(C.byref(graph), C.c_int64(shard.v_lo), ptr(got), ptr(m_dev),
                                                     C.c_int64(m_cap), C.c_int32(world), seg, call0, C.c_int32(k),
                                                     C.c_int32(self.sampler), C.c_uint64(seed), C.c_int32(self.slot_bv),
                                                     C.c_int32(self.slot_bp), C.c_int32(4 if h == 0 else 1), ptr(slots), ptr(sws),
                                                     C.c_int64(sws.numel() * 8), stream))
                if self.exchange:
                    slots_back = self._a2a(slots[:m_cap], recv, send, "slots_recv%d" % W)
                    if slots_back.numel() == 0:
                        slots_back = self._buf("slots_recv%d" % W, 1, torch.int64, W // 2)
                else:
                    slots_back = slots
                _cabi.check(lib.tg_part_emit_slots(C.byref(so), C.c_int64(nb), C.c_int64(B), C.c_int64(self.request_cap),
                                                   C.c_int32(world), C.c_int32(k), C.c_int32(h), C.c_int32(H), ptr(self.ws),
                                                   ptr(slots_back), C.c_int32(self.slot_bv), C.c_int32(self.slot_bp),
                                                   self.e_lo_of, stream))
                continue
            if self.general:
                cnt, off, reply = self._owner_general(got, got_states, m_dev, m_cap, seg, call0, k, seed, stream)
            else:
                cnt, off, reply = self._owner_uniform(graph, got, m_dev, m_cap, seg, call0, k, seed, stream)
            if not self.exchange:
                cnt_back, reply_back = cnt, reply
            else:
                pending = self._sizes_begin(self.reply_counts[:world], 1)
                yield ("reply sizes", h)
                rc_send, rc_recv = self._sizes_end(pending)
                cnt_back = self._buf("cnt_back", self.request_cap, torch.int32)
                n_back = int(sum(send))
                cnt_back[:n_back] = self._a2a(cnt[:m_cap], recv, send, "cnt_recv")
                reply_back = self._a2a(reply[:int(sum(rc_send))], rc_send, rc_recv, "reply_recv")
                if reply_back.numel() == 0:
                    reply_back = self._buf("reply_recv", 1, torch.int64, self.reply_words)
            _cabi.check(lib.tg_part_emit(C.byref(so), C.c_int64(nb), C.c_int64(B), C.c_int64(self.request_cap),
                                         C.c_int64(cap), C.c_int32(world), C.c_int32(k), C.c_int32(h), C.c_int32(H),
                                         ptr(self.ws), ptr(cnt_back), None if self.exchange else ptr(off), ptr(reply_back),
                                         C.c_int32(self.reply_format), stream))

    def _owner_uniform(self, graph, got, m_dev, m_cap, seg, call0, k, seed, stream):
        """owner side, unweighted and unfiltered: the dedicated count / sample kernels of csrc/partition.hip"""
        C, lib, ptr = self.C, _cabi.lib, _cabi.ptr
        shard, world = self.shard, self.world
        cnt = self._buf("cnt", m_cap, torch.int32)
        off = self._buf("off", m_cap + 1, torch.int64)
        reply = self._buf("reply", m_cap * k, torch.int64, self.reply_words)
        tmp_bytes = C.c_int64(0)
        _cabi.check(lib.tg_part_scan_workspace_bytes(C.c_int64(m_cap), C.byref(tmp_bytes)))
        tmp = self._buf("scan_tmp", tmp_bytes.value // 8 + 1, torch.int64)
        _cabi.check(lib.tg_part_count(C.byref(graph), C.c_int64(shard.v_lo), ptr(got), ptr(m_dev), C.c_int64(m_cap),
                                      C.c_int32(world), seg, call0, C.c_int32(k), C.c_int32(self.sampler), ptr(cnt),
                                      ptr(off), ptr(self.reply_counts), ptr(tmp), C.c_int64(tmp.numel() * 8), stream))
        ws_bytes = C.c_int64(0)   # many requests against a large shard: sampled in window order (tg_part_sample_ws)
        _cabi.check(lib.tg_part_sample_workspace_bytes(C.c_int64(m_cap), C.byref(ws_bytes)))
        sws = self._buf("sample_ws", ws_bytes.value // 8 + 64, torch.int64)
        _cabi.check(lib.tg_part_sample_ws(C.byref(graph), C.c_int64(shard.v_lo), C.c_int64(shard.e_lo), ptr(got),
                                          ptr(m_dev), C.c_int64(m_cap), C.c_int32(world), seg, call0, C.c_int32(k),
                                          C.c_int32(self.sampler), C.c_uint64(seed), ptr(cnt), ptr(off), ptr(reply),
                                          C.c_int32(self.reply_format), ptr(sws), C.c_int64(sws.numel() * 8), stream))
        return cnt, off, reply


def interleave(jobs, lanes, start, finish=None):
    """Drives `jobs` generators, at most `lanes` at a time, from ONE host thread in a FIXED order: job i runs on lane
    i % lanes; every tick advances the active lanes 0, 1, ... by one step each (a step = up to the generator's next yield, i.e.
    up to its next blocking read-back).  The order depends on nothing but (number of jobs, lanes, steps per job), which are the
    same on every rank -- so the collectives the steps enqueue on ONE communicator meet in the same order everywhere, which
    thread-per-lane schedules cannot promise (SURVEY.md 8(e) mode 2: the exchange of one super-batch under the sampling of
    the next).  start(i, lane) -> generator; finish(i, lane) is called when job i is exhausted, before its lane is reused."""
    active = [None] * lanes            # (job index, generator) per lane
    nxt = 0
    while nxt < jobs or any(a is not None for a in active):
        for lane in range(lanes):
            if active[lane] is None and nxt < jobs and nxt % lanes == lane:
                active[lane] = (nxt, start(nxt, lane))
                nxt += 1
            if active[lane] is None:
                continue
            i, gen = active[lane]
            try:
                next(gen)
            except StopIteration:
                if finish is not None:
                    finish(i, lane)
                active[lane] = None


class PipelinedPartitionedSampler:
    """`lanes` PartitionedSamplers (each with its own buffers, output slabs and HIP stream) driven by `interleave` over ONE
    communicator: while the host waits for the split sizes of one super-batch, the kernels and collectives of the other
    are already enqueued and run.  Legal with any number of ranks (round 3's thread-per-lane / communicator-per-lane form
    was verified for one rank only and refused beyond)."""

    def __init__(self, shard, n_batches, n_seeds, fanout, lanes=2, groups=None, **kw):
        """groups: optionally one process group (communicator) per lane.  With ONE communicator the lanes' collectives queue
        behind each other on its stream (a lane's tiny size exchange waits for the other lane's reply all-to-all and the
        kernels that one waits for); a communicator per lane lets them pass each other -- and stays deadlock-free here,
        because `interleave` enqueues every collective of every communicator in the same order on every rank (what
        thread-per-lane schedules could not promise)."""
        n_l = max(1, int(lanes))
        if groups is not None:
            assert len(groups) == n_l and "group" not in kw
        self.samplers = [PartitionedSampler(shard, n_batches, n_seeds, fanout, **(dict(kw, group=groups[j]) if groups else kw))
                         for j in range(n_l)]
        dev = shard.ptrs.device
        self.dev = dev
        self.streams = [torch.cuda.Stream(device=dev) for _ in self.samplers] if dev.type == "cuda" else [None] * len(self.samplers)

    def sample_many(self, n_jobs, seeds_of, seed, call_ids_of, consume, seeds_state_of=None):
        """Super-batch i: seeds_of(i) -> [n_batches, n_seeds] seeds, call_ids_of(i) -> (this rank's first call id, every
        rank's first call ids or None).  consume(i, out) runs when super-batch i is complete, on its lane's stream, before
        the lane's slabs are reused.  Plain samplers only need this; under a filter / with weights the call's status word
        is read when a super-batch ends (overflow or a non-positive weight sum raise: no silent retry inside a pipeline)."""
        cur = torch.cuda.current_stream(self.dev) if self.dev.type == "cuda" else None
        for st in self.streams:
            if st is not None:
                st.wait_stream(cur)

        def on(lane):
            return torch.cuda.stream(self.streams[lane]) if self.streams[lane] is not None else _Null()

        def start(i, lane):
            def run():
                ps = self.samplers[lane]
                with on(lane):
                    ps._status_acc.zero_()
                    first, all_first = call_ids_of(i)
                    gen = ps.steps(seeds_of(i), seed, first, all_first, seeds_state_of(i) if seeds_state_of else None)
                while True:
                    with on(lane):
                        try:
                            next(gen)
                        except StopIteration:
                            return
                    yield
            return run()

        def finish(i, lane):
            ps = self.samplers[lane]
            with on(lane):
                if ps.general:
                    word = ps._agreed_status()
                    if word & 2:
                        raise RuntimeError("weighted sampling met a non-positive running weight sum (the reference panics here)")
                    if word & 1:
                        raise RuntimeError("column-group workspace overflow inside a pipelined call: sample this super-batch "
                                           "with PartitionedSampler.sample(), which retries")
                consume(i, ps.out)

        interleave(n_jobs, len(self.samplers), start, finish)
        for st in self.streams:
            if st is not None:
                cur.wait_stream(st)


class _Null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def ns_homo_partitioned_device(shard, seeds, fanout, seed, first_call_id, sampler=SAMPLER_UNIFORM, group=None,
                               filter_mode=_cabi.FILTER_NONE, forward=False, window=(0, 0), seeds_state=None,
                               packed_replies=None, slot_replies=None):
    """One call of the device form (a throw-away PartitionedSampler; keep one around to reuse its buffers)."""
    ps = PartitionedSampler(shard, seeds.shape[0], seeds.shape[1], fanout, sampler=sampler, group=group,
                            filter_mode=filter_mode, forward=forward, window=window, packed_replies=packed_replies,
                            slot_replies=slot_replies)
    return ps.sample(seeds, seed, first_call_id, seeds_state=seeds_state)


def ns_homo_partitioned(shard, seeds, fanout, seed, first_call_id, sampler=SAMPLER_UNIFORM, group=None,
                        _hop_fn=None):
    """`ns_homo_partitioned_steps` run to its end."""
    gen = ns_homo_partitioned_steps(shard, seeds, fanout, seed, first_call_id, sampler, group, _hop_fn)
    while True:
        try:
            next(gen)
        except StopIteration as done:
            return done.value


def ns_homo_partitioned_steps(shard, seeds, fanout, seed, first_call_id, sampler=SAMPLER_UNIFORM, group=None,
                              _hop_fn=None):
    """A generator (it yields before every size exchange, like PartitionedSampler.steps, so that `interleave` can drive
    several calls over one communicator); its return value is the result.
    seeds: [n_batches, B] int64 on the shard's device; batch j of this rank has call id first_call_id + j.

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
            yield ("request sizes", len(hops))
            recv_counts = _exchange_counts(send_counts, group, dev)
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
            yield ("reply sizes", len(hops))
            rep_recv = _exchange_counts(rep_send, group, dev)
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
