"""Range-partitioned sampling with the all-to-alls emulated in ONE process (tests of csrc/partition.hip for any world
size on a one-GPU box): bucket p of the origin's requests is answered from shard p, counts and replies are concatenated
in bucket order -- exactly what `all_to_all_single` delivers back to the origin rank."""
import ctypes as C

import torch


def emulated_world_sample(cabi, shards, seeds, fan, seed, first_call, sampler=0, filter_mode=-1, forward=False,
                          window=(0, 0), seeds_state=None, packed=False, slots=False):
    """-> (NsBatchedOut filled through tg_part_begin / requests / [count + sample | unpack + flat hop + pack] / emit,
    requests that left shard 0).  Filters / weights take the general owner path.  packed: one-word reply entries
    (TG_PART_REPLY_PACKED / _PACKED_STATE) instead of pairs / triples.  slots: the fixed-size slot replies
    (tg_part_sample_slots / tg_part_emit_slots; unweighted, unfiltered sampling only)."""
    filtered = filter_mode != -1
    general = filtered or sampler == 2
    fmt = (4 if packed else 3) if filtered else (1 if packed else 2)     # tchgeo.h TG_PART_REPLY_*
    words = 2 if fmt == 4 else fmt
    lib, ptr = cabi.lib, cabi.ptr
    dev, world = seeds.device, len(shards)
    nb, B = seeds.shape
    H = len(fan)
    out = cabi.NsBatchedOut(nb, B, fan, dev, with_states=filtered)
    so, stream = out.struct(), cabi.stream_ptr(dev)
    hop_cap, cap = [], nb * B
    for k in fan:
        hop_cap.append(cap)
        cap *= k
    request_cap = max(hop_cap + [1])
    nbytes = C.c_int64(0)
    cabi.check(lib.tg_part_workspace_bytes(C.c_int64(nb), C.c_int64(request_cap), C.c_int32(world), C.byref(nbytes)))
    i64 = dict(dtype=torch.int64, device=dev)
    ws = torch.empty(nbytes.value // 8 + 1, **i64)
    requests = torch.empty((request_cap, 2), **i64)
    send_counts = torch.zeros(world + 1, **i64)
    seeds = seeds.contiguous()
    req_states = torch.empty(request_cap, **i64) if filtered else None
    if filtered:
        seeds_state = seeds_state.contiguous()
    cabi.check(lib.tg_part_begin(ptr(seeds), ptr(seeds_state) if filtered else None, C.c_int64(nb), C.c_int64(B),
                                 C.c_int32(H), C.byref(so), C.c_int64(request_cap), C.c_int32(world), ptr(ws), stream))
    crossed = 0
    call0 = (C.c_uint64 * 64)(first_call)                   # owner side sees ONE requesting rank: the origin
    if slots:
        assert not general
        longest = max(int((sh.ptrs[1:] - sh.ptrs[:-1]).max()) if sh.ptrs.numel() > 1 else 0 for sh in shards)
        bv, bp = max(1, (shards[0].n_nodes - 1).bit_length()), max(1, (max(longest, 1) - 1).bit_length())
        e_lo_of = (C.c_int64 * 64)(*[sh.e_lo for sh in shards])
    for h, k in enumerate(fan):
        cabi.check(lib.tg_part_requests(C.byref(so), C.c_int64(nb), C.c_int64(request_cap),
                                        C.c_int64(shards[0].shard_size), C.c_int32(world), ptr(ws), ptr(requests),
                                        ptr(req_states) if filtered else None, ptr(send_counts), stream))
        sizes = send_counts.tolist()
        assert sum(sizes[:world]) == sizes[world]
        cnts, replies, lo = [], [], 0
        for p, m in enumerate(sizes[:world]):               # "all-to-all": bucket p goes to the owner of shard p
            mine = requests[lo:lo + m].contiguous()
            if m:
                owner = torch.clamp(mine[:, 0] // shards[0].shard_size, max=world - 1)
                assert bool((owner == p).all())
                crossed += m if p else 0
            m_dev = torch.tensor([m], **i64)
            if slots:
                W = C.c_int32(0)
                cabi.check(lib.tg_part_slot_words(C.c_int32(k), C.c_int32(bv), C.c_int32(bp), C.byref(W)))
                assert W.value in (16, 32)
                mine_slots = torch.empty((max(m, 1), W.value // 2), **i64)
                seg = (C.c_int64 * 65)(0, m)
                g = shards[p].graph_view()
                wb = C.c_int64(0)
                cabi.check(lib.tg_part_sample_workspace_bytes(C.c_int64(m), C.byref(wb)))
                sws = torch.empty(wb.value // 8 + 64, **i64)
                cabi.check(lib.tg_part_sample_slots(C.byref(g), C.c_int64(shards[p].v_lo), ptr(mine), ptr(m_dev),
                                                    C.c_int64(m), C.c_int32(1), seg, call0, C.c_int32(k),
                                                    C.c_int32(sampler), C.c_uint64(seed), C.c_int32(bv), C.c_int32(bp),
                                                    C.c_int32(1), ptr(mine_slots), ptr(sws), C.c_int64(sws.numel() * 8), stream))
                replies.append(mine_slots[:m])
                lo += m
                continue
            if general:
                cnt, rep = _general_owner(cabi, shards[p], mine, req_states[lo:lo + m].contiguous() if filtered else None,
                                          m_dev, m, k, sampler, filter_mode, forward, window, seed, call0, fmt, stream)
                cnts.append(cnt[:m])
                replies.append(rep)
                lo += m
                continue
            cnt = torch.empty(max(m, 1), dtype=torch.int32, device=dev)
            off = torch.empty(max(m, 1) + 1, **i64)
            rc = torch.zeros(2, **i64)
            tb = C.c_int64(0)
            cabi.check(lib.tg_part_scan_workspace_bytes(C.c_int64(m), C.byref(tb)))
            tmp = torch.empty(tb.value // 8 + 1, **i64)
            seg = (C.c_int64 * 65)(0, m)
            g = shards[p].graph_view()
            cabi.check(lib.tg_part_count(C.byref(g), C.c_int64(shards[p].v_lo), ptr(mine), ptr(m_dev), C.c_int64(m),
                                         C.c_int32(1), seg, call0, C.c_int32(k), C.c_int32(sampler), ptr(cnt), ptr(off),
                                         ptr(rc), ptr(tmp), C.c_int64(tmp.numel() * 8), stream))
            total = int(rc[1])
            assert total == int(rc[0]) and (m == 0 or total == int(cnt[:m].sum()))
            rep = torch.empty((max(total, 1), words), **i64)
            cabi.check(lib.tg_part_sample(C.byref(g), C.c_int64(shards[p].v_lo), C.c_int64(shards[p].e_lo), ptr(mine),
                                          ptr(m_dev), C.c_int64(m), C.c_int32(1), seg, call0, C.c_int32(k),
                                          C.c_int32(sampler), C.c_uint64(seed), ptr(cnt), ptr(off), ptr(rep),
                                          C.c_int32(fmt), stream))
            cnts.append(cnt[:m])
            replies.append(rep[:total])
            lo += m
        if slots:
            back = torch.cat(replies).contiguous()
            if back.numel() == 0:
                back = torch.empty((1, W.value // 2), **i64)
            cabi.check(lib.tg_part_emit_slots(C.byref(so), C.c_int64(nb), C.c_int64(B), C.c_int64(request_cap),
                                              C.c_int32(world), C.c_int32(k), C.c_int32(h), C.c_int32(H), ptr(ws), ptr(back),
                                              C.c_int32(bv), C.c_int32(bp), e_lo_of, stream))
            continue
        cnt_back = torch.zeros(request_cap, dtype=torch.int32, device=dev)
        allc = torch.cat(cnts)
        cnt_back[:allc.numel()] = allc
        back = torch.cat(replies).contiguous()
        if back.numel() == 0:
            back = torch.empty((1, words), **i64)
        cabi.check(lib.tg_part_emit(C.byref(so), C.c_int64(nb), C.c_int64(B), C.c_int64(request_cap),
                                    C.c_int64(hop_cap[h]), C.c_int32(world), C.c_int32(k), C.c_int32(h), C.c_int32(H),
                                    ptr(ws), ptr(cnt_back), None, ptr(back), C.c_int32(fmt), stream))
    torch.cuda.synchronize()
    return out, crossed


def _general_owner(cabi, shard, mine, states, m_dev, m, k, sampler, filter_mode, forward, window, seed, call0, fmt, stream):
    """tg_part_unpack -> tg_ns_hop_scan / tg_ns_hop_weighted -> tg_part_pack for one bucket of requests"""
    lib, ptr = cabi.lib, cabi.ptr
    dev = mine.device
    i64 = dict(dtype=torch.int64, device=dev)
    words = 2 if fmt == 4 else fmt
    mc = max(m, 1)
    vert, ids, calls = (torch.empty(mc, **i64) for _ in range(3))
    seg = (C.c_int64 * 65)(0, m)
    if m == 0:
        return torch.zeros(1, dtype=torch.int32, device=dev), torch.empty((0, words), **i64)
    cabi.check(lib.tg_part_unpack(C.c_int64(shard.v_lo), C.c_int64(shard.v_hi - shard.v_lo), ptr(mine), ptr(m_dev),
                                  C.c_int64(m), C.c_int32(1), seg, call0, ptr(vert), ptr(ids), ptr(calls), stream))
    hcnt, hoff = torch.empty(mc, **i64), torch.empty(mc + 1, **i64)
    nbr, ep, par, st_out = (torch.empty(mc * k, **i64) for _ in range(4))
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    hin, hout, flt = cabi.TgHopIn(), cabi.TgHopOut(), cabi.TgHopFilter()
    hin.vertices, hin.ids, hin.call_ids = vert.data_ptr(), ids.data_ptr(), calls.data_ptr()
    hin.m, hin.id_base, hin.fanout, hin.sampler, hin.rng_tag = m, 0, k, sampler, 0
    hout.cnt, hout.offsets = hcnt.data_ptr(), hoff.data_ptr()
    hout.neighbors, hout.edge_ptrs, hout.parents = nbr.data_ptr(), ep.data_ptr(), par.data_ptr()
    flt.filter_mode, flt.forward = filter_mode, int(bool(forward))
    flt.win_lo, flt.win_hi = window
    flt.states = states.data_ptr() if states is not None else None
    g = shard.graph_view()
    rng = cabi.TgRng(seed, 0)
    group_cap = 1 if sampler == 2 else max(1024, g.n_edges // 512 + 2 * m + 2) * 8
    nbytes = C.c_int64(0)
    cabi.check(lib.tg_ns_hop_scan_workspace_bytes(C.c_int64(m), C.c_int32(k), C.c_int64(group_cap), C.byref(nbytes)))
    ws = torch.empty(nbytes.value // 8 + 1, **i64)
    if sampler == 2:
        cabi.check(lib.tg_ns_hop_weighted(C.byref(g), C.byref(hin), C.byref(flt), C.byref(rng), C.byref(hout), ptr(st_out),
                                          ptr(status), ptr(ws), C.c_int64(nbytes.value), stream))
    else:
        cabi.check(lib.tg_ns_hop_scan(C.byref(g), C.byref(hin), C.byref(flt), C.byref(rng), C.byref(hout), ptr(st_out),
                                      ptr(status), ptr(ws), C.c_int64(nbytes.value), C.c_int64(group_cap), stream))
    assert int(status[0]) == 0
    total = int(hoff[m])
    cnt = torch.empty(mc, dtype=torch.int32, device=dev)
    rep = torch.empty((max(total, 1), words), **i64)
    rc = torch.zeros(2, **i64)
    cabi.check(lib.tg_part_pack(C.byref(hout), ptr(st_out) if fmt in (3, 4) else None, ptr(m_dev), C.c_int64(m),
                                C.c_int64(shard.e_lo), C.c_int32(1), seg, ptr(cnt), ptr(rep), C.c_int32(fmt), ptr(rc),
                                stream))
    assert int(rc[0]) == total == int(rc[1])
    return cnt, rep[:total]
