"""Range-partitioned sampling with the all-to-alls emulated in ONE process (tests of csrc/partition.hip for any world
size on a one-GPU box): bucket p of the origin's requests is answered from shard p, counts and replies are concatenated
in bucket order -- exactly what `all_to_all_single` delivers back to the origin rank."""
import ctypes as C

import torch


def emulated_world_sample(cabi, shards, seeds, fan, seed, first_call, sampler=0):
    """-> (NsBatchedOut filled through tg_part_begin / requests / count / sample / emit, requests that left shard 0)."""
    lib, ptr = cabi.lib, cabi.ptr
    dev, world = seeds.device, len(shards)
    nb, B = seeds.shape
    H = len(fan)
    out = cabi.NsBatchedOut(nb, B, fan, dev)
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
    cabi.check(lib.tg_part_begin(ptr(seeds), C.c_int64(nb), C.c_int64(B), C.c_int32(H), C.byref(so),
                                 C.c_int64(request_cap), C.c_int32(world), ptr(ws), stream))
    crossed = 0
    call0 = (C.c_uint64 * 64)(first_call)                   # owner side sees ONE requesting rank: the origin
    for h, k in enumerate(fan):
        cabi.check(lib.tg_part_requests(C.byref(so), C.c_int64(nb), C.c_int64(request_cap),
                                        C.c_int64(shards[0].shard_size), C.c_int32(world), ptr(ws), ptr(requests),
                                        ptr(send_counts), stream))
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
            rep = torch.empty((max(total, 1), 2), **i64)
            cabi.check(lib.tg_part_sample(C.byref(g), C.c_int64(shards[p].v_lo), C.c_int64(shards[p].e_lo), ptr(mine),
                                          ptr(m_dev), C.c_int64(m), C.c_int32(1), seg, call0, C.c_int32(k),
                                          C.c_int32(sampler), C.c_uint64(seed), ptr(cnt), ptr(off), ptr(rep), stream))
            cnts.append(cnt[:m])
            replies.append(rep[:total])
            lo += m
        cnt_back = torch.zeros(request_cap, dtype=torch.int32, device=dev)
        allc = torch.cat(cnts)
        cnt_back[:allc.numel()] = allc
        back = torch.cat(replies).contiguous()
        if back.numel() == 0:
            back = torch.empty((1, 2), **i64)
        cabi.check(lib.tg_part_emit(C.byref(so), C.c_int64(nb), C.c_int64(B), C.c_int64(request_cap),
                                    C.c_int64(hop_cap[h]), C.c_int32(world), C.c_int32(k), C.c_int32(h), C.c_int32(H),
                                    ptr(ws), ptr(cnt_back), None, ptr(back), stream))
    torch.cuda.synchronize()
    return out, crossed
