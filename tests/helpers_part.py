"""Range-partitioned sampling with the all-to-alls emulated in ONE process (tests of csrc/partition.hip for any world
size on a one-GPU box): bucket p of the requests is answered from shard p, replies are concatenated in bucket order --
exactly what `all_to_all_single` delivers to the origin rank."""
import ctypes as C

import torch


def emulated_world_sample(cabi, shards, seeds, fan, seed, first_call, sampler=0):
    """-> (NsBatchedOut filled through tg_part_begin/requests/sample/emit, number of requests that left shard 0)."""
    lib, ptr = cabi.lib, cabi.ptr
    dev, world = seeds.device, len(shards)
    nb, B = seeds.shape
    out = cabi.NsBatchedOut(nb, B, fan, dev)
    so, stream = out.struct(), cabi.stream_ptr(dev)
    nbytes = C.c_int64(0)
    cabi.check(lib.tg_part_workspace_bytes(C.c_int64(nb), C.c_int32(world), C.byref(nbytes)))
    ws = torch.zeros(nbytes.value // 8, dtype=torch.int64, device=dev)
    seeds = seeds.contiguous()
    cabi.check(lib.tg_part_begin(ptr(seeds), C.c_int64(nb), C.c_int64(B), C.byref(so), ptr(ws), stream))
    cap, crossed = nb * B, 0
    for h, k in enumerate(fan):
        req = torch.empty(cap * 3, dtype=torch.int64, device=dev)
        req_pos = torch.empty(cap, dtype=torch.int64, device=dev)
        cabi.check(lib.tg_part_requests(C.byref(so), C.c_int64(nb), C.c_int64(cap), C.c_int64(shards[0].shard_size),
                                        C.c_int32(world), C.c_uint64(first_call), ptr(ws), ptr(req), ptr(req_pos), stream))
        sizes = ws[5 * nb + 1:5 * nb + 1 + world].tolist()
        assert sum(sizes) == int(ws[5 * nb])                # batch_off[n_batches] = number of requests
        r3 = req[:sum(sizes) * 3].reshape(-1, 3)
        replies, lo = [], 0
        for p, m in enumerate(sizes):                       # "all-to-all": bucket p goes to the owner of shard p
            mine = r3[lo:lo + m].contiguous()
            if m:
                owner = torch.clamp(mine[:, 0] // shards[0].shard_size, max=world - 1)
                assert bool((owner == p).all())
                crossed += m if p else 0
            rep = torch.empty(max(m, 1) * k * 2, dtype=torch.int64, device=dev)
            g = shards[p].graph_view()
            cabi.check(lib.tg_part_sample(C.byref(g), C.c_int64(shards[p].v_lo), C.c_int64(shards[p].e_lo), ptr(mine),
                                          C.c_int64(m), C.c_int32(k), C.c_int32(sampler), C.c_uint64(seed), ptr(rep),
                                          stream))
            replies.append(rep[:m * k * 2])
            lo += m
        back = torch.cat(replies).contiguous()
        if back.numel() == 0:
            back = torch.empty(2, dtype=torch.int64, device=dev)
        cabi.check(lib.tg_part_emit(C.byref(so), C.c_int64(nb), C.c_int64(B), C.c_int32(k), C.c_int32(h),
                                    C.c_int32(len(fan)), ptr(ws), ptr(req_pos), ptr(back), stream))
        cap *= k
    if not fan:
        out.counts[:, 0] = B
    return out, crossed
