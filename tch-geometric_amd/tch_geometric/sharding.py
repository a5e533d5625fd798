"""Seed-batch sharding across the GPUs of one node (mode 1: replicated CSC).

Every seed batch is independent given counter-addressed draws (call_id = global batch id), so rank r
simply owns a contiguous block of global batch ids and no data-path collective exists.  The only
collectives are those of the measurement protocol: a barrier, MAX over ranks of the elapsed time, SUM of
the work counters."""
import torch
import torch.distributed as dist


def rank_batch_range(rank: int, world: int, batches_per_rank: int):
    """Global batch ids [first, last) owned by `rank`; the union over ranks is [0, world * batches_per_rank)."""
    if not (0 <= rank < world):
        raise ValueError("rank %d outside world of %d" % (rank, world))
    return rank * batches_per_rank, (rank + 1) * batches_per_rank


def reduce_measurement(elapsed_s: float, counters: torch.Tensor):
    """-> (max elapsed over ranks, counters summed over ranks).  No-op without a process group."""
    active = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    where = torch.device("cpu") if active and dist.get_backend() == "gloo" else counters.device  # gloo: host tensors
    t = torch.tensor([elapsed_s], dtype=torch.float64, device=where)
    tot = counters.clone().to(where)
    if active:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    return float(t.item()), tot.to(counters.device)


def fence(device=None):
    """barrier + device synchronize on both sides of a timed region."""
    if device is not None and torch.cuda.is_available():
        torch.cuda.synchronize(device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
    if device is not None and torch.cuda.is_available():
        torch.cuda.synchronize(device)
