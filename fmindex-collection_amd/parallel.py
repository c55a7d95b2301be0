"""Multi-GPU plumbing of the path: the index is replicated, the query batch is sharded contiguously, and the only exchange
is a gather of the per-query results to one rank (RCCL over xGMI on GPUs; the same code runs over gloo in the CPU tests).
The reference has no distributed component — its queries are independent (search/SearchNg26.h:407-423 loops over qidx)."""
import torch
import torch.distributed as dist


def shard_range(n_items, world, rank):
    """contiguous, balanced split: ranks [0, n_items % world) get one extra item"""
    base, rest = divmod(n_items, world)
    lo = rank * base + min(rank, rest)
    return lo, lo + base + (1 if rank < rest else 0)


def gather_fixed(local, dst=0, group=None):
    """fixed-size per-rank payload (exact search: [lb | len] of the rank's shard) -> list of tensors on `dst`, else None"""
    world = dist.get_world_size(group)
    out = [torch.empty_like(local) for _ in range(world)] if dist.get_rank(group) == dst else None
    dist.gather(local, out, dst=dst, group=group)
    return out


def gather_ragged(local, dst=0, group=None):
    """shards of unequal length (last shard shorter, or k-mismatch hit records): sizes first, then one padded gather"""
    world = dist.get_world_size(group)
    n = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(s.item()) for s in sizes]
    pad = max(sizes)
    buf = torch.zeros((pad,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    buf[: local.shape[0]] = local
    out = [torch.empty_like(buf) for _ in range(world)] if dist.get_rank(group) == dst else None
    dist.gather(buf, out, dst=dst, group=group)
    if out is None:
        return None
    return [o[:s] for o, s in zip(out, sizes)]
