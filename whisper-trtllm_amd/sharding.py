"""Utterance-parallel sharding helpers: utterances are independent, so a batch is cut into
contiguous per-rank chunks, every rank holds a full weight replica, and NO data-path collective is needed.
torch.distributed (RCCL on the GPU box, gloo in CPU tests) is used only for the timing barrier, the
max-over-ranks of the elapsed time and the final gather of the (tiny) token-id arrays."""
from __future__ import annotations

from typing import List, Sequence, Tuple


def utterance_shard(total: int, world: int, rank: int) -> Tuple[int, int]:
    """[begin, end) of the contiguous chunk of `total` utterances owned by `rank`; sizes differ by at most 1."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world {world}")
    base, extra = divmod(total, world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def batches(begin: int, end: int, batch: int) -> List[Tuple[int, int]]:
    """Cut a shard into engine batches of at most `batch` utterances (8 per call on the fast path)."""
    return [(i, min(i + batch, end)) for i in range(begin, end, batch)]


def max_over_ranks(value: float, dist=None) -> float:
    """bench.py contract: the reported time is the MAX over ranks."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    import torch
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([value], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_ids(local_ids: Sequence[Sequence[int]], dist=None) -> List[List[int]]:
    """All ranks' token-id rows in global utterance order (host-side, <= 448 int32 per utterance)."""
    rows = [list(map(int, r)) for r in local_ids]
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return rows
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, rows)
    return [r for part in out for r in part]
