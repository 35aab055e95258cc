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


def length_sorted_batches(lengths: Sequence[float], batch: int, world: int = 1, rank: int = 0) -> List[List[int]]:
    """Length-aware batching for real data: a batch decodes until its LONGEST row stops (examples/whisper/run.py:219-226 is batch 1,
    so the reference never pays for a neighbour), hence utterances of similar length should share a batch.  `lengths` is any
    monotone proxy of the transcript length -- the audio duration, e.g. `audio.valid_frames(mel)`.  Utterances are sorted by length
    (descending, ties in dataset order), cut into batches of `batch`, and the batches are dealt round-robin over the ranks so every
    GPU sees the same mix of long and short batches.  Returns this rank's batches as lists of dataset indices."""
    if batch < 1 or not (0 <= rank < world):
        raise ValueError(f"batch {batch}, rank {rank}, world {world}")
    order = sorted(range(len(lengths)), key=lambda i: (-float(lengths[i]), i))
    groups = [order[i:i + batch] for i in range(0, len(order), batch)]
    return groups[rank::world]


def slot_utilisation(row_steps: Sequence[int], groups: Sequence[Sequence[int]]) -> float:
    """Fraction of decoder (row, step) slots that produce a token some row still needs: sum of row lengths / sum over batches of
    rows x longest row.  1.0 = no row ever waits for a neighbour."""
    used = sum(int(row_steps[i]) for g in groups for i in g)
    paid = sum(len(g) * max(int(row_steps[i]) for i in g) for g in groups if len(g))
    return used / paid if paid else 1.0


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


def init_from_env(backend: str = None):
    """One process per GPU under `python -m torch.distributed.run` (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the env).

    Returns (rank, world, device_index, dist_or_None).  Without those variables: (0, 1, 0, None) -- a plain single-process run.
    Device = LOCAL_RANK modulo the visible GPUs, so a one-GPU box can rehearse N ranks on its one card.  Backend: "nccl"
    (= RCCL) when every rank has its own GPU, else "gloo" (RCCL cannot put two ranks on one device); the collectives here only
    carry a barrier, one float64 and the token-id lists, never activations."""
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1 or "RANK" not in os.environ:
        return 0, 1, 0, None
    import torch
    import torch.distributed as dist
    rank, local = int(os.environ["RANK"]), int(os.environ.get("LOCAL_RANK", os.environ["RANK"]))
    ndev = torch.cuda.device_count()
    if ndev < 1:
        raise RuntimeError("whisper-trtllm_amd has no CPU fallback: no GPU is visible to this rank")
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world))
    device = local % ndev
    if backend is None:
        backend = "nccl" if ndev >= local_world else "gloo"
    torch.cuda.set_device(device)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if not dist.is_initialized():
        kw = {"device_id": torch.device("cuda", device)} if backend == "nccl" else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world, device, dist


def gather_objects(local: list, dist=None) -> list:
    """Concatenate every rank's list in rank order (host-side; hypotheses / references / id rows)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return list(local)
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, list(local))
    return [x for part in out for x in part]
