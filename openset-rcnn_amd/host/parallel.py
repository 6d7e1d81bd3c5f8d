"""Data-parallel plumbing for the hot path on one node: one process per GPU over torch.distributed
(backend "nccl" = RCCL on ROCm over xGMI; "gloo" on CPU for tests). The reference's only strategy is data
parallelism (train.py:201-205, :287-294). Inference shards images and needs NO data-path collective: each rank
takes a contiguous slice of the global batch, results are gathered for the evaluator on rank 0
([d2] comm.gather at pascal_voc_evaluation.py:106) and timings are reduced with MAX."""
from __future__ import annotations

from typing import Any, List, Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(global_n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous split of `global_n` images over `world` ranks; the first global_n % world ranks get one extra."""
    base, extra = divmod(global_n, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def is_dist() -> bool:
    return dist.is_available() and dist.is_initialized()


def world_info() -> Tuple[int, int]:
    return (dist.get_rank(), dist.get_world_size()) if is_dist() else (0, 1)


def barrier() -> None:
    if is_dist():
        dist.barrier()


def max_over_ranks(value: float, device: Optional[torch.device] = None) -> float:
    if not is_dist():
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_to_rank0(obj: Any) -> Optional[List[Any]]:
    """Per-rank python results (lists of per-image dicts) -> list over ranks on rank 0, None elsewhere."""
    if not is_dist():
        return [obj]
    rank, world = world_info()
    out: Optional[List[Any]] = [None] * world if rank == 0 else None
    dist.gather_object(obj, out, dst=0)
    return out


def merge_sharded(results: List[List[Any]]) -> List[Any]:
    """Undo shard_range: concatenating the rank-ordered shards restores the global image order."""
    merged: List[Any] = []
    for r in results:
        merged.extend(r)
    return merged


def all_reduce_sum_(flat: torch.Tensor) -> int:
    """In-place sum of one flat buffer over all ranks (the training step's gradient exchange: a single RCCL all-reduce of
    the 166 MB fp32 gradient buffer; gloo on CPU in the tests). Returns the world size (1 when not distributed)."""
    if not is_dist() or dist.get_world_size() == 1:
        return 1
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    return dist.get_world_size()
