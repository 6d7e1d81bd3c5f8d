"""Data-parallel plumbing for the hot path on one node: one process per GPU over torch.distributed
(backend "nccl" = RCCL on ROCm over xGMI; "gloo" on CPU for tests). The reference's only strategy is data
parallelism (train.py:201-205, :287-294). Inference shards images and needs NO data-path collective: each rank
takes a contiguous slice of the global batch, results are gathered for the evaluator on rank 0
([d2] comm.gather at pascal_voc_evaluation.py:106) and timings are reduced with MAX."""
from __future__ import annotations

from typing import Any, Dict, List, Optional, Sequence, Tuple

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


def poison_unless_(ok: torch.Tensor, grad_elem: torch.Tensor) -> None:
    """Make a RANK-LOCAL verdict global without a collective of its own: when the one-element flag `ok` is 0, add +inf to `grad_elem`
    (a one-element view into this rank's flat gradient) BEFORE its bucket is reduced. The sum over ranks is then non-finite on every
    rank, and the finiteness check of the all-reduced buffer -- the one verdict that gates the update, the momentum and the loss scale
    -- comes out the same everywhere. No host sync; two element-sized launches."""
    inf = torch.full_like(grad_elem, float("inf"))
    grad_elem.add_(torch.where(ok.to(torch.bool).view_as(grad_elem), torch.zeros_like(grad_elem), inf))


def all_reduce_sum_(flat: torch.Tensor) -> int:
    """In-place sum of one flat buffer over all ranks (the training step's gradient exchange: a single RCCL all-reduce of
    the 166 MB fp32 gradient buffer; gloo on CPU in the tests). Returns the world size (1 when not distributed)."""
    if not is_dist() or dist.get_world_size() == 1:
        return 1
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    return dist.get_world_size()


def reduce_dict(input_dict: dict, average: bool = True) -> dict:
    """[d2] comm.reduce_dict (train.py:139): the scalar tensors of a dict reduced to rank 0 in one collective (sorted keys, one
    stacked tensor), averaged by default; the other ranks get their partial sums back, as in detectron2. Not distributed: the
    dict itself."""
    if not is_dist() or dist.get_world_size() < 2:
        return input_dict
    with torch.no_grad():
        names = sorted(input_dict.keys())
        values = torch.stack([input_dict[k].detach().reshape(()) for k in names])
        dist.reduce(values, dst=0)
        if dist.get_rank() == 0 and average:
            values = values / dist.get_world_size()
        return {k: v for k, v in zip(names, values)}


class GradBuckets:
    """Bucketed gradient all-reduce overlapped with the backward (what DDP gives train.py:201-205 for free).

    The trainer lays its flat fp32 gradient buffer out in REVERSE order of completion (the heads, whose gradients the backward
    finishes first, sit at the end), so the finished part of the buffer grows from the end towards the start. The buffer is cut
    from the end into contiguous buckets of >= bucket_bytes (25 MB default: on xGMI's point-to-point links a ring is per-link
    bound, so few large collectives beat many small ones; 166.5 MB -> 6 buckets). `mark_done(name)` is called by the backward
    when a parameter's gradient is final; the call that completes a bucket issues its all-reduce with async_op=True -- the
    collective runs on the backend's own stream, ordered after everything enqueued so far on the compute stream, while the
    remaining data- and weight-gradient launches keep the compute stream busy. `finish()` waits for every outstanding bucket
    (a stream dependency for RCCL, a host wait for gloo) and returns the world size. Sums are associative-order independent
    here: every element is reduced exactly once, so bucketed == single-shot bit for bit for two ranks and up to the backend's
    own reduction order beyond."""

    def __init__(self, flat: torch.Tensor, layout: Sequence[Tuple[str, int, int]], bucket_bytes: int = 25 << 20):
        """layout: (name, offset, numel) of every parameter's view in `flat`, ascending offsets."""
        self.flat = flat
        self.buckets: List[Dict[str, Any]] = []
        self.owner: Dict[str, int] = {}
        esz = flat.element_size()
        hi, names, lo = flat.numel(), [], flat.numel()
        for name, off, numel in reversed(list(layout)):
            names.append(name)
            lo = off
            if (hi - lo) * esz >= bucket_bytes:
                self._close(lo, hi, names)
                hi, names = lo, []
        if names or hi > 0:
            self._close(0, hi, names)
        self.reset()

    def _close(self, lo: int, hi: int, names: List[str]) -> None:
        if hi <= lo:
            return
        for n in names:
            self.owner[n] = len(self.buckets)
        self.buckets.append(dict(lo=lo, hi=hi, names=tuple(names)))

    def reset(self) -> None:
        self.pending = [set(b["names"]) for b in self.buckets]
        self.works: List[Any] = []
        self.issued = [False] * len(self.buckets)

    def mark_done(self, name: str) -> None:
        for b in self.mark_ready((name,)):
            self.issue(b)

    def mark_ready(self, names: Sequence[str]) -> List[int]:
        """Record that these parameters' gradients are final; returns the buckets this completes (not yet issued): the caller
        issues them with `issue(b)` from the stream it wants the collective ordered behind."""
        ready: List[int] = []
        for name in names:
            b = self.owner[name]
            self.pending[b].discard(name)
            if not self.pending[b] and not self.issued[b] and b not in ready:
                ready.append(b)
        return ready

    def issue(self, b: int) -> None:
        if self.issued[b]:
            return
        self.issued[b] = True
        if is_dist() and dist.get_world_size() > 1:
            bk = self.buckets[b]
            self.works.append(dist.all_reduce(self.flat[bk["lo"]:bk["hi"]], op=dist.ReduceOp.SUM, async_op=True))

    def finish(self) -> int:
        """Issue whatever the backward did not mark (a bucket with an untracked parameter), wait for all, re-arm."""
        for b in range(len(self.buckets)):
            if not self.issued[b]:
                self.issue(b)
        for w in self.works:
            w.wait()
        self.reset()
        return dist.get_world_size() if is_dist() else 1
