"""Data-parallel sharding of independent clouds over the GPUs of one node (one process per GPU).

The reference has no inference-time multi-GPU code (its only distributed path is DDP training, train.py:163-176);
clouds never interact on the hot path, so rank r simply takes clouds [r*B/R, (r+1)*B/R) with replicated weights
and the only exchange is one all_gather of the per-cloud results (RCCL over xGMI when the tensors live on GPUs;
the same code runs over gloo on CPU tensors in the tests).
"""
import os
from typing import Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: str = None) -> Tuple[int, int, int]:
    """Initialises torch.distributed from torchrun's environment; returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"  # "nccl" IS RCCL on ROCm
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous shard [lo, hi) of ``total`` clouds for ``rank``; remainders go to the lowest ranks."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_results(local: torch.Tensor, total: int, async_op: bool = False):
    """all_gather of per-cloud results [b_local, ...] -> [total, ...] in cloud order (every rank gets all).
    Uneven shards are padded to the largest shard for the collective and trimmed afterwards."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local if not async_op else (local, None)
    world, rank = dist.get_world_size(), dist.get_rank()
    sizes = [shard_range(total, r, world)[1] - shard_range(total, r, world)[0] for r in range(world)]
    mx = max(sizes)
    if local.shape[0] != sizes[rank]:
        raise ValueError(f"rank {rank}: local batch {local.shape[0]} != shard size {sizes[rank]}")
    send = local
    if local.shape[0] < mx:
        send = torch.cat([local, local.new_zeros((mx - local.shape[0],) + tuple(local.shape[1:]))], 0)
    out = local.new_empty((world * mx,) + tuple(local.shape[1:]))
    work = dist.all_gather_into_tensor(out, send.contiguous(), async_op=async_op)

    def finish():
        parts = [out[r * mx: r * mx + sizes[r]] for r in range(world)]
        return parts[0] if world == 1 else torch.cat(parts, 0) if min(sizes) != mx else out

    if async_op:
        return finish, work
    return finish()


class SideStreamGather:
    """The per-step all_gather of the results on its own HIP stream, so that the collective (RCCL over xGMI; latency-bound: 3.1 MB per
    rank at BASELINE config #4) overlaps the dense stage of the following batches instead of sitting between them.

        g = SideStreamGather(device)
        h = g.start((masks, iou), total)      # on the caller's stream, after the results were produced
        ...                                   # enqueue more work
        masks_all, iou_all = g.finish(h)      # makes the caller's stream wait for the collective

    With one process (no process group) start/finish pass the tensors through."""

    def __init__(self, device=None):
        self.enabled = dist.is_initialized() and dist.get_world_size() > 1
        # GPU tensors: the collective gets its own HIP stream.  CPU tensors (gloo, tests): the gather runs in place, synchronously.
        # (a stream chosen by probing: beside the pipelines' streams a fresh one usually lands on the hardware queue the tokenizer stream starves, streams.py)
        if self.enabled and device is not None and torch.cuda.is_available():
            from .streams import side_stream
            self.stream = side_stream(device)
        else:
            self.stream = None

    def start(self, tensors, total: int):
        if not self.enabled:
            return tuple(tensors), None
        if self.stream is None:
            return tuple(gather_results(t, total) for t in tensors), None
        cur = torch.cuda.current_stream()
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            outs = tuple(gather_results(t, total) for t in tensors)
            for t in tensors:
                t.record_stream(self.stream)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        return outs, ev

    def finish(self, handle):
        outs, ev = handle
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)
            for t in outs:
                t.record_stream(torch.cuda.current_stream())
        return outs
