"""Data-parallel helpers (one process per GPU, torch.distributed; backend "nccl" is RCCL
over xGMI on ROCm, "gloo" in the CPU tests).

The reference is single-GPU, one view per step (brush-train/src/train.rs:176-186): data
parallelism over cameras is a capability this build adds.  Semantics:
every rank holds a full replica of the splats + Adam state, renders its own view, and
between backward and Adam the fused gradient buffer is SUM-all-reduced and scaled by
1/world (mean gradient over the K views); the refine statistics (refine weight, visible
flag, max screen radius) are MAX-all-reduced so every rank applies the identical update
and keeps identical RefineRecords.  K = 1 is bit-identical to the single-GPU step.
"""
import torch


def view_for_rank(step, rank, world, num_views):
    """Round-robin sharding of a view list over ranks: at `step` rank r takes view
    (step * world + r) mod num_views, so one pass over the dataset visits every view once
    when num_views is a multiple of world."""
    return (step * world + rank) % num_views


def allreduce_step_buffers(grads: torch.Tensor, stats: torch.Tensor, group=None):
    """In place: grads <- sum over ranks, stats <- max over ranks (one collective each:
    the gradient buffer is a single fused [N*(10+3C+1)] tensor, 56 MB at 1 M splats / SH0,
    so the all-reduce is one large message — the right shape for xGMI's per-link bound)."""
    import torch.distributed as dist
    dist.all_reduce(grads, op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(stats, op=dist.ReduceOp.MAX, group=group)
    return dist.get_world_size(group)
