"""Multi-GPU sharding of independent sequences.

One process per GPU (torch.distributed; backend "nccl" == RCCL over xGMI on ROCm,
"gloo" on CPU for tests).  Sequences are independent MonoSLAM instances, so the
data path has NO collective: each rank generates/loads and steps its own shard.
RCCL is used only at the edges: a barrier + MAX-reduce of the timing and an
all-gather of the small per-sequence results (final vehicle states).
"""
import os

import numpy as np


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def shard_range(total, world, rank):
    """Contiguous block partition of `total` sequences: returns (first, count); sizes differ by at most 1."""
    base, rem = divmod(int(total), int(world))
    first = rank * base + min(rank, rem)
    return first, base + (1 if rank < rem else 0)


def global_sequence_ids(per_rank, world, rank):
    """Weak scaling: every rank owns `per_rank` sequences; ids are globally unique and dense."""
    return np.arange(rank * per_rank, (rank + 1) * per_rank, dtype=np.int64)


def max_over_ranks(value, device=None):
    """MAX all-reduce of a python float (identity when torch.distributed is not initialised)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device=None):
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def gather_states(local_states, device=None):
    """All-gather of per-sequence result rows ([n_local][k] float64, same n_local on every rank).
    Returns [world * n_local][k] ordered by rank == ordered by global sequence id."""
    import torch
    import torch.distributed as dist
    a = np.ascontiguousarray(local_states, dtype=np.float64)
    if not (dist.is_available() and dist.is_initialized()):
        return a.copy()
    t = torch.from_numpy(a)
    if device is not None:
        t = t.to(device)
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return torch.cat(out, dim=0).cpu().numpy()


def scatter_frames(frames_root, per_rank, frame_shape, root=0, device=None):
    """Scatter of one step's frames when they originate on ONE rank (a single grabber feeding a node): the root holds
    uint8 [world * per_rank, H, W] in global sequence order, every rank receives its block [per_rank, H, W].
    One grouped point-to-point exchange per destination (xGMI is point to point: a root-sourced scatter is bounded by
    the root's links - SURVEY.md 8(e) prefers per-rank loading; this exists for the case where that is impossible).
    Identity without torch.distributed."""
    import torch
    import torch.distributed as dist
    H, W = frame_shape
    if not (dist.is_available() and dist.is_initialized()):
        return np.ascontiguousarray(frames_root, dtype=np.uint8).reshape(per_rank, H, W)
    world, rank = dist.get_world_size(), dist.get_rank()
    out = torch.empty((per_rank, H, W), dtype=torch.uint8, device=device if device is not None else "cpu")
    if rank == root:
        src = torch.from_numpy(np.ascontiguousarray(frames_root, dtype=np.uint8).reshape(world, per_rank, H, W))
        if device is not None:
            src = src.to(device)
        chunks = [src[r].contiguous() for r in range(world)]
        dist.scatter(out, chunks, src=root)
    else:
        dist.scatter(out, None, src=root)
    return out if device is not None else out.numpy()
