"""Batch sharding across the GPUs of one node: the path is embarrassingly parallel
(no cross-trajectory coupling in the reference), so rank r simply owns the contiguous
instance range [r*B, (r+1)*B) and no data-path collective exists. The only
collectives are the timing barrier and a MAX over ranks (bench.py)."""
import os


def rank_info():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def shard_range(rank, per_rank_batch):
    """Instance index range owned by `rank` (weak scaling: fixed batch per GPU)."""
    return rank * per_rank_batch, (rank + 1) * per_rank_batch


def max_over_ranks(value, dist=None, device=None):
    """MAX-reduce a python float over the process group (identity without one)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, dist=None, device=None):
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())
