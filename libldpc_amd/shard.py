"""Frame sharding across GPUs and the one collective of the path.

Frames are independent, so a simulation shards by contiguous frame ranges of the SAME noise stream: rank r of N
owns frames [r*per_rank, (r+1)*per_rank) of mt19937_64(seed) — frame f keeps its global identity, results do
not depend on N.  The only exchange is a sum of the counters {frames, fec, bec, iters, converged}
(the reference's shared OpenMP counters, ldpcsim.cpp:175-200), one all-reduce of 5 x int64 per step.
"""
import os


def rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def frame_range(rank, world, per_rank):
    """Contiguous global frame range owned by `rank`."""
    return rank * per_rank, (rank + 1) * per_rank


def reduce_counters(counters, dist=None):
    """Sum a 1-D int64 tensor of counters over all ranks (no-op for a single process)."""
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(counters)
    return counters


def counters_from_outputs(torch, iters, bit_errors, max_iters, early_term):
    """{frames, fec, bec, iters, converged} of one batch from the per-frame outputs (device tensors)."""
    it = iters.to(torch.int64)
    be = bit_errors.to(torch.int64)
    conv = (it < max_iters).sum() if early_term else torch.zeros((), dtype=torch.int64, device=it.device)
    n = torch.full((), it.numel(), dtype=torch.int64, device=it.device)
    return torch.stack([n, (be > 0).sum(), be.sum(), it.sum(), conv])
