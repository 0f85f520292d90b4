"""Rank bookkeeping of the multi-GPU path (one process per GPU).

The RAW noise stream mt19937_64(seed) is what is sharded: a global step is world x m whole generator chunks, rank r turns
its own chunks into accepted polar pairs, ONE all-gather of three words per rank (pairs in the piece, pairs including
the margin, status) places every piece in the pair sequence, and a frame belongs to the rank whose piece holds its first
pair (libldpc_amd/csrc/shard_place.hpp; `Comm.place` runs that step by itself, without a GPU).  Frames keep their identity
in the one stream, so results do not depend on the number of ranks.  The counters {frames, fec, bec,
iters, converged} (the reference's shared OpenMP counters, ldpcsim.cpp:175-200) are summed per rank on the device and
gathered once after the last step.
"""
import os


def rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def counters_from_outputs(torch, iters, bit_errors, max_iters, early_term):
    """{frames, fec, bec, iters, converged} of one batch from the per-frame outputs (device tensors): the torch spelling of
    ldpc_hip_batch_counters."""
    it = iters.to(torch.int64)
    be = bit_errors.to(torch.int64)
    conv = (it < max_iters).sum() if early_term else torch.zeros((), dtype=torch.int64, device=it.device)
    n = torch.full((), it.numel(), dtype=torch.int64, device=it.device)
    return torch.stack([n, (be > 0).sum(), be.sum(), it.sum(), conv])
