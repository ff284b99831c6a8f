"""Data-parallel plumbing: one process per GPU, independent 30 s windows sharded across ranks, ONE collective at load.

The reference has no distributed layer at all (no NCCL/MPI call sites; its only multi-instance
facility is iModel::clone on one adapter). Windows transcribed with NoContext are complete, independent encode+decode
units (ContextImpl.cpp:476-477), so the path shards with no collective inside the step:
  * rank 0 reads the ggml file and fills the packed weight arena; every other rank allocates an arena of the same size
    (the layout is a pure function of the hparams, wh_model_arena_bytes) and receives it with one broadcast -- RCCL over
    xGMI on GPUs (backend "nccl"), gloo in the CPU tests;
  * window indices are dealt out in contiguous, balanced ranges; each rank transcribes its range as lock-step batches;
  * the per-window token ids (a few hundred bytes) are gathered on rank 0 in window order.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced [begin, end) of `n_items` for `rank`: the first n_items % world ranks get one extra item."""
    if world <= 0 or not (0 <= rank < world) or n_items < 0:
        raise ValueError("bad shard arguments")
    q, r = divmod(n_items, world)
    begin = rank * q + min(rank, r)
    return begin, begin + q + (1 if rank < r else 0)


def broadcast_arena(arena, src: int = 0):
    """Broadcast the packed weight arena (a uint8 torch tensor, CUDA with nccl / CPU with gloo) from `src` in place."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(arena, src=src)
    return arena


def gather_window_tokens(local_tokens: np.ndarray, n_windows_total: int, max_len: int):
    """Every rank passes its [n_local][<= max_len] int32 token ids (padded with -1); rank 0 gets them back in window order
    as an [n_windows_total][max_len] array, the other ranks get None."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    b, e = shard_range(n_windows_total, rank, world)
    pad = np.full((e - b, max_len), -1, np.int32)
    if e > b:           # a rank with an empty shard (world > n_windows) still has to enter the collective below
        lt = np.asarray(local_tokens, np.int32).reshape(e - b, -1)
        pad[:, :lt.shape[1]] = lt[:, :max_len]
    if world == 1:
        return pad
    # equal-sized buffers for all_gather: pad each rank's block to the largest shard (at least one row, so that no rank
    # contributes a zero-sized tensor)
    biggest = max(1, shard_range(n_windows_total, 0, world)[1])
    buf = np.full((biggest, max_len), -1, np.int32)
    buf[:e - b] = pad
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    mine = torch.from_numpy(buf).to(dev)
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine)
    if rank != 0:
        return None
    out = np.full((n_windows_total, max_len), -1, np.int32)
    for r, p in enumerate(parts):
        rb, re_ = shard_range(n_windows_total, r, world)
        out[rb:re_] = p.cpu().numpy()[:re_ - rb]
    return out


def transcribe_sharded(n_windows_total: int, transcribe_local, max_len: int):
    """The whole data-parallel step: this rank's contiguous range of window indices -> transcribe_local(begin, end) ->
    [end - begin][<= max_len] token ids -> gathered on rank 0 in window order (None elsewhere). No collective runs between
    the two ends: windows are independent (ContextImpl.cpp:476-477, NoContext)."""
    import torch.distributed as dist
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    b, e = shard_range(n_windows_total, rank, world)
    local = transcribe_local(b, e) if e > b else np.zeros((0, max_len), np.int32)
    return gather_window_tokens(local, n_windows_total, max_len)
