"""CPU tests of the multi-GPU path (gloo, world_size 2): shard partition, the weight-arena broadcast and the result gather."""
import os
import socket

import numpy as np
import pytest

from whisper_amd import distributed as wd


def test_shard_range_is_a_balanced_partition():
    for n in (0, 1, 7, 8, 255, 256, 257):
        for world in (1, 2, 3, 8):
            spans = [wd.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)
    with pytest.raises(ValueError):
        wd.shard_range(4, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_windows, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # 1. weight arena: rank 0 holds the bytes, the others receive them in place
        n = 1 << 20
        arena = torch.zeros(n, dtype=torch.uint8)
        if rank == 0:
            arena = torch.from_numpy(np.random.default_rng(5).integers(0, 256, n, dtype=np.uint8))
        wd.broadcast_arena(arena, 0)
        want = np.random.default_rng(5).integers(0, 256, n, dtype=np.uint8)
        ok_arena = bool((arena.numpy() == want).all())
        # 2. each rank "transcribes" its windows: token ids derived from the window index
        b, e = wd.shard_range(n_windows, rank, world)
        local = np.array([[1000 * w + j for j in range(5)] for w in range(b, e)], np.int32).reshape(e - b, 5)
        out = wd.gather_window_tokens(local, n_windows, 8)
        if rank == 0:
            ok = ok_arena and out.shape == (n_windows, 8) and all(out[w, 0] == 1000 * w and out[w, 5] == -1 for w in range(n_windows))
            q.put(("rank0", bool(ok)))
        else:
            q.put(("rank%d" % rank, ok_arena and out is None))
    finally:
        dist.destroy_process_group()


def test_gloo_world_size_2_broadcast_and_gather():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 7, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert results == {"rank0": True, "rank1": True}
