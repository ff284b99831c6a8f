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


def _transcribe_windows_cpu(model, mels, begin, end, n_steps):
    """The oracle's numpy model as the per-rank compute stand-in (there is no GPU here): greedy token ids of windows
    [begin, end) -- encode, the 3-token prompt, n_steps argmax steps. Same role as HipContext.decode_window_* on a GPU rank."""
    from oracle import whisper_np as wn
    from whisper_amd import ggml_format as gf
    sp = gf.special_tokens(model.hparams)
    n = wn.WhisperNP(model)
    rows = []
    for w in range(begin, end):
        n.encode(mels[w], 0)
        toks = [sp["sot"], sp["sot"] + 1, sp["transcribe"]] if model.hparams.n_vocab >= 51865 else [sp["sot"]]
        logits, _ = n.decode(toks, 0, exact_pv=False)
        n_past, out = len(toks), []
        for _ in range(n_steps):
            t = int(np.argmax(logits[-1]))
            out.append(t)
            logits, _ = n.decode([t], n_past, exact_pv=False)
            n_past += 1
        rows.append(out)
    return np.array(rows, np.int32).reshape(end - begin, n_steps)


def _model_worker(rank, world, port, n_windows, n_steps, path, q):
    """Rank 0 reads the ggml file and broadcasts its BYTES (the stand-in for the packed arena); every rank parses the model
    from what it received, transcribes its contiguous shard of windows and enters the gather."""
    import tempfile
    import torch
    import torch.distributed as dist
    from whisper_amd import ggml_format as gf
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        size = torch.zeros(1, dtype=torch.int64)
        if rank == 0:
            raw = np.fromfile(path, dtype=np.uint8)
            size[0] = raw.size
        dist.broadcast(size, 0)
        arena = torch.from_numpy(raw.copy()) if rank == 0 else torch.zeros(int(size[0]), dtype=torch.uint8)
        wd.broadcast_arena(arena, 0)
        with tempfile.TemporaryDirectory() as td:
            mine = os.path.join(td, "m.bin")
            arena.numpy().tofile(mine)
            model = gf.read_model(mine)
        rng = np.random.default_rng(17)
        mels = rng.uniform(-1, 1, (n_windows, model.hparams.n_mels, 3000)).astype(np.float32)      # same on every rank (seeded)
        out = wd.transcribe_sharded(n_windows, lambda b, e: _transcribe_windows_cpu(model, mels, b, e, n_steps), n_steps)
        q.put((rank, None if out is None else out.tolist()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_windows", [3, 1])
def test_gloo_world_size_2_shard_transcribe_gather(tmp_path, n_windows):
    """The whole data-parallel step on two ranks with real window tokens: broadcast of the model bytes, contiguous shards
    (3 windows -> 2 + 1; 1 window -> 1 + 0: a rank with an empty shard still has to enter the collective), per-rank compute,
    gather on rank 0 in window order -- equal to one process transcribing every window."""
    import torch.multiprocessing as mp
    from whisper_amd import ggml_format as gf
    n_steps = 2
    model = gf.synth_model("test-d128", seed=77, n_audio_ctx=200)
    path = str(tmp_path / "m.bin")
    gf.write_model(path, model)
    rng = np.random.default_rng(17)
    mels = rng.uniform(-1, 1, (n_windows, model.hparams.n_mels, 3000)).astype(np.float32)
    want = _transcribe_windows_cpu(gf.read_model(path), mels, 0, n_windows, n_steps)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_model_worker, args=(r, 2, port, n_windows, n_steps, path, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert results[1] is None
    assert np.array_equal(np.array(results[0], np.int32), want)


def test_the_collective_library_is_packaged_as_the_loader_expects():
    """No GPU needed: wh_comm_* open librccl.so by name on first use (nothing is linked). The names they try must be what the image ships under
    /opt/rocm/lib, and every entry point they call must resolve -- so that the first multi-GPU lease cannot fail before its first timed step for a
    packaging reason."""
    import ctypes as C
    from whisper_amd import binding
    L = binding.lib()
    L.wh_comm_runtime_check.argtypes = [C.c_char_p, C.c_size_t]
    buf = C.create_string_buffer(512)
    rc = L.wh_comm_runtime_check(buf, 512)
    print(buf.value.decode())
    shipped = set(os.listdir("/opt/rocm/lib")) if os.path.isdir("/opt/rocm/lib") else set()
    assert {"librccl.so", "librccl.so.1"} & shipped, "the image does not ship librccl under /opt/rocm/lib"
    assert rc == 0, buf.value.decode()
    assert b"ncclBroadcast" in buf.value and b"librccl.so" in buf.value

