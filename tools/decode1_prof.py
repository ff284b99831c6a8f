"""Batch-1 decode steps of the medium shape alone (random weights), for rocprofv3 --kernel-trace --stats and for wall-clock per token:
    python tools/decode1_prof.py            # prints us per token through the captured greedy graph, prompt-step time, encode time
Env: D1_MODEL (medium), D1_STEPS (200), D1_PROMPT (105 prompt tokens), D1_BATCH (1)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from whisper_amd import binding, ggml_format as gf  # noqa: E402


def main():
    kind = os.environ.get("D1_MODEL", "medium")
    steps = int(os.environ.get("D1_STEPS", "200"))
    n_prompt = int(os.environ.get("D1_PROMPT", "105"))
    batch = int(os.environ.get("D1_BATCH", "1"))
    model = gf.synth_model(kind, seed=3)
    hp = model.hparams
    m = binding.HipModel.from_ggml(model)
    del model
    ctx = binding.HipContext(m, batch)
    g = torch.Generator(device="cuda").manual_seed(5)
    mel = torch.rand((batch, hp.n_mels, 3000), generator=g, device="cuda") * 2.0 - 1.0
    for _ in range(2):
        ctx.encode(mel)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        ctx.encode(mel)
    ctx.synchronize()
    print("encode batch %d: %.3f ms" % (batch, (time.perf_counter() - t0) / 5 * 1e3), flush=True)
    rng = np.random.default_rng(1)
    prompt = rng.integers(1000, 40000, size=(batch, n_prompt)).astype(np.int32)
    ctx.decode(prompt, 0, want_logits=False, want_probs=False)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        ctx.decode(prompt, 0, want_logits=False, want_probs=False)
    ctx.synchronize()
    print("prompt step, %d tokens: %.3f ms" % (n_prompt, (time.perf_counter() - t0) / 5 * 1e3), flush=True)
    first = np.full((batch,), 1234, np.int32)
    ctx.decode_greedy(first, n_prompt, 8)        # captures the graph
    for rep in range(3):
        t0 = time.perf_counter()
        ctx.decode_greedy(first, n_prompt, steps)
        dt = time.perf_counter() - t0
        print("greedy graph: %d steps from n_past %d: %.1f us per token" % (steps, n_prompt, dt / steps * 1e6), flush=True)
    # the host loop's pattern: one captured step per launch, an event behind each; (a) nobody waits, (b) the host fetches every
    # sample while one step runs ahead of the one it reads (WindowDecoder in host/whisperImpl.cpp)
    n = min(steps, hp.n_text_ctx - n_prompt - 8)
    for lead in (0, 1, 2, 3, 4):
        ctx.decode_window_start(prompt[:, :n_prompt], 1)
        ctx.decode_window_fetch(0, 1)
        t0 = time.perf_counter()
        enq = 2                                   # samples enqueued (the first one and one greedy step)
        for _ in range(lead - 1):
            ctx.decode_window_continue(1)
            enq += 1
        for i in range(1, n):
            if lead:
                ctx.decode_window_fetch(i, 1)       # sample i exists ...
            ctx.decode_window_continue(1)           # ... and `lead` steps are in flight behind it again
            enq += 1
        ctx.decode_window_finish()
        dt = time.perf_counter() - t0
        print("window loop (%s): %.1f us per token" % ("no fetch" if not lead else "fetch, %d step(s) ahead" % lead, dt / (enq - 1) * 1e6), flush=True)
    ctx.close()
    m.close()


if __name__ == "__main__":
    main()
