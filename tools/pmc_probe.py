"""Eager (no hipGraph) launches of the hot path for a rocprofv3 counter pass -- one batch of the bench's shape:
    cd /tmp && rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out_fetch -- python tools/pmc_probe.py
    cd /tmp && rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d out_write -- python tools/pmc_probe.py
Medium shape, PMC_WINDOWS windows (default 28 = the bench's lock-step batch): mel + encoder once, the 3-token prompt step, then
PMC_STEPS single-token steps (wh_decode launches its kernels directly). Writes the library's own per-class ALGORITHMIC bytes /
flops of exactly these launches to $PMC_ALGO (JSON), which tools/pmc_summary.py puts next to the counters."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    import torch
    from whisper_amd import binding, ggml_format as gf
    kind = os.environ.get("PMC_MODEL", "medium")
    B = int(os.environ.get("PMC_WINDOWS", "28"))
    steps = int(os.environ.get("PMC_STEPS", "3"))
    hp = gf.hparams_for(kind)
    sp = gf.special_tokens(hp)
    m = binding.HipModel.from_ggml(gf.synth_model(kind, seed=1))
    ctx = binding.HipContext(m, B)
    pcm = torch.from_numpy(bench.synth_pcm(7, seed=100)).cuda()
    mel = torch.stack([ctx.mel_spectrogram(pcm[b % 7]) for b in range(B)])
    ctx.profile(True)
    ctx.encode(mel)
    prompt = np.tile(np.array([sp["sot"], sp["sot"] + 1, sp["transcribe"]], np.int32), (B, 1))
    ctx.decode(prompt, 0, want_logits=False, want_probs=False)
    for i in range(steps):
        ctx.decode(np.full((B, 1), 1000 + i, np.int32), 3 + i, want_logits=False, want_probs=False)
    ctx.synchronize()
    prof = ctx.profile_read()
    out = os.environ.get("PMC_ALGO")
    if out:
        with open(out, "w") as f:
            json.dump({"model": kind, "windows": B, "steps": steps, "classes": prof}, f)
    print("done")


if __name__ == "__main__":
    main()
