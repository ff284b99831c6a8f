"""Helper of tests/test_00_canary.py::test_results_do_not_depend_on_stale_device_memory (run as a subprocess).

Runs the hot path on the d = 128 test model -- spectrogram, encoder, a parity-mode and a fast-path decode step, a greedy
window through the captured graph at batch 1, and the lock-step batch path at 10 windows (fused self-attention block) --
and prints one JSON line of digests. The parent runs it with and without WH_DEBUG_POISON and compares the lines.
"""
import hashlib
import json
import sys

import numpy as np
import torch

from whisper_amd import binding, ggml_format as gf


def digest(a) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


def main():
    out = {}
    model = gf.synth_model("test-d128", seed=7, attn_sharpness=2.0)
    sp = gf.special_tokens(model.hparams)
    hm = binding.HipModel.from_ggml(model)
    rng = np.random.default_rng(3)
    pcm = (0.1 * rng.standard_normal(16000 * 11 + 77)).astype(np.float32)
    for batch in (1, 10):
        ctx = binding.HipContext(hm, batch)
        mel = ctx.mel_spectrogram(torch.from_numpy(pcm).cuda())
        out["mel"] = digest(mel.cpu().numpy())
        melb = mel.unsqueeze(0).repeat(batch, 1, 1).contiguous()
        offs = [100 * i for i in range(batch)]
        for parity in (1, 0):
            ctx.set_parity(parity)
            ctx.encode(melb, offs)
            toks = np.tile(np.asarray([[sp["sot"], sp["not_"]]], np.int32), (batch, 1))
            logits, probs = ctx.decode(toks, 0)
            out["logits_b%d_p%d" % (batch, parity)] = digest(logits)
            logits1, _ = ctx.decode(np.full((batch, 1), sp["beg"], np.int32), 2)
            out["logits1_b%d_p%d" % (batch, parity)] = digest(logits1)
            assert np.isfinite(logits).all() and np.isfinite(logits1).all()
        ctx.encode(melb, offs)
        ctx.decode_window_start(np.full((batch, 1), sp["sot"], np.int32), 6, force_first_timestamp=True, first_is_initial=True)
        ctx.decode_window_continue(3)
        ids, ps = ctx.decode_window_finish()
        out["ids_b%d" % batch] = [int(x) for x in ids.reshape(-1)]
        out["ps_b%d" % batch] = digest(ps)
        # the streamed spectrogram (runStreamed's per-window normalisation)
        w = ctx.mel_spectrogram_window(torch.from_numpy(pcm).cuda(), 200, 700)
        out["melwin_b%d" % batch] = digest(w.cpu().numpy())
        ctx.close()
    hm.close()
    print("PROBE " + json.dumps(out, sort_keys=True))


if __name__ == "__main__":
    sys.exit(main())
