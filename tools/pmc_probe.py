"""A few eager (no hipGraph) launches of the decode kernels for a counter pass:
    cd /tmp && rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- python tools/pmc_probe.py
Medium shape, 7 windows: encoder once, the 3-token prompt step, then two single-token steps (wh_decode launches its
kernels directly). FETCH_SIZE per gemvFused dispatch against its algorithmic weight bytes is `roofline.traffic`."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    import torch
    from whisper_amd import binding, ggml_format as gf
    kind = os.environ.get("PMC_MODEL", "medium")
    hp = gf.hparams_for(kind)
    sp = gf.special_tokens(hp)
    m = binding.HipModel.from_ggml(gf.synth_model(kind, seed=1))
    B = 7
    ctx = binding.HipContext(m, B)
    mel = torch.from_numpy(np.random.default_rng(0).uniform(-1, 1, (B, hp.n_mels, 3000)).astype(np.float32)).cuda()
    ctx.encode(mel)
    prompt = np.tile(np.array([sp["sot"], sp["sot"] + 1, sp["transcribe"]], np.int32), (B, 1))
    ctx.decode(prompt, 0, want_logits=False, want_probs=False)
    for i in range(2):
        ctx.decode(np.full((B, 1), 1000 + i, np.int32), 3 + i, want_logits=False, want_probs=False)
    ctx.synchronize()
    print("done")


if __name__ == "__main__":
    main()
