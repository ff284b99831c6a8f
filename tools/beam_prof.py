"""Kernel table of one beam-search window batch (BASELINE configs[2] shape: large-v2, WINDOWS windows x HYP hypotheses, STEPS forced steps), eager launches
with hipEvent pairs:   BEAM_WINDOWS=8 BEAM_HYP=5 BEAM_STEPS=50 python tools/beam_prof.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.r5_sweep import kernel_table  # noqa: E402


def main():
    import torch
    from whisper_amd import binding, ggml_format as gf
    kind = os.environ.get("BEAM_MODEL", "large-v2")
    k, hyp, n_steps = int(os.environ.get("BEAM_WINDOWS", "8")), int(os.environ.get("BEAM_HYP", "5")), int(os.environ.get("BEAM_STEPS", "50"))
    hp = gf.hparams_for(kind)
    sp = gf.special_tokens(hp)
    hm = binding.HipModel.from_ggml(gf.synth_model(kind, seed=1))
    c = binding.HipContext(hm, k, hypotheses=hyp)
    g = torch.Generator(device="cuda").manual_seed(1000)
    mel = torch.rand((k, hp.n_mels, 3000), generator=g, device="cuda") * 2.0 - 1.0
    base = np.asarray([sp["sot"], sp["sot"] + 1, sp["transcribe"]], np.int32)
    for rep in range(2):
        c.encode(mel)
        c.beam_window_start(np.tile(base, (k, 1)), hyp, n_steps)
        c.beam_window_status()
    import time
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    c.encode(mel, sync=False)
    c.beam_window_start(np.tile(base, (k, 1)), hyp, n_steps)
    c.beam_window_status()
    print("%s, %d windows x %d hypotheses, %d steps through the captured graph: %.2f ms" % (kind, k, hyp, n_steps, 1e3 * (time.perf_counter() - t0)))
    c.profile(True)
    c.encode(mel)
    c.beam_window_start(np.tile(base, (k, 1)), hyp, n_steps)
    c.beam_window_status()
    print(kernel_table(c.profile_read()))
    c.profile(False)


if __name__ == "__main__":
    main()
