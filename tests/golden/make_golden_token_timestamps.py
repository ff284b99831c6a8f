"""Golden token-level timestamps (TokenTimestamps flag, thold_pt / thold_ptsum / max_len) of the reference's CPU model
(whisper_exp_compute_token_level_timestamps + whisper_wrap_segment, whisper.cpp:3374-3575, 2711-2760) on scripted models.
For every case the reference runs twice: max_len = 0 gives the unsplit segments (the INPUT of the host post-processing:
segment times and the tokens' id / tid / p / pt / ptsum) and their token times; max_len = L gives the wrapped segments.
Run in the build container: make -C oracle && python tests/golden/make_golden_token_timestamps.py"""
import json
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from whisper_amd import ggml_format as gf  # noqa: E402
from oracle import ref  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def bursty_pcm(n, seed):
    """Noise in speech-like bursts, so that the voice-activity step has edges to snap to."""
    rng = np.random.default_rng(seed)
    t = np.arange(n, dtype=np.float64) / 16000.0
    env = 0.004 + 0.25 * (np.sin(2 * np.pi * t / 0.83) > 0.2) * (0.6 + 0.4 * np.sin(2 * np.pi * t / 0.31) ** 2)
    return (rng.standard_normal(n) * env).astype(np.float32)


def cases():
    hp = gf.hparams_for("test-d128-ml")
    sp = gf.special_tokens(hp)
    beg, eot = sp["beg"], sp["eot"]
    a = [beg, 300, 301, 302, 303, beg + 120, beg + 120, 400, 401, 402, beg + 250, beg + 250, 500, 501, 502, 503, 504, beg + 420, eot]
    b = [beg + 10, 600, 601, 602, beg + 200, beg + 200, 610, 611, 612, 613, eot]
    return [
        dict(name="default_tholds", script=a, seconds=12.0, thold_pt=0.01, thold_ptsum=0.01, max_len=14),
        dict(name="anchors_on_any_probability", script=a, seconds=12.0, thold_pt=0.0, thold_ptsum=0.0, max_len=9),
        dict(name="open_ended_two_windows", script=b, seconds=43.0, thold_pt=0.01, thold_ptsum=0.01, max_len=11),
        dict(name="tiny_max_len", script=a, seconds=9.5, thold_pt=0.01, thold_ptsum=0.01, max_len=1),
    ]


def main():
    out = []
    for k, c in enumerate(cases()):
        model = gf.scripted_model(c["script"], 3)
        n = int(16000 * c["seconds"])
        pcm = bursty_pcm(n, 40 + k)
        with tempfile.TemporaryDirectory() as td:
            path = os.path.join(td, "m.bin")
            gf.write_model(path, model)
            w = ref.RefWhisper(path, n_threads=4, log_level=0)
            plain = w.full_token_timestamps(pcm, thold_pt=c["thold_pt"], thold_ptsum=c["thold_ptsum"], max_len=0)
            wrapped = w.full_token_timestamps(pcm, thold_pt=c["thold_pt"], thold_ptsum=c["thold_ptsum"], max_len=c["max_len"])
            w.close()
        print(c["name"], len(plain), "segments ->", len(wrapped), "after wrapping to", c["max_len"],
              [(t["t0"], t["t1"]) for t in plain[0]["tokens"]])
        out.append(dict(name=c["name"], script=c["script"], prompt_len=3, n_samples=n, pcm_seed=40 + k, thold_pt=c["thold_pt"],
                        thold_ptsum=c["thold_ptsum"], max_len=c["max_len"], plain=plain, wrapped=wrapped))
    with open(os.path.join(HERE, "ref_token_timestamps.json"), "w") as f:
        json.dump(dict(pcm="bursty_pcm(n, seed) of make_golden_token_timestamps.py", cases=out), f, indent=1)


if __name__ == "__main__":
    main()
