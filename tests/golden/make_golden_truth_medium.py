"""Generates tests/golden/truth_medium.npz: the exact-arithmetic yardstick AT THE MEASURED SHAPE.

Run in the build container: make -C oracle && python tests/golden/make_golden_truth_medium.py   (~3 min, ~20 GB of RAM)

ggml-medium shape, the bench's random model (seed 1) and the bench's window 0 (bench.synth_pcm(1, seed=100)): the float64 spectrogram of
oracle/whisper_np.py, then WhisperTruth (float64, no intermediate rounding: what the reference's graph computes in exact arithmetic) for the
3-token prompt and three teacher-forced steps with fixed ids. Next to it the reference CPU path (oracle/_ref, 8 threads) on the same inputs:
its distance from the exact result is the yardstick the HIP path is held against (tests/test_gpu_model.py::test_medium_shape_against_exact_arithmetic);
north_star's 1e-3 on logits is a statement about two implementations, each of which sits several 1e-3 from the exact result at this shape."""
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from whisper_amd import ggml_format as gf  # noqa: E402
from oracle import ref, whisper_np as wn  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
STEPS_EXTRA = [1000, 1001, 1002]          # teacher-forced ids after the prompt's sample (text tokens: the steps do not depend on a sampler)


def main():
    t0 = time.time()
    model = gf.synth_model("medium", seed=1)
    hp = model.hparams
    sp = gf.special_tokens(hp)
    prompt = [sp["sot"], sp["sot"] + 1, sp["transcribe"]]
    pcm = bench.synth_pcm(1, seed=100)[0]
    mel = wn.log_mel_spectrogram(pcm, model.filters).astype(np.float32)
    steps = [prompt] + [[t] for t in STEPS_EXTRA]
    out = {"mel": mel, "prompt": np.array(prompt, np.int32), "extra": np.array(STEPS_EXTRA, np.int32)}
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "m.bin")
        gf.write_model(path, model)
        w = ref.RefWhisper(path, n_threads=8, log_level=0)
        w.set_mel(mel)
        w.encode(0)
        ref_logits, n_past = [], 0
        for toks in steps:
            ref_logits.append(w.decode(toks, n_past)[0][-1].astype(np.float64))
            n_past += len(toks)
        w.close()
    print("reference done after %.0f s" % (time.time() - t0), flush=True)
    tr = wn.WhisperTruth(model)
    del model
    tr.encode(mel.astype(np.float64), 0)
    print("exact encoder done after %.0f s" % (time.time() - t0), flush=True)
    stats, n_past = [], 0
    for i, toks in enumerate(steps):
        tl = tr.decode(toks, n_past)[-1]
        n_past += len(toks)
        d = np.abs(ref_logits[i] - tl)
        out["truth_logits%d" % i] = tl.astype(np.float32)
        stats.append(dict(step=i, ref8_vs_truth_max=float(d.max()), ref8_vs_truth_mean=float(d.mean()), span=float(tl.max() - tl.min()),
                          truth_top1=int(np.argmax(tl)), ref8_top1=int(np.argmax(ref_logits[i])), truth_top2_margin=float(np.sort(tl)[-1] - np.sort(tl)[-2])))
        print(stats[-1], flush=True)
    out["stats"] = np.asarray(json.dumps(stats))
    np.savez_compressed(os.path.join(HERE, "truth_medium.npz"), **out)
    print("wrote truth_medium.npz, %.2f MB, %.0f s" % (os.path.getsize(os.path.join(HERE, "truth_medium.npz")) / 1e6, time.time() - t0))


if __name__ == "__main__":
    main()
