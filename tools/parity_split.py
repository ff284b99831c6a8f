"""GPU: where the timed path's distance to the reference comes from, measured on the device against WH_FLAG_PARITY_EXACT (== oracle/_ref bit for bit).

For a model shape: one context, the same windows and teacher-forced tokens through
  T   the timed path (what bench.py measures),
  E1  the exact mode at 1 thread        (== the reference at 1 thread),
  E16 the exact mode at 16 threads      (== the reference at 16 threads),
  E0  the exact mode with the decoder's P.V rounded once (what every thread count of the reference approximates),
  and the two crossings: exact encoder + timed decoder, timed encoder + exact (E0) decoder.
Prints max / mean |logit differences| per pair.  Usage: python tools/parity_split.py medium [n_win] [steps] [sharpness]
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from whisper_amd import binding, ggml_format as gf
import bench


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "medium"
    n_win = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    n_steps = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    sharp = float(sys.argv[4]) if len(sys.argv) > 4 else 1.0
    model = gf.synth_model(kind, seed=1, attn_sharpness=sharp)
    hp = model.hparams
    sp = gf.special_tokens(hp)
    m = binding.HipModel.from_ggml(model)
    del model
    ctx = binding.HipContext(m, n_win)
    pcm_dev = torch.from_numpy(bench.synth_pcm(n_win, seed=100)).cuda()
    mels = torch.stack([ctx.mel_spectrogram(pcm_dev[b]) for b in range(n_win)])
    prompt = np.array([[sp["sot"], sp["sot"] + 1, sp["transcribe"]]] * n_win, np.int32)

    def run(enc_flags, dec_flags, tokens=None):
        """Returns (logits [steps + 1][n_win][vocab], the token fed at every step)."""
        t0 = time.time()
        ctx.set_flags(*enc_flags)
        ctx.encode(mels)
        ctx.synchronize()
        t1 = time.time()
        ctx.set_flags(*dec_flags)
        out, fed = [], []
        toks, n_past = prompt, 0
        for st in range(n_steps + 1):
            gl, _ = ctx.decode(toks, n_past)
            out.append(gl.copy())
            fed.append(toks.copy())
            n_past += toks.shape[1]
            toks = (tokens[st + 1] if tokens is not None else np.argmax(gl, axis=1).astype(np.int32).reshape(-1, 1))
        print("    (encode %.2f s, decode %.2f s)" % (t1 - t0, time.time() - t1), flush=True)
        return np.stack(out), fed

    X = binding.WH_FLAG_PARITY_EXACT
    T, fed = run((0, 1), (0, 1))
    fed = fed + [None]
    runs = {"T": T}
    for name, ef, df in (("E1", (X, 1), (X, 1)), ("E16", (X, 16), (X, 16)), ("E0", (X, 0), (X, 0)), ("encE+decT", (X, 1), (0, 1)), ("encT+decE0", (0, 1), (X, 0))):
        print("running", name, flush=True)
        runs[name], _ = run(ef, df, tokens=fed)

    print("running E0alt (the weight products' 32 chains added left to right instead of in ggml's tree)", flush=True)
    binding.set_option("exact_alt_order", 1)
    runs["E0alt"], _ = run((X, 0), (X, 0), tokens=fed)
    binding.set_option("exact_alt_order", 0)

    def show(a, b):
        d = np.abs(runs[a] - runs[b])
        eq = int((runs[a].argmax(-1) == runs[b].argmax(-1)).sum())
        print("%-12s vs %-12s logits max %.3e  mean %.3e  | per step max %s | top-1 equal %d / %d" % (a, b, d.max(), d.mean(), " ".join("%.1e" % x for x in d.max(axis=(1, 2))), eq, d.shape[0] * d.shape[1]), flush=True)

    span = float((runs["E0"].max(-1) - runs["E0"].min(-1)).mean())
    print("%s, attn_sharpness %.1f, %d windows, prompt + %d steps; logit span %.2f, |logit| max %.2f" % (kind, sharp, n_win, n_steps, span, float(np.abs(runs["E0"]).max())))
    for a, b in (("E0alt", "E0"), ("E1", "E16"), ("E1", "E0"), ("E16", "E0"), ("T", "E0"), ("T", "E1"), ("T", "E16"), ("encE+decT", "E0"), ("encT+decE0", "E0")):
        show(a, b)


if __name__ == "__main__":
    main()
