"""In-process A/B of kernel variants (wh_debug_set_tuning): same model, same clip, the bench's transcribe_clip loop, the
variants interleaved so box-to-box and clock drift cancel. Usage: python tools/ab_bench.py [--model medium] [--rounds 3]
Prints one line per (round, mask): ms per clip pass."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

NAMES = {2: "rows4", 8: "gemmBig", 32: "gemvSmallReg", 64: "gemmGlds", 128: "lnSeparateBigM", 256: "attnXcd"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="medium")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--masks", default="490,488,426,362,234")
    args = ap.parse_args()
    import torch
    from whisper_amd import binding, ggml_format as gf
    hp = gf.hparams_for(args.model)
    sp = gf.special_tokens(hp)
    prompt = [sp["sot"], sp["sot"] + 1, sp["transcribe"]] if hp.is_multilingual else [sp["sot"], sp["not_"], sp["beg"]]
    model = gf.synth_model(args.model, seed=1)
    hip_model = binding.HipModel.from_ggml(model)
    B = 7
    pcm = torch.from_numpy(bench.synth_pcm(B, seed=100)).cuda()
    mel = torch.empty((B, hp.n_mels, 3000), dtype=torch.float32, device="cuda")
    masks = [int(m) for m in args.masks.split(",")]
    ctxs = {}
    for m in masks:
        binding.lib().wh_debug_set_tuning(m)
        ctxs[m] = binding.HipContext(hip_model, B)
        bench.transcribe_clip([(ctxs[m], pcm, mel)], prompt, bench.N_GREEDY)      # capture the graph under this mask
    torch.cuda.synchronize()
    res = {m: [] for m in masks}
    for r in range(args.rounds):
        for m in masks:
            binding.lib().wh_debug_set_tuning(m)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                toks = bench.transcribe_clip([(ctxs[m], pcm, mel)], prompt, bench.N_GREEDY)
            torch.cuda.synchronize()
            ms = 1e3 * (time.perf_counter() - t0) / args.steps
            res[m].append(ms)
            print("round %d mask %6d %-36s %8.2f ms  checksum %d" % (r, m, "+".join(n for b, n in NAMES.items() if m & b) or "-", ms,
                                                                      int(np.asarray(toks, np.int64).sum() % 1000003)), flush=True)
    for m in masks:
        print("mask %2d best %8.2f ms  median %8.2f ms" % (m, min(res[m]), float(np.median(res[m]))))
    # per-kernel tables for the full and the empty mask
    for m in (masks[0], masks[-1]):
        binding.lib().wh_debug_set_tuning(m)
        ctxs[m].profile(True)
        bench.transcribe_clip([(ctxs[m], pcm, mel)], prompt, bench.N_GREEDY)
        print("mask", m, {k: (v["calls"], round(v["ms"], 2)) for k, v in ctxs[m].profile_read().items()})
        ctxs[m].profile(False)


if __name__ == "__main__":
    main()
