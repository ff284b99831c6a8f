"""In-process A/B of kernel variants (wh_debug_set_tuning): same model, same clips, bench.py's clip_start / clip_finish loop,
the variants interleaved so box-to-box and clock drift cancel.
    python tools/ab_bench.py [--model medium] [--rounds 3] [--windows 28] [--inflight 1|3] [--masks default,-512,-1024,...]
A mask is an absolute value, "default", or "-BIT" (default with that bit cleared) / "+BIT". Prints ms per batch pass."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

NAMES = {2: "rows4", 8: "gemmBig", 32: "gemvSmallReg", 64: "gemmGlds", 128: "lnSeparateBigM", 256: "attnXcd", 512: "attnDecG",
         1024: "fuseCrossQ", 2048: "gemmGroupM", 4096: "selfBlock", 16384: "gemvK8", 65536: "attnEncF", 131072: "wideEpi", 262144: "fragPf", 524288: "attnEnc2Sweep",
         1048576: "allRows", 2097152: "rowGroups", 4194304: "selfMfma"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="medium")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--windows", type=int, default=28)
    ap.add_argument("--inflight", type=int, default=1)
    ap.add_argument("--masks", default="default,-512,-1024,-2048")
    ap.add_argument("--kernels", action="store_true", help="per-kernel tables for the first and the last mask")
    args = ap.parse_args()
    import torch
    from whisper_amd import binding, ggml_format as gf
    hp = gf.hparams_for(args.model)
    sp = gf.special_tokens(hp)
    prompt = [sp["sot"], sp["sot"] + 1, sp["transcribe"]] if hp.is_multilingual else [sp["sot"], sp["not_"], sp["beg"]]
    model = gf.synth_model(args.model, seed=1)
    hip_model = binding.HipModel.from_ggml(model)
    B = args.windows

    def parse(m):
        if m == "default":
            return binding.TUNE_DEFAULT
        if m.startswith("-"):
            return binding.TUNE_DEFAULT & ~int(m[1:])
        if m.startswith("+"):
            return binding.TUNE_DEFAULT | int(m[1:])
        return int(m)

    masks = [parse(m) for m in args.masks.split(",")]
    pcm = torch.from_numpy(np.concatenate([bench.synth_pcm(7, seed=100 + 1000 * j) for j in range((B + 6) // 7)])[:B]).cuda()
    groups = {}
    for m in masks:
        binding.lib().wh_debug_set_tuning(m)
        groups[m] = [(binding.HipContext(hip_model, B), None, pcm, torch.empty((B, hp.n_mels, 3000), dtype=torch.float32, device="cuda"))
                     for _ in range(args.inflight)]
        bench.run_passes(groups[m], prompt, bench.N_GREEDY, args.inflight)      # capture the graphs under this mask
    torch.cuda.synchronize()
    res = {m: [] for m in masks}
    for r in range(args.rounds):
        for m in masks:
            binding.lib().wh_debug_set_tuning(m)
            seq = [groups[m][i % args.inflight] for i in range(args.steps * args.inflight)]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            toks = bench.run_passes(seq, prompt, bench.N_GREEDY, args.inflight)
            torch.cuda.synchronize()
            ms = 1e3 * (time.perf_counter() - t0) / len(seq)
            res[m].append(ms)
            print("round %d mask %6d %-60s %8.2f ms/batch  checksum %d" % (r, m, "+".join(n for b, n in NAMES.items() if m & b) or "-", ms,
                                                                            int(np.asarray(toks, np.int64).sum() % 1000003)), flush=True)
    for m in masks:
        print("mask %5d best %8.2f ms  median %8.2f ms   (%d windows per batch, %d in flight)" % (m, min(res[m]), float(np.median(res[m])), B, args.inflight))
    if args.kernels:
        for m in (masks[0], masks[-1]):
            binding.lib().wh_debug_set_tuning(m)
            g = groups[m][0]
            g[0].profile(True)
            bench.run_passes([g], prompt, bench.N_GREEDY, 1)
            prof = g[0].profile_read()
            pair = prof.get("eventPair", {"ms": 0, "calls": 1})
            cal = pair["ms"] / max(pair["calls"], 1)
            print("mask", m, "event pair %.2f us" % (1e3 * cal))
            for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"]):
                if k == "eventPair":
                    continue
                net = max(v["ms"] - v["calls"] * cal, 0.0)
                print("   %-18s calls %6d  ms %9.3f  avg %8.2f us  %8.1f GB/s  %8.1f TF/s" % (k, v["calls"], net, 1e3 * net / v["calls"],
                                                                                           v["bytes"] / max(net, 1e-9) / 1e6, v["flops"] / max(net, 1e-9) / 1e9))
            g[0].profile(False)


if __name__ == "__main__":
    main()
