"""Round 5: lock-step batch size x contexts in flight x the big-batch decode options, one model build, bench.py's own measure_batched.

    python tools/r5_sweep.py plans  "16x2:32,32x1:32,64x1:64,32x2:64,20x1:20,10x2:20"     # CxI:steps  -> ms per clip pass, audio-s/s
    python tools/r5_sweep.py options 32 "default;dec_tile=1;dec_tile=44;dec_tile=42;self_nq=4;self_nq=8;self_fuse_max_rows=128;vocab_decrows=1"
                                                                                             # clips in ONE batch; per option: ms per batch + kernel table
SWEEP_MODEL=large-v2 for the other shape. The options are csrc/kernels.h struct Options (wh_debug_set_option)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

DEFAULTS = {"dec_tile": 0, "dec_depth": 0, "dec_wide_rows": 1, "dec_deep_rows": 0, "vocab_decrows": 0, "enc_chunk": 128, "self_fuse_max_rows": 32, "self_nq": 0, "self_wave_min_rows": 32}


def kernel_table(prof, batches=1):
    pair = prof.get("eventPair", {"ms": 0, "calls": 1})
    cal = max(0.0, pair["ms"] / max(pair["calls"], 1) - 1.9e-3)
    rows = []
    for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"]):
        if k == "eventPair" or not v["calls"]:
            continue
        net = max(v["ms"] - v["calls"] * cal, 0.0)
        rows.append("   %-18s calls %6d  ms %9.3f  avg %8.2f us  %8.1f GB/s  %8.1f TF/s" % (k, v["calls"], net / batches, 1e3 * net / v["calls"],
                                                                                         v["bytes"] / max(net, 1e-9) / 1e6, v["flops"] / max(net, 1e-9) / 1e9))
    return "\n".join(rows)


def main():
    import torch
    import torch.distributed as dist
    from whisper_amd import binding, ggml_format as gf
    mode = sys.argv[1]
    kind = os.environ.get("SWEEP_MODEL", "medium")
    hp = gf.hparams_for(kind)
    sp = gf.special_tokens(hp)
    prompt = [sp["sot"], sp["sot"] + 1, sp["transcribe"]]
    hm = binding.HipModel.from_ggml(gf.synth_model(kind, seed=1))
    if mode == "plans":
        for item in sys.argv[2].split(","):
            ci, steps = item.split(":")
            explicit = None
            if ci.startswith("p"):          # "p9+11:2" = the explicit batch plan [9, 11] with 2 contexts in flight
                explicit = [int(x) for x in ci[1:].split("+")]
                C, I, steps = max(explicit), int(steps), sum(explicit)
            else:
                C, I = (int(x) for x in ci.split("x"))
                steps = int(steps)
            t0 = time.time()
            best = 1e9
            for rep in range(int(os.environ.get("SWEEP_REPS", "2"))):
                m = bench.measure_batched(hm, hp, prompt, steps, 1 if rep == 0 else 0, 7, C, I, 0, 1, dist, want_kernels=False, plan=explicit)
                best = min(best, m["elapsed"])
                plan = m["plan"]
                for s in m["slots"]:
                    s[0].close()
                del m
                torch.cuda.empty_cache()
            ms = 1e3 * best / steps
            print("%s clips/batch <= %2d in flight %d, %3d passes, plan %-14s: %7.2f ms per clip pass  %8.1f audio-s/s   (%.1f s incl. setup)"
                  % (kind, C, I, steps, plan, ms, bench.CLIP_SECONDS / (ms * 1e-3), time.time() - t0), flush=True)
        return
    clips = int(sys.argv[2])
    B = clips * 7
    specs = sys.argv[3].split(";")
    pcm = torch.from_numpy(np.concatenate([bench.synth_pcm(7, seed=100 + 1000 * j) for j in range(clips)])).cuda()
    mel = torch.empty((B, hp.n_mels, 3000), dtype=torch.float32, device="cuda")
    rounds = int(os.environ.get("SWEEP_REPS", "2"))
    res = {s: [] for s in specs}
    sums, tables = {}, {}
    for r in range(rounds):
        for spec in specs:
            # one context at a time (a 448-window context holds ~90 GB): created, warmed (graph capture), timed, closed
            opts = dict(kv.split("=") for kv in spec.split(",") if "=" in kv)
            for k, v in DEFAULTS.items():
                binding.set_option(k, int(opts.get(k, v)))
            g = (binding.HipContext(hm, B), None, pcm, mel)
            bench.run_passes([g], prompt, bench.N_GREEDY, 1)
            for _ in range(2):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                toks = bench.run_passes([g], prompt, bench.N_GREEDY, 1)
                torch.cuda.synchronize()
                res[spec].append(1e3 * (time.perf_counter() - t0))
            sums[spec] = int(np.asarray(toks, np.int64).sum() % 1000003)
            if r == 0:
                g[0].profile(True)
                bench.run_passes([g], prompt, bench.N_GREEDY, 1)
                tables[spec] = kernel_table(g[0].profile_read())
                g[0].profile(False)
            g[0].close()
            torch.cuda.empty_cache()
    for spec in specs:
        ms = min(res[spec])
        print("%s %3d windows in ONE lock-step batch, %-34s best %8.2f ms  median %8.2f ms  = %8.1f audio-s/s   ids checksum %d"
              % (kind, B, spec, ms, float(np.median(res[spec])), clips * bench.CLIP_SECONDS / (ms * 1e-3), sums[spec]), flush=True)
    for spec in specs:
        print("-- kernel table, %s, %d windows, %s" % (kind, B, spec))
        print(tables[spec], flush=True)
    for k, v in DEFAULTS.items():
        binding.set_option(k, v)


if __name__ == "__main__":
    main()
