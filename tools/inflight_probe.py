"""Where does the time go with several clip passes in flight? Host enqueue time of one pass (mel + encoder + 52 graph
launches) and throughput with one host thread per in-flight context (ctypes releases the GIL inside the C ABI calls).
Usage: python tools/inflight_probe.py [--model medium]"""
import argparse
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="medium")
    ap.add_argument("--clips", type=int, default=24)
    args = ap.parse_args()
    import torch
    from whisper_amd import binding, ggml_format as gf
    hp = gf.hparams_for(args.model)
    sp = gf.special_tokens(hp)
    prompt = [sp["sot"], sp["sot"] + 1, sp["transcribe"]] if hp.is_multilingual else [sp["sot"], sp["not_"], sp["beg"]]
    hip_model = binding.HipModel.from_ggml(gf.synth_model(args.model, seed=1))
    B = 7
    pcm = torch.from_numpy(bench.synth_pcm(B, seed=100)).cuda()
    NMAX = 8
    slots = [(binding.HipContext(hip_model, B), pcm, torch.empty((B, hp.n_mels, 3000), dtype=torch.float32, device="cuda")) for _ in range(NMAX)]
    for s in slots:
        bench.transcribe_clip([s], prompt, bench.N_GREEDY)
    torch.cuda.synchronize()
    # host enqueue time of one pass
    t0 = time.perf_counter()
    bench.clip_start(slots[0], prompt, bench.N_GREEDY)
    t1 = time.perf_counter()
    bench.clip_finish(slots[0])
    t2 = time.perf_counter()
    print("one pass: host enqueue %.2f ms, then %.2f ms until the results are back" % (1e3 * (t1 - t0), 1e3 * (t2 - t1)), flush=True)
    for n in (1, 2, 3, 4, 6, 8):
        # single host thread, n in flight
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        bench.run_passes([slots[i % n] for i in range(args.clips)], prompt, bench.N_GREEDY, n)
        torch.cuda.synchronize()
        single = 1e3 * (time.perf_counter() - t0) / args.clips
        # one host thread per context
        per = args.clips // n
        def worker(slot):
            for _ in range(per):
                bench.clip_start(slot, prompt, bench.N_GREEDY)
                bench.clip_finish(slot)
        ths = [threading.Thread(target=worker, args=(slots[i],)) for i in range(n)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        torch.cuda.synchronize()
        multi = 1e3 * (time.perf_counter() - t0) / (per * n)
        print("in flight %d: one host thread %.2f ms per clip pass; one thread per context %.2f ms per clip pass" % (n, single, multi), flush=True)


if __name__ == "__main__":
    main()
