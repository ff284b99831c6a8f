"""Batch-shape sweep on the GPU box: clips per lock-step batch x batches in flight (x the split-stream option), bench.py's own
measure_batched on the medium shape.  python tools/shape_sweep.py "4x3,8x2,8x3,12x2,16x1,16x2" [--split]
Prints ms per clip pass (7 windows) and audio-s/s per configuration."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    import torch
    import torch.distributed as dist
    from whisper_amd import binding, ggml_format as gf
    cfgs = [tuple(int(x) for x in c.split("x")) for c in (sys.argv[1] if len(sys.argv) > 1 else "4x3,8x2,8x3,16x1,16x2").split(",")]
    kind = os.environ.get("SWEEP_MODEL", "medium")
    hp = gf.hparams_for(kind)
    sp = gf.special_tokens(hp)
    prompt = [sp["sot"], sp["sot"] + 1, sp["transcribe"]]
    hm = binding.HipModel.from_ggml(gf.synth_model(kind, seed=1))
    masks = [("default", binding.TUNE_DEFAULT)]
    if "--split" in sys.argv:
        masks.append(("split-streams", binding.TUNE_DEFAULT | binding.TUNE_SPLIT_STREAMS))
    for name, mask in masks:
        binding.lib().wh_debug_set_tuning(mask)
        for C, I in cfgs:
            steps = C * I * 2
            t0 = time.time()
            m = bench.measure_batched(hm, hp, prompt, steps, 1, 7, C, I, 0, 1, dist, want_kernels=False)
            ms = 1e3 * m["elapsed"] / steps
            print("%-14s clips/batch %2d (%3d windows)  in flight %d : %7.2f ms per clip pass  %8.1f audio-s/s   (%.1f s incl. setup)"
                  % (name, C, 7 * C, I, ms, bench.CLIP_SECONDS / (ms * 1e-3), time.time() - t0), flush=True)
            for s in m["slots"]:
                s[0].close()
            del m
            torch.cuda.empty_cache()
    binding.lib().wh_debug_set_tuning(binding.TUNE_DEFAULT)


if __name__ == "__main__":
    main()
