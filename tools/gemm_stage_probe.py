"""Tiled-GEMM staging depth on the GPU box: 2 LDS stages (wait for everything per K step) vs 3 / 4 stages with counted vmcnt and a raw
barrier, per encoder shape. Every variant is verified against the production kernel inside wh_debug_probe.
    python tools/gemm_stage_probe.py > gpurun_out/gemm_stage.txt"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from whisper_amd import binding, ggml_format as gf  # noqa: E402


def main():
    m = binding.HipModel.from_ggml(gf.synth_model("test-d128", seed=1))
    ctx = binding.HipContext(m, 1)
    shapes = [(42000, 1024, 1024), (42000, 3072, 1024), (42000, 4096, 1024), (42000, 1024, 4096), (48000, 3840, 1280), (48000, 5120, 1280)]
    tiles = {12: "256x256x64 2 stages", 25: "256x256x64 2 stages fragpf", 27: "256x256x32 3 stages fragpf", 20: "256x256x32 3 stages", 26: "128x128x32 fragpf",
             11: "128x128x32 2 stages"}
    for (M, N, K) in shapes:
        row = []
        for v, name in tiles.items():
            try:
                ms = ctx.probe(1, v, M, N, K, iters=20)
                row.append("%s %.0f TF" % (name, 2.0 * M * N * K / (ms * 1e-3) / 1e12))
            except Exception as e:
                row.append("%s FAILED (%s)" % (name, str(e)[-60:]))
        print("GEMM %6d x %5d x %4d  %s" % (M, N, K, " | ".join(row)), flush=True)


if __name__ == "__main__":
    main()
