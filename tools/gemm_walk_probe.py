"""Tiled-GEMM block walk on the GPU box: rows of tiles vs bands of groupM M tiles (gemm.hip), per shape of the 28-window encoder.
    python tools/gemm_walk_probe.py > gpurun_out/gemm_walk.txt      (variant = tile variant + 100 * groupM, wh_debug_probe kind 1)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from whisper_amd import binding, ggml_format as gf  # noqa: E402


def main():
    m = binding.HipModel.from_ggml(gf.synth_model("test-d128", seed=1))
    ctx = binding.HipContext(m, 1)
    shapes = [(42000, 1024, 1024), (42000, 3072, 1024), (42000, 4096, 1024), (42000, 1024, 4096), (42000, 49152, 1024),
              (10500, 3072, 1024), (10500, 4096, 1024), (10500, 49152, 1024), (48000, 3840, 1280), (48000, 5120, 1280)]
    tiles = {12: "GL 256x256x64", 11: "GL 128x128x32", 13: "GL 256x128x64"}
    for (M, N, K) in shapes:
        for v, name in tiles.items():
            row = []
            for g in (1, 2, 4, 8, 16):
                it = 3 if N > 10000 else 20
                ms = ctx.probe(1, v + 100 * g, M, N, K, iters=it)
                row.append("g%d %.0f TF" % (g, 2.0 * M * N * K / (ms * 1e-3) / 1e12))
            print("GEMM %6d x %5d x %4d  %-16s %s" % (M, N, K, name, " | ".join(row)), flush=True)


if __name__ == "__main__":
    main()
