"""Micro-benchmarks on the GPU box: dispatch floor (empty-kernel chains, eager vs hipGraph) and tiled-GEMM tile shapes.
    python tools/gemm_probe.py > gpurun_out/gemm_probe.txt
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from whisper_amd import binding, ggml_format as gf  # noqa: E402

VARIANTS = {0: "128x128x64 pf2", 1: "128x128x64 pf1", 2: "128x128x32 pf1", 3: "256x128x64 pf1", 4: "256x128x64 pf2",
            5: "256x128x32 pf1", 6: "256x256x64 pf1", 7: "128x256x64 pf1", 8: "256x256x32 pf1",
            10: "GL 128x128x64", 11: "GL 128x128x32", 12: "GL 256x256x64", 13: "GL 256x128x64",
            14: "GL 256x128x32 w128x64", 15: "GL 256x128x64 w128x64", 16: "GL 256x256x64 w128x64", 17: "GL 256x256x32 w128x64",
            18: "GL 128x256x32 w64x128"}
if os.environ.get("PROBE_VARIANTS"):
    VARIANTS = {int(v): VARIANTS[int(v)] for v in os.environ["PROBE_VARIANTS"].split(",")}


def main():
    m = binding.HipModel.from_ggml(gf.synth_model("test-d128", seed=1))
    ctx = binding.HipContext(m, 1)
    for grid in (() if os.environ.get("PROBE_VARIANTS") else (1, 64, 256, 1024)):
        e = ctx.probe(0, grid, iters=2000) * 1e3
        g = ctx.probe(2, grid, iters=2000) * 1e3
        print("empty kernel chain, %4d workgroups: eager %.2f us/kernel, hipGraph %.2f us/kernel" % (grid, e, g), flush=True)
    shapes = [(10500, 1024, 1024), (10500, 3072, 1024), (10500, 4096, 1024), (10500, 1024, 4096), (10500, 49152, 1024), (1500, 1024, 1024),
              (42000, 1024, 1024), (42000, 3072, 1024), (42000, 4096, 1024), (42000, 1024, 4096)]
    for (M, N, K) in shapes:
        row = []
        for v, name in VARIANTS.items():
            it = 5 if N > 10000 else 30
            ms = ctx.probe(1, v, M, N, K, iters=it)
            tf = 2.0 * M * N * K / (ms * 1e-3) / 1e12
            row.append((tf, name, ms))
        best = max(row)
        print("GEMM %6d x %5d x %4d: " % (M, N, K) + " | ".join("%s %.0f TF" % (n, t) for t, n, _ in row) + "  -> best %s" % best[1], flush=True)


if __name__ == "__main__":
    main()
