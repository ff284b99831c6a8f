"""A/B of the tiled-GEMM kernels on the encoder's shapes, interleaved rounds in one process (random FP16 operands).
    python tools/gemm8_probe.py            # variant 25 = 16-wave 256x256x64 (round 2), 40 = gemmTiled8
Every variant is checked against the register-staged 128x128x32 kernel inside wh_debug_probe (max |diff| <= 1e-3)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from whisper_amd import binding, ggml_format as gf  # noqa: E402

VARIANTS = [int(v) for v in os.environ.get("PROBE_VARIANTS", "25,40").split(",")]
ROUNDS = int(os.environ.get("PROBE_ROUNDS", "3"))


def main():
    m = binding.HipModel.from_ggml(gf.synth_model("test-d128", seed=1))
    ctx = binding.HipContext(m, 1)
    shapes = [(168000, 1024, 1024), (168000, 3072, 1024), (168000, 4096, 1024), (168000, 1024, 4096), (168000, 49152, 1024),
              (16500, 4096, 1024), (42000, 1024, 1024), (168000, 1280, 1280), (168000, 1024, 256), (16397, 1000, 192)]
    if os.environ.get("PROBE_SHAPES"):
        shapes = [tuple(int(x) for x in s.split("x")) for s in os.environ["PROBE_SHAPES"].split(",")]
    for (M, N, K) in shapes:
        res = {v: [] for v in VARIANTS}
        for _ in range(ROUNDS):
            for v in VARIANTS:
                it = 3 if N > 10000 else 10
                ms = ctx.probe(1, v, M, N, K, iters=it)
                res[v].append(2.0 * M * N * K / (ms * 1e-3) / 1e12)
        print("GEMM %6d x %5d x %4d: " % (M, N, K) + " | ".join("v%d median %.0f max %.0f TF" % (v, sorted(r)[len(r) // 2], max(r)) for v, r in res.items()), flush=True)


if __name__ == "__main__":
    main()
