"""Cost of a grid-wide barrier inside one persistent kernel on MI355X (wh_debug_probe kind 3), next to the cost of a kernel
boundary (kinds 0 / 2: dependent empty kernels, eager / hipGraph). Usage: python tools/barrier_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from whisper_amd import binding, ggml_format as gf  # noqa: E402


def main():
    m = binding.HipModel.from_ggml(gf.synth_model("test-d128"))
    ctx = binding.HipContext(m, 1)
    for grid in (256,):
        print("empty kernel chain, grid %d: eager %.2f us, graph %.2f us" % (grid, 1e3 * ctx.probe(0, grid, iters=2000), 1e3 * ctx.probe(2, grid, iters=2000)))
    for grid in (64, 128, 256):
        for mode in (0, 1):
            n = 2000
            ms = ctx.probe(3, grid, M=mode, iters=n)        # returns ms per barrier (elapsed / iters)
            print("grid barrier: %3d workgroups, mode %d (%s): %.2f us per barrier" % (grid, mode, "flat" if mode == 0 else "per-XCD", 1e3 * ms))


if __name__ == "__main__":
    main()
