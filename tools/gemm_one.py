"""One tiled-GEMM variant on one shape, for counter passes:  rocprofv3 --pmc ... -- python tools/gemm_one.py 25 42000 4096 1024"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from whisper_amd import binding, ggml_format as gf  # noqa: E402

v, M, N, K = [int(x) for x in sys.argv[1:5]]
m = binding.HipModel.from_ggml(gf.synth_model("test-d128", seed=1))
ctx = binding.HipContext(m, 1)
ms = ctx.probe(1, v, M, N, K, iters=10)
print("variant %d %dx%dx%d: %.0f TF" % (v, M, N, K, 2.0 * M * N * K / (ms * 1e-3) / 1e12))
