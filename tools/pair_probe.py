"""GPU: the encoder product next to the cross-attention, each alone and together (wh_debug_probe kind 4).
Usage: python tools/pair_probe.py [windows=224] [M=28672] [N=4096] [K=1024] [launches=40]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from whisper_amd import binding, ggml_format as gf
a = [int(x) for x in sys.argv[1:]]
wins, M, N, K, it = (a + [224, 28672, 4096, 1024, 40][len(a):])[:5]
m = binding.HipModel.from_ggml(gf.synth_model("test-d128", seed=1))
ctx = binding.HipContext(m, 1)
print("pair (256 CUs, no masks): %.3f ms per product launch equivalent" % ctx.probe(4, wins, M, N, K, it))
