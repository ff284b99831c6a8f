"""Where do gemmTiled4's FP16 (GELU) outputs differ from gemmTiled8's? (debug aid)"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from whisper_amd import binding

def ptr(t): return t.data_ptr()

M, N, K = [int(x) for x in os.environ.get("SHAPE", "16500x4608x1024").split("x")]
g = torch.Generator(device="cuda").manual_seed(1)
a = torch.randn((M, K), generator=g, device="cuda").half()
w = (0.05 * torch.randn((N, K), generator=g, device="cuda")).half()
bias = torch.randn(N, generator=g, device="cuda")
L = binding.lib()
outs = []
for mask in (binding.TUNE_DEFAULT & ~binding.TUNE_GEMM_4WAVE, binding.TUNE_DEFAULT | binding.TUNE_GEMM_4WAVE, binding.TUNE_DEFAULT | binding.TUNE_GEMM_4WAVE):
    L.wh_debug_set_tuning(mask)
    o = torch.full((M, N), float("nan"), dtype=torch.float16, device="cuda")
    binding.check(L.wh_op_mul_mat_gelu(None, ptr(a), ptr(w), ptr(bias), ptr(o), M, N, K))
    torch.cuda.synchronize()
    outs.append(o.cpu().numpy())
for name, x, y in (("8 vs 4", outs[0], outs[1]), ("4 vs 4", outs[1], outs[2])):
    d = (x.view(np.uint16) != y.view(np.uint16))
    print(name, "differing:", int(d.sum()), "of", d.size, "nan in 4-wave:", int(np.isnan(outs[1].astype(np.float32)).sum()))
    if d.sum():
        r, c = np.nonzero(d)
        print("  rows mod 256 histogram (by 32):", np.bincount((r % 256) // 32, minlength=8))
        print("  cols mod 256 histogram (by 32):", np.bincount((c % 256) // 32, minlength=8))
        print("  tile rows:", np.unique(r // 256)[:20], "tile cols:", np.unique(c // 256)[:20])
        print("  row mod 32 hist:", np.bincount(r % 32, minlength=32))
        print("  col mod 64 hist:", np.bincount(c % 64, minlength=64))
        for i in range(min(6, len(r))):
            print("   ", r[i], c[i], x[r[i], c[i]], y[r[i], c[i]])
