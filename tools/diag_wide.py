"""Why do the one-row-tile GELU outputs differ from gemvFused's by bits? FP32 accumulators of both, and both GELU outputs against the table of each."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from whisper_amd import binding
    L = binding.lib()
    g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "ref_test_d128.npz"))
    table = torch.from_numpy(g["table_gelu"].astype(np.int32)).cuda()
    ptr = lambda t: C.c_void_p(t.data_ptr())
    for (M, N, K) in ((70, 4096, 1024), (112, 3072, 1024), (128, 2048, 512)):
        gen = torch.Generator(device="cuda").manual_seed(M + N)
        a = torch.randn((M, K), generator=gen, device="cuda").half()
        w = (0.1 * torch.randn((N, K), generator=gen, device="cuda")).half()
        bias = torch.randn(N, generator=gen, device="cuda")
        f32, gel = {}, {}
        for opt in (0, 2):
            binding.set_option("dec_wide_rows", opt)
            o = torch.zeros((M, N), dtype=torch.float32, device="cuda")
            binding.check(L.wh_op_mul_mat(None, ptr(a), ptr(w), ptr(bias), None, ptr(o), M, N, K))
            h = torch.zeros((M, N), dtype=torch.float16, device="cuda")
            binding.check(L.wh_op_mul_mat_gelu(None, ptr(a), ptr(w), ptr(bias), ptr(h), M, N, K))
            torch.cuda.synchronize()
            f32[opt], gel[opt] = o, h
        binding.set_option("dec_wide_rows", 0)
        d32 = (f32[0] - f32[2]).abs()
        print("%dx%dx%d  FP32 pre-activations: equal %s, max diff %.3e, %d of %d differ" % (M, N, K, bool(torch.equal(f32[0], f32[2])), float(d32.max()), int((d32 > 0).sum()), M * N))
        for opt in (0, 2):
            idx = f32[opt].half().view(torch.int16).to(torch.int32) & 0xFFFF
            want = table[idx.long()].to(torch.int16).view(torch.float16)
            print("   option %d: GELU output vs the table of ITS OWN pre-activation: %d differ; vs the table of the OTHER's: %d" % (
                opt, int((gel[opt] != want).sum()),
                int((gel[opt] != table[(f32[2 - opt].half().view(torch.int16).to(torch.int32) & 0xFFFF).long()].to(torch.int16).view(torch.float16)).sum())))
        dg = gel[0] != gel[2]
        print("   GELU outputs differ at %d places; of those, pre-activations differ at %d" % (int(dg.sum()), int((dg & (d32 > 0)).sum())))
        if int(dg.sum()):
            ii = torch.nonzero(dg)[:5]
            for r, c in ii.tolist():
                print("      [%d,%d] pre %.8f / %.8f  gelu %.6f / %.6f" % (r, c, float(f32[0][r, c]), float(f32[2][r, c]), float(gel[0][r, c]), float(gel[2][r, c])))


if __name__ == "__main__":
    main()
