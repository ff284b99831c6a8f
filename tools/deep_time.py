"""The deep decode products (FP32 + bias + residual, the MLP down-projection) at 33 .. 128 rows: us per launch for dec_split 0 (gemvFused, eight waves over K) and 1
(gemmDecTile<SPLIT = 8> + decSplitCombine).  python tools/deep_time.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from whisper_amd import binding
    L = binding.lib()
    p = lambda t: C.c_void_p(t.data_ptr())
    for M in (40, 64, 70, 96, 128):
        for (N, K) in ((1024, 4096), (1280, 5120)):
            pool = max(2, min(64, int(400e6 / (N * K * 2))))
            w = (0.05 * torch.randn((pool, N, K), device="cuda")).half()
            a = torch.randn((M, K), device="cuda").half()
            bias = torch.randn(N, device="cuda")
            res = torch.randn((M, N), device="cuda")
            out = torch.zeros((M, N), device="cuda")
            row, outs = [], {}
            for split in (0, 1):
                binding.set_option("dec_split", split)
                for i in range(8):
                    L.wh_op_mul_mat(None, p(a), p(w[i % pool]), p(bias), p(res), p(out), M, N, K)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(200):
                    L.wh_op_mul_mat(None, p(a), p(w[i % pool]), p(bias), p(res), p(out), M, N, K)
                e1.record()
                torch.cuda.synchronize()
                row.append("dec_split %d %.1f us" % (split, e0.elapsed_time(e1) * 1e3 / 200))
                L.wh_op_mul_mat(None, p(a), p(w[0]), p(bias), p(res), p(out), M, N, K)
                torch.cuda.synchronize()
                outs[split] = out.clone()
            binding.set_option("dec_split", binding.get_option_default("dec_split"))
            print("M=%3d N=%4d K=%4d  %s | same bits %s" % (M, N, K, " | ".join(row), bool(torch.equal(outs[0], outs[1]))), flush=True)


if __name__ == "__main__":
    main()
