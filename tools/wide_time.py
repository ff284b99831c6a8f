"""The wide decode products (FP16 GELU epilogue) at 33 .. 128 rows: us per launch for dec_lds 0 (gemmDecRows, one row tile), 1 (gemmDecTile) and gemmDecTile with one K tile per ring
slot (dec_lds_ks 1).  python tools/wide_time.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from whisper_amd import binding
    L = binding.lib()
    p = lambda t: C.c_void_p(t.data_ptr())
    for M in (40, 64, 70, 96, 112, 128):
        for (N, K) in ((4096, 1024), (3072, 1024), (5120, 1280), (3840, 1280)):
            pool = max(2, min(64, int(400e6 / (N * K * 2))))
            w = (0.05 * torch.randn((pool, N, K), device="cuda")).half()
            a = torch.randn((M, K), device="cuda").half()
            bias = torch.randn(N, device="cuda")
            out = torch.zeros((M, N), device="cuda", dtype=torch.float16)
            row, outs = [], {}
            for lds in (0, 1, 2):
                binding.set_option("dec_lds", min(lds, 1))
                binding.set_option("dec_lds_ks", 1 if lds == 2 else binding.get_option_default("dec_lds_ks"))
                for i in range(8):
                    L.wh_op_mul_mat_gelu(None, p(a), p(w[i % pool]), p(bias), p(out), M, N, K)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(200):
                    L.wh_op_mul_mat_gelu(None, p(a), p(w[i % pool]), p(bias), p(out), M, N, K)
                e1.record()
                torch.cuda.synchronize()
                row.append("%s %.1f us" % (("dec_lds 0", "dec_lds 1", "dec_lds 1 / dec_lds_ks 1")[lds], e0.elapsed_time(e1) * 1e3 / 200))
                L.wh_op_mul_mat_gelu(None, p(a), p(w[0]), p(bias), p(out), M, N, K)
                torch.cuda.synchronize()
                outs[lds] = out.clone()
            binding.set_option("dec_lds", binding.get_option_default("dec_lds"))
            binding.set_option("dec_lds_ks", binding.get_option_default("dec_lds_ks"))
            print("M=%3d N=%4d K=%4d  %s | same bits %s" % (M, N, K, " | ".join(row), bool(torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]))), flush=True)


if __name__ == "__main__":
    main()
