"""The vocabulary product of a decode step at 33 .. 128 rows (FP32 out, no bias): us per launch for vocab_lds 0 (gemmAllRows) and 1 (gemmDecTile, 64 x 64 tiles).  python tools/vocab_time.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from whisper_amd import binding
    L = binding.lib()
    p = lambda t: C.c_void_p(t.data_ptr())
    for (N, K) in ((51865, 1280), (51865, 1024)):
        w = (0.05 * torch.randn((4, N, K), device="cuda")).half()
        for M in (40, 64, 70, 100, 128):
            a = torch.randn((M, K), device="cuda").half()
            out = torch.zeros((M, N), device="cuda")
            row, outs = [], {}
            for mode in (0, 1):
                binding.set_option("vocab_lds", mode)
                for i in range(4):
                    binding.check(L.wh_op_mul_mat(None, p(a), p(w[i % 4]), None, None, p(out), M, N, K))
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(40):
                    L.wh_op_mul_mat(None, p(a), p(w[i % 4]), None, None, p(out), M, N, K)
                e1.record()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) * 1e3 / 40
                row.append("vocab_lds %d %.1f us (%.2f TB/s)" % (mode, us, 2.0 * N * K / us / 1e6))
                L.wh_op_mul_mat(None, p(a), p(w[0]), None, None, p(out), M, N, K)
                torch.cuda.synchronize()
                outs[mode] = out.clone()
            binding.set_option("vocab_lds", binding.get_option_default("vocab_lds"))
            print("M=%3d N=%5d K=%4d  %s | same bits %s" % (M, N, K, " | ".join(row), bool(torch.equal(outs[0], outs[1]))), flush=True)


if __name__ == "__main__":
    main()
