"""Decode-rows products in isolation: us per launch at M rows for the decoder's shapes, weights rotated through a pool larger than
the 256 MB MALL so that every launch streams them from HBM.   python tools/gemv_time.py [M]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from whisper_amd import binding
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 112
    L = binding.lib()
    p = lambda t: C.c_void_p(t.data_ptr())
    shapes = [(1024, 1024, "out-proj"), (4096, 1024, "mlp-up"), (1024, 4096, "mlp-down"), (51865, 1024, "logits")]
    for (N, K, name) in shapes:
        pool = max(2, min(64, int(400e6 / (N * K * 2))))
        w = (0.05 * torch.randn((pool, N, K), device="cuda")).half()
        a = torch.randn((M, K), device="cuda").half()
        bias = torch.randn(N, device="cuda")
        res = torch.randn((M, N), device="cuda")
        out = torch.zeros((M, N), device="cuda")
        row = []
        outs = {}
        for label, mask in (("dec_lds 0", binding.TUNE_DEFAULT), ("dec_lds 1", binding.TUNE_DEFAULT), ("ks 2", binding.TUNE_DEFAULT)):
            L.wh_debug_set_tuning(mask)
            binding.set_option("dec_lds", 0 if label == "dec_lds 0" else 1)
            binding.set_option("dec_lds_ks", 2 if label == "ks 2" else 1)
            iters = 200 if N < 10000 else 40
            for i in range(8):
                L.wh_op_mul_mat(None, p(a), p(w[i % pool]), p(bias), p(res), p(out), M, N, K)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(iters):
                L.wh_op_mul_mat(None, p(a), p(w[i % pool]), p(bias), p(res), p(out), M, N, K)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / iters
            row.append("%s %.1f us (%.0f GB/s)" % (label, us, N * K * 2 / us / 1e3))
            L.wh_op_mul_mat(None, p(a), p(w[0]), p(bias), p(res), p(out), M, N, K)
            torch.cuda.synchronize()
            outs[label] = out.clone()
        L.wh_debug_set_tuning(binding.TUNE_DEFAULT)
        binding.set_option("dec_lds", binding.get_option_default("dec_lds"))
        binding.set_option("dec_lds_ks", binding.get_option_default("dec_lds_ks"))
        row.append("same bits: %s %s" % (bool(torch.equal(outs["dec_lds 0"], outs["dec_lds 1"])), bool(torch.equal(outs["dec_lds 0"], outs["ks 2"]))))
        print("M=%d %-9s N=%5d K=%4d  %s" % (M, name, N, K, " | ".join(row)), flush=True)


if __name__ == "__main__":
    main()
