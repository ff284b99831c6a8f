"""The cross-attention of a beam step (hypothesis groups sharing a window's keys, fused LayerNorm + query projection): us per launch for cross_mfma 0 (attentionDecG<NQ, true>)
and 1 (attentionDecM), rotating over a pool of K/V buffers larger than the Infinity Cache.  python tools/cross_time.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from whisper_amd import binding
    L = binding.lib()
    p = lambda t: C.c_void_p(t.data_ptr())
    for (blocks, heads, group) in ((8, 20, 5), (8, 16, 5), (16, 20, 5), (8, 20, 2), (8, 20, 8)):
        d, seqs, keys = heads * 64, blocks * group, 1500
        pool = 10
        K = (0.8 * torch.randn((pool, blocks, heads, keys, 64), device="cuda")).half()
        V = torch.randn((pool, blocks, heads, keys, 64), device="cuda").half()
        x = torch.randn((seqs, d), device="cuda") * 2 + 0.3
        lw, lb = 1 + 0.1 * torch.randn(d, device="cuda"), 0.1 * torch.randn(d, device="cuda")
        wq = (torch.randn((d, d), device="cuda") / d ** 0.5).half()
        bq = 0.1 * torch.randn(d, device="cuda")
        out = torch.zeros((seqs, d), device="cuda", dtype=torch.float16)
        scale = C.c_float(64.0 ** -0.25)
        row, outs = [], {}
        call = lambda i: L.wh_op_decoder_cross_attention(None, p(x), p(lw), p(lb), p(wq), p(bq), scale, p(K[i % pool]), p(V[i % pool]), p(out), seqs, heads, keys, keys, group)
        for mode in (0, 1):
            binding.set_option("cross_mfma", mode)
            for i in range(5):
                binding.check(call(i))
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(100):
                call(i)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / 100
            row.append("cross_mfma %d %.1f us (%.2f TB/s)" % (mode, us, 2.0 * blocks * heads * keys * 64 * 2 / us / 1e6))
            call(0)
            torch.cuda.synchronize()
            outs[mode] = out.float().clone()
        binding.set_option("cross_mfma", binding.get_option_default("cross_mfma"))
        print("windows=%2d heads=%2d group=%d  %s | max |diff| %.2e" % (blocks, heads, group, " | ".join(row), float((outs[0] - outs[1]).abs().max())), flush=True)


if __name__ == "__main__":
    main()
