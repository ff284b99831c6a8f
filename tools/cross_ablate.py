"""attentionDecM<5> at 8 windows x 20 heads with parts removed (option cross_ablate; results wrong, times only). The ablation instances left the source after session r6v: check out commit 1576939 to run this.  python tools/cross_ablate.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from whisper_amd import binding
    L = binding.lib()
    p = lambda t: C.c_void_p(t.data_ptr())
    blocks, heads, group = 8, 20, 5
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
    call = lambda i: L.wh_op_decoder_cross_attention(None, p(x), p(lw), p(lb), p(wq), p(bq), scale, p(K[i % pool]), p(V[i % pool]), p(out), seqs, heads, keys, keys, group)
    names = {0: "the kernel", 1: "no exponentials", 2: "no V transposes", 4: "no V loads", 6: "no V loads, no transposes", 7: "no V loads, transposes, exponentials", 16: "no K loads",
             23: "no K / V loads, transposes, exponentials (LayerNorm + query projection + MFMAs + reductions)"}
    for abl in (0, 1, 2, 4, 6, 7, 16, 23, 0):
        binding.set_option("cross_ablate", abl)
        for i in range(5):
            binding.check(call(i))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(100):
            call(i)
        e1.record()
        torch.cuda.synchronize()
        print("cross_ablate %2d  %5.1f us  %s" % (abl, e0.elapsed_time(e1) * 1e3 / 100, names[abl]), flush=True)
    binding.set_option("cross_ablate", 0)


if __name__ == "__main__":
    main()
