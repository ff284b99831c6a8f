"""GPU: encoder attention launch time per mode (enc_exp 0 = table in LDS, 1 = v_exp_f32) at the bench's chunk: 112 windows x 16 heads x 1500."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from whisper_amd import binding
batch, heads, T = int(sys.argv[1]) if len(sys.argv) > 1 else 112, 16, 1500
Tpad = (T + 255) // 256 * 256
q = (torch.randn(batch * heads, T, 64, device="cuda") * 1.5).half()
k = (torch.randn(batch * heads, T, 64, device="cuda") * 1.5).half()
v = torch.randn(batch * heads, 64 * Tpad, device="cuda").half()
out = torch.empty(batch, T, heads * 64, device="cuda", dtype=torch.float16)
L = binding.lib()
import ctypes as C
p = lambda t: C.c_void_p(t.data_ptr())
flops = 4.0 * batch * heads * T * T * 64
modes = [int(x) for x in os.environ.get('ENC_MODES', '0,1,2,3,0,1,2,3').split(',')]
for mode, abl in [(m_, 0) for m_ in modes]:
    binding.set_option("enc_exp", mode)
    binding.set_option("enc_ablate", abl)
    for _ in range(2):
        binding.check(L.wh_op_flash_attention(None, p(q), p(k), p(v), p(out), batch, heads, T))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 10
    for _ in range(n):
        binding.check(L.wh_op_flash_attention(None, p(q), p(k), p(v), p(out), batch, heads, T))
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print("enc_exp %d ablate %d: %.1f us per launch of %d windows = %.0f TFLOP/s = %.3f of 2.5 PF" % (mode, abl, ms * 1e3, batch, flops / ms * 1e-9, flops / ms * 1e-9 / 2500))
binding.set_option("enc_exp", binding.get_option_default("enc_exp"))
binding.set_option("enc_ablate", 0)
