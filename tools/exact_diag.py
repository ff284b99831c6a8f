"""GPU: where does WH_FLAG_PARITY_EXACT first leave the CPU build of the same primitives (tests/exact_model.py == oracle/_ref bit for bit)?
Stops the exact-order encoder after 0 and 1 layers and compares every buffer."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from whisper_amd import binding, ggml_format as gf
import exact_model as em
import bench

model = gf.synth_model("test-d128", seed=1234, attn_sharpness=2.0)
m = binding.HipModel.from_ggml(model)
ctx = binding.HipContext(m, 1)
ctx.set_flags(binding.WH_FLAG_PARITY_EXACT, 1)
pcm = bench.synth_pcm(1, seed=100)
mel = ctx.mel_spectrogram(torch.from_numpy(pcm).cuda()[0])
x = em.WhisperExact(model)

def cmp(name, got, want):
    got = np.asarray(got, np.float32).reshape(want.shape)
    d = np.abs(got - want)
    bad = int((got != want).sum())
    print("%-8s mismatches %8d of %8d  max |diff| %.3g  (|want| max %.3g)%s" % (name, bad, want.size, d.max(), np.abs(want).max(),
          "" if bad == 0 else "   first at %s" % (np.argwhere(got != want)[0],)), flush=True)

for n in (0, 1):
    binding.set_option("exact_enc_layers", n)
    ctx.encode(mel[None])
    tr = {}
    x.encode(mel.cpu().numpy(), trace=tr, n_layers=n)
    print("== after %d encoder layer(s)" % n)
    if n == 0:
        cmp("conv1", ctx.debug_read("exact:conv1")[0], tr["conv1"])
        cmp("x", ctx.debug_read("exact:x")[0], tr["x"])
    else:
        for name in ("q", "k", "v", "kqv", "h", "x"):
            cmp(name, ctx.debug_read("exact:" + name)[0], tr[name])
        cmp("cur=ln2", ctx.debug_read("exact:cur")[0], tr["ln2"])
binding.set_option("exact_enc_layers", -1)
