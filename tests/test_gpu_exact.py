"""WH_FLAG_PARITY_EXACT on the device: the reference CPU path's bits, and the timed kernels measured against them.

north_star asks for logits within 1e-3 of the reference CPU path. The reference is not one function: its decoder sums P.V in FP16 over
per-thread key ranges (Whisper/source/ggml.c:4689-4735), so its own logits move by 0.05 .. 0.5 with the thread count on random-weight models
and by >= 1.5e-3 however peaky the attention is made (tools/ref_band_sweep.py, profiles/r06_reference_band.txt). A tolerance of 1e-3 against
"the reference" is therefore only meaningful against the reference AT A GIVEN THREAD COUNT, computed in its own summation order -- which is what
the exact mode does (whisper_amd/csrc/exact.hip, primitives pinned on the CPU by tests/test_exact_cpu.py):

  1. exact mode == oracle/_ref, BIT FOR BIT: cross-attention caches of every decoder layer, self-attention caches, logits and probabilities, at 1 and
     at 3 threads (d = 128), and at the shape BASELINE measures (ggml-medium: 24 + 24 layers, d = 1024) at 1 and 16 threads.
  2. the timed path against the exact mode ON THE DEVICE at the medium and large-v2 shapes: the same windows, the same tokens; the difference is
     what the MFMA summation order, FP32 LayerNorm sums and the FP32 P.V product (more accurate than the reference's FP16 one) amount to.
"""
import os

import numpy as np
import pytest
import torch

from whisper_amd import binding
from whisper_amd import ggml_format as gf

pytestmark = pytest.mark.gpu


def _prompt(sp):
    return [sp["sot"], sp["sot"] + 1, sp["transcribe"]]


def _bits_equal(a, b):
    return np.array_equal(np.asarray(a, np.float32), np.asarray(b, np.float32))


def test_exact_tables_are_the_reference_tables(ref_lib_available):
    """All 65536 entries of both tables ggml_init builds (ggml.c:1375-1385): built here on the host with the same expressions."""
    if not ref_lib_available:
        pytest.skip("oracle/_ref/libwhisper_ref.so not present")
    from oracle import ref
    g, e = ref.lookup_tables()
    model = gf.synth_model("test-d128", seed=1234, attn_sharpness=2.0)
    m = binding.HipModel.from_ggml(model)
    ctx = binding.HipContext(m, 1)
    assert np.array_equal(ctx.debug_read("exact-gelu-table"), g.view(np.float16).astype(np.float32), equal_nan=True)
    assert np.array_equal(ctx.debug_read("exact-exp-table"), e.view(np.float16).astype(np.float32), equal_nan=True)
    ctx.close()
    m.close()


def exact_against_the_reference(kind, model, tmp_path, n_threads, n_win, windows_checked, n_steps, layers=None):
    from oracle import ref
    import bench
    hp = model.hparams
    sp = gf.special_tokens(hp)
    path = str(tmp_path / (kind + ".bin"))
    gf.write_model(path, model)
    m = binding.HipModel.from_ggml(model)
    ctx = binding.HipContext(m, n_win)
    ctx.set_flags(binding.WH_FLAG_PARITY_EXACT, n_threads)
    pcm = bench.synth_pcm(n_win, seed=100)
    pcm_dev = torch.from_numpy(pcm).cuda()
    mels = torch.stack([ctx.mel_spectrogram(pcm_dev[b]) for b in range(n_win)])
    ctx.encode(mels)
    layers = layers if layers is not None else range(hp.n_text_layer)
    gk = {il: ctx.debug_read("cross-k", il) for il in layers}
    gv = {il: ctx.debug_read("cross-v", il) for il in layers}
    toks = np.array([_prompt(sp)] * n_win, np.int32)
    steps = []
    n_past = 0
    for step in range(n_steps):
        gl, gp = ctx.decode(toks, n_past)
        steps.append((toks.copy(), n_past, gl.copy(), gp.copy()))
        n_past += toks.shape[1]
        toks = np.argmax(gl, axis=1).astype(np.int32).reshape(-1, 1)
    sk = ctx.debug_read("self-k", 0, n_past)
    sv = ctx.debug_read("self-v", hp.n_text_layer - 1, n_past)
    for b in windows_checked:
        w = ref.RefWhisper(path, n_threads=n_threads, log_level=0)
        w.set_mel_any(mels[b].cpu().numpy())
        w.encode(0)
        for il in layers:
            k, v = w.cross_kv(il)
            assert _bits_equal(gk[il][b], k), "%s window %d: cross-K of layer %d differs (max %g)" % (kind, b, il, np.abs(gk[il][b] - k).max())
            assert _bits_equal(gv[il][b], v), "%s window %d: cross-V of layer %d differs (max %g)" % (kind, b, il, np.abs(gv[il][b] - v).max())
        for step, (tk, npast, gl, gp) in enumerate(steps):
            rl, rp = w.decode([int(t) for t in tk[b]], npast)
            d = np.abs(gl[b] - rl[-1])
            print("%s, %d thread(s), window %d step %d: logits max |diff| %.3g, %d of %d differ" % (kind, n_threads, b, step, d.max(), int((d != 0).sum()), d.size))
            assert _bits_equal(gl[b], rl[-1]), "logits differ"
            assert _bits_equal(gp[b], rp[-1]), "probabilities differ"
        k0, _ = w.self_kv(0, n_past)
        _, v1 = w.self_kv(hp.n_text_layer - 1, n_past)
        assert _bits_equal(sk[b], k0) and _bits_equal(sv[b], v1), "self-attention caches differ"
        w.close()
    ctx.close()
    m.close()


@pytest.mark.parametrize("n_threads", [1, 3])
def test_exact_mode_is_the_reference_bit_for_bit(ref_lib_available, tmp_path, n_threads):
    """d = 128, three windows with different audio in one batch, windows 0 and 2 against the live reference: every cross-attention cache, the
    3-token prompt and 4 greedy steps, self-attention caches of the first and last layer. IDENTICAL, at 1 and at 3 threads."""
    if not ref_lib_available:
        pytest.skip("oracle/_ref/libwhisper_ref.so not present")
    model = gf.synth_model("test-d128", seed=1234, attn_sharpness=2.0)
    exact_against_the_reference("test-d128", model, tmp_path, n_threads, 3, (0, 2), 5)


def test_exact_mode_at_the_large_v3_shape(ref_lib_available, tmp_path):
    """128 mel bins (conv1's channels padded 128 -> 128: no padding chain) and the 51866-entry vocabulary."""
    if not ref_lib_available:
        pytest.skip("oracle/_ref/libwhisper_ref.so not present")
    model = gf.synth_model("test-d128-v3", seed=77, attn_sharpness=2.0)
    exact_against_the_reference("test-d128-v3", model, tmp_path, 2, 1, (0,), 3)


@pytest.mark.parametrize("n_threads", [1, 16])
def test_exact_mode_at_the_medium_shape(ref_lib_available, tmp_path, n_threads):
    """The shape BASELINE's metric is quoted on (ggml-medium: d = 1024, 16 heads, 24 + 24 layers, 51865 tokens): cross-attention caches of the first
    and last decoder layer bit-identical to the reference's (24 encoder layers upstream of them), the prompt and 3 steps bit-identical on logits and
    probabilities -- at 1 thread and at the 16 threads the other full-shape tests run the reference with."""
    if not ref_lib_available:
        pytest.skip("oracle/_ref/libwhisper_ref.so not present")
    model = gf.synth_model("medium", seed=1)
    exact_against_the_reference("medium", model, tmp_path, n_threads, 1, (0,), 4, layers=(0, model.hparams.n_text_layer - 1))
