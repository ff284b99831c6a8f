"""The arithmetic of WH_FLAG_PARITY_EXACT held against the reference's own code on the CPU, bit for bit.

whisper_amd/csrc/exact_ops.h restates the summation ORDER of the reference CPU path (ggml_vec_dot_f16's 32 chains and reduction tree,
ggml_vec_mad_f16's FP16 accumulation per thread range, the double LayerNorm sums with the contraction gcc applies, the table softmax). The
HIP kernels of the exact mode (whisper_amd/csrc/exact.hip) are built from these primitives; here the same header is compiled for the host
(tests/exact_cpu/ops.cpp), driven through the reference's graph (tests/exact_model.py) and compared with oracle/_ref (Whisper/source/ggml.c +
whisper.cpp compiled unmodified): cross-attention caches, logits and probabilities must be IDENTICAL -- not close -- at 1 and at 3 threads
(the thread count changes the reference's decoder P.V sums, ggml.c:4689-4735). The GPU counterpart is tests/test_gpu_exact.py."""
import os

import numpy as np
import pytest

from whisper_amd import ggml_format as gf
import exact_model as em


@pytest.mark.parametrize("n_threads", [1, 3])
def test_exact_order_primitives_reproduce_the_reference_bit_for_bit(ref_lib_available, tmp_path, n_threads):
    if not ref_lib_available:
        pytest.skip("oracle/_ref/libwhisper_ref.so not present")
    if not os.path.exists(em.CLANG):
        pytest.skip("no host compiler with _Float16")
    from oracle import ref
    import bench
    model = gf.synth_model("test-d128", seed=1234, attn_sharpness=2.0)
    sp = gf.special_tokens(model.hparams)
    path = str(tmp_path / "m.bin")
    gf.write_model(path, model)
    pcm = bench.synth_pcm(1, seed=100)[0]
    w = ref.RefWhisper(path, n_threads=n_threads, log_level=0)
    mel = w.pcm_to_mel(pcm)
    w.set_mel(mel)
    w.encode(0)
    x = em.WhisperExact(model)
    x.encode(mel)
    for il in range(model.hparams.n_text_layer):
        k, v = w.cross_kv(il)
        assert np.array_equal(k, x.cross_k[il].astype(np.float32)), "cross-K of layer %d differs" % il
        assert np.array_equal(v, x.cross_v[il].astype(np.float32)), "cross-V of layer %d differs" % il
    toks = [sp["sot"], sp["sot"] + 1, sp["transcribe"]]
    n_past = 0
    for step in range(4):
        rl, rp = w.decode(toks, n_past)
        xl, xp = x.decode(toks, n_past, n_threads=n_threads)
        assert np.array_equal(rl, xl), "logits differ at step %d: max %g" % (step, np.abs(rl - xl).max())
        assert np.array_equal(rp, xp), "probabilities differ at step %d" % step
        n_past += len(toks)
        toks = [int(np.argmax(rl[-1]))]
    k, v = w.self_kv(0, n_past)
    assert np.array_equal(k, x.self_k[0][:n_past].astype(np.float32)) and np.array_equal(v, x.self_v[0][:n_past].astype(np.float32))
    w.close()


def test_dot_order_matters():
    """The order is not a formality: the same dot product summed left to right in FP32 differs from the reference's 32-chain order in a
    large share of random rows -- which is why 'same rounding points' (oracle/whisper_np.py) is only close and this mode is identical."""
    if not os.path.exists(em.CLANG):
        pytest.skip("no host compiler with _Float16")
    rng = np.random.default_rng(3)
    w = (0.05 * rng.standard_normal((64, 1024))).astype(np.float16)
    x = rng.standard_normal((8, 1024)).astype(np.float32)
    m = type("M", (), {"hparams": None, "tensors": None})
    xm = em.WhisperExact.__new__(em.WhisperExact)
    got = em.WhisperExact.mul_mat(xm, w, x)
    x16 = x.astype(np.float16).astype(np.float32)
    seq = np.zeros((8, 64), np.float32)
    for k in range(1024):
        seq = (seq + (x16[:, k:k + 1] * w[:, k].astype(np.float32)[None, :]).astype(np.float32)).astype(np.float32)
    exact = x16.astype(np.float64) @ w.astype(np.float64).T
    assert np.abs(got - exact).max() < 1e-4 and np.abs(seq - exact).max() < 1e-4
    assert (got != seq).mean() > 0.2
