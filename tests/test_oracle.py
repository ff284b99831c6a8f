"""CPU tests: pin the numpy restatement (oracle/whisper_np.py) against the reference's own outputs.

Two pins: (a) tests/golden/ref_test_d128.npz -- produced by the reference CPU path compiled unmodified (oracle/_ref),
always available; (b) oracle/_ref run live when the .so is present. The reference tree holds no golden vectors of its
own for this path.

Tolerances. A single stage fed with the reference's own input matches to FP32 round-off (~1e-6). End to end, ANY
implementation whose FP32 summation order differs from ggml's accumulates FP16 rounding flips (activations, GELU/exp
table arguments); measured restatement-vs-reference noise on this model: encoder output max 2e-3 / mean 3e-4, logits
max 2e-3 / mean 4e-4 (|logit| <= 2.5). The end-to-end bounds below are 2.5x that floor.
"""
import os
import numpy as np
import pytest

from oracle import whisper_np as wn
from whisper_amd import ggml_format as gf

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

E2E_MAX, E2E_MEAN = 4e-3, 8e-4   # same bounds as tests/test_gpu_model.py


def test_lookup_tables_match_reference(golden):
    """gelu16 / exp16 restate table_gelu_f16 / table_exp_f16 (ggml.c:1375-1385) for every one of the 65536 inputs."""
    bits = np.arange(65536, dtype=np.uint16)
    x = bits.view(np.float16).astype(np.float32)
    finite = np.isfinite(x)
    g = wn.gelu16(x).astype(np.float16).view(np.uint16)
    e = wn.exp16(x).astype(np.float16).view(np.uint16)
    # numpy's tanh/exp vs glibc's: allow at most a handful of 1-ulp differences
    gd = np.abs(g[finite].astype(np.int32) - golden["table_gelu"][finite].astype(np.int32))
    neg = finite & (x <= 0)
    ed = np.abs(e[neg].astype(np.int32) - golden["table_exp"][neg].astype(np.int32))
    assert gd.max() <= 1 and (gd > 0).mean() < 1e-3
    assert ed.max() <= 1 and (ed > 0).mean() < 1e-3


def test_mel_restatement(golden, tiny_model):
    pcm = golden["pcm16"].astype(np.float32) / 32768.0
    mel = wn.log_mel_spectrogram(pcm, tiny_model.filters)
    assert mel.shape == golden["mel"].shape == (80, 1100)
    d = np.abs(mel - golden["mel"])
    # the reference's FP32 recursive FFT carries its own noise on near-silent bins
    assert d.max() < 5e-4 and d.mean() < 5e-6


def test_encoder_restatement(golden, tiny_model):
    n = wn.WhisperNP(tiny_model)
    tr = {}
    out = n.encode(golden["mel"], 0, trace=tr)
    d = np.abs(out - golden["encode_out"])
    assert d.max() < E2E_MAX and d.mean() < E2E_MEAN
    d = np.abs(tr["enc-KQV"] - golden["enc_kqv0"].astype(np.float32))
    assert d.max() < 4e-3 and d.mean() < 1e-4        # first layer only: few flips yet
    for il in (0, 3):
        for nm, mine in (("k", n.kv.cross_k[il]), ("v", n.kv.cross_v[il])):
            d = np.abs(mine - golden["cross_%s%d" % (nm, il)].astype(np.float32))
            assert d.max() < 5e-3 and d.mean() < E2E_MEAN


def test_decoder_restatement(golden, tiny_model):
    """Teacher-forced token steps with the FP16 thread-partitioned P.V emulation at the fixture's n_threads = 1."""
    n = wn.WhisperNP(tiny_model)
    n.encode(golden["mel"], 0)
    pos, n_past = 0, 0
    sp = gf.special_tokens(tiny_model.hparams)
    for i, ln in enumerate(golden["step_lens"]):
        toks = golden["steps"][pos:pos + ln]
        logits, probs = n.decode(toks, n_past, n_threads=1)
        d = np.abs(logits[-1] - golden["logits%d" % i])
        assert d.max() < E2E_MAX and d.mean() < E2E_MEAN, (i, d.max(), d.mean())
        assert abs(float(probs[-1].astype(np.float64).sum()) - 1.0) < 1e-4
        sb = wn.sample_best(probs[-1], sp["beg"], sp["sot"], sp["solm"], sp["not_"])
        st = wn.sample_best(probs[-1], sp["beg"], sp["sot"], sp["solm"], sp["not_"], True, i == 0)
        # token choice can only differ where the reference's own top-2 probabilities are within the noise
        ref_ids = golden["sample%d" % i]
        if sb["id"] != ref_ids[0]:
            assert abs(probs[-1][sb["id"]] - probs[-1][ref_ids[0]]) < 4e-3 * probs[-1][ref_ids[0]]
        if st["id"] != ref_ids[2]:
            assert abs(probs[-1][st["id"]] - probs[-1][ref_ids[2]]) < 4e-3 * probs[-1][ref_ids[2]]
        pos += ln
        n_past += ln


def test_pv_thread_partition_semantics():
    """The FP16 accumulate emulation: 1 thread == sequential, n threads == per-range partials summed in FP32."""
    rng = np.random.default_rng(0)
    P = rng.random((2, 37)).astype(np.float32)
    P /= P.sum(axis=1, keepdims=True)
    V = rng.standard_normal((37, 64)).astype(np.float16).astype(np.float32)
    a1 = wn.WhisperNP.pv_f16_accumulate(P, V, 1)
    a4 = wn.WhisperNP.pv_f16_accumulate(P, V, 4)
    exact = (P.astype(np.float64) @ V.astype(np.float64))
    assert np.abs(a1 - exact).max() < 5e-3 and np.abs(a4 - exact).max() < 5e-3
    # results are FP16-representable sums of <= 4 FP16 partials
    assert np.all(a1 == a1.astype(np.float16).astype(np.float32))
    parts = [wn.WhisperNP.pv_f16_accumulate(P[:, 10 * i:10 * (i + 1)], V[10 * i:10 * (i + 1)], 1) for i in range(4)]
    assert np.array_equal(a4, ((parts[0] + parts[1]) + parts[2]) + parts[3])


def test_live_reference_matches_golden(golden, tiny_model, ref_lib_available, tmp_path):
    """When oracle/_ref is built, re-run the reference and require bit-identical results to the committed fixture."""
    if not ref_lib_available:
        pytest.skip("oracle/_ref/libwhisper_ref.so not built (needs /root/reference)")
    from oracle import ref
    path = str(tmp_path / "m.bin")
    gf.write_model(path, tiny_model)
    w = ref.RefWhisper(path, n_threads=1, log_level=0)
    w.set_mel(golden["mel"])
    w.encode(0)
    k, _ = w.cross_kv(0)
    assert np.array_equal(k.astype(np.float16), golden["cross_k0"])
    ln = int(golden["step_lens"][0])
    logits, _ = w.decode(golden["steps"][:ln], 0)
    assert np.array_equal(logits[-1], golden["logits0"])


def test_streamed_spectrogram_restatement(golden, tiny_model):
    """MelStreamerNP (MelStreamer.cpp:125-245): a one-window clip is the whole-buffer spectrogram (FP32 vs double arithmetic
    only), a later window is normalised by its own maximum, and a request ending where the last one ended re-uses that maximum."""
    pcm = golden["pcm16"].astype(np.float32) / 32768.0
    st = wn.MelStreamerNP(pcm, tiny_model.filters)
    whole = wn.log_mel_spectrogram(pcm, tiny_model.filters)
    assert np.abs(st.make_buffer(0, st.length) - whole).max() < 1e-6
    raw = wn.log_mel_raw(pcm, tiny_model.filters)
    a = st.make_buffer(600, 500)          # ends at 1100 like the request before it: the stored (global) maximum is re-used
    assert np.abs(a - whole[:, 600:]).max() < 1e-6
    b = st.make_buffer(700, 300)          # ends at 1000: a fresh, window-local maximum
    lo = max(np.float32(1e-20), raw[:, 700:1000].max()) - np.float32(8)
    assert np.array_equal(b, ((np.maximum(raw[:, 700:1000], lo) + np.float32(4)) * np.float32(0.25)).astype(np.float32))
    st.n_chunks = 1000
    c = st.make_buffer(900, 200)          # frames 1000.. have no chunk: zero before normalisation
    assert np.all(c[:, 100:] == c[0, 100]) and c.shape == (80, 200)


def _melstreamer_fixture():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden_melstreamer", os.path.join(ROOT, "tests", "golden", "make_golden_melstreamer.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    return mg, dict(np.load(os.path.join(ROOT, "tests", "golden", "ref_melstreamer.npz")))


def test_melstreamer_restatement_pinned_on_the_reference(tiny_model):
    """MelStreamerNP against outputs of the REFERENCE's streaming spectrogram (Whisper/Whisper/MelStreamer.cpp +
    melSpectrogram.cpp compiled unmodified, oracle/Makefile; fixture tests/golden/ref_melstreamer.npz from make_golden_melstreamer.py):
    the request sequence of iContext::runStreamed on a 63917-sample stream (399 chunks + a partial one) with a quiet tail -- a fresh
    maximum, the re-used maximum of a request that ends where the last one ended, window-local maxima, and a request past the
    stream's length (partial chunk's frame computed, frames after it zero before normalisation). Values: the reference's FP32
    split-radix FFT against float64 (max 1.3e-4, mean 3e-7 measured; the whole-buffer path of row a1 shows the same noise);
    the clamp floor (which maximum was used) and the positions of the zero frames must agree exactly."""
    mg, g = _melstreamer_fixture()
    pcm = mg.clip()
    assert len(pcm) == int(g["n_samples"])
    st = wn.MelStreamerNP(pcm, tiny_model.filters)
    assert st.length == 399 and st.n_chunks == 400
    floors = []
    for i, (off, ln) in enumerate(g["requests"]):
        want = g["window%d" % i]
        got = st.make_buffer(int(off), int(ln))
        d = np.abs(got - want)
        assert got.shape == want.shape and d.max() < 4e-4 and d.mean() < 2e-6, (i, d.max(), d.mean())
        if i < 2:       # requests 0 and 1 are clamped at (the stream's maximum - 8): the floor says which maximum was used
            assert abs(float(got.min()) - float(want.min())) < 2e-6
        floors.append(float(want.min()))
    # window0 is also Spectrogram::pcmToMel of the GPU model's runFull (Spectrogram.cpp:64-122; the generator asserts the two equal bit
    # for bit): the whole-buffer restatement of row a1 against the file SURVEY.md 8(a1) cites, next to whisper.cpp's (test_mel_*)
    whole = wn.log_mel_spectrogram(pcm, tiny_model.filters)
    d = np.abs(whole - g["window0"])
    assert whole.shape == (80, 399) and d.max() < 4e-4 and d.mean() < 2e-6
    # request 1 ends where request 0 ended: the stream-wide maximum is re-used although the window itself is quiet;
    # request 2 ends elsewhere: its own maximum (nothing is clamped any more, the floor is the data's)
    assert floors[1] == floors[0] and floors[2] < floors[0] - 0.3
    off, ln = (int(x) for x in g["past_end"])
    got, want = st.make_buffer(off, ln), g["past_end_simple"]
    assert np.abs(got - want).max() < 4e-4
    tail = want[:, st.n_chunks - off:]
    assert tail.size and np.all(tail == tail[0, 0]) and np.array_equal(got[:, st.n_chunks - off:], tail)
    # the reference's two streamers differ only past the length runStreamed clamps its requests to (MelInputTensor.cpp:37-39)
    differs = np.where(np.abs(g["past_end_simple"] - g["past_end_thread"]).max(axis=0) > 0)[0]
    assert list(differs) == [st.length - off]


def test_live_melstreamer_if_present(tiny_model):
    """When oracle/_ref/libmelstreamer_ref.so is built: the reference's streamers (on demand / background thread) reproduce the
    committed fixture bit for bit."""
    from oracle import ref
    if not ref.melstreamer_available():
        pytest.skip("oracle/_ref/libmelstreamer_ref.so not built (needs /root/reference)")
    mg, g = _melstreamer_fixture()
    pcm = mg.clip()
    for threads in (1, 3):
        st = ref.RefMelStreamer(pcm, tiny_model.filters, threads=threads)
        assert st.length == 399
        for i, (off, ln) in enumerate(g["requests"]):
            got = st.make_buffer(int(off), int(ln))
            if i == 0 and threads > 1 and not np.array_equal(got, g["window0"]):
                # the reference's background streamer lost its start-up race (zeros for frames not produced yet, MelStreamer.cpp:436-452;
                # oracle/shim/melstreamer/stdafx.h CreateThread): possible on a machine too busy to run a new thread for 30 ms
                pytest.skip("MelStreamerThread's start-up race was lost on this machine")
            assert np.array_equal(got, g["window%d" % i])
        assert np.array_equal(ref.spectrogram_pcm_to_mel(pcm, tiny_model.filters, threads=threads), g["window0"])
        past = st.make_buffer(*(int(x) for x in g["past_end"]))
        assert np.array_equal(past, g["past_end_simple" if threads == 1 else "past_end_thread"])
        st.close()


def test_large_v3_shape_restatement_against_the_reference(ref_lib_available, tmp_path):
    """BASELINE config 5's shape (128 mel bins, vocabulary 51866). No entry point of the reference accepts it -- whisper_set_mel and the
    spectrogram are tied to WHISPER_N_MEL = 80 (whisper.h:23, whisper.cpp:2289, :2318) -- but its ENCODER AND DECODER take the counts from
    the model file (whisper.cpp:1097-1106, :486): with the context's spectrogram written directly (oracle/ref_harness.cpp ref_set_mel_any)
    the reference's own arithmetic runs a model of that shape. The restatement (the oracle of tests/test_gpu_model.py::test_large_v3_shape)
    is held against it: cross-K/V and logits at the noise level of the 80-mel shapes (max 2e-3 / mean 4e-4 there; same bounds)."""
    if not ref_lib_available:
        pytest.skip("oracle/_ref/libwhisper_ref.so not built (needs /root/reference)")
    from oracle import ref
    model = gf.synth_model("test-d128-v3", seed=77, attn_sharpness=2.0)
    hp = model.hparams
    assert hp.n_mels == 128 and hp.n_vocab == 51866
    sp = gf.special_tokens(hp)
    pcm = (0.1 * np.random.default_rng(8).standard_normal(16000 * 4)).astype(np.float32)
    mel = wn.log_mel_spectrogram(pcm, model.filters)
    assert mel.shape == (128, 400)
    path = str(tmp_path / "v3.bin")
    gf.write_model(path, model)
    w = ref.RefWhisper(path, n_threads=1, log_level=0)
    assert w.n_mels == 128
    w.set_mel_any(mel)
    w.encode(0)
    n = wn.WhisperNP(model)
    n.encode(mel, 0)
    for layer in (0, hp.n_text_layer - 1):
        k, v = w.cross_kv(layer)
        assert np.abs(k - n.kv.cross_k[layer]).max() < E2E_MAX and np.abs(v - n.kv.cross_v[layer]).max() < 2 * E2E_MAX
    toks = [sp["sot"], sp["sot"] + 1, sp["transcribe"]]
    n_past = 0
    for step in range(4):
        logits, _ = w.decode(toks, n_past)
        nl, _ = n.decode(toks, n_past, n_threads=1)
        d = np.abs(logits - nl)
        assert logits.shape[1] == 51866 and d.max() < E2E_MAX and d.mean() < E2E_MEAN, (step, d.max(), d.mean())
        assert int(np.argmax(logits[-1])) == int(np.argmax(nl[-1]))
        n_past += len(toks)
        toks = [int(np.argmax(logits[-1]))]
    w.close()


def test_truth_model_orders_the_references(golden, golden_e2e):
    """The yardstick of SURVEY.md 8(c)(ii), from committed fixtures: exact arithmetic (WhisperTruth) vs the reference at 1 and
    8 threads. The 1-thread reference is 20x further from the truth than the 8-thread one (sequential FP16 accumulation over
    1500 keys, ggml.c:4689-4735) -- which is why parity with "the reference" is stated against the truth."""
    for i in range(len(golden["step_lens"])):
        truth = golden_e2e["truth_logits%d" % i].astype(np.float64)
        d1 = np.abs(golden["logits%d" % i] - truth)
        d8 = np.abs(golden_e2e["ref8_logits%d" % i] - truth)
        assert 1e-2 < d1.max() < 1e-1 and d8.max() < 2.5e-3 and d8.mean() < 4e-4
    assert len(golden_e2e["greedy_ids"]) == 33 and float(golden_e2e["greedy_min_margin"][0]) > 0.02


def test_truth_model_matches_restatement_structure(golden, tiny_model):
    """WhisperTruth is the same graph as the pinned restatement: with the restatement's FP32 P.V choice the two agree to the
    FP16-rounding noise of the restatement (1.3-1.7e-3 max on the logits), on encoder output and first decode step."""
    tr = wn.WhisperTruth(tiny_model)
    eo = tr.encode(golden["mel"].astype(np.float64), 0)
    n = wn.WhisperNP(tiny_model)
    d = np.abs(n.encode(golden["mel"], 0) - eo)
    assert d.max() < 5e-3 and d.mean() < 5e-4
    ln = int(golden["step_lens"][0])
    lt = tr.decode(golden["steps"][:ln], 0)[-1]
    lf = n.decode(golden["steps"][:ln], 0, exact_pv=False)[0][-1]
    d = np.abs(lf - lt)
    assert d.max() < 2.5e-3 and d.mean() < 4e-4


def test_split_cross_attention_is_the_table_softmax():
    """The single-stream cross-attention (decode1.hip: 8 key ranges per head, the maximum exchanged between the two launches) is
    the reference's softmax: the unnormalised exponentials are bit-identical to softmax_table's (a split with LOCAL maxima is
    not: exp16 rounds its argument to FP16, so a different maximum rounds differently), and the output differs from
    softmax_table(s) @ V by FP32 summation order only."""
    rng = np.random.default_rng(17)
    for n_keys in (1500, 1499, 37, 8, 3):
        q = wn.r16(rng.standard_normal(64).astype(np.float32) * 0.5)
        K = wn.r16(rng.standard_normal((n_keys, 64)).astype(np.float32))
        V = wn.r16(rng.standard_normal((n_keys, 64)).astype(np.float32))
        out, e = wn.cross_attention_split(q, K, V)
        s = (K @ q).astype(np.float32)
        m = s.max()
        e_ref = wn.exp16((s - m).astype(np.float32))
        assert np.array_equal(e, e_ref)
        P = wn.softmax_table(s[None, :])[0]
        want = (P.astype(np.float64) @ V.astype(np.float64)).astype(np.float32)
        assert np.abs(out - want).max() < 2e-6 * max(1.0, np.abs(want).max())
        if n_keys >= 37:
            # local maxima + rescale (an online softmax) does NOT reproduce the table's exponentials
            per = ((n_keys + 7) // 8 + 3) & ~3
            differs = 0
            for i in range(8):
                a, b = i * per, min((i + 1) * per, n_keys)
                if b > a and s[a:b].max() < m:
                    local = wn.exp16((s[a:b] - s[a:b].max()).astype(np.float32)) * np.exp(np.float64(s[a:b].max()) - np.float64(m))
                    differs += int((np.abs(local - e_ref[a:b]) > 0).sum())
            assert differs > 0
