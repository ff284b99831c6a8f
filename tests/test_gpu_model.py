"""GPU parity tests, graph level: wh_mel_spectrogram / wh_encode / wh_decode / wh_sample_best through the C ABI against
(a) the committed outputs of the reference's CPU path (tests/golden, produced by oracle/_ref), (b) the numpy
restatement on the same inputs, (c) oracle/_ref run live when it travelled with the snapshot, and -- at full model
sizes, where the oracle would take minutes -- size-independent properties (batch invariance, probability mass,
KV-cache consistency between a prompt step and token-by-token steps).

End-to-end tolerance (see tests/test_oracle.py): two faithful implementations of the reference's FP16-table numerics
that differ only in FP32 summation order sit ~2e-3 (max) / 4e-4 (mean) apart on the logits of this model; the bounds
here are 2.5x that measured floor. north_star asks for 1e-3 on real weights; stage-level tests (test_gpu_ops.py) hold
each kernel to FP32 round-off.
"""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")

from oracle import whisper_np as wn  # noqa: E402
from whisper_amd import binding, ggml_format as gf  # noqa: E402

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# End-to-end bounds of the TIMED path against the reference at 8 threads / its numpy restatement on the d = 128 test model. north_star's 1e-3 is asserted where it can
# hold: bit-exactly for the exact mode (tests/test_gpu_exact.py: 0 against the live reference at any thread count) and in the MEAN here (8e-4 < 1e-3). In the max norm no
# implementation that does not sum in ggml's own order can hold it -- re-ordering only the last reduction of the reference's dot products moves the logits by 1.7e-3 at this
# shape and 4.8e-3 at the medium shape (test_timed_path_against_the_exact_mode) --, and the reference differs from ITSELF by 0.2 between 1 and 8 threads. 4e-3 = the sum of the
# two distances that meet in this comparison: timed path to the correctly rounded result (1.7e-3) + reference at 8 threads to the same (2.0e-3), profiles/r06_evidence/split_d128.txt.
E2E_MAX, E2E_MEAN = 4e-3, 8e-4


def report(name, got, want):
    d = np.abs(np.asarray(got, np.float64) - np.asarray(want, np.float64))
    print("%-34s max|want|=%9.4f maxdiff=%.3e meandiff=%.3e" % (name, np.abs(want).max(), d.max(), d.mean()))
    return d


@pytest.fixture(scope="module")
def hip_tiny(tiny_model):
    m = binding.HipModel.from_ggml(tiny_model)
    yield m
    m.close()


@pytest.fixture(scope="module")
def np_tiny(tiny_model, golden):
    n = wn.WhisperNP(tiny_model)
    n.encode(golden["mel"], 0)
    return n


def test_device_is_mi355x():
    info = binding.device_info(0)
    print(info)
    assert binding.device_count() >= 1 and "gfx950" in info["name"]


def test_mel_spectrogram(hip_tiny, golden, tiny_model):
    ctx = binding.HipContext(hip_tiny, 1)
    pcm = golden["pcm16"].astype(np.float32) / 32768.0
    mel = ctx.mel_spectrogram(torch.from_numpy(pcm).cuda()).cpu().numpy()
    assert mel.shape == (80, 1100)
    d = report("mel vs reference", mel, golden["mel"])
    assert d.max() < 5e-4 and d.mean() < 5e-6           # the reference's own FP32-FFT noise on near-silent bins
    d = report("mel vs float64 restatement", mel, wn.log_mel_spectrogram(pcm, tiny_model.filters))
    assert d.max() < 2e-5
    # ragged / edge lengths: shorter than one frame, not a multiple of the hop, zeros
    for n in (0, 159, 160, 401, 16000 + 77):
        x = np.zeros(max(n, 1), np.float32)[:n] if n < 400 else pcm[:n]
        if n == 0:
            continue
        got = ctx.mel_spectrogram(torch.from_numpy(np.ascontiguousarray(x)).cuda()).cpu().numpy()
        assert got.shape == (80, n // 160)
        if n >= 160:
            want = wn.log_mel_spectrogram(x, tiny_model.filters)
            assert np.abs(got - want).max() < 2e-5, n
    ctx.close()


def test_mel_matrix_core_kernel_against_the_valu_kernel(hip_tiny, golden, tiny_model):
    """TUNE_MEL_MFMA: DFT + filterbank as FP64 MFMA tiles vs the FP64 FMA kernel -- same arithmetic, different summation order
    (1e-15 relative before the log) -- on the sample clip, ragged lengths and the streamed-window entry point; both against
    the float64 restatement."""
    ctx = binding.HipContext(hip_tiny, 1)
    pcm = golden["pcm16"].astype(np.float32) / 32768.0
    L = binding.lib()
    for n in (len(pcm), 16000 + 77, 401, 160 * 37):
        x = np.ascontiguousarray(pcm[:n])
        want = wn.log_mel_spectrogram(x, tiny_model.filters)
        outs = {}
        for name, mask in (("mfma", binding.TUNE_DEFAULT | binding.TUNE_MEL_MFMA), ("valu", binding.TUNE_DEFAULT & ~binding.TUNE_MEL_MFMA)):
            L.wh_debug_set_tuning(mask)
            try:
                outs[name] = ctx.mel_spectrogram(torch.from_numpy(x).cuda()).cpu().numpy()
                win = ctx.mel_spectrogram_window(torch.from_numpy(x).cuda(), 0, n // 160).cpu().numpy() if n >= 1600 else None
            finally:
                L.wh_debug_set_tuning(binding.TUNE_DEFAULT)
            d = report("mel %s kernel, %d samples, vs float64 restatement" % (name, n), outs[name], want)
            assert outs[name].shape == want.shape and d.max() < 2e-5
            if win is not None:
                assert np.abs(win - outs[name]).max() < 1e-6
        assert np.abs(outs["mfma"] - outs["valu"]).max() < 2e-6
    ctx.close()


def test_encoder(hip_tiny, golden, np_tiny):
    ctx = binding.HipContext(hip_tiny, 2)
    mel = torch.from_numpy(golden["mel"]).cuda()
    ctx.encode(mel)
    out = ctx.debug_read("encode-out")[0]
    d = report("encode-out vs reference", out, golden["encode_out"])
    assert d.max() < E2E_MAX + 2e-3 and d.mean() < E2E_MEAN      # + FP16 rounding of the read-back
    for il in (0, 3):
        for nm in ("k", "v"):
            got = ctx.debug_read("cross-" + nm, il)[0]
            want = golden["cross_%s%d" % (nm, il)].astype(np.float32)
            d = report("cross-%s[%d] vs reference" % (nm, il), got, want)
            assert d.max() < 5e-3 and d.mean() < E2E_MEAN
            mine = np_tiny.kv.cross_k[il] if nm == "k" else np_tiny.kv.cross_v[il]
            d = report("cross-%s[%d] vs restatement" % (nm, il), got, mine)
            assert d.max() < 5e-3 and d.mean() < E2E_MEAN
    ctx.close()


@pytest.mark.gpu
def test_batched_spectrogram_equals_one_call_per_buffer(hip_tiny):
    """wh_mel_spectrogram_batch (three launches for `batch` independent buffers, a maximum each) against one wh_mel_spectrogram call per buffer: the same kernels
    with a buffer index in the grid, so the same bits -- including a silent buffer (its own maximum, not its neighbour's) and a row stride beyond the samples."""
    rng = np.random.default_rng(11)
    n = 176000
    pcm = (0.1 * rng.standard_normal((5, n + 320))).astype(np.float32)
    pcm[2] *= 1e-4          # a quiet buffer: normalised by ITS maximum
    pcm[3] = 0.0            # silence
    dev_pcm = torch.from_numpy(pcm).cuda()
    ctx = binding.HipContext(hip_tiny, 1)
    view = dev_pcm[:, :n]   # rows n + 320 apart
    got = ctx.mel_spectrogram_batch(view).cpu().numpy()
    for b in range(5):
        one = ctx.mel_spectrogram(dev_pcm[b, :n].contiguous()).cpu().numpy()
        assert np.array_equal(one, got[b]), b
    assert not np.array_equal(got[0], got[2])
    ctx.close()



def test_encoder_batch_and_offsets(hip_tiny, golden):
    """Windows in one batch are independent: same window at different batch slots / offsets gives identical caches."""
    ctx = binding.HipContext(hip_tiny, 3)
    mel = golden["mel"]
    shifted = np.zeros((80, 1100 + 40), np.float32)
    shifted[:, 40:] = mel
    pad = np.zeros((80, 1140), np.float32)
    pad[:, :1100] = mel
    batch = torch.from_numpy(np.stack([pad, shifted, pad])).cuda()
    ctx.encode(batch, offsets=[0, 40, 0])
    k = ctx.debug_read("cross-k", 3)
    assert np.array_equal(k[0], k[2])
    assert np.array_equal(k[0], k[1])          # offset slicing reproduces the same window
    ctx1 = binding.HipContext(hip_tiny, 1)
    ctx1.encode(torch.from_numpy(mel).cuda())
    assert np.array_equal(ctx1.debug_read("cross-k", 3)[0], k[0])      # batch size does not change results
    ctx.close()
    ctx1.close()


def run_steps(ctx, golden, batch=1):
    outs = []
    pos, n_past = 0, 0
    for i, ln in enumerate(golden["step_lens"]):
        toks = golden["steps"][pos:pos + ln]
        logits, probs = ctx.decode(np.tile(toks, (batch, 1)), n_past)
        outs.append((logits, probs))
        pos += ln
        n_past += ln
    return outs


def test_decoder_parity_mode(hip_tiny, golden, np_tiny, tiny_model):
    """Forced-token steps with the reference's FP16 thread-partitioned P.V emulated (n_threads = 1, as the fixture)."""
    ctx = binding.HipContext(hip_tiny, 1)
    ctx.encode(torch.from_numpy(golden["mel"]).cuda())
    ctx.set_parity(1)
    sp = gf.special_tokens(tiny_model.hparams)
    pos, n_past = 0, 0
    for i, ln in enumerate(golden["step_lens"]):
        ln = int(ln)
        logits, probs = ctx.decode(golden["steps"][pos:pos + ln][None, :], n_past)
        d = report("logits step %d vs reference" % i, logits[0], golden["logits%d" % i])
        assert d.max() < E2E_MAX and d.mean() < E2E_MEAN
        nl, npr = np_tiny.decode(golden["steps"][pos:pos + ln], n_past, n_threads=1)
        d = report("logits step %d vs restatement" % i, logits[0], nl[-1])
        assert d.max() < E2E_MAX and d.mean() < E2E_MEAN
        assert abs(float(probs[0].astype(np.float64).sum()) - 1.0) < 1e-4
        # token ids: identical to the reference unless its own top-2 are within the implementation noise
        ref_ids = golden["sample%d" % i]
        sb = ctx.sample_best(1)[0]
        st = ctx.sample_best(1, True, i == 0)[0]
        print("step", i, "sample", sb, "timestamp", st, "reference", ref_ids, golden["samplep%d" % i])
        for mine, ref_id in ((sb, ref_ids[0]), (st, ref_ids[2])):
            if mine["id"] != ref_id:
                # near-uniform random-weight distribution: a swap is only legitimate inside the implementation noise
                assert abs(probs[0][mine["id"]] - probs[0][ref_id]) < 4e-3 * probs[0][ref_id]
        # the device sampler against the host restatement on the SAME probabilities: exact
        hb = wn.sample_best(probs[0], sp["beg"], sp["sot"], sp["solm"], sp["not_"])
        ht = wn.sample_best(probs[0], sp["beg"], sp["sot"], sp["solm"], sp["not_"], True, i == 0)
        assert (sb["id"], sb["tid"]) == (hb["id"], hb["tid"]) and (st["id"], st["tid"]) == (ht["id"], ht["tid"])
        assert abs(sb["p"] - hb["p"]) < 1e-9 and abs(sb["ptsum"] - hb["ptsum"]) < 1e-7 and abs(st["pt"] - ht["pt"]) < 1e-6
        pos += ln
        n_past += ln
    rows = int(golden["step_lens"].sum())
    for nm in ("k", "v"):
        got = ctx.debug_read("self-" + nm, 0, rows)[0]
        d = report("self-%s[0] vs reference" % nm, got, golden["self_%s0" % nm].astype(np.float32))
        assert d.max() < 5e-3 and d.mean() < E2E_MEAN
    ctx.close()


def test_decoder_fast_path(hip_tiny, golden, golden_e2e, np_tiny):
    """The MEASURED path (FP32 P.V, what bench.py times and what the reference's own GPU shaders do, mulMatByRowTiled.hlsl).
    The reference's CPU decoder accumulates P.V in FP16, key by key, per thread (ggml.c:4689-4735): its logits move by 3-5e-2
    between 1 and 8 threads on this model, so "the reference" is a band, not a point. The yardstick is the exact-arithmetic
    result (oracle/whisper_np.py WhisperTruth, float64, no intermediate rounding; fixtures from tests/golden/make_golden_e2e.py):
        |HIP - truth| must not exceed |reference(1 thread) - truth|, must stay within 1.25x of |reference(8 threads) - truth|
        (measured: 1.3-1.7e-3 vs 1.7-2.0e-3 max, 2.4e-4 vs 3.3e-4 mean), and the distance to the 8-thread reference itself is
        bounded by the sum of the two (asserted at 4e-3 max / 6e-4 mean; north_star's 1e-3 holds in the mean)."""
    ctx = binding.HipContext(hip_tiny, 1)
    ctx.encode(torch.from_numpy(golden["mel"]).cuda())
    ctx.set_parity(0)
    outs = run_steps(ctx, golden)
    pos, n_past = 0, 0
    for i, (logits, _) in enumerate(outs):
        ln = int(golden["step_lens"][i])
        nl, _ = np_tiny.decode(golden["steps"][pos:pos + ln], n_past, exact_pv=False)
        d = report("fast logits step %d vs restatement(fp32 PV)" % i, logits[0], nl[-1])
        assert d.max() < E2E_MAX and d.mean() < E2E_MEAN
        truth, ref1, ref8 = golden_e2e["truth_logits%d" % i], golden["logits%d" % i], golden_e2e["ref8_logits%d" % i]
        dt = report("fast logits step %d vs truth (float64)" % i, logits[0], truth)
        d1 = report("   reference, 1 thread, vs truth", ref1, truth)
        d8 = report("   reference, 8 threads, vs truth", ref8, truth)
        report("   reference, 1 vs 8 threads", ref1, ref8)
        assert dt.max() < 2.5e-3 and dt.mean() < 4e-4
        assert dt.max() <= d1.max() and dt.mean() <= d1.mean()
        assert dt.max() <= 1.25 * d8.max() and dt.mean() <= 1.25 * d8.mean()
        dr = report("fast logits step %d vs reference (8 threads)" % i, logits[0], ref8)
        assert dr.max() < 4e-3 and dr.mean() < 6e-4
        assert int(np.argmax(logits[0])) == int(np.argmax(ref8)) or abs(np.sort(ref8)[-1] - np.sort(ref8)[-2]) < 4e-3
        pos += ln
        n_past += ln
    ctx.close()


def test_greedy_token_ids_on_jfk_wav(golden, golden_e2e):
    """north_star's end-to-end criterion on the measured path: SampleClips/jfk.wav -> GPU mel -> encoder -> prompt -> 32 greedy
    steps (device-side sampler, captured hipGraph) must give the token ids the reference gives when IT chooses its own tokens
    (whisper_decode + whisper_sample_timestamp / whisper_sample_best, fixture greedy_ids; identical at 1 and 8 reference
    threads). Model: test-d128 with the tied token embedding scaled by 4 (peaked distributions; plain random weights make
    every sample a timestamp by the sum rule)."""
    model = gf.synth_model("test-d128", seed=1234, attn_sharpness=2.0)
    te = model.tensors["decoder.token_embedding.weight"].astype(np.float32) * float(golden_e2e["greedy_gain"][0])
    model.tensors["decoder.token_embedding.weight"] = te.astype(np.float16)
    sp = gf.special_tokens(model.hparams)
    want = [int(x) for x in golden_e2e["greedy_ids"]]
    m = binding.HipModel.from_ggml(model)
    pcm = torch.from_numpy(golden["pcm16"].astype(np.float32) / 32768.0).cuda()
    for batch in (1, 3):
        ctx = binding.HipContext(m, batch)
        mel = ctx.mel_spectrogram(pcm)
        ctx.encode(torch.stack([mel] * batch))
        ctx.decode_window_start(np.array([[sp["sot"]]] * batch, np.int32), len(want) - 1, force_first_timestamp=True, first_is_initial=True)
        ids, ps = ctx.decode_window_finish()
        print("greedy ids (batch %d):" % batch, [int(x) for x in ids[:, 0]][:12], "reference:", want[:12], "min margin", float(golden_e2e["greedy_min_margin"][0]))
        for b in range(batch):
            assert [int(x) for x in ids[:, b]] == want
        assert np.abs(ps[:, 0] - golden_e2e["greedy_p"]).max() < 5e-3
        ctx.close()
    m.close()


def test_stage_level_probe_points(hip_tiny, golden, np_tiny, tiny_model):
    """The intermediates at the reference's Tracing probe points (WhisperContext.cpp:142-638 / whisper.cpp:1121-1869),
    read back under WH_FLAG_DEBUG_CAPTURE: localises a parity failure to conv front-end / first attention / decoder attention."""
    ctx = binding.HipContext(hip_tiny, 1)
    ctx.set_flags(binding.WH_FLAG_DEBUG_CAPTURE | binding.WH_FLAG_PARITY_PV, 1)
    ctx.encode(torch.from_numpy(golden["mel"]).cuda())
    n = wn.WhisperNP(tiny_model)
    tr = {}
    n.encode(golden["mel"], 0, trace=tr)
    got = ctx.debug_read("enc.temp1")[0]                     # [2*n_ctx][d]
    d = report("enc.temp1 (conv1 + GELU) vs restatement", got, wn.r16(tr["enc.temp1"].T))
    assert d.max() < 2e-3 and (d > 0).mean() < 0.02          # FP16 flips of the GELU argument only
    got = ctx.debug_read("enc.layer0.in")[0]
    d = report("enc.layer[0].in (conv2 + GELU + pos) vs restatement", got, tr["enc.layer[ 0 ].in"])
    assert d.max() < 4e-3 and d.mean() < 1e-4
    got = ctx.debug_read("enc-KQV")[0]                       # [n_ctx][H*64]
    want = golden["enc_kqv0"].astype(np.float32).transpose(1, 0, 2).reshape(got.shape)      # reference: [H][n_ctx][64]
    d = report("enc-KQV (layer 0 attention) vs reference", got, wn.r16(want))
    assert d.max() < 4e-3 and d.mean() < 1e-4
    ln = int(golden["step_lens"][0])
    ctx.decode(golden["steps"][:ln][None, :], 0)
    for nm, key in (("dec-KQV", "dec_kqv_self0"), ("dec-KQV#2", "dec_kqv_cross0")):
        got = ctx.debug_read(nm, rows=ln)                    # [rows][H*64]
        want = golden[key].astype(np.float32).transpose(1, 0, 2).reshape(got.shape)
        d = report("%s (decoder layer 0) vs reference" % nm, got, wn.r16(want))
        assert d.max() < 4e-3 and d.mean() < 3e-4
    ctx.close()


def test_hypotheses_share_cross_attention(hip_tiny, golden):
    """5 decoder sequences per window on ONE pass over the window's cross-attention K/V (wh_context_create_hyp): every
    hypothesis computes exactly what a lone sequence fed the same tokens computes -- the greedy-equivalence at b = 1 that
    SURVEY.md 8(d) config 3 asks for, since the reference declares beam search without implementing it (sFullParams.h:12-13)."""
    pad = np.zeros((80, 3000), np.float32)
    pad[:, :1100] = golden["mel"]
    mel = torch.from_numpy(pad).cuda()
    rng = np.random.default_rng(3)
    mel2 = torch.from_numpy(rng.uniform(-1, 1, (80, 3000)).astype(np.float32)).cuda()
    toks = [[50257, 1000 + 7 * j, 2000 + j] for j in range(5)]
    lone = {}
    for wi, mm in enumerate((mel, mel2)):
        c1 = binding.HipContext(hip_tiny, 1)
        c1.encode(mm)
        for j in range(5):
            c1.encode(mm)
            la, _ = c1.decode(np.array([toks[j]], np.int32), 0)
            lb, _ = c1.decode(np.array([[300 + j]], np.int32), 3)
            lone[(wi, j)] = (la[0], lb[0])
        c1.close()
    ch = binding.HipContext(hip_tiny, 2, hypotheses=5)
    ch.encode(torch.stack([mel, mel2]))
    la, _ = ch.decode(np.array(toks + toks, np.int32), 0)                       # rows window-major: w0h0..w0h4, w1h0..w1h4
    lb, _ = ch.decode(np.array([[300 + j] for j in range(5)] * 2, np.int32), 3)
    for wi in range(2):
        for j in range(5):
            d1 = report("hyp prompt step w%d h%d vs lone" % (wi, j), la[wi * 5 + j], lone[(wi, j)][0])
            d2 = report("hyp token step  w%d h%d vs lone" % (wi, j), lb[wi * 5 + j], lone[(wi, j)][1])
            # the 15-row prompt step of the lone run and the 30-row one here take different kernels (FP32 summation order)
            assert d1.max() < E2E_MAX and d2.max() < E2E_MAX and d1.mean() < E2E_MEAN
    # device-side greedy loop with hypotheses: identical first tokens -> identical streams within a window
    ch.encode(torch.stack([mel, mel2]))
    ch.decode_window_start(np.array([[50257, 50362, 50363]] * 10, np.int32), 6)
    ids, _ = ch.decode_window_finish()
    for wi in range(2):
        assert all(np.array_equal(ids[:, wi * 5], ids[:, wi * 5 + j]) for j in range(5))
    ch.close()


@pytest.mark.parametrize("bit,batch", [("TUNE_FUSE_CROSS_Q", 2), ("TUNE_FUSE_SELF_BLOCK", 9), ("TUNE_FUSE_SELF_BLOCK", 25), ("TUNE_GEMV_LN_BLOCK", 20),
                                       ("TUNE_GEMV_K8", 3),
                                       # 2 heads x 170 / 390 sequences: 2 / 4 sequences per workgroup of the fused self-attention block
                                       ("TUNE_FUSE_SELF_BLOCK", 170), ("TUNE_FUSE_SELF_BLOCK", 390),
                                       # its projection on the matrix cores vs on the VALU (1 and 4 sequences per workgroup)
                                       ("TUNE_SELF_MFMA", 10), ("TUNE_SELF_MFMA", 390),
                                       # 40 and 100 rows: 32 instead of 64 rows per workgroup in the decode products
                                       ("TUNE_GEMV_ROWGROUPS", 40), ("TUNE_GEMV_ROWGROUPS", 100),
                                       # one stream (1 .. 4 sequences): the chip-wide launches of decode1.hip against the batch kernels;
                                       # their prefetch workgroups on / off must not change a bit
                                       ("TUNE_DECODE_SMALL", 1), ("TUNE_DECODE_SMALL", 2), ("TUNE_DECODE_SMALL", 3), ("TUNE_DECODE_SMALL", 4),
                                       ("TUNE_DECODE_PREFETCH", 1), ("TUNE_DECODE_PREFETCH", 3),
                                       # 65 .. 128 rows, N / 16 >= 256 (the MLP up-projection at d = 128: N = 512 -- too narrow; the test model
                                       # never takes the variant, the medium-shape model of test_gemv_all_rows_variant does)
                                       ("TUNE_GEMV_MT8", 100)])
def test_fused_launches_match_separate_launches(hip_tiny, golden, bit, batch):
    """Decode steps with a fusion switched on against the same steps with the separate launches (tuning bit off):
    LayerNorm + cross-attention query inside the attention kernel; LayerNorm + per-head QKV + cache append + self-attention
    as one kernel (1, 2 or 4 sequences per workgroup), its Q/K/V projection as MFMA tiles; the workgroup-wide LayerNorm
    prologue of the 17..32-row gemv; 8 waves splitting K in the MLP down-projection; the row grouping of the 33..128-row
    products; the single-stream path (gemvSmall + cross-attention over 8 key ranges, decode1.hip). Only FP32 summation
    order may differ."""
    pad = np.zeros((80, 3000), np.float32)
    pad[:, :1100] = golden["mel"]
    mel = torch.from_numpy(pad).cuda()
    L = binding.lib()
    res = {}
    # the fused self-attention block serves 9 .. 32 sequences by default (round 5: beyond that the self-attention is its own launch, selfAttnDecWave);
    # its 2- and 4-sequence workgroups only exist at larger batches, so the bound is lifted for these cases
    lift = bit in ("TUNE_FUSE_SELF_BLOCK", "TUNE_SELF_MFMA")
    for name, mask in (("fused", binding.TUNE_DEFAULT | getattr(binding, bit)), ("separate", binding.TUNE_DEFAULT & ~getattr(binding, bit))):
        L.wh_debug_set_tuning(mask)
        if lift:
            binding.set_option("self_fuse_max_rows", 512)
        try:
            ctx = binding.HipContext(hip_tiny, batch)
            ctx.encode(torch.stack([mel] * batch))
            ctx.decode(np.array([[50257, 50362, 50363]] * batch, np.int32), 0)
            a = ctx.decode(np.array([[1234]] * batch, np.int32), 3)[0]
            b = ctx.decode(np.array([[777]] * batch, np.int32), 4)[0]
            res[name] = (a, b, ctx.debug_read("self-k", 3, 5), ctx.debug_read("self-v", 0, 5))
            ctx.close()
        finally:
            L.wh_debug_set_tuning(binding.TUNE_DEFAULT)
            binding.set_option("self_fuse_max_rows", 32)
    for i in range(2):
        d = report("%s on vs off, step %d" % (bit, i), res["fused"][i][0], res["separate"][i][0])
        assert d.max() < 3e-3 and d.mean() < 4e-4
        if bit == "TUNE_DECODE_PREFETCH":
            assert d.max() == 0.0
        assert all(np.array_equal(res["fused"][i][0], res["fused"][i][k]) for k in range(batch))
    for i in (2, 3):
        d = report("%s on vs off, self cache" % bit, res["fused"][i][0], res["separate"][i][0])
        assert d.max() < 4e-3 and d.mean() < 2e-4


@pytest.mark.parametrize("d,heads,batch", [(1280, 20, 1), (1280, 20, 3), (768, 12, 2), (384, 6, 4)])
def test_single_stream_path_at_other_widths(d, heads, batch):
    """decode1.hip at the widths of the other model sizes (large: K = 1280 is 2.5 KiB pieces per weight row, 20 heads; small
    768 / 12; tiny 384 / 6 = less than one piece per row), two layers: decode steps with the chip-wide launches against the
    batch kernels, step by step. Only FP32 summation order may differ; the sequences of a batch stay bit-identical."""
    hp = gf.HParams(n_vocab=51865, n_audio_ctx=1500, n_audio_state=d, n_audio_head=heads, n_audio_layer=1,
                    n_text_ctx=448, n_text_state=d, n_text_head=heads, n_text_layer=2, n_mels=80, f16=1)
    model = gf.synth_model(hp=hp, seed=21, attn_sharpness=2.0)
    hm = binding.HipModel.from_ggml(model)
    g = torch.Generator(device="cuda").manual_seed(9)
    mel = torch.rand((80, 3000), generator=g, device="cuda") * 2.0 - 1.0
    L = binding.lib()
    res = {}
    for name, mask in (("small", binding.TUNE_DEFAULT | binding.TUNE_DECODE_SMALL), ("batch", binding.TUNE_DEFAULT & ~binding.TUNE_DECODE_SMALL)):
        L.wh_debug_set_tuning(mask)
        try:
            ctx = binding.HipContext(hm, batch)
            ctx.encode(torch.stack([mel] * batch))
            ctx.decode(np.array([[50258, 50259, 50359]] * batch, np.int32), 0)
            a = ctx.decode(np.array([[1234]] * batch, np.int32), 3)[0]
            b = ctx.decode(np.array([[777]] * batch, np.int32), 4)[0]
            res[name] = (a, b)
            ctx.close()
        finally:
            L.wh_debug_set_tuning(binding.TUNE_DEFAULT)
    for i in range(2):
        d_ = report("d=%d, %d sequence(s), step %d" % (d, batch, i), res["small"][i][0], res["batch"][i][0])
        span = float(res["batch"][i][0].max() - res["batch"][i][0].min())
        assert np.isfinite(res["small"][i]).all() and d_.max() < 1e-3 * max(span, 1.0) and d_.mean() < 1e-4 * max(span, 1.0)
        assert all(np.array_equal(res["small"][i][0], res["small"][i][k]) for k in range(batch))
    hm.close()


def test_large_v3_shape(tmp_path):
    """128 mel bins, vocabulary 51866 (BASELINE config 5). The reference's GPU model cannot load this shape (N_MEL is a constexpr 80,
    audioConstants.h:13; special ids keyed on 51865) and no entry point of its CPU model accepts a 128-bin spectrogram -- but its CPU
    encoder / decoder take the counts from the model file: tests/test_oracle.py::test_large_v3_shape_restatement_against_the_reference
    runs them on this very model (spectrogram written into the context directly) and pins the numpy restatement on them AT THIS SHAPE.
    Here: conv1 with K = 384, the special ids the sampler uses, and parity with that restatement."""
    model = gf.synth_model("test-d128-v3", seed=77, attn_sharpness=2.0)
    hp = model.hparams
    assert hp.n_mels == 128 and hp.n_vocab == 51866
    sp = gf.special_tokens(hp)
    assert (sp["eot"], sp["sot"], sp["translate"], sp["transcribe"], sp["prev"], sp["solm"], sp["not_"], sp["beg"]) == \
        (50257, 50258, 50359, 50360, 50362, 50363, 50364, 50365)
    m = binding.HipModel.from_ggml(model)
    ctx = binding.HipContext(m, 1)
    rng = np.random.default_rng(8)
    pcm = (0.1 * rng.standard_normal(16000 * 4)).astype(np.float32)
    mel = ctx.mel_spectrogram(torch.from_numpy(pcm).cuda()).cpu().numpy()
    assert mel.shape == (128, 400)
    want = wn.log_mel_spectrogram(pcm, model.filters)
    assert np.abs(mel - want).max() < 2e-5
    n = wn.WhisperNP(model)
    n.encode(want, 0)
    ctx.encode(torch.from_numpy(np.ascontiguousarray(want)).cuda())
    d = report("v3-shape encode-out vs restatement", ctx.debug_read("encode-out")[0], n.encode(want, 0))
    assert d.max() < E2E_MAX + 2e-3 and d.mean() < E2E_MEAN
    ctx.set_parity(1)
    toks = [sp["sot"], sp["sot"] + 1, sp["transcribe"]]
    logits, probs = ctx.decode(np.array([toks], np.int32), 0)
    nl, npr = n.decode(toks, 0, n_threads=1)
    d = report("v3-shape logits vs restatement", logits[0], nl[-1])
    assert logits.shape[1] == 51866 and d.max() < E2E_MAX and d.mean() < E2E_MEAN
    sb = ctx.sample_best(1, True, True)[0]
    hb = wn.sample_best(probs[0], sp["beg"], sp["sot"], sp["solm"], sp["not_"], True, True)
    assert (sb["id"], sb["tid"]) == (hb["id"], hb["tid"]) and sp["beg"] <= sb["id"] <= sp["beg"] + 100
    # ... and against the reference's own encoder / decoder on this model, live (oracle/_ref travels to the GPU box)
    from oracle import ref
    if ref.available():
        path = str(tmp_path / "v3.bin")
        gf.write_model(path, model)
        w = ref.RefWhisper(path, n_threads=1, log_level=0)
        w.set_mel_any(want)
        w.encode(0)
        rl, _ = w.decode(toks, 0)
        d = report("v3-shape logits vs the reference's decoder (1 thread)", logits[0], rl[-1])
        assert d.max() < E2E_MAX and d.mean() < E2E_MEAN and int(np.argmax(logits[0])) == int(np.argmax(rl[-1]))
        w.close()
    ctx.close()
    m.close()


# Bounds of the full-shape comparisons of the TIMED path with the live reference at 16 threads (absolute, on logits of magnitude ~7, span ~12; cross-K |k| <= 2.2,
# cross-V |v| <= 5.2 where one FP16 ulp is 3.9e-3). Since round 6 they have a derivation instead of "2 x measured": both sides are measured on the device against the exact
# mode's correctly rounded P.V variant E0 (tests/test_gpu_exact.py, profiles/r06_evidence/split_medium.txt / split_large_v2.txt): reference(16 threads) - E0 = 5.9e-3 / 9.0e-4
# (medium), 6.4e-3 / 1.1e-3 (large-v2); timed - E0 = 5.1e-3 / 7.8e-4, 5.6e-3 / 9.9e-4 -- the floor ANY re-ordered summation shows (4.8e-3 / 5.8e-3). The logit bounds below
# are the sums of the two (1.2e-2, 1.3e-2): a triangle inequality, not a tolerance chosen to pass. The stated 1e-3 is asserted bit-exactly where it is meaningful (the exact
# mode against the reference at the same thread count: 0) and in the mean for the timed path against E0 (test_timed_path_against_the_exact_mode).
SHAPE_BOUNDS = {
    # kind: (K max, K mean, V max, V mean, logits max, logits mean, steps, min top-1 agreement)
    "medium": (4e-3, 6e-4, 1e-2, 1.5e-3, 1.2e-2, 2e-3, 7, 6),
    # measured in round 4 (gpurun r4A, profiles/r04_evidence/parity_r4A.txt): K 2.9e-3 / 3.3e-4 (|k| <= 2.3), V 7.8e-3 / 9.5e-4 (|v| <= 7.2: two FP16 ulps
    # of 3.9e-3), logits 6.6e-3 / 1.14e-3 on magnitudes up to 9, top-1 4 / 4
    "large-v2": (6e-3, 7e-4, 1.6e-2, 1.9e-3, 1.3e-2, 2.3e-3, 4, 3),
}


def full_shape_against_the_reference(kind, tmp_path, n_threads=16):
    from oracle import ref
    import bench
    k_max, k_mean, v_max, v_mean, l_max, l_mean, n_steps, min_agree = SHAPE_BOUNDS[kind]
    model = gf.synth_model(kind, seed=1)
    hp = model.hparams
    sp = gf.special_tokens(hp)
    path = str(tmp_path / (kind + ".bin"))
    gf.write_model(path, model)
    m = binding.HipModel.from_ggml(model)
    del model
    n_win = 11
    ctx = binding.HipContext(m, n_win)
    pcm = bench.synth_pcm(n_win, seed=100)
    pcm_dev = torch.from_numpy(pcm).cuda()
    mels = torch.stack([ctx.mel_spectrogram(pcm_dev[b]) for b in range(n_win)])
    ctx.encode(mels)
    w = ref.RefWhisper(path, n_threads=n_threads, log_level=0)
    w.set_mel(mels[0].cpu().numpy())
    w.encode(0)
    for il in (0, hp.n_text_layer - 1):
        k, v = w.cross_kv(il)
        dk = report("%s cross-k[%d] vs reference" % (kind, il), ctx.debug_read("cross-k", il)[0], k)
        dv = report("%s cross-v[%d] vs reference" % (kind, il), ctx.debug_read("cross-v", il)[0], v)
        scale_k, scale_v = float(np.abs(k).max()), float(np.abs(v).max())
        assert scale_k < 4.0 and scale_v < 10.0
        assert dk.max() < k_max and dk.mean() < k_mean
        assert dv.max() < v_max and dv.mean() < v_mean
    prompt = [sp["sot"], sp["sot"] + 1, sp["transcribe"]]
    toks = np.array([prompt] * n_win, np.int32)
    n_past = 0
    agree = 0
    worst = 0.0
    for step in range(n_steps):
        gl, _ = ctx.decode(toks, n_past)
        rl, _ = w.decode([int(t) for t in toks[0]], n_past)
        rl = rl[-1]
        d = report("%s logits step %d vs reference (%d threads)" % (kind, step, n_threads), gl[0], rl)
        span = float(rl.max() - rl.min())
        print("    logit span %.3f, top-1 gpu %d ref %d" % (span, int(np.argmax(gl[0])), int(np.argmax(rl))))
        worst = max(worst, d.max() / max(span, 1e-6))
        assert np.isfinite(gl).all()
        assert d.max() < l_max and d.mean() < l_mean
        top2 = np.sort(rl)[-2:]
        same = int(np.argmax(gl[0]) == np.argmax(rl))
        # a top-1 disagreement is admissible only where the reference's own top-2 margin is inside the error band
        assert same or float(top2[1] - top2[0]) < 1e-2, ("top-1 differs with a reference margin of", float(top2[1] - top2[0]))
        agree += same
        n_past += toks.shape[1]
        nxt = ctx.sample_best(n_win, step == 0, step == 0)
        toks = np.array([[t["id"]] for t in nxt], np.int32)
    print("%s shape: top-1 agreement %d / %d steps, worst max-diff / span %.2e" % (kind, agree, n_steps, worst))
    assert agree >= min_agree
    w.close()
    ctx.close()
    m.close()


def test_medium_shape_against_the_reference(ref_lib_available, tmp_path):
    """Parity at the shape and through the kernel instances BASELINE measures: ggml-medium shape (the bench's model), an 11-window
    lock-step batch -- M = 16500 rows, so the encoder GEMMs take the persistent 256x256x64 tiles with the banded block
    walk (EPI_QKV_ENC, EPI_F32, EPI_F16_GELU, EPI_CROSS_KV) -- and the measured FP32-P.V decoder, against the reference's own
    CPU path (oracle/_ref, 16 threads) on window 0: cross-attention caches of the first and last decoder layer, then the
    logits of the 3-token prompt and of 6 teacher-forced greedy steps (the GPU's own ids). ~25 s of host CPU."""
    if not ref_lib_available:
        pytest.skip("oracle/_ref/libwhisper_ref.so not present")
    full_shape_against_the_reference("medium", tmp_path)


def test_large_v2_shape_against_the_reference(ref_lib_available, tmp_path):
    """The same at the other model BASELINE names (configs[2], [3]): ggml-large-v2 shape, d = 1280 / 20 heads / 32 + 32 layers, through the
    same kernel instances at their large-shape parameters (head count not a power of two, K = 1280 / 5120 products, N = 2*32*1280 cross-K/V
    product), window 0 of an 11-window batch against the live reference: cross-K/V of the first and last decoder layer, the 3-token
    prompt and 3 teacher-forced steps. The reference analogue is GpuEncTest / GpuDecTest on whatever model is loaded
    (Whisper/whisperCom.cpp:929-1088). ~60-90 s of host CPU (model build + a 3 GB file + the reference's encoder on 16 threads)."""
    if not ref_lib_available:
        pytest.skip("oracle/_ref/libwhisper_ref.so not present")
    full_shape_against_the_reference("large-v2", tmp_path)


def test_encoder_is_bit_identical_under_the_gemm_variants():
    """The encoder products' epilogues that only the model reaches at their full size -- head-split Q / K, fragment-major V, the cross-K/V caches of
    all decoder layers, conv1's segmented FP16 GELU output -- through gemmTiled8 with the general block epilogues, gemmTiled8 with the lean
    interior-tile epilogue (TUNE_GEMM_FAST_EPI, the default) and gemmTiled4 (TUNE_GEMM_4WAVE): the same MFMAs in the same order and the same
    arithmetic per element, so the cross-attention caches of an 11-window batch (M = 16500 rows, ragged last tile row; every encoder layer is
    upstream of them) must agree bit for bit."""
    import bench
    model = gf.synth_model("medium", seed=3)
    hp = model.hparams
    m = binding.HipModel.from_ggml(model)
    del model
    n_win = 11
    ctx = binding.HipContext(m, n_win)
    pcm_dev = torch.from_numpy(bench.synth_pcm(n_win, seed=7)).cuda()
    mels = torch.stack([ctx.mel_spectrogram(pcm_dev[b]) for b in range(n_win)])
    L = binding.lib()
    base = binding.TUNE_DEFAULT & ~(binding.TUNE_GEMM_4WAVE | binding.TUNE_GEMM_FAST_EPI)
    got = []
    try:
        # (tuning mask, gemm_mf16): the last two are gemmTiled8 with v_mfma_f32_16x16x32_f16 in its K loop (round 6) -- general and lean epilogues.
        # The matrix cores add a k-block of 8 at a time in both instruction shapes, so even the other MFMA gives the same bits.
        for mask, mf16 in ((base, 0), (base | binding.TUNE_GEMM_FAST_EPI, 0), (base | binding.TUNE_GEMM_4WAVE, 0), (base, 1), (base | binding.TUNE_GEMM_FAST_EPI, 1)):
            L.wh_debug_set_tuning(mask)
            binding.set_option("gemm_mf16", mf16)
            ctx.encode(mels)
            got.append([(ctx.debug_read("cross-k", il).copy(), ctx.debug_read("cross-v", il).copy()) for il in (0, hp.n_text_layer // 2, hp.n_text_layer - 1)])
    finally:
        L.wh_debug_set_tuning(binding.TUNE_DEFAULT)
        binding.set_option("gemm_mf16", binding.get_option_default("gemm_mf16"))
    assert np.isfinite(got[0][0][0]).all() and float(np.abs(got[0][0][1]).max()) > 0.1
    for other in got[1:]:
        for (k0, v0), (k1, v1) in zip(got[0], other):
            assert np.array_equal(k0, k1) and np.array_equal(v0, v1)
    ctx.close()
    m.close()


def test_decoder_batch_invariance_and_kv_consistency(hip_tiny, golden):
    """(1) every sequence of a lock-step batch gets the result of a batch of one; (2) feeding the prompt token by token
    through the KV cache gives the same last-row logits as one multi-token step (same kernels, same order)."""
    mel = torch.from_numpy(golden["mel"]).cuda()
    ctx1 = binding.HipContext(hip_tiny, 1)
    ctx1.encode(mel)
    single = run_steps(ctx1, golden)
    ctx3 = binding.HipContext(hip_tiny, 3)
    ctx3.encode(torch.stack([mel, mel, mel]))
    triple = run_steps(ctx3, golden, batch=3)
    for (l1, p1), (l3, p3) in zip(single, triple):
        for b in range(3):
            assert np.array_equal(l1[0], l3[b]) and np.array_equal(p1[0], p3[b])
    # 20 sequences: single-token steps use both MFMA column tiles of the gemv (rows 0-15 and 16-19), the prompt step the tiled GEMM
    ctx20 = binding.HipContext(hip_tiny, 20)
    ctx20.encode(torch.stack([mel] * 20))
    many = run_steps(ctx20, golden, batch=20)
    for (l1, p1), (l20, p20) in zip(single, many):
        # the 60-row prompt step runs on the tiled GEMM: its FP32 summation order differs from the gemv's of the 3-row one, which
        # flips FP16 roundings of activations and cached K/V (the implementation noise floor of the module docstring), so
        # against the batch of one this is a noise bound; within the batch it is exact
        d = report("batch of 20 vs batch of 1", l20[0], l1[0])
        assert d.max() < E2E_MAX and d.mean() < E2E_MEAN
        assert all(np.array_equal(l20[0], l20[b]) and np.array_equal(p20[0], p20[b]) for b in range(20))
    ctx20.close()
    # prompt in one step vs token by token
    prompt = golden["steps"][:int(golden["step_lens"][0])]
    ctx1.encode(mel)
    la, _ = ctx1.decode(prompt[None, :], 0)
    ctx1.encode(mel)
    for j, t in enumerate(prompt):
        lb, _ = ctx1.decode(np.array([[t]], np.int32), j)
    d = report("prompt step vs token-by-token", la[0], lb[0])
    # the 3-token step runs LayerNorm / QKV / attention as separate launches, the single-token steps the fused kernels: same
    # arithmetic, different FP32 summation grouping, so an FP16 rounding of a cached K/V row or an activation can flip and
    # move every logit a little (measured 6e-4 max / 1.2e-4 mean; the module's implementation noise floor bounds it)
    assert d.max() < 1.5e-3 and d.mean() < 3e-4
    ctx1.close()
    ctx3.close()


def test_live_reference_if_present(hip_tiny, tiny_model, golden, ref_lib_available, tmp_path):
    """When oracle/_ref travelled with the snapshot, run the reference on the GPU box's CPU and compare a fresh input."""
    if not ref_lib_available:
        pytest.skip("oracle/_ref/libwhisper_ref.so not present")
    from oracle import ref
    path = str(tmp_path / "m.bin")
    gf.write_model(path, tiny_model)
    w = ref.RefWhisper(path, n_threads=1, log_level=0)
    rng = np.random.default_rng(99)
    mel = rng.uniform(-1, 1, (80, 3000)).astype(np.float32)          # config-3 style synthetic mel
    w.set_mel(mel)
    w.encode(0)
    sp = gf.special_tokens(tiny_model.hparams)
    rl, _ = w.decode([sp["sot"], sp["not_"]], 0)
    ctx = binding.HipContext(hip_tiny, 1)
    ctx.encode(torch.from_numpy(mel).cuda())
    ctx.set_parity(1)
    gl, _ = ctx.decode(np.array([[sp["sot"], sp["not_"]]], np.int32), 0)
    d = report("live reference logits (uniform mel)", gl[0], rl[-1])
    assert d.max() < E2E_MAX and d.mean() < E2E_MEAN
    k, v = w.cross_kv(2)
    d = report("live reference cross-k[2]", ctx.debug_read("cross-k", 2)[0], k)
    assert d.max() < 5e-3 and d.mean() < E2E_MEAN
    ctx.close()


def test_full_size_properties():
    """ggml-medium shape (random weights; the oracle needs ~10 s per window on the host): properties that do not need it.
    batch invariance across 2 windows, probabilities sum to one, no NaN, decode determinism."""
    model = gf.synth_model("medium", seed=1)
    m = binding.HipModel.from_ggml(model)
    ctx = binding.HipContext(m, 2)
    rng = np.random.default_rng(2)
    mel = rng.uniform(-1, 1, (2, 80, 3000)).astype(np.float32)
    mel[1] = mel[0]
    ctx.encode(torch.from_numpy(mel).cuda())
    k = ctx.debug_read("cross-k", 23)
    assert np.isfinite(k).all() and np.array_equal(k[0], k[1])
    sp = gf.special_tokens(model.hparams)
    toks = np.array([[sp["sot"], sp["sot"] + 1, sp["transcribe"]]] * 2, np.int32)
    logits, probs = ctx.decode(toks, 0)
    assert np.isfinite(logits).all() and np.array_equal(logits[0], logits[1])
    assert abs(float(probs[0].astype(np.float64).sum()) - 1.0) < 1e-4
    l2, _ = ctx.decode(np.array([[123], [123]], np.int32), 3)
    ctx.encode(torch.from_numpy(mel).cuda())
    ctx.decode(toks, 0)
    l3, _ = ctx.decode(np.array([[123], [123]], np.int32), 3)
    assert np.array_equal(l2, l3)
    print("medium-shape context VRAM: %.1f MB" % (ctx.vram_bytes() / 1e6))
    ctx.close()
    m.close()


def test_device_greedy_loop_matches_host_loop(hip_tiny, golden, tiny_model):
    """wh_decode_greedy (captured hipGraph replayed per token, position + flags in device memory, fused softmax+sampler)
    must produce exactly the tokens of the host-driven loop wh_decode -> wh_sample_best -> feed back."""
    sp = gf.special_tokens(tiny_model.hparams)
    mel = torch.from_numpy(golden["mel"]).cuda()
    prompt = np.array([[sp["sot"], sp["transcribe"], sp["not_"]]] * 2, np.int32)
    n_steps = 12
    ctx = binding.HipContext(hip_tiny, 2)

    def start():
        ctx.encode(torch.stack([mel, mel]))
        ctx.decode(prompt, 0, want_logits=False, want_probs=False)
        return ctx.sample_best(2, True, True)

    first = start()
    toks = np.array([t["id"] for t in first], np.int32)
    host_ids, host_p = [], []
    cur = toks.copy()
    for s in range(n_steps):
        ctx.decode(cur[:, None], 3 + s, want_logits=False, want_probs=False)
        sb = ctx.sample_best(2)
        host_ids.append([t["id"] for t in sb])
        host_p.append([t["p"] for t in sb])
        cur = np.array(host_ids[-1], np.int32)
    host_ids = np.array(host_ids, np.int32)

    for flags in (0, binding.WH_FLAG_NO_GRAPH):
        ctx.set_flags(flags)
        assert [t["id"] for t in start()] == list(toks)
        ids, data = ctx.decode_greedy(toks, 3, n_steps)
        print("greedy flags=%d ids[:,0]=%s" % (flags, ids[:, 0]))
        assert np.array_equal(ids, host_ids)
        assert np.allclose([[t["p"] for t in row] for row in data], host_p, rtol=0, atol=1e-9)
        assert np.array_equal(ids[:, 0], ids[:, 1])
        # replaying the captured graph a second time (fresh state) is deterministic
        start()
        ids2, _ = ctx.decode_greedy(toks, 3, n_steps)
        assert np.array_equal(ids2, ids)
    # forced initial timestamp on the first device-side sample: one step from the prompt's last token must reproduce
    # sampleTimestamp(true) on the same probabilities
    ctx.set_flags(0)
    start()
    ctx.decode(prompt[:, :2], 0, want_logits=False, want_probs=False)
    ids3, data3 = ctx.decode_greedy(prompt[:, 2], 2, 1, force_first_timestamp=True, first_is_initial=True)
    assert list(ids3[0]) == list(toks)
    ctx.close()


def test_async_window_decode_and_concurrent_contexts(hip_tiny, golden, tiny_model):
    """wh_decode_window_start/finish (prompt + first sample + greedy steps, no host sync) equals the blocking path, and
    two contexts driven back to back from one host thread (they overlap on the GPU) do not disturb each other."""
    sp = gf.special_tokens(tiny_model.hparams)
    mel = torch.from_numpy(golden["mel"]).cuda()
    rng = np.random.default_rng(4)
    mel2 = torch.from_numpy(rng.uniform(-1, 1, (80, 3000)).astype(np.float32)).cuda()
    prompt = np.array([sp["sot"], sp["transcribe"], sp["not_"]], np.int32)
    n_steps = 10

    def blocking(m):
        ctx = binding.HipContext(hip_tiny, 1)
        ctx.encode(m)
        ctx.decode(prompt[None, :], 0, want_logits=False, want_probs=False)
        first = ctx.sample_best(1, True, True)[0]["id"]
        ids, _ = ctx.decode_greedy([first], 3, n_steps)
        ctx.close()
        return [first] + [int(x) for x in ids[:, 0]]

    want1, want2 = blocking(mel), blocking(mel2)
    a, b = binding.HipContext(hip_tiny, 1), binding.HipContext(hip_tiny, 1)
    for rep in range(2):                      # second repetition replays the captured graphs
        a.encode(mel)
        a.decode_window_start(prompt, n_steps)
        b.encode(mel2)
        b.decode_window_start(prompt, n_steps)
        ids_a, p_a = a.decode_window_finish()
        ids_b, p_b = b.decode_window_finish()
        assert [int(x) for x in ids_a[:, 0]] == want1 and [int(x) for x in ids_b[:, 0]] == want2
        assert (p_a > 0).all() and (p_b > 0).all()
    a.close()
    b.close()


@pytest.mark.gpu
@pytest.mark.parametrize("batch", [1, 3, 6])
def test_fetch_through_the_host_mailbox_equals_the_event_path(hip_tiny, golden, tiny_model, batch):
    """wh_decode_window_fetch while further steps are queued: for up to 4 sequences the sampler stamps a pinned host mailbox that
    the host polls (nothing is enqueued on the decode stream), larger batches and WH_NO_MAILBOX contexts wait for an event and
    copy. Both must hand back, sample by sample, what wh_decode_window_finish returns at the end -- across two windows, so
    that stamps of the first window cannot be taken for the second's."""
    sp = gf.special_tokens(tiny_model.hparams)
    mel = torch.stack([torch.from_numpy(golden["mel"]).cuda()] * batch)
    prompt = np.tile(np.array([sp["sot"], sp["transcribe"], sp["not_"]], np.int32), (batch, 1))
    results = {}
    for mode in ("mailbox", "events"):
        if mode == "events":
            os.environ["WH_NO_MAILBOX"] = "1"
        try:
            ctx = binding.HipContext(hip_tiny, batch)
        finally:
            os.environ.pop("WH_NO_MAILBOX", None)
        got = []
        for window in range(2):
            ctx.encode(mel)
            ctx.decode_window_start(prompt, 1)
            fetched = [ctx.decode_window_fetch(0, 1)]
            for i in range(1, 9):
                ctx.decode_window_continue(1)              # one step queued behind the one about to be read
                fetched.append(ctx.decode_window_fetch(i, 1))
            ids, _ = ctx.decode_window_finish()
            assert np.array_equal(np.concatenate(fetched), ids[:len(fetched)])
            got.append(ids)
        ctx.close()
        results[mode] = got
    for w in range(2):
        assert np.array_equal(results["mailbox"][w], results["events"][w])


@pytest.mark.gpu
def test_pipelined_passes_in_flight(hip_tiny, golden, tiny_model):
    """bench.py's steady state: three contexts, each pass (encoder + prompt + greedy steps) enqueued without a host sync while
    the two before it are still running; every pass must return what a lone blocking pass returns."""
    sp = gf.special_tokens(tiny_model.hparams)
    rng = np.random.default_rng(11)
    mels = [torch.from_numpy(golden["mel"]).cuda(),
            torch.from_numpy(rng.uniform(-1, 1, (80, 3000)).astype(np.float32)).cuda(),
            torch.from_numpy(rng.uniform(-1, 1, (80, 3000)).astype(np.float32)).cuda()]
    torch.cuda.synchronize()
    prompt = np.array([sp["sot"], sp["transcribe"], sp["not_"]], np.int32)
    n_steps = 12
    want = []
    for m in mels:
        ctx = binding.HipContext(hip_tiny, 1)
        ctx.encode(m)
        ctx.decode_window_start(prompt, n_steps)
        want.append([int(x) for x in ctx.decode_window_finish()[0][:, 0]])
        ctx.close()
    ctxs = [binding.HipContext(hip_tiny, 1) for _ in range(3)]
    pending, got = [], []
    for i in range(9):
        if len(pending) == 3:
            j, c = pending.pop(0)
            got.append((j, [int(x) for x in c.decode_window_finish()[0][:, 0]]))
        c = ctxs[i % 3]
        c.encode(mels[(i * 2) % 3], sync=False)           # the mel a context sees changes from pass to pass
        c.decode_window_start(prompt, n_steps)
        pending.append(((i * 2) % 3, c))
    for j, c in pending:
        got.append((j, [int(x) for x in c.decode_window_finish()[0][:, 0]]))
    assert len(got) == 9
    for j, ids in got:
        assert ids == want[j]
    for c in ctxs:
        c.close()


def test_streamed_mel_window(hip_tiny, golden, tiny_model):
    """wh_mel_spectrogram_window against the restatement of MelStreamer::makeBuffer (oracle/whisper_np.py MelStreamerNP): per-window
    maximum, the re-used maximum when a request ends where the last one ended, zero frames past the reader's chunks; and the
    invariant that ties it to the pinned path: on a clip of one window the streamed spectrogram is the whole-buffer one."""
    ctx = binding.HipContext(hip_tiny, 1)
    pcm = golden["pcm16"].astype(np.float32) / 32768.0
    loud = pcm.copy()
    loud[16000 * 6:] *= 0.02                        # quiet tail: window-local and global maxima differ
    dev = torch.from_numpy(loud).cuda()
    st = wn.MelStreamerNP(loud, tiny_model.filters)
    assert st.length == 1100
    for off, ln, reuse in ((0, 1100, False), (600, 500, True), (700, 300, False), (900, 200, False), (950, 150, True)):
        want = st.make_buffer(off, ln)
        got = ctx.mel_spectrogram_window(dev, off, ln, reuse_previous_max=reuse).cpu().numpy()
        d = report("streamed mel window off=%d len=%d reuse=%d" % (off, ln, reuse), got, want)
        assert got.shape == (80, ln) and d.max() < 2e-5
    whole = ctx.mel_spectrogram(dev).cpu().numpy()
    one = ctx.mel_spectrogram_window(dev, 0, 1100).cpu().numpy()
    assert np.abs(whole - one).max() < 1e-6
    # the window normalised by its own maximum is NOT the slice of the whole-buffer spectrogram when the maxima differ
    tail = ctx.mel_spectrogram_window(dev, 700, 300).cpu().numpy()
    assert np.abs(tail - whole[:, 700:1000]).max() > 1e-2
    # a reader that over-estimated its length: frames beyond its chunks are zero before normalisation -> (max(0, mmax-8)+4)/4
    short = ctx.mel_spectrogram_window(dev, 1000, 150, n_chunks=1100).cpu().numpy()
    ref_short = wn.MelStreamerNP(loud, tiny_model.filters)
    ref_short.n_chunks = 1100
    assert np.abs(short - ref_short.make_buffer(1000, 150)).max() < 2e-5
    ctx.close()


def test_streamed_mel_window_against_the_reference(hip_tiny, tiny_model):
    """SURVEY.md 8 row f1 on the HIP path: wh_mel_spectrogram_window against outputs of the REFERENCE's streaming spectrogram
    (Whisper/Whisper/MelStreamer.cpp + melSpectrogram.cpp compiled unmodified; tests/golden/ref_melstreamer.npz, generated by
    tests/golden/make_golden_melstreamer.py) on the request sequence iContext::runStreamed makes over a 63917-sample stream: fresh
    maximum, re-used maximum (the request ends where the last one ended, MelStreamer.cpp:158-172), window-local maxima, and a
    request past the stream's length (MelStreamerSimple: the partial chunk's frame is computed, frames after it are zero before
    normalisation). Bound: the reference's FP32 FFT against the FP64 matrix-core transform, as in row a1 (5e-4 max, 5e-6 mean);
    the live streamer, when oracle/_ref carries it, must give the fixture's bits."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden_melstreamer", os.path.join(ROOT, "tests", "golden", "make_golden_melstreamer.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", "ref_melstreamer.npz")))
    pcm = mg.clip()
    assert len(pcm) == int(g["n_samples"])
    ctx = binding.HipContext(hip_tiny, 1)
    dev = torch.from_numpy(pcm).cuda()
    last_end = None
    for i, (off, ln) in enumerate(g["requests"]):
        off, ln = int(off), int(ln)
        reuse = last_end == off + ln           # the host loop's rule (whisperImpl.cpp encodeWindow; MelStreamer.cpp:158-166)
        if not reuse:
            last_end = off + ln
        got = ctx.mel_spectrogram_window(dev, off, ln, reuse_previous_max=reuse).cpu().numpy()
        want = g["window%d" % i]
        d = report("streamed mel window vs the reference's MelStreamer off=%d len=%d reuse=%d" % (off, ln, reuse), got, want)
        assert got.shape == want.shape and d.max() < 5e-4 and d.mean() < 5e-6
        if i < 2:       # requests 0 and 1 are clamped at (the stream's maximum - 8): the floor says which maximum was used
            assert abs(float(got.min()) - float(want.min())) < 5e-6
    # window0 is also Spectrogram::pcmToMel of the reference's GPU model (Spectrogram.cpp:64-122, asserted equal by the generator):
    # row a1's whole-buffer kernel against the file SURVEY.md 8(a1) names
    whole = ctx.mel_spectrogram(dev).cpu().numpy()
    d = report("whole-buffer mel vs the reference's Spectrogram::pcmToMel", whole, g["window0"])
    assert whole.shape == (80, 399) and d.max() < 5e-4 and d.mean() < 5e-6
    off, ln = (int(x) for x in g["past_end"])
    got = ctx.mel_spectrogram_window(dev, off, ln).cpu().numpy()
    want = g["past_end_simple"]
    d = report("streamed mel window past the stream's end", got, want)
    assert d.max() < 5e-4
    n_chunks = (len(pcm) + 159) // 160
    assert np.array_equal(got[:, n_chunks - off:], want[:, n_chunks - off:])       # zero frames: one constant, the same one
    ctx.close()
    from oracle import ref
    if ref.melstreamer_available():
        st = ref.RefMelStreamer(pcm, tiny_model.filters, threads=1)      # MelStreamerSimple: no background thread, no start-up race
        assert np.array_equal(st.make_buffer(0, 399), g["window0"])
        st.close()


@pytest.mark.parametrize("batch", [3, 6, 12, 37])
def test_ragged_prompts_equal_each_sequence_alone(hip_tiny, golden, tiny_model, batch):
    """wh_decode_window_start_ragged: the sequences of a lock-step batch carry prompts of different lengths (the streams of
    Whisper::runFullBatch: [prev] + own past text + task tokens, ContextImpl.cpp:565-576) and stand at different decoder positions.
    Sequence b must compute what it computes when every sequence of the batch has ITS prompt (same slot, same window, same kernels):
    token ids identical, probabilities to FP32 round-off, its self-attention cache rows to one FP16 ulp (the prompt step's products
    run at a different row count, hence through differently tiled kernels). Batch 3 = the single-stream launches (decode1.hip),
    6 = separate QKV product + attention, 12 / 37 = the fused self-attention block (NQ sequences per workgroup at different positions)."""
    sp = gf.special_tokens(tiny_model.hparams)
    hp = tiny_model.hparams
    rng = np.random.default_rng(17)
    base = golden["mel"]
    mels = np.stack([np.roll(base, 37 * b, axis=1) if b % 3 else (base * (1.0 - 0.02 * (b % 5))).astype(np.float32) for b in range(batch)])
    kinds = [[sp["sot"], sp["not_"]],
             [sp["prev"], 700, 701, 702, 703, sp["sot"], sp["not_"]],
             [sp["prev"]] + [int(x) for x in rng.integers(300, 5000, 17)] + [sp["sot"], sp["not_"]],
             [sp["prev"], 4242, sp["sot"], sp["not_"]]]
    prompts = [kinds[(b * 7 + b // 4) % len(kinds)] for b in range(batch)]
    n_steps, n_more = 6, 5
    ctx = binding.HipContext(hip_tiny, batch)
    mel_dev = torch.from_numpy(mels).cuda()

    def run(ps, ragged):
        ctx.encode(mel_dev)
        if ragged:
            ctx.decode_window_start_ragged(ps, n_steps)
        else:
            ctx.decode_window_start(np.asarray(ps, np.int32), n_steps)
        first = ctx.decode_window_fetch_data(0, 1 + n_steps)
        ctx.decode_window_continue(n_more)
        while not ctx.decode_window_ready(1 + n_steps, n_more):
            pass
        more = ctx.decode_window_fetch_data(1 + n_steps, n_more)
        data = {k: np.concatenate([first[k], more[k]]) for k in first}
        rows = max(len(p) for p in ps) + n_steps + n_more
        kv = [(ctx.debug_read("self-k", il, rows), ctx.debug_read("self-v", il, rows)) for il in (0, hp.n_text_layer - 1)]
        return data, kv

    got, got_kv = run(prompts, True)
    assert np.isfinite(got["p"]).all()
    worst_p = worst_kv = 0.0
    for kind in kinds:
        who = [b for b in range(batch) if prompts[b] == kind]
        if not who:
            continue
        want, want_kv = run([kind] * batch, False)
        n = len(kind) + n_steps + n_more
        for b in who:
            assert [int(x) for x in got["id"][:, b]] == [int(x) for x in want["id"][:, b]], (batch, b, len(kind))
            assert [int(x) for x in got["tid"][:, b]] == [int(x) for x in want["tid"][:, b]]
            for f in ("p", "pt", "ptsum"):
                worst_p = max(worst_p, float(np.abs(got[f][:, b] - want[f][:, b]).max()))
            for (gk, gv), (wk, wv) in zip(got_kv, want_kv):
                worst_kv = max(worst_kv, float(np.abs(gk[b, :n] - wk[b, :n]).max()), float(np.abs(gv[b, :n] - wv[b, :n]).max()))
    print("ragged batch %d: prompt lengths %s; max |p - p_alone| %.2e, max |self-KV - alone| %.2e" % (batch, sorted(set(len(p) for p in prompts)), worst_p, worst_kv))
    assert worst_p < 2e-4 and worst_kv < 4e-3
    # the same window again with uniform prompts through the ragged entry point == the plain entry point, bit for bit
    a, _ = run([kinds[1]] * batch, True)
    b_, _ = run([kinds[1]] * batch, False)
    assert all(np.array_equal(a[k], b_[k]) for k in a)
    ctx.close()


def test_encode_windows_equals_encode(hip_tiny, golden):
    """wh_encode_windows: every window of the batch from its OWN spectrogram (length, offset) -- what the batch scheduler feeds the
    encoder with -- fills the cross-attention caches exactly as wh_encode does for that spectrogram alone; a null window is zeros."""
    rng = np.random.default_rng(8)
    mel_a = torch.from_numpy(golden["mel"]).cuda()                                         # [80][1100]
    mel_b = torch.from_numpy(rng.uniform(-1, 1, (80, 4321)).astype(np.float32)).cuda()
    ctx = binding.HipContext(hip_tiny, 4)
    one = binding.HipContext(hip_tiny, 1)
    want = []
    for mel, off in ((mel_a, 0), (mel_b, 2000), (mel_a, 700)):
        one.encode(mel, offsets=[off])
        want.append((one.debug_read("cross-k", 1)[0].copy(), one.debug_read("cross-v", 3)[0].copy()))
    one.encode(torch.zeros((80, 3000), device="cuda"))
    want.append((one.debug_read("cross-k", 1)[0].copy(), one.debug_read("cross-v", 3)[0].copy()))
    ctx.encode_windows([(mel_a, 0), (mel_b, 2000), (mel_a, 700), (None, 0)])
    k, v = ctx.debug_read("cross-k", 1), ctx.debug_read("cross-v", 3)
    for b in range(4):
        assert np.array_equal(k[b], want[b][0]) and np.array_equal(v[b], want[b][1]), b
    ctx.close()
    one.close()


def test_beam_candidates_and_cache_reorder(hip_tiny, golden, tiny_model):
    """The data path of beam search on hypothesis groups (extension, whisper_hip.h): wh_beam_candidates gives every sequence's `width` best
    continuations under sampleBest's rules -- candidate 0 IS wh_sample_best's token, probabilities do not increase, no token twice -- and
    wh_reorder_self_cache makes sequence j continue sequence parents[j]: its self-attention cache rows are the parent's (every layer, any
    permutation), and its next step computes what the parent's next step computes for the same token."""
    sp = gf.special_tokens(tiny_model.hparams)
    hp = tiny_model.hparams
    hyp, windows = 5, 2
    S = hyp * windows
    mel = torch.from_numpy(golden["mel"]).cuda()
    ctx = binding.HipContext(hip_tiny, windows, hypotheses=hyp)
    ctx.encode(torch.stack([mel, torch.roll(mel, 100, 1)]))
    prompt = np.array([[sp["sot"], sp["not_"], 300 + s] for s in range(S)], np.int32)       # every hypothesis its own third token
    ctx.decode(prompt, 0, want_logits=False, want_probs=False)
    for first in (True, False):
        cand = ctx.beam_candidates(S, 5, force_timestamp=first, is_initial=first)
        best = ctx.sample_best(S, first, first)
        assert [int(x) for x in cand["id"][:, 0]] == [b["id"] for b in best]
        assert np.allclose(cand["p"][:, 0], [b["p"] for b in best], rtol=0, atol=0)
        assert (np.diff(cand["p"], axis=1) <= 0).all()
        assert all(len(set(row)) == 5 for row in cand["id"])
        assert not np.isin(cand["id"], [sp["sot"], sp["solm"], sp["not_"]]).any()
    # one more token, so that position 3 differs between hypotheses too
    toks = cand["id"][:, 1].astype(np.int32)[:, None]
    ctx.decode(toks, 3, want_logits=False, want_probs=False)
    rows = 4
    before = [(ctx.debug_read("self-k", il, rows).copy(), ctx.debug_read("self-v", il, rows).copy()) for il in (0, hp.n_text_layer - 1)]
    parents = np.array([2, 2, 0, 4, 1, 5 + 4, 5 + 0, 5 + 0, 5 + 3, 5 + 2], np.int32)      # a permutation with a cycle in window 0, a fan-out in window 1
    # what every parent computes next for token 777 (before anything moves)
    nxt = np.full((S, 1), 777, np.int32)
    want_logits, _ = ctx.decode(nxt, rows)
    # undo that step's cache row (position 4 is simply overwritten again below), reorder, repeat the step
    ctx.reorder_self_cache(parents, rows)
    after = [(ctx.debug_read("self-k", il, rows), ctx.debug_read("self-v", il, rows)) for il in (0, hp.n_text_layer - 1)]
    for (bk, bv), (ak, av) in zip(before, after):
        for j in range(S):
            assert np.array_equal(ak[j], bk[parents[j]]) and np.array_equal(av[j], bv[parents[j]]), j
    got_logits, _ = ctx.decode(nxt, rows)
    for j in range(S):
        assert np.array_equal(got_logits[j], want_logits[parents[j]]), j
    ctx.reorder_self_cache(np.arange(S, dtype=np.int32), rows)                                # identity: nothing to do
    ctx.close()


def test_beam_search_steps_on_the_device(hip_tiny, golden, tiny_model):
    """wh_beam_window_*: 3 windows x 5 hypotheses, FORCED steps (no stop rules: what bench.py --workload beam5 runs on random weights), the ranking
    as a kernel inside the captured step graph -- against the same search ranked on the host after every step through wh_beam_candidates /
    wh_reorder_self_cache (numpy: parent score + log p, stable order; round 4's data path): the best hypothesis' ids of every window, its score,
    and the chain of every surviving hypothesis are the same. Chunked enqueueing (start + continue) equals one enqueue."""
    hp = tiny_model.hparams
    sp = gf.special_tokens(hp)
    k, hyp, n_steps = 3, 5, 20
    rng = np.random.default_rng(21)
    mel = np.stack([np.roll(golden["mel"], 53 * b, axis=1) + 0.02 * rng.standard_normal(golden["mel"].shape).astype(np.float32) for b in range(k)]).astype(np.float32)
    mel_dev = torch.from_numpy(mel).cuda()
    base = np.asarray([sp["sot"], sp["transcribe"], sp["not_"]], np.int32)
    S = k * hyp
    c = binding.HipContext(hip_tiny, k, hypotheses=hyp)
    # ---- host-ranked ----
    c.encode(mel_dev)
    c.decode(np.tile(base, (S, 1)), 0, want_logits=False, want_probs=False)
    cand = c.beam_candidates(S, hyp, True, True)
    p0 = cand["p"][::hyp].astype(np.float64)
    order = np.argsort(-np.log(np.maximum(p0, 1e-30)), axis=1, kind="stable")
    score = np.take_along_axis(np.log(np.maximum(p0, 1e-30)), order, axis=1)
    tok = np.take_along_axis(cand["id"][::hyp], order, axis=1).astype(np.int32)
    parents = (np.arange(k)[:, None] * hyp + np.zeros((1, hyp), np.int64)).astype(np.int32)
    hist = tok[:, :, None]
    for s_ in range(n_steps):
        c.reorder_self_cache(parents.reshape(-1), len(base) + s_)
        c.decode(tok.reshape(-1, 1), len(base) + s_, want_logits=False, want_probs=False)
        cand = c.beam_candidates(S, hyp)
        pool = score[:, :, None] + np.log(np.maximum(cand["p"].reshape(k, hyp, hyp).astype(np.float64), 1e-30))
        flat = pool.reshape(k, hyp * hyp)
        order = np.argsort(-flat, axis=1, kind="stable")[:, :hyp]
        par_local = order // hyp
        score = np.take_along_axis(flat, order, axis=1)
        tok = np.take_along_axis(cand["id"].reshape(k, hyp * hyp), order, axis=1).astype(np.int32)
        parents = (np.arange(k)[:, None] * hyp + par_local).astype(np.int32)
        hist = np.concatenate([np.take_along_axis(hist, par_local[:, :, None], axis=1), tok[:, :, None]], axis=2)
    # ---- device-ranked, in one enqueue and in chunks ----
    for chunks in ([n_steps], [7, 1, 12]):
        c.encode(mel_dev)
        c.beam_window_start(np.tile(base, (k, 1)), hyp, chunks[0])
        for more in chunks[1:]:
            c.beam_window_continue(more)
        st = c.beam_window_status()
        rec = c.beam_window_records(0, n_steps + 1)
        for w in range(k):
            assert st[w]["step"] == n_steps + 1 and st[w]["nLive"] == hyp and st[w]["nFinished"] == 0 and not st[w]["done"]
            for j in range(hyp):
                h = st[w]["live"][j]
                ids = c.beam_chain(rec, w, h["rec"])
                assert ids == [int(x) for x in hist[w, j]], (chunks, w, j)
                assert abs(h["sum"] - score[w, j]) < 1e-9 * max(1.0, abs(score[w, j])) and h["nTok"] == n_steps + 1
    print("device-ranked beam search == host-ranked on %d windows x %d hypotheses x %d steps; best scores %s" % (k, hyp, n_steps, np.round(score[:, 0], 4)))
    c.close()


@pytest.mark.parametrize("M", [65, 100, 112, 128])
def test_gemv_all_rows_variant(M):
    """TUNE_GEMV_MT8: the 65 .. 128-row decode product with ALL rows in one workgroup (weights streamed once), against the default
    row-grouped kernel on the MLP up-projection's shape (N = 4096, K = 1024, GELU epilogue) and a plain FP32 product: the same MFMA
    tiles and the same K split, so the results are identical bit for bit."""
    rng = np.random.default_rng(M)
    N, K = 4096, 1024
    a = (rng.standard_normal((M, K)) * 0.5).astype(np.float16)
    w = (rng.standard_normal((N, K)) * 0.05).astype(np.float16)
    bias = rng.standard_normal(N).astype(np.float32)
    ad, wd, bd = (torch.from_numpy(x).cuda() for x in (a, w, bias))
    L = binding.lib()
    outs = {}
    for name, mask in (("default", binding.TUNE_DEFAULT & ~binding.TUNE_GEMV_MT8), ("mt8", binding.TUNE_DEFAULT | binding.TUNE_GEMV_MT8)):
        L.wh_debug_set_tuning(mask)
        try:
            o16 = torch.zeros((M, N), dtype=torch.float16, device="cuda")
            binding.check(L.wh_op_mul_mat_gelu(None, ad.data_ptr(), wd.data_ptr(), bd.data_ptr(), o16.data_ptr(), M, N, K))
            o32 = torch.zeros((M, N), dtype=torch.float32, device="cuda")
            binding.check(L.wh_op_mul_mat(None, ad.data_ptr(), wd.data_ptr(), bd.data_ptr(), None, o32.data_ptr(), M, N, K))
            torch.cuda.synchronize()
            outs[name] = (o16.cpu().numpy(), o32.cpu().numpy())
        finally:
            L.wh_debug_set_tuning(binding.TUNE_DEFAULT)
    want = a.astype(np.float32) @ w.astype(np.float32).T + bias
    assert np.abs(outs["mt8"][1] - want).max() < 2e-3
    assert np.array_equal(outs["default"][0], outs["mt8"][0]) and np.array_equal(outs["default"][1], outs["mt8"][1])


@pytest.mark.parametrize("batch", [1, 2, 4])
def test_spread_sampler_equals_the_one_workgroup_sampler(golden, batch):
    """TUNE_SAMPLE_SPREAD (measured, not the default): for the few rows of one stream the sampler's row is cut into 64 slices over the chip
    (slice maxima -> exponentials, slice sums / argmaxima / top-4 lists -> one wave merges them and applies sampleBest's rules,
    ContextImpl.cpp:71-157) instead of one workgroup on one CU. Same numbers: token ids, timestamp ids and p identical; pt / ptsum add the same
    products in another order (double). Two models: random weights (every sample a timestamp by the sum rule) and an audio-conditioned one
    (text tokens between timestamps, the forced first timestamp)."""
    mel = torch.from_numpy(golden["mel"]).cuda()
    mels = torch.stack([torch.roll(mel, 53 * b, 1) for b in range(batch)])
    L = binding.lib()
    hp_ml = gf.hparams_for("test-d128-ml")
    cond = gf.conditioned_model(gf.conditioned_layout(hp_ml), 4, kind="test-d128-ml", seed=10)
    for model, text in ((gf.synth_model("test-d128", seed=77), False), (cond, True)):
        sp = gf.special_tokens(model.hparams)
        prompt = [sp["prev"], sp["sot"], sp["sot"] + 1, sp["transcribe"]] if text else [sp["sot"]]
        m = binding.HipModel.from_ggml(model)
        res = {}
        for name, mask in (("spread", binding.TUNE_DEFAULT | binding.TUNE_SAMPLE_SPREAD), ("one", binding.TUNE_DEFAULT & ~binding.TUNE_SAMPLE_SPREAD)):
            L.wh_debug_set_tuning(mask)
            try:
                ctx = binding.HipContext(m, batch)
                ctx.encode(mels)
                ctx.decode_window_start(np.array([prompt] * batch, np.int32), 14, force_first_timestamp=True, first_is_initial=True)
                res[name] = ctx.decode_window_fetch_data(0, 15)
                ctx.close()
            finally:
                L.wh_debug_set_tuning(binding.TUNE_DEFAULT)
        a, b = res["spread"], res["one"]
        assert np.array_equal(a["id"], b["id"]) and np.array_equal(a["tid"], b["tid"]) and np.array_equal(a["p"], b["p"])
        assert np.allclose(a["pt"], b["pt"], rtol=1e-6, atol=0) and np.allclose(a["ptsum"], b["ptsum"], rtol=1e-6, atol=0)
        if text:
            assert (a["id"] < sp["eot"]).any() and (a["id"] > sp["beg"]).any()
        m.close()


def test_medium_shape_against_exact_arithmetic():
    """The yardstick at the MEASURED shape (tests/golden/truth_medium.npz, make_golden_truth_medium.py): the bench's ggml-medium-shape model and
    window, the prompt and three teacher-forced steps, in float64 with no intermediate rounding (oracle WhisperTruth). north_star asks for 1e-3
    against the reference; the reference (8 threads) itself sits `ref8_vs_truth` from the exact result at this shape, so the HIP path is held to
    that distance: not further from exact arithmetic than 1.25 x the reference is (max and mean), top-1 equal to the exact result's."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "truth_medium.npz")
    g = np.load(path)
    import json as _json
    stats = _json.loads(str(g["stats"]))
    model = gf.synth_model("medium", seed=1)
    m = binding.HipModel.from_ggml(model)
    del model
    ctx = binding.HipContext(m, 1)
    ctx.encode(torch.from_numpy(g["mel"]).cuda())
    steps = [[int(t) for t in g["prompt"]]] + [[int(t)] for t in g["extra"]]
    n_past = 0
    for i, toks in enumerate(steps):
        gl, _ = ctx.decode(np.asarray([toks], np.int32), n_past)
        n_past += len(toks)
        tl = g["truth_logits%d" % i].astype(np.float64)
        d = np.abs(gl[0].astype(np.float64) - tl)
        s = stats[i]
        print("medium step %d: |HIP - exact| max %.2e mean %.2e   |reference(8 threads) - exact| max %.2e mean %.2e   span %.2f, exact top-2 margin %.3f" %
              (i, d.max(), d.mean(), s["ref8_vs_truth_max"], s["ref8_vs_truth_mean"], s["span"], s["truth_top2_margin"]))
        assert d.max() <= 1.25 * s["ref8_vs_truth_max"] and d.mean() <= 1.25 * s["ref8_vs_truth_mean"]
        assert int(np.argmax(gl[0])) == s["truth_top1"] or s["truth_top2_margin"] < 2.0 * d.max()
    ctx.close()
    m.close()
