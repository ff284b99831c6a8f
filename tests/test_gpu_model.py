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
import numpy as np
import pytest

torch = pytest.importorskip("torch")

from oracle import whisper_np as wn  # noqa: E402
from whisper_amd import binding, ggml_format as gf  # noqa: E402

pytestmark = pytest.mark.gpu

E2E_MAX, E2E_MEAN = 6e-3, 1e-3


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
            assert d.max() < 8e-3 and d.mean() < E2E_MEAN
            mine = np_tiny.kv.cross_k[il] if nm == "k" else np_tiny.kv.cross_v[il]
            d = report("cross-%s[%d] vs restatement" % (nm, il), got, mine)
            assert d.max() < 8e-3 and d.mean() < E2E_MEAN
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
        assert d.max() < 8e-3 and d.mean() < E2E_MEAN
    ctx.close()


def test_decoder_fast_path(hip_tiny, golden, np_tiny):
    """FP32 P.V (what the reference's own GPU shaders do) against the restatement with the same choice."""
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
        d = report("fast logits step %d vs reference (1 thread)" % i, logits[0], golden["logits%d" % i])
        pos += ln
        n_past += ln
    ctx.close()


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
    assert d.max() < 2e-4            # skinny kernels both ways; only the accumulation grouping of the attention differs
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
    assert d.max() < 8e-3 and d.mean() < E2E_MEAN
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
