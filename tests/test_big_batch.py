"""GPU parity tests of the BIG lock-step batches (round 5): one context of 129 .. 512 windows instead of two of 112.

What is new on that path and what pins it:
  * gemmDecRows (decode products of 129 .. 512 rows, 64 x 64 output tiles)  -- op level against float64, every tile shape bit-identical,
    and bit-identical with gemvFused on the rows both can compute (same K split, same summation order);
  * the encoder in chunks (activations sized for one chunk, cross-attention caches for the whole batch) -- equal caches whatever the chunk;
  * the whole decode step at > 128 sequences (selfBlockDec at 8 sequences per workgroup, the vocabulary product on the M-tiled kernel,
    sampler, captured graph) -- every window of a 150-window toy batch against the same window in a small batch;
  * THE PARITY CHAIN OF WHAT bench.py TIMES ("What's weak" 1): reference (16 threads) <-> 11-window context is
    test_gpu_model.py::test_medium_shape_against_the_reference; here 11-window context <-> 112- / 224- / 448-window context driven step by step
    (logits of eleven windows spread over the batch at every step) <-> the same big context through the captured greedy graph (ids identical):
    the kernel instances of the timed region are tied to the live reference link by link;
  * parity mode at the medium shape against the reference at ONE thread, where the reference is a point and not a band, with the stage probes.
"""
import ctypes as C
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")

from whisper_amd import binding, ggml_format as gf  # noqa: E402

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def ptr(t):
    return C.c_void_p(t.data_ptr())


def report(name, got, want):
    d = np.abs(np.asarray(got, np.float64) - np.asarray(want, np.float64))
    print("%-58s max|want|=%9.4f maxdiff=%.3e meandiff=%.3e" % (name, np.abs(want).max(), d.max(), d.mean()))
    return d


class option:
    """with option("dec_tile", 44): ...  -- an integer knob of the library for the duration of a block."""

    def __init__(self, name, value, default):
        self.name, self.value, self.default = name, value, default

    def __enter__(self):
        binding.set_option(self.name, self.value)

    def __exit__(self, *a):
        binding.set_option(self.name, self.default)


TILES = (0, 44, 42, 24, 22)


@pytest.mark.parametrize("M,N,K", [(129, 1024, 1024), (224, 1024, 4096), (448, 4096, 1024), (512, 1024, 1024), (300, 1280, 5120), (336, 3840, 1280),
                                   (200, 1000, 384), (448, 52, 2048), (150, 128, 128),
                                   # round 6, gemmDecTile's edges: ragged last row / column tiles, an odd N (element-wise epilogue), a quarter of K of ONE tile,
                                   # two K tiles per ring slot with the fewest slots (K = 2048 on 64 x 32 tiles)
                                   (450, 4100, 256), (449, 1031, 768), (320, 1024, 2048), (330, 2052, 1024)])
def test_mul_mat_big_batch_decode_rows(M, N, K):
    """129 .. 512 activation rows through gemmDecTile (the default where its tiles fill the chip, round 6) and gemmDecRows (every pinned tile shape): against float64, repeated launches and every tile shape bit-identical (the tile only
    decides which workgroup computes an element, never the order of its sum), and rows [0, 112) bit-identical with what gemvFused gives for a
    112-row batch when both split K over 4 waves (K < 2048: at K >= 2048 the 112-row kernel splits K over 8)."""
    rng = np.random.default_rng(M * 3 + N)
    a = rng.standard_normal((M, K)).astype(np.float16)
    w = (0.05 * rng.standard_normal((N, K))).astype(np.float16)
    bias = rng.standard_normal(N).astype(np.float32)
    res = rng.standard_normal((M, N)).astype(np.float32)
    want = (a.astype(np.float64) @ w.astype(np.float64).T + bias + res).astype(np.float32)
    ad, wd, bd, rd = dev(a), dev(w), dev(bias), dev(res)
    L = binding.lib()
    outs = {}
    for tile in TILES + (1,):
        with option("dec_tile", tile, 0):
            for rep in range(2):
                out = torch.full((M, N), float("nan"), dtype=torch.float32, device="cuda")
                binding.check(L.wh_op_mul_mat(None, ptr(ad), ptr(wd), ptr(bd), ptr(rd), ptr(out), M, N, K))
                torch.cuda.synchronize()
                if tile in outs:
                    assert torch.equal(out, outs[tile])
                outs[tile] = out
    d = report("mul_mat big-batch decode rows %dx%dx%d" % (M, N, K), outs[0].cpu().numpy(), want)
    assert d.max() < 2e-5 * max(1.0, np.sqrt(K / 128))
    for tile in TILES[1:]:
        assert torch.equal(outs[tile], outs[0]), "tile %d differs from the default" % tile
    d1 = report("   gemvFused row groups (dec_tile 1)", outs[1].cpu().numpy(), want)
    assert d1.max() < 2e-5 * max(1.0, np.sqrt(K / 128))
    if K < 2048 and K % 128 == 0:
        small = torch.full((112, N), float("nan"), dtype=torch.float32, device="cuda")
        binding.check(L.wh_op_mul_mat(None, ptr(ad), ptr(wd), ptr(bd), ptr(rd), ptr(small), 112, N, K))
        torch.cuda.synchronize()
        same = torch.equal(small, outs[0][:112])
        print("   rows [0, 112) equal to the 112-row decode kernel's: %s" % same)
        if N < 16384:       # (the vocabulary-sized N takes gemmAllRows at 112 rows: same K split, asserted too)
            assert same


@pytest.mark.parametrize("M,N,K", [(448, 4096, 1024), (224, 5120, 1280), (130, 2000, 512)])
def test_mul_mat_gelu_big_batch_decode_rows(M, N, K, golden):
    """The MLP up-projection of a 129 .. 512-row decode step: FP16 GELU-table epilogue (8-byte stores), every tile shape."""
    g = torch.Generator(device="cuda").manual_seed(M + N)
    a = torch.randn((M, K), generator=g, device="cuda").half()
    w = (0.1 * torch.randn((N, K), generator=g, device="cuda")).half()
    bias = torch.randn(N, generator=g, device="cuda")
    pre = (a.double() @ w.double().T + bias.double()).float()
    table = torch.from_numpy(golden["table_gelu"].astype(np.int32)).cuda()
    idx = pre.half().view(torch.int16).to(torch.int32) & 0xFFFF
    want = table[idx.long()].to(torch.int16).view(torch.float16).float()
    L = binding.lib()
    first = None
    for tile in TILES:
        with option("dec_tile", tile, 0):
            out = torch.zeros((M, N), dtype=torch.float16, device="cuda")
            binding.check(L.wh_op_mul_mat_gelu(None, ptr(a), ptr(w), ptr(bias), ptr(out), M, N, K))
            torch.cuda.synchronize()
        d = (out.float() - want).abs()
        frac = float((d > 0).float().mean())
        print("mul_mat_gelu big-batch rows tile %d %dx%dx%d: %.4f %% of entries differ, max %.3e" % (tile, M, N, K, 100 * frac, float(d.max())))
        assert frac < 0.02 and bool((d <= torch.maximum(torch.tensor(4e-3, device="cuda"), want.abs() * 2.0 ** -10)).all())
        if first is None:
            first = out
        assert torch.equal(out, first)


@pytest.mark.parametrize("seqs,heads,n_past,stride", [(150, 16, 54, 448), (70, 20, 0, 448), (129, 3, 63, 448), (130, 2, 64, 448), (66, 4, 200, 448), (65, 2, 447, 448)])
def test_self_attention_a_wave_per_sequence_and_head(seqs, heads, n_past, stride):
    """selfAttnDecWave: the causal self-attention of a single-token step for more than 64 sequences (what a lock-step batch beyond 128 runs behind its QKV
    product) against the restatement of mulMat(K,Q) -> diagMaskInf -> softMax -> mulMat(V,.) (whisper.cpp:1618-1660), and against attentionDecG -- the
    kernel the same call takes with the wave kernel switched off -- to the last FP16 ulp of the output. Key counts on both sides of the 64-key and
    8-slot boundaries, the first token (one key) and the last position of the context."""
    from test_gpu_ops import _np_decoder_attention
    from oracle import whisper_np as wn
    rng = np.random.default_rng(seqs * 100 + n_past)
    d = heads * 64
    n_keys = n_past + 1
    q = (rng.standard_normal((seqs, d)) * 0.8).astype(np.float16)
    K = (rng.standard_normal((seqs, heads, stride, 64)) * 0.8).astype(np.float16)
    V = rng.standard_normal((seqs, heads, stride, 64)).astype(np.float16)
    want = _np_decoder_attention(q, K, V, 1, n_keys, 1, n_past, 1, 0)
    qd, kd, vd = dev(q), dev(K), dev(V)
    L = binding.lib()
    got = {}
    for name, min_rows in (("wave", 32), ("attentionDecG", 1 << 20)):
        with option("self_wave_min_rows", min_rows, 32):
            out = torch.full((seqs, d), float("nan"), dtype=torch.float16, device="cuda")
            binding.check(L.wh_op_decoder_attention(None, ptr(qd), ptr(kd), ptr(vd), ptr(out), seqs, heads, 1, n_keys, stride, 1, n_past, 1, 0))
            torch.cuda.synchronize()
        got[name] = out.cpu().numpy().astype(np.float32)
        assert np.isfinite(got[name]).all()
        dd = report("self-attention %s %d seqs x %d heads, %d keys" % (name, seqs, heads, n_keys), got[name], wn.r16(want))
        assert dd.max() < 2e-3 and dd.mean() < 2e-4
    dk = np.abs(got["wave"] - got["attentionDecG"])
    print("    wave kernel vs attentionDecG: max %.3e, %.4f %% of the outputs differ" % (dk.max(), 100.0 * (dk > 0).mean()))
    assert dk.max() <= 2e-3 and (dk > 0).mean() < 0.02


def test_encoder_in_chunks(tiny_model, golden):
    """An 11-window batch encoded in chunks of 4 + 4 + 3 windows (option enc_chunk) against the same batch in one pass: the cross-attention caches
    of every window (first, middle and last decoder layer) agree bit for bit -- a window's rows never meet another window's -- and so do the logits
    of a prompt step. Offsets and per-window sources (wh_encode_windows) are cut at the chunk boundaries too."""
    m = binding.HipModel.from_ggml(tiny_model)
    hp = tiny_model.hparams
    n_win = 11
    rng = np.random.default_rng(5)
    mel = np.stack([np.roll(golden["mel"], 37 * b, axis=1) + 0.01 * rng.standard_normal(golden["mel"].shape).astype(np.float32) for b in range(n_win)])
    mel_dev = torch.from_numpy(mel.astype(np.float32)).cuda()
    offs = [(3 * b) % 50 for b in range(n_win)]
    sp = gf.special_tokens(hp)
    toks = np.array([[sp["sot"], sp["transcribe"], sp["not_"]]] * n_win, np.int32)
    got = {}
    for chunk in (128, 4):
        with option("enc_chunk", chunk, 128):
            ctx = binding.HipContext(m, n_win)
        ctx.encode(mel_dev, offsets=offs)
        caches = [(ctx.debug_read("cross-k", il).copy(), ctx.debug_read("cross-v", il).copy()) for il in (0, hp.n_text_layer // 2, hp.n_text_layer - 1)]
        one = ctx.debug_read("cross-k1", hp.n_text_layer - 1, rows=n_win - 1)
        assert np.array_equal(one, caches[-1][0][n_win - 1])
        logits, _ = ctx.decode(toks, 0)
        ctx.encode_windows([(mel_dev[b], offs[b]) for b in range(n_win)])
        caches_w = [(ctx.debug_read("cross-k", il).copy(), ctx.debug_read("cross-v", il).copy()) for il in (0, hp.n_text_layer - 1)]
        got[chunk] = (caches, logits, caches_w, ctx.vram_bytes())
        ctx.close()
    assert np.isfinite(got[128][1]).all() and float(np.abs(got[128][0][0][1]).max()) > 0.05
    for (k0, v0), (k1, v1) in zip(got[128][0], got[4][0]):
        assert np.array_equal(k0, k1) and np.array_equal(v0, v1)
    for (k0, v0), (k1, v1) in zip(got[128][2], got[4][2]):
        assert np.array_equal(k0, k1) and np.array_equal(v0, v1)
    assert np.array_equal(got[128][2][0][0], got[128][0][0][0])
    assert np.array_equal(got[128][1], got[4][1])
    print("context of 11 windows: %.1f MB with one encoder pass, %.1f MB with chunks of 4" % (got[128][3] / 1e6, got[4][3] / 1e6))
    assert got[4][3] < got[128][3]
    m.close()


def _window_inputs(n_small, seed):
    import bench
    return bench.synth_pcm(n_small, seed=seed)


def _mels(ctx, pcm_dev, idx):
    return torch.stack([ctx.mel_spectrogram(pcm_dev[i]) for i in idx])


def test_big_lock_step_batch_on_the_toy_model(tiny_model, golden):
    """150 sequences in ONE lock-step batch (d = 128 model; the encoder in two chunks of 75): every window's greedy ids through the captured graph equal
    the ids of the same window decoded in a batch of 5, except where that window's own top-2 margin is inside the noise band of a different
    summation order; the probabilities of the chosen tokens agree to the module's noise floor. Self-attention block at 8 and at 4 sequences per
    workgroup and as separate launches, vocabulary product on the M-tiled kernel and on gemmDecRows: the same ids."""
    m = binding.HipModel.from_ggml(tiny_model)
    hp = tiny_model.hparams
    sp = gf.special_tokens(hp)
    n_big, n_small, n_steps = 150, 5, 24
    rng = np.random.default_rng(11)
    base = np.stack([np.roll(golden["mel"], 101 * b, axis=1) + 0.02 * rng.standard_normal(golden["mel"].shape).astype(np.float32) for b in range(n_small)]).astype(np.float32)
    small_dev = torch.from_numpy(base).cuda()
    big_dev = torch.from_numpy(base[np.arange(n_big) % n_small]).cuda()
    prompt = [sp["sot"], sp["transcribe"], sp["not_"]]
    cs = binding.HipContext(m, n_small)
    cs.encode(small_dev)
    cs.decode_window_start(np.tile(np.asarray(prompt, np.int32), (n_small, 1)), n_steps)
    ids_s, ps_s = cs.decode_window_finish()
    cs.close()
    results = {}
    for name, opts in (("default", {}), ("selfBlockDec, 8 sequences per workgroup", {"self_fuse_max_rows": 512, "self_nq": 8}), ("selfBlockDec, 4 per workgroup", {"self_fuse_max_rows": 512, "self_nq": 4}),
                       ("self-attention through attentionDecG", {"self_wave_min_rows": 1 << 20}),
                       ("vocabulary on gemmDecRows", {"vocab_decrows": 1}), ("gemvFused row groups", {"dec_tile": 1})):
        defaults = {"self_nq": 0, "self_fuse_max_rows": 32, "vocab_decrows": 0, "dec_tile": 0, "self_wave_min_rows": 32}
        try:
            for k, v in opts.items():
                binding.set_option(k, v)
            cb = binding.HipContext(m, n_big)
            cb.encode(big_dev)
            cb.decode_window_start(np.tile(np.asarray(prompt, np.int32), (n_big, 1)), n_steps)
            ids_b, ps_b = cb.decode_window_finish()
            cb.close()
        finally:
            for k in opts:
                binding.set_option(k, defaults[k])
        results[name] = ids_b
        assert ids_b.shape == (n_steps + 1, n_big) and ((ids_b >= 0) & (ids_b < hp.n_vocab)).all() and np.isfinite(ps_b).all()
        diverged = 0
        worst_p = 0.0
        for w in range(n_big):
            ref_ids, ref_p = ids_s[:, w % n_small], ps_s[:, w % n_small]
            same = ids_b[:, w] == ref_ids
            first = int(np.argmin(same)) if not same.all() else n_steps + 1
            worst_p = max(worst_p, float(np.abs(ps_b[:first, w] - ref_p[:first]).max()) if first else 0.0)
            diverged += int(first <= n_steps)
        print("%-36s %3d of %d windows leave the small batch's greedy path (a near-tie decided by summation order); max |p - p_small| before that %.2e"
              % (name, diverged, n_big, worst_p))
        assert worst_p < 4e-3
        assert diverged <= n_big // 10
        # within the big batch: windows with the same audio give the same ids (batch invariance inside one launch sequence)
        for w in range(n_small, n_big):
            assert np.array_equal(ids_b[:, w], ids_b[:, w % n_small])
    m.close()


# ----------------------------------------------------------------------------------------------------------------------
# the parity chain of the timed batch
# ----------------------------------------------------------------------------------------------------------------------
CHAIN_BOUNDS = {
    # logits of one of our contexts against another of our contexts on the same window, teacher-forced: (max, mean). Different kernel instances
    # (LayerNorm fused or not, 4 or 8 waves over K, 1 / 4 / 8 sequences per self-attention workgroup) = different FP32 summation order = FP16
    # roundings of activations and cache rows that flip. Bounds = the medium / large-v2 bounds of the comparison with the reference itself
    # (test_gpu_model.SHAPE_BOUNDS); measured values are printed.
    "medium": (1.2e-2, 2e-3),
    "large-v2": (1.3e-2, 2.3e-3),
}


@pytest.fixture(scope="module")
def hip_medium():
    """The bench's model (ggml-medium shape, seed 1) once for the module."""
    model = gf.synth_model("medium", seed=1)
    m = binding.HipModel.from_ggml(model)
    del model
    yield m
    m.close()


def _chain(kind, n_big, n_small=11, n_cmp_steps=10, n_steps=51, m=None):
    import bench
    own = m is None
    if own:
        model = gf.synth_model(kind, seed=1)
        m = binding.HipModel.from_ggml(model)
        del model
    hp = m.hp
    sp = gf.special_tokens(hp)
    prompt = [sp["sot"], sp["sot"] + 1, sp["transcribe"]]
    pcm = bench.synth_pcm(n_small, seed=100)           # the windows test_*_shape_against_the_reference ties to the live reference
    pcm_dev = torch.from_numpy(pcm).cuda()
    big_of = np.arange(n_big) % n_small                # window i of the big batch = small window i % n_small
    # the small context's row k is compared with the LAST big window that holds its audio (k = 0: with window 0, the first)
    pick = [0] + [int(np.where(big_of == k)[0][-1]) for k in range(1, n_small)]
    cb = binding.HipContext(m, n_big)
    cs = binding.HipContext(m, n_small)
    print("%s shape: %d-window context %.1f GB (encoder chunk sized), %d-window context %.1f GB" % (kind, n_big, cb.vram_bytes() / 1e9, n_small, cs.vram_bytes() / 1e9))
    mel_small = _mels(cs, pcm_dev, range(n_small))
    mel_big = mel_small[torch.from_numpy(big_of).cuda()].contiguous()
    # (1) the big context through the captured greedy graph: exactly bench.py's clip_start sequence
    cb.encode(mel_big, sync=True)
    cb.decode_window_start(np.tile(np.asarray(prompt, np.int32), (n_big, 1)), n_steps, force_first_timestamp=True, first_is_initial=True)
    ids_graph, _ = cb.decode_window_finish()
    last_logits_graph = cb.debug_read("logits", rows=n_big)
    # (2) the big context step by step (host-driven wh_decode + wh_sample_best: the same kernel instances, positions as launch arguments)
    cb.encode(mel_big)
    cs.encode(mel_small)
    toks_b = np.tile(np.asarray(prompt, np.int32), (n_big, 1))
    n_past = 0
    l_max, l_mean = CHAIN_BOUNDS[kind]
    worst_max = worst_mean = 0.0
    ids_host = []
    agree = total = 0
    for step in range(n_steps + 1):
        want_logits = step < n_cmp_steps or step == n_steps
        gl, _ = cb.decode(toks_b, n_past, want_logits=want_logits, want_probs=False)
        if want_logits and step < n_cmp_steps:
            # (3) the small context teacher-forced with the tokens of the picked big windows
            sl, _ = cs.decode(toks_b[pick], n_past, want_probs=False)
            for k in range(n_small):
                d = np.abs(gl[pick[k]].astype(np.float64) - sl[k].astype(np.float64))
                worst_max, worst_mean = max(worst_max, float(d.max())), max(worst_mean, float(d.mean()))
                top2 = np.sort(sl[k])[-2:]
                same = int(np.argmax(gl[pick[k]]) == np.argmax(sl[k]))
                assert same or float(top2[1] - top2[0]) < 2 * l_max, ("top-1 differs with a margin of", float(top2[1] - top2[0]), "step", step, "window", pick[k])
                agree += same
                total += 1
        if step == n_steps:
            # the graph's last step left the same logits (bit for bit: same kernels, same inputs)
            assert np.array_equal(gl, last_logits_graph), float(np.abs(gl - last_logits_graph).max())
        n_past += toks_b.shape[1]
        nxt = cb.sample_best(n_big, step == 0, step == 0)
        toks_b = np.array([[t["id"]] for t in nxt], np.int32)
        ids_host.append(toks_b[:, 0].copy())
    ids_host = np.stack(ids_host)
    print("%s shape, %d windows in lock step vs the %d-window context on the same audio, %d teacher-forced steps x %d windows (incl. window 0 and %d): "
          "logits max %.3e mean %.3e, top-1 equal %d / %d" % (kind, n_big, n_small, n_cmp_steps, n_small, pick[1], worst_max, worst_mean, agree, total))
    assert worst_max < l_max and worst_mean < l_mean
    assert agree >= total - max(1, total // 20)
    # (4) graph replay == host-driven steps, every window, every step
    assert np.array_equal(ids_graph, ids_host), "captured greedy graph and host-driven steps disagree at %s" % (np.argwhere(ids_graph != ids_host)[:4],)
    # within the batch: the same audio gives the same ids whatever the row (and the chunk of the encoder) it sits in
    for w in range(n_small, n_big):
        assert np.array_equal(ids_graph[:, w], ids_graph[:, w % n_small]), w
    checksum = int(ids_graph.T.astype(np.int64)[:7].sum() % 1000003)
    print("    ids of windows 0..6 checksum %d" % checksum)
    cb.close()
    cs.close()
    if own:
        m.close()


@pytest.mark.parametrize("n_big", [112, 224, 448])
def test_timed_batch_parity_chain_medium(n_big, hip_medium):
    """ggml-medium shape: the lock-step batch sizes bench.py can time (112 = two contexts of round 4; 224 / 448 = ONE context, round 5) against the
    11-window context of test_medium_shape_against_the_reference, and the captured graph against the step-by-step decode. Reference analogue:
    the whole-model A/B hooks GpuEncTest / GpuDecTest (Whisper/whisperCom.cpp:929-1088)."""
    _chain("medium", n_big, m=hip_medium)


def test_timed_batch_parity_chain_large_v2():
    """The same at the ggml-large-v2 shape with the batch bench.py's large_v2 object times."""
    _chain("large-v2", 224, n_cmp_steps=6)


@pytest.mark.parametrize("M,N,K", [(40, 4096, 1024), (70, 4096, 1024), (100, 5120, 1280), (112, 3072, 1024), (128, 2048, 512), (33, 4096, 1024)])
def test_wide_decode_products_in_one_row_tile(M, N, K, golden):
    """Option dec_wide_rows (default on): 33 .. 128 rows against N >= 2048 (the MLP up-projection and the QKV product of a decode step) as gemmDecRows with ALL
    rows in one row tile per 32 columns, against gemvFused's 16-column workgroups. Same K split and summation order: the FP32 accumulators are the same bits
    (dec_wide_rows = 2 routes the FP32 epilogue through the same instances to show it). The FP16 GELU outputs agree except where gelu16 -- 9 instructions on
    v_exp_f32 / v_rcp_f32, table-exact to one ulp on < 0.2 % of the inputs -- lands on a rounding midpoint and the two kernels' instruction schedules differ
    in the last bit of the FP32 value (measured: 14-20 of ~300 000 outputs, each one FP16 ulp; tools/diag_wide.py)."""
    g = torch.Generator(device="cuda").manual_seed(M + N)
    a = torch.randn((M, K), generator=g, device="cuda").half()
    w = (0.1 * torch.randn((N, K), generator=g, device="cuda")).half()
    bias = torch.randn(N, generator=g, device="cuda")
    L = binding.lib()
    acc, outs = {}, {}
    default = binding.get_option_default("dec_wide_rows")
    try:
        for mode in (0, 1, 2):
            binding.set_option("dec_wide_rows", mode)
            out = torch.zeros((M, N), dtype=torch.float16, device="cuda")
            binding.check(L.wh_op_mul_mat_gelu(None, ptr(a), ptr(w), ptr(bias), ptr(out), M, N, K))
            o32 = torch.zeros((M, N), dtype=torch.float32, device="cuda")
            binding.check(L.wh_op_mul_mat(None, ptr(a), ptr(w), ptr(bias), None, ptr(o32), M, N, K))
            torch.cuda.synchronize()
            outs[mode], acc[mode] = out, o32
        # round 6: the one-tile path is gemmDecTile (LDS-staged, 4 / 6 / 8 row tiles) by default; dec_lds 0 = gemmDecRows' one-tile instances -- the same bits
        binding.set_option("dec_lds", 0)
        binding.set_option("dec_wide_rows", 2)
        alt16 = torch.zeros((M, N), dtype=torch.float16, device="cuda")
        binding.check(L.wh_op_mul_mat_gelu(None, ptr(a), ptr(w), ptr(bias), ptr(alt16), M, N, K))
        alt32 = torch.zeros((M, N), dtype=torch.float32, device="cuda")
        binding.check(L.wh_op_mul_mat(None, ptr(a), ptr(w), ptr(bias), None, ptr(alt32), M, N, K))
        torch.cuda.synchronize()
    finally:
        binding.set_option("dec_wide_rows", default)
        binding.set_option("dec_lds", binding.get_option_default("dec_lds"))
    assert torch.equal(acc[2], alt32) and torch.equal(outs[2], alt16), "gemmDecTile's one-tile instances differ from gemmDecRows'"
    assert torch.equal(acc[0], acc[2]), "FP32 accumulators of the one-tile instances differ from gemvFused's"
    table = torch.from_numpy(golden["table_gelu"].astype(np.int32)).cuda()
    want = table[(acc[0].half().view(torch.int16).to(torch.int32) & 0xFFFF).long()].to(torch.int16).view(torch.float16).float()
    for mode in (0, 1):
        d = (outs[mode].float() - want).abs()
        assert float((d > 0).float().mean()) < 0.002 and bool((d <= torch.maximum(torch.tensor(4e-3, device="cuda"), want.abs() * 2.0 ** -10)).all())
    differ = int((outs[0] != outs[1]).sum())
    print("GELU %dx%dx%d: one-tile vs gemvFused differ at %d of %d outputs (each one FP16 ulp)" % (M, N, K, differ, M * N))
    assert differ < 2e-4 * M * N and torch.equal(outs[1], outs[2])
    dd = (outs[0].float() - outs[1].float()).abs()
    assert bool((dd <= outs[0].float().abs() * 2.0 ** -9 + 1e-7).all())


@pytest.mark.parametrize("M,N,K", [(40, 1024, 4096), (70, 1024, 4096), (100, 1280, 5120), (112, 1024, 4096), (128, 1024, 2048), (33, 2048, 4096), (70, 1000, 4096)])
def test_deep_decode_products_all_rows_per_workgroup(M, N, K):
    """Option dec_deep_rows: 33 .. 128 rows against N <= 2048, K >= 2048 (the MLP down-projection of a decode step) as gemmDecRows with all rows per 16-column
    workgroup and EIGHT waves over K -- gemvFused's own split for this product (TUNE_GEMV_K8) -- against gemvFused: FP32 + bias + residual, the same bits
    (N = 1000 is not a multiple of 16: stays on gemvFused in both settings)."""
    rng = np.random.default_rng(M * 3 + N)
    a = rng.standard_normal((M, K)).astype(np.float16)
    w = (0.05 * rng.standard_normal((N, K))).astype(np.float16)
    bias = rng.standard_normal(N).astype(np.float32)
    res = rng.standard_normal((M, N)).astype(np.float32)
    want = (a.astype(np.float64) @ w.astype(np.float64).T + bias + res).astype(np.float32)
    ad, wd, bd, rd = dev(a), dev(w), dev(bias), dev(res)
    L = binding.lib()
    outs = {}
    default = binding.get_option_default("dec_deep_rows")
    try:
        binding.set_option("dec_split", 0)         # (round 6's default route for this shape; its own test follows)
        for on in (0, 1):
            binding.set_option("dec_deep_rows", on)
            out = torch.full((M, N), float("nan"), dtype=torch.float32, device="cuda")
            binding.check(L.wh_op_mul_mat(None, ptr(ad), ptr(wd), ptr(bd), ptr(rd), ptr(out), M, N, K))
            torch.cuda.synchronize()
            outs[on] = out
    finally:
        binding.set_option("dec_deep_rows", default)
        binding.set_option("dec_split", binding.get_option_default("dec_split"))
    d = report("deep decode product %dx%dx%d" % (M, N, K), outs[1].cpu().numpy(), want)
    assert d.max() < 2e-5 * max(1.0, np.sqrt(K / 128))
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("M,N,K", [(40, 1280, 5120), (33, 1024, 4096), (64, 1024, 4096), (70, 1280, 5120), (96, 1024, 4096), (100, 1280, 5120), (128, 1280, 5120), (128, 1024, 2048),
                                   (47, 2048, 4096), (70, 1000, 4096), (40, 1024, 2304)])
def test_deep_decode_products_k_split_over_workgroups(M, N, K):
    """Option dec_split (round 6, default): the MLP down-projection of 33 .. 128 rows as gemmDecTile<SPLIT = 8> -- the eight K shares of gemvFused's eight waves on
    eight workgroups per 32 columns, partial tiles in the context's scratch -- plus decSplitCombine adding them in wave order: FP32 + bias + residual, the SAME BITS as
    gemvFused (dec_split 0), also with the residual updated in place as the decoder does, and without bias / residual. N = 1000 and K = 2304 are not covered by the split
    kernel (N % 32, K % 512): both settings run gemvFused."""
    rng = np.random.default_rng(M * 5 + N)
    a = rng.standard_normal((M, K)).astype(np.float16)
    w = (0.05 * rng.standard_normal((N, K))).astype(np.float16)
    bias = rng.standard_normal(N).astype(np.float32)
    res = rng.standard_normal((M, N)).astype(np.float32)
    want = (a.astype(np.float64) @ w.astype(np.float64).T + bias + res).astype(np.float32)
    ad, wd, bd, rd = dev(a), dev(w), dev(bias), dev(res)
    L = binding.lib()
    outs, inplace, bare = {}, {}, {}
    try:
        for on in (0, 1):
            binding.set_option("dec_split", on)
            out = torch.full((M, N), float("nan"), dtype=torch.float32, device="cuda")
            binding.check(L.wh_op_mul_mat(None, ptr(ad), ptr(wd), ptr(bd), ptr(rd), ptr(out), M, N, K))
            x = rd.clone()
            binding.check(L.wh_op_mul_mat(None, ptr(ad), ptr(wd), ptr(bd), ptr(x), ptr(x), M, N, K))
            y = torch.full((M, N), float("nan"), dtype=torch.float32, device="cuda")
            binding.check(L.wh_op_mul_mat(None, ptr(ad), ptr(wd), None, None, ptr(y), M, N, K))
            torch.cuda.synchronize()
            outs[on], inplace[on], bare[on] = out, x, y
    finally:
        binding.set_option("dec_split", binding.get_option_default("dec_split"))
    d = report("K-split decode product %dx%dx%d" % (M, N, K), outs[1].cpu().numpy(), want)
    assert d.max() < 2e-5 * max(1.0, np.sqrt(K / 128))
    assert torch.equal(outs[0], outs[1]), "the K-split product differs from gemvFused's eight-wave sums"
    assert torch.equal(inplace[0], inplace[1]) and torch.equal(inplace[1], outs[1])
    assert torch.equal(bare[0], bare[1])


def test_wide_qkv_product_appends_the_same_cache_rows(hip_medium):
    """The QKV product of a single-token step through the one-tile instances (EPI_QKV_DEC: scaled query out, K and V rows appended to the self-attention
    cache): at 70 and 112 windows per context the rows decoder layer 0 appends on the first single-token step are bit-identical to gemvFused's (no GELU
    upstream of them: LayerNorm of the embedding, then the product)."""
    import bench
    hp = hip_medium.hp
    sp = gf.special_tokens(hp)
    prompt = [sp["sot"], sp["sot"] + 1, sp["transcribe"]]
    default = binding.get_option_default("dec_wide_rows")
    for n_win in (70, 112):
        rows = {}
        for on in (0, 1):
            binding.set_option("dec_wide_rows", on)
            try:
                ctx = binding.HipContext(hip_medium, n_win)
                pcm_dev = torch.from_numpy(bench.synth_pcm(7, seed=100)).cuda()
                mels = _mels(ctx, pcm_dev, range(7))
                ctx.encode(mels[torch.arange(n_win, device="cuda") % 7].contiguous())
                ctx.decode(np.tile(np.asarray(prompt, np.int32), (n_win, 1)), 0, want_logits=False, want_probs=False)
                ctx.decode(np.full((n_win, 1), 1234, np.int32), 3, want_logits=False, want_probs=False)
                rows[on] = (ctx.debug_read("self-k", 0, 4), ctx.debug_read("self-v", 0, 4))
                ctx.close()
            finally:
                binding.set_option("dec_wide_rows", default)
        assert np.isfinite(rows[1][0]).all() and float(np.abs(rows[1][0][:, 3]).max()) > 0.01
        assert np.array_equal(rows[0][0], rows[1][0]) and np.array_equal(rows[0][1], rows[1][1]), n_win


def test_parity_mode_at_medium_shape_vs_one_thread(ref_lib_available, tmp_path):
    """north_star's 1e-3 at the measured shape where the reference IS a point: its decoder at ONE thread (FP16 P.V accumulated key by key in one
    partition, ggml.c:4689-4735) against WH_FLAG_PARITY_PV with one emulated thread (and the reference's fp16(e / sum) operand in the encoder).
    The encoder is thread-count invariant (bit-identical at 1 and 16 threads), so it runs at 16. Reports max / mean over the prompt and
    6 teacher-forced steps, and the stage probes (conv front end, first encoder attention, encoder output, first decoder self- and cross-attention)
    so that what exceeds 1e-3 is attributable."""
    if not ref_lib_available:
        pytest.skip("oracle/_ref/libwhisper_ref.so not present")
    from oracle import ref, whisper_np as wn
    import bench
    kind = "medium"
    model = gf.synth_model(kind, seed=1)
    hp = model.hparams
    sp = gf.special_tokens(hp)
    path = str(tmp_path / (kind + ".bin"))
    gf.write_model(path, model)
    m = binding.HipModel.from_ggml(model)
    del model
    ctx = binding.HipContext(m, 1)
    ctx.set_flags(binding.WH_FLAG_DEBUG_CAPTURE | binding.WH_FLAG_PARITY_PV, 1)
    pcm = bench.synth_pcm(1, seed=100)
    mel = ctx.mel_spectrogram(torch.from_numpy(pcm[0]).cuda())
    ctx.encode(mel)
    w = ref.RefWhisper(path, n_threads=16, log_level=0)
    w.set_mel(mel.cpu().numpy())
    w.trace(True)
    w.encode(0)
    tr = w.traced()
    w.trace(False)
    stage = {}
    got = ctx.debug_read("enc-KQV")[0]
    want = tr["enc-KQV"].astype(np.float32).transpose(1, 0, 2).reshape(got.shape)
    d = report("enc-KQV (encoder layer 0 attention) vs reference", got, wn.r16(want))
    stage["enc-KQV"] = (float(d.max()), float(d.mean()))
    got = ctx.debug_read("encode-out")[0]
    want = tr["encode-out"].astype(np.float32).reshape(got.shape[::-1]).T if tr["encode-out"].shape != got.shape else tr["encode-out"]
    d = report("encode-out (ln_post of the encoder) vs reference", got, wn.r16(want))
    stage["encode-out"] = (float(d.max()), float(d.mean()))
    for il in (0, hp.n_text_layer - 1):
        k, v = w.cross_kv(il)
        dk = report("cross-k[%d] vs reference" % il, ctx.debug_read("cross-k", il)[0], k)
        dv = report("cross-v[%d] vs reference" % il, ctx.debug_read("cross-v", il)[0], v)
        stage["cross-k[%d]" % il] = (float(dk.max()), float(dk.mean()))
        stage["cross-v[%d]" % il] = (float(dv.max()), float(dv.mean()))
    w.n_threads = 1
    prompt = [sp["sot"], sp["sot"] + 1, sp["transcribe"]]
    toks = np.array([prompt], np.int32)
    n_past = 0
    worst_max = worst_mean = 0.0
    agree = 0
    n_steps = 7
    for step in range(n_steps):
        w.trace(True)
        rl, _ = w.decode([int(t) for t in toks[0]], n_past)
        trd = w.traced()
        w.trace(False)
        rl = rl[-1]
        gl, _ = ctx.decode(toks, n_past)
        if step == 0:
            for nm in ("dec-KQV", "dec-KQV#2"):
                got = ctx.debug_read(nm, rows=len(prompt))
                want = trd[nm].astype(np.float32).transpose(1, 0, 2).reshape(got.shape)
                d = report("%s (decoder layer 0) vs reference, 1 thread" % nm, got, wn.r16(want))
                stage[nm] = (float(d.max()), float(d.mean()))
        d = report("parity mode, medium shape, step %d vs reference at 1 thread" % step, gl[0], rl)
        worst_max, worst_mean = max(worst_max, float(d.max())), max(worst_mean, float(d.mean()))
        agree += int(np.argmax(gl[0]) == np.argmax(rl))
        n_past += toks.shape[1]
        nxt = ctx.sample_best(1, step == 0, step == 0)
        toks = np.array([[nxt[0]["id"]]], np.int32)
    print("PARITY MODE at the medium shape vs the reference at 1 thread: logits max %.3e mean %.3e over %d steps, top-1 %d / %d; stages: %s"
          % (worst_max, worst_mean, n_steps, agree, n_steps, {k: "%.1e / %.1e" % v for k, v in stage.items()}))
    # bounds: 2x the values measured on MI355X in round 5 (see DESIGN.md section 2 for which op carries the excess over 1e-3)
    assert worst_max < 1.2e-2 and worst_mean < 2e-3
    assert agree >= n_steps - 1
    w.close()
    ctx.close()
    m.close()
