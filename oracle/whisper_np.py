"""numpy restatement of the reference's Whisper CPU path -- TEST INFRASTRUCTURE ONLY (the checker, never the product).

Every function restates, with the same rounding points, what the reference's CPU implementation computes
(Whisper/source/whisper.cpp + Whisper/source/ggml.c, the code behind eModelImplementation::Reference).  It is
PINNED in tests/test_oracle.py against (a) oracle/_ref (the reference sources compiled unmodified, run live when
present) and (b) the fixtures under tests/golden/ that were produced by oracle/_ref (script:
tests/golden/make_golden.py).  The reference tree itself holds no golden vectors for this path.

Numerics that matter (all line numbers in Whisper/source/ggml.c unless noted):
  * FP16 <-> FP32 conversions are IEEE round-to-nearest-even (F16C, :150-160)            -> np.float16 casts
  * weight GEMMs round the activations to FP16 first, FP32 accumulate (:4588-4611, :751-790)
  * GELU and the softmax exponent go through 65536-entry FP16 tables:
        gelu16(x) = fp16( gelu_double( fp32( fp16(x) ) ) ),  exp16(x) = fp16( exp( fp32( fp16(x) ) ) )
    (:1375-1385 build, :1014-1021 and :5069-5080 use)
  * LayerNorm statistics in double, eps 1e-5, no affine (:4098-4156); affine is mul then add (whisper.cpp:1195-1199)
  * encoder attention = ggml_flash_attn_f16 (:5912-6097): S = K.Q (FP16 operands, FP32 acc) * 1/sqrt(D),
    table softmax, P rounded to FP16, O = V.P (FP32 acc)
  * decoder attention (whisper.cpp:1618-1660, :1715-1748): Q rounded to FP16 by the K.Q mul_mat, table softmax
    in FP32, and P.V through the transposed-src0 branch of mul_mat_f16_f32 (:4689-4735, :4615-4644) which
    accumulates in FP16, one key at a time, per thread over a contiguous key range; partials summed in FP32.
    `n_threads` therefore changes the result and is an explicit argument here.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np

F32 = np.float32
F16 = np.float16


def r16(x):
    """FP32 -> FP16 (RNE) -> FP32: the value an F16C round trip leaves (ggml.c:150-160)."""
    return np.asarray(x, F32).astype(F16).astype(F32)


def gelu16(x):
    """ggml_vec_gelu_f32 with GGML_GELU_FP16 (ggml.c:83, :1003-1021): table lookup on fp16(x)."""
    f = r16(x).astype(np.float64)
    y = 0.5 * f * (1.0 + np.tanh(0.79788456080286535587989211986876 * f * (1.0 + 0.044715 * f * f)))
    return y.astype(F32).astype(F16).astype(F32)


def exp16(x):
    """table_exp_f16[fp16(x)] (ggml.c:1382, used at :5069-5080 and :6065-6067)."""
    f = r16(x).astype(np.float64)
    with np.errstate(over="ignore"):
        return np.exp(f).astype(F32).astype(F16).astype(F32)


def softmax_table(s):
    """ggml_compute_forward_soft_max_f32 (ggml.c:5030-5090) over the last axis. -inf entries become 0."""
    s = np.asarray(s, F32)
    m = s.max(axis=-1, keepdims=True)
    with np.errstate(invalid="ignore"):
        val = np.where(np.isneginf(s), F32(0), exp16((s - m).astype(F32)))
    tot = val.astype(np.float64).sum(axis=-1, keepdims=True)
    inv = (1.0 / tot).astype(F32)
    return (val * inv).astype(F32)


def norm(x):
    """ggml_compute_forward_norm_f32 (ggml.c:4098-4156): rows of the last axis, double statistics, eps=1e-5."""
    x = np.asarray(x, F32)
    xd = x.astype(np.float64)
    mean = xd.sum(axis=-1, keepdims=True) / x.shape[-1]
    v = xd - mean
    y = v.astype(F32)
    sum2 = (v * v).sum(axis=-1, keepdims=True)
    scale = (1.0 / np.sqrt(sum2 / x.shape[-1] + np.float64(F32(1e-5)))).astype(F32)
    return (y * scale).astype(F32)


def layer_norm(x, w, b):
    """norm, then w*x + b as two separate FP32 ops (whisper.cpp:1190-1199)."""
    return ((norm(x) * w.astype(F32)).astype(F32) + b.astype(F32)).astype(F32)


def mul_mat_w(w16, x):
    """ggml_mul_mat with an FP16 weight [out][in] and FP32 activations [rows][in] (ggml.c:4588-4611, :4645-4687):
    activations rounded to FP16, products accumulated in FP32. Returns [rows][out]."""
    return (r16(x) @ w16.astype(F32).T).astype(F32)


# ----------------------------------------------------------------------------------------------------------------------
# mel spectrogram
# ----------------------------------------------------------------------------------------------------------------------
def log_mel_raw(pcm, filters, n_fft=400, hop=160, n_len=None):
    """The spectrogram before its normalisation: log10(max(mel power, 1e-10)) per frame, [n_mel][n_len] float32.
    SpectrogramContext::fft / log_mel_spectrogram's per-frame part (melSpectrogram.cpp:318-391, whisper.cpp:2080-2150):
    n_len = n_samples // hop frames (or as given), no centre padding, samples past the end are zero; periodic Hann(400)
    multiplied in float; power spectrum with the reference's fold p[j] += p[400-j], j = 1..199; filterbank with a double sum."""
    pcm = np.asarray(pcm, F32)
    if n_len is None:
        n_len = len(pcm) // hop
    hann = (0.5 * (1.0 - np.cos((2.0 * np.pi * np.arange(n_fft)) / n_fft))).astype(F32)
    padded = np.concatenate([pcm, np.zeros(n_len * hop + n_fft, F32)])
    idx = (np.arange(n_len) * hop)[:, None] + np.arange(n_fft)[None, :]
    frames = (padded[idx] * hann[None, :]).astype(F32)          # the reference multiplies in float
    spec = np.fft.fft(frames.astype(np.float64), axis=1)
    p = (spec.real ** 2 + spec.imag ** 2)
    half = n_fft // 2
    folded = p[:, :half + 1].copy()
    folded[:, 1:half] += p[:, n_fft - 1:half:-1]
    mel = folded @ filters.astype(np.float64).T                 # [n_len][n_mel], double sum as the reference
    mel = np.log10(np.maximum(mel, 1e-10))
    return np.ascontiguousarray(mel.T.astype(F32))              # stored into a float vector before the clamp


def log_mel_spectrogram(pcm, filters, n_fft=400, hop=160):
    """log_mel_spectrogram (whisper.cpp:2060-2180), mirrored by Spectrogram::pcmToMel (Whisper/Whisper/Spectrogram.cpp:64-122):
    log_mel_raw, then clamp to (global max - 8) and (x + 4) / 4. The DFT here is evaluated in float64 (the reference uses a
    float32 recursive FFT, so it carries ~1e-6 relative noise of its own; tests compare with a stated tolerance).
    Returns [n_mel][n_len] float32."""
    return normalize_mel(log_mel_raw(pcm, filters, n_fft, hop))


class MelStreamerNP:
    """MelStreamer::makeBuffer + makeTransposedBuffer (Whisper/Whisper/MelStreamer.cpp:189-245, :125-187), what
    iContext::runStreamed feeds the encoder with: frames [off, off + len) of the stream; frames the reader has no 160-sample chunk
    for are 0 BEFORE normalisation; maximum over the window with a floor of 1e-20; when a request ends at the frame the previous
    one ended at, the previous maximum is re-used; clamp and (x + 4) * 0.25 in FP32. Pinned on outputs of the reference's own
    MelStreamer.cpp / melSpectrogram.cpp compiled unmodified (oracle/Makefile -> _ref/libmelstreamer_ref.so; fixture
    tests/golden/ref_melstreamer.npz; tests/test_oracle.py::test_melstreamer_restatement_pinned_on_the_reference). These are
    MelStreamerSimple's semantics; MelStreamerThread returns zeros for the partial last chunk's frame too, which lies past the
    length runStreamed clamps its requests to."""

    def __init__(self, pcm, filters):
        self.pcm = np.asarray(pcm, F32)
        self.filters = filters
        self.length = len(self.pcm) // 160                 # PcmReader::getLength
        self.n_chunks = (len(self.pcm) + 159) // 160       # readChunk pads the last chunk
        self.last_end, self.last_max = None, F32(0)

    def make_buffer(self, off, length):
        raw = log_mel_raw(self.pcm[off * 160:], self.filters, n_len=length)
        valid = max(0, min(length, self.n_chunks - off))
        raw[:, valid:] = 0.0
        mmax = max(F32(1e-20), raw.max())
        if self.last_end == off + length:
            mmax = self.last_max
        else:
            self.last_end, self.last_max = off + length, mmax
        lo = F32(mmax) - F32(8.0)
        return ((np.maximum(raw, lo) + F32(4.0)) * F32(0.25)).astype(F32)


def normalize_mel(mel):
    """Global (whole-buffer) max-8 clamp and (x+4)/4 (whisper.cpp:2160-2176)."""
    mmax = np.float64(mel.max()) - 8.0
    out = np.maximum(mel.astype(np.float64), mmax)
    return ((out + 4.0) / 4.0).astype(F32)


# ----------------------------------------------------------------------------------------------------------------------
# model container
# ----------------------------------------------------------------------------------------------------------------------
@dataclass
class KVState:
    """Per-stream caches, FP16 values held as float32 arrays (model.memory_* in whisper.cpp:960-996)."""
    cross_k: List[np.ndarray] = field(default_factory=list)      # per decoder layer [n_audio_ctx][d]
    cross_v: List[np.ndarray] = field(default_factory=list)
    self_k: List[np.ndarray] = field(default_factory=list)       # per decoder layer [n_text_ctx][d]
    self_v: List[np.ndarray] = field(default_factory=list)


class WhisperNP:
    def __init__(self, model):
        """model: whisper_amd.ggml_format.GgmlModel (hparams + name->array tensors)."""
        self.hp = model.hparams
        self.t = model.tensors
        self.filters = model.filters
        self.kv = KVState()

    # ---- encoder ----
    def conv_1d(self, w16, x, stride):
        """ggml_conv_1d_1s / _2s, FP16 kernel (ggml.c:5199-5318, :5465-5584): k=3, zero padding 1, input rounded to FP16,
        one FP32 dot product per tap, taps added in FP32. w16: [out][in][3]; x: [in][T] -> [out][T/stride]."""
        ic, T = x.shape
        xp = np.zeros((ic, T + 2), F32)
        xp[:, 1:T + 1] = r16(x)
        w = w16.astype(F32)
        n_out = T // stride
        out = np.zeros((w.shape[0], n_out), F32)
        for k in range(3):
            seg = xp[:, k:k + T:stride][:, :n_out]               # input position stride*t + k - 1
            out = (out + (w[:, :, k] @ seg).astype(F32)).astype(F32)
        return out

    def encode(self, mel, mel_offset=0, n_threads=1, trace: Optional[Dict[str, np.ndarray]] = None):
        """whisper_encode (whisper.cpp:1084-1496). mel: [n_mel][n_len]. Fills the cross-attention caches and returns
        the encoder output [n_ctx][d]."""
        hp, t = self.hp, self.t
        n_ctx, d, H = hp.n_audio_ctx, hp.n_audio_state, hp.n_audio_head
        D = d // H
        inp = np.zeros((hp.n_mels, 2 * n_ctx), F32)
        i0 = min(mel_offset, mel.shape[1])
        i1 = min(mel_offset + 2 * n_ctx, mel.shape[1])
        inp[:, :i1 - i0] = mel[:, i0:i1]

        cur = self.conv_1d(t["encoder.conv1.weight"], inp, 1)
        if trace is not None:
            trace["enc.conv1"] = cur.copy()
        cur = gelu16((cur + t["encoder.conv1.bias"].reshape(-1, 1)).astype(F32))
        if trace is not None:
            trace["enc.temp1"] = cur.copy()
        cur = self.conv_1d(t["encoder.conv2.weight"], cur, 2)
        cur = gelu16((cur + t["encoder.conv2.bias"].reshape(-1, 1)).astype(F32))
        x = (t["encoder.positional_embedding"][:n_ctx] + cur.T).astype(F32)      # [n_ctx][d]

        scale = F32(1.0 / np.sqrt(np.float64(D)))
        for il in range(hp.n_audio_layer):
            p = f"encoder.blocks.{il}"
            if trace is not None:
                trace[f"enc.layer[ {il} ].in"] = x.copy()
            cur = layer_norm(x, t[p + ".attn_ln.weight"], t[p + ".attn_ln.bias"])
            q = (mul_mat_w(t[p + ".attn.query.weight"], cur) + t[p + ".attn.query.bias"]).astype(F32)
            k = mul_mat_w(t[p + ".attn.key.weight"], cur)
            v = (mul_mat_w(t[p + ".attn.value.weight"], cur) + t[p + ".attn.value.bias"]).astype(F32)
            if trace is not None and il == 0:
                trace["enc-Qcur-b"], trace["enc-Kcur"], trace["enc-Vcur-b"] = q.copy(), k.copy(), v.copy()
            # ggml_flash_attn_f16 (ggml.c:5912-6097): all three operands copied to FP16 first (whisper.cpp:1242-1264)
            q16 = r16(q).reshape(n_ctx, H, D).transpose(1, 0, 2)
            k16 = r16(k).reshape(n_ctx, H, D).transpose(1, 0, 2)
            v16 = r16(v).reshape(n_ctx, H, D).transpose(1, 0, 2)
            kqv = np.zeros((H, n_ctx, D), F32)
            for h in range(H):
                S = ((q16[h] @ k16[h].T).astype(F32) * scale).astype(F32)
                P = softmax_table(S)
                kqv[h] = (r16(P) @ v16[h]).astype(F32)
            if trace is not None and il == 0:
                trace["enc-KQV"] = kqv.copy()
            cur = kqv.transpose(1, 0, 2).reshape(n_ctx, d)
            cur = (mul_mat_w(t[p + ".attn.out.weight"], cur) + t[p + ".attn.out.bias"]).astype(F32)
            x = (cur + x).astype(F32)
            cur = layer_norm(x, t[p + ".mlp_ln.weight"], t[p + ".mlp_ln.bias"])
            cur = gelu16((mul_mat_w(t[p + ".mlp.0.weight"], cur) + t[p + ".mlp.0.bias"]).astype(F32))
            cur = (mul_mat_w(t[p + ".mlp.2.weight"], cur) + t[p + ".mlp.2.bias"]).astype(F32)
            x = (cur + x).astype(F32)
        if trace is not None:
            trace["enc.layers"] = x.copy()
        out = layer_norm(x, t["encoder.ln_post.weight"], t["encoder.ln_post.bias"])
        if trace is not None:
            trace["encode-out"] = out.copy()

        # cross-attention caches (whisper.cpp:1448-1487): K scaled by (d/H)^-0.25, both stored FP16
        kscale = F32(np.power(np.float64(F32(d) / F32(H)), -0.25))
        self.kv.cross_k, self.kv.cross_v = [], []
        for il in range(hp.n_text_layer):
            p = f"decoder.blocks.{il}.cross_attn"
            kc = (mul_mat_w(t[p + ".key.weight"], out) * kscale).astype(F32)
            vc = (mul_mat_w(t[p + ".value.weight"], out) + t[p + ".value.bias"]).astype(F32)
            self.kv.cross_k.append(r16(kc))
            self.kv.cross_v.append(r16(vc))
        self._reset_self_kv()
        return out

    def _reset_self_kv(self):
        hp = self.hp
        if not self.kv.self_k:
            self.kv.self_k = [np.zeros((hp.n_text_ctx, hp.n_text_state), F32) for _ in range(hp.n_text_layer)]
            self.kv.self_v = [np.zeros((hp.n_text_ctx, hp.n_text_state), F32) for _ in range(hp.n_text_layer)]

    # ---- decoder ----
    @staticmethod
    def pv_f16_accumulate(P, V16, n_threads):
        """The transposed-src0 branch of ggml_compute_forward_mul_mat_f16_f32 (ggml.c:4689-4735 + FINALIZE :4615-4644).

        P: [N][keys] FP32 probabilities (not rounded); V16: [keys][D] FP16 values as float32.
        Thread ith owns keys [dc*ith, min(dc*(ith+1), keys)), dc = ceil(keys/n_threads), and accumulates
        y = fp16( fma( fp32(v), p, fp32(y) ) ) key by key (ggml_vec_mad_f16, :871-891, F16C + FMA build);
        the per-thread FP16 partials are then added in FP32, thread 0 first. Returns [N][D] FP32."""
        N, keys = P.shape
        D = V16.shape[1]
        dc = (keys + n_threads - 1) // n_threads
        total = None
        for ith in range(n_threads):
            y = np.zeros((N, D), F32)
            for kk in range(dc * ith, min(dc * (ith + 1), keys)):
                acc = V16[kk].astype(np.float64)[None, :] * P[:, kk].astype(np.float64)[:, None] + y.astype(np.float64)
                y = acc.astype(F32).astype(F16).astype(F32)
            total = y if total is None else (total + y).astype(F32)
        return total

    def _attention_dec(self, q, Kc, Vc, n_keys, mask_past: Optional[int], n_threads, exact_pv=True):
        """q: [N][d] FP32 already scaled; Kc/Vc: [>=n_keys][d] FP16-valued. Returns [N][d] (KQV_merged)."""
        hp = self.hp
        H = hp.n_text_head
        D = hp.n_text_state // H
        N = q.shape[0]
        out = np.zeros((N, H * D), F32)
        q16 = r16(q)
        for h in range(H):
            sl = slice(h * D, (h + 1) * D)
            S = (q16[:, sl] @ Kc[:n_keys, sl].T).astype(F32)             # mul_mat(K, Q): Q rounded to FP16
            if mask_past is not None:                                     # ggml_diag_mask_inf (ggml.c:4967-5020)
                j = np.arange(n_keys)[None, :]
                i = np.arange(N)[:, None]
                S = np.where(j > mask_past + i, F32(-np.inf), S)
            P = softmax_table(S)
            if exact_pv:
                out[:, sl] = self.pv_f16_accumulate(P, Vc[:n_keys, sl], n_threads)
            else:
                out[:, sl] = (P.astype(np.float64) @ Vc[:n_keys, sl].astype(np.float64)).astype(F32)
        return out

    def decode(self, tokens: Sequence[int], n_past: int, n_threads=1, exact_pv=True, trace=None):
        """whisper_decode (whisper.cpp:1508-1872). Returns (logits, probs), each [N][n_vocab]."""
        hp, t = self.hp, self.t
        d, H = hp.n_text_state, hp.n_text_head
        N = len(tokens)
        M = hp.n_audio_ctx
        self._reset_self_kv()
        tok = np.asarray(tokens, np.int64)
        x = (t["decoder.token_embedding.weight"][tok].astype(F32) +
             t["decoder.positional_embedding"][n_past:n_past + N]).astype(F32)
        if trace is not None:
            trace["dec-rows"] = x.copy()
        s = F32(np.power(np.float64(F32(d) / F32(H)), -0.25))
        for il in range(hp.n_text_layer):
            p = f"decoder.blocks.{il}"
            cur = layer_norm(x, t[p + ".attn_ln.weight"], t[p + ".attn_ln.bias"])
            q = ((mul_mat_w(t[p + ".attn.query.weight"], cur) + t[p + ".attn.query.bias"]).astype(F32) * s).astype(F32)
            k = (mul_mat_w(t[p + ".attn.key.weight"], cur) * s).astype(F32)
            v = (mul_mat_w(t[p + ".attn.value.weight"], cur) + t[p + ".attn.value.bias"]).astype(F32)
            self.kv.self_k[il][n_past:n_past + N] = r16(k)
            self.kv.self_v[il][n_past:n_past + N] = r16(v)
            a = self._attention_dec(q, self.kv.self_k[il], self.kv.self_v[il], n_past + N, n_past, n_threads, exact_pv)
            if trace is not None and il == 0:
                trace["dec-KQV"] = a.copy()
            cur = (mul_mat_w(t[p + ".attn.out.weight"], a) + t[p + ".attn.out.bias"]).astype(F32)
            x = (cur + x).astype(F32)

            cur = layer_norm(x, t[p + ".cross_attn_ln.weight"], t[p + ".cross_attn_ln.bias"])
            q = ((mul_mat_w(t[p + ".cross_attn.query.weight"], cur) + t[p + ".cross_attn.query.bias"]).astype(F32) * s).astype(F32)
            a = self._attention_dec(q, self.kv.cross_k[il], self.kv.cross_v[il], M, None, n_threads, exact_pv)
            if trace is not None and il == 0:
                trace["dec-KQV#2"] = a.copy()
            cur = (mul_mat_w(t[p + ".cross_attn.out.weight"], a) + t[p + ".cross_attn.out.bias"]).astype(F32)
            x = (cur + x).astype(F32)

            cur = layer_norm(x, t[p + ".mlp_ln.weight"], t[p + ".mlp_ln.bias"])
            cur = gelu16((mul_mat_w(t[p + ".mlp.0.weight"], cur) + t[p + ".mlp.0.bias"]).astype(F32))
            cur = (mul_mat_w(t[p + ".mlp.2.weight"], cur) + t[p + ".mlp.2.bias"]).astype(F32)
            x = (cur + x).astype(F32)
        cur = layer_norm(x, t["decoder.ln.weight"], t["decoder.ln.bias"])
        logits = mul_mat_w(t["decoder.token_embedding.weight"], cur)
        probs = softmax_table(logits)
        return logits, probs


# ----------------------------------------------------------------------------------------------------------------------
# "truth": the same graph in float64 with NO intermediate rounding
# ----------------------------------------------------------------------------------------------------------------------
class WhisperTruth:
    """What the reference's graph computes in exact arithmetic, as close as float64 gets: the stored weights (FP16 / FP32
    values as they are in the file) but no FP16 rounding of activations, no GELU / exp tables, no FP16 K/V caches, no FP16
    accumulation. It restates whisper_encode / whisper_decode (whisper.cpp:1084-1496, :1508-1872) op for op; it is NOT the
    reference's numerics -- it is the yardstick both the reference (any thread count) and the HIP path are measured
    against: an implementation is "as good as the reference" when |impl - truth| <= |reference - truth|."""

    def __init__(self, model):
        self.hp = model.hparams
        self.t = {k: np.asarray(v, np.float64) for k, v in model.tensors.items()}
        self.cross_k, self.cross_v, self.self_k, self.self_v = [], [], [], []

    @staticmethod
    def _gelu(x):
        return 0.5 * x * (1.0 + np.tanh(0.79788456080286535587989211986876 * x * (1.0 + 0.044715 * x * x)))

    @staticmethod
    def _ln(x, w, b):
        mean = x.mean(axis=-1, keepdims=True)
        v = x - mean
        return v / np.sqrt((v * v).mean(axis=-1, keepdims=True) + np.float64(F32(1e-5))) * w + b

    @staticmethod
    def _softmax(s):
        m = s.max(axis=-1, keepdims=True)
        with np.errstate(invalid="ignore"):
            e = np.where(np.isneginf(s), 0.0, np.exp(s - m))
        return e / e.sum(axis=-1, keepdims=True)

    def _conv(self, w, x, stride):
        ic, T = x.shape
        xp = np.zeros((ic, T + 2))
        xp[:, 1:T + 1] = x
        n_out = T // stride
        out = np.zeros((w.shape[0], n_out))
        for k in range(3):
            out += w[:, :, k] @ xp[:, k:k + T:stride][:, :n_out]
        return out

    def encode(self, mel, mel_offset=0):
        hp, t = self.hp, self.t
        n_ctx, d, H = hp.n_audio_ctx, hp.n_audio_state, hp.n_audio_head
        D = d // H
        inp = np.zeros((hp.n_mels, 2 * n_ctx))
        i0, i1 = min(mel_offset, mel.shape[1]), min(mel_offset + 2 * n_ctx, mel.shape[1])
        inp[:, :i1 - i0] = mel[:, i0:i1]
        cur = self._gelu(self._conv(t["encoder.conv1.weight"], inp, 1) + t["encoder.conv1.bias"].reshape(-1, 1))
        cur = self._gelu(self._conv(t["encoder.conv2.weight"], cur, 2) + t["encoder.conv2.bias"].reshape(-1, 1))
        x = t["encoder.positional_embedding"][:n_ctx] + cur.T
        for il in range(hp.n_audio_layer):
            p = f"encoder.blocks.{il}"
            cur = self._ln(x, t[p + ".attn_ln.weight"], t[p + ".attn_ln.bias"])
            q = (cur @ t[p + ".attn.query.weight"].T + t[p + ".attn.query.bias"]).reshape(n_ctx, H, D).transpose(1, 0, 2)
            k = (cur @ t[p + ".attn.key.weight"].T).reshape(n_ctx, H, D).transpose(1, 0, 2)
            v = (cur @ t[p + ".attn.value.weight"].T + t[p + ".attn.value.bias"]).reshape(n_ctx, H, D).transpose(1, 0, 2)
            kqv = np.stack([self._softmax((q[h] @ k[h].T) / np.sqrt(np.float64(D))) @ v[h] for h in range(H)])
            cur = kqv.transpose(1, 0, 2).reshape(n_ctx, d)
            x = x + cur @ t[p + ".attn.out.weight"].T + t[p + ".attn.out.bias"]
            cur = self._ln(x, t[p + ".mlp_ln.weight"], t[p + ".mlp_ln.bias"])
            cur = self._gelu(cur @ t[p + ".mlp.0.weight"].T + t[p + ".mlp.0.bias"])
            x = x + cur @ t[p + ".mlp.2.weight"].T + t[p + ".mlp.2.bias"]
        out = self._ln(x, t["encoder.ln_post.weight"], t["encoder.ln_post.bias"])
        ks = np.power(np.float64(d) / np.float64(H), -0.25)
        self.cross_k, self.cross_v = [], []
        for il in range(hp.n_text_layer):
            p = f"decoder.blocks.{il}.cross_attn"
            self.cross_k.append(out @ t[p + ".key.weight"].T * ks)
            self.cross_v.append(out @ t[p + ".value.weight"].T + t[p + ".value.bias"])
        self.self_k = [np.zeros((hp.n_text_ctx, d)) for _ in range(hp.n_text_layer)]
        self.self_v = [np.zeros((hp.n_text_ctx, d)) for _ in range(hp.n_text_layer)]
        return out

    def _attn(self, q, K, V, n_keys, mask_past):
        H = self.hp.n_text_head
        D = self.hp.n_text_state // H
        N = q.shape[0]
        out = np.zeros((N, H * D))
        for h in range(H):
            sl = slice(h * D, (h + 1) * D)
            S = q[:, sl] @ K[:n_keys, sl].T
            if mask_past is not None:
                S = np.where(np.arange(n_keys)[None, :] > mask_past + np.arange(N)[:, None], -np.inf, S)
            out[:, sl] = self._softmax(S) @ V[:n_keys, sl]
        return out

    def decode(self, tokens, n_past):
        """Returns the logits [N][n_vocab] in float64."""
        hp, t = self.hp, self.t
        d, H = hp.n_text_state, hp.n_text_head
        N = len(tokens)
        x = t["decoder.token_embedding.weight"][np.asarray(tokens, np.int64)] + t["decoder.positional_embedding"][n_past:n_past + N]
        s = np.power(np.float64(d) / np.float64(H), -0.25)
        for il in range(hp.n_text_layer):
            p = f"decoder.blocks.{il}"
            cur = self._ln(x, t[p + ".attn_ln.weight"], t[p + ".attn_ln.bias"])
            q = (cur @ t[p + ".attn.query.weight"].T + t[p + ".attn.query.bias"]) * s
            self.self_k[il][n_past:n_past + N] = cur @ t[p + ".attn.key.weight"].T * s
            self.self_v[il][n_past:n_past + N] = cur @ t[p + ".attn.value.weight"].T + t[p + ".attn.value.bias"]
            a = self._attn(q, self.self_k[il], self.self_v[il], n_past + N, n_past)
            x = x + a @ t[p + ".attn.out.weight"].T + t[p + ".attn.out.bias"]
            cur = self._ln(x, t[p + ".cross_attn_ln.weight"], t[p + ".cross_attn_ln.bias"])
            q = (cur @ t[p + ".cross_attn.query.weight"].T + t[p + ".cross_attn.query.bias"]) * s
            a = self._attn(q, self.cross_k[il], self.cross_v[il], hp.n_audio_ctx, None)
            x = x + a @ t[p + ".cross_attn.out.weight"].T + t[p + ".cross_attn.out.bias"]
            cur = self._ln(x, t[p + ".mlp_ln.weight"], t[p + ".mlp_ln.bias"])
            cur = self._gelu(cur @ t[p + ".mlp.0.weight"].T + t[p + ".mlp.0.bias"])
            x = x + cur @ t[p + ".mlp.2.weight"].T + t[p + ".mlp.2.bias"]
        cur = self._ln(x, t["decoder.ln.weight"], t["decoder.ln.bias"])
        return cur @ t["decoder.token_embedding.weight"].T


# ----------------------------------------------------------------------------------------------------------------------
# sampling (host logic; restates ContextImpl::sampleBest, Whisper/Whisper/ContextImpl.cpp:71-157 == whisper.cpp:1875-1960)
# ----------------------------------------------------------------------------------------------------------------------
def cross_attention_split(q, K, V, n_splits=8):
    """Restatement of the single-stream cross-attention of whisper_amd/csrc/decode1.hip (crossScores -> crossSoftmaxPV -> the
    combine prologue of gemvSmall) for ONE head: q [64] FP32 (already scaled and rounded to FP16), K, V [n_keys][64] FP16-valued.
    The keys are cut into n_splits ranges (the kernel's rounding of the range length to a multiple of 4); every range computes
    its scores and its maximum, the exponentials use the maximum over ALL ranges -- which is what makes the result the
    reference's table softmax (softmax_table, ggml.c:5030-5090) and not an online softmax with local maxima -- and each range
    contributes sum(e) in double and sum(e * V) in FP32; the ranges are added in order and scaled by float(1 / sum).
    Returns (out [64] FP32, e [n_keys] FP32): out differs from  softmax_table(s) @ V  by FP32 summation order only, e is
    bit-identical to the reference's unnormalised exponentials."""
    q = np.asarray(q, F32)
    K = np.asarray(K, F32)
    V = np.asarray(V, F32)
    n_keys = K.shape[0]
    per = ((n_keys + n_splits - 1) // n_splits + 3) & ~3
    s = (K @ q).astype(F32)
    ranges = [(i * per, min((i + 1) * per, n_keys)) for i in range(n_splits)]
    maxima = [s[a:b].max() if b > a else F32(-np.inf) for a, b in ranges]
    gmax = F32(max(maxima))
    e = exp16((s - gmax).astype(F32))
    acc = np.zeros(V.shape[1], F32)
    tot = 0.0
    for a, b in ranges:                                   # combine: the ranges in order
        if b > a:
            acc = (acc + (e[a:b, None] * V[a:b]).astype(F32).sum(axis=0, dtype=F32)).astype(F32)
            tot += float(e[a:b].astype(np.float64).sum())
    inv = F32(1.0 / tot)
    return (acc * inv).astype(F32), e


def sample_best(probs, token_beg, token_sot, token_solm, token_not, force_timestamp=False, is_initial=False):
    """probs: [n_vocab] of the last row. Returns dict(id, tid, p, pt, ptsum).

    Text-vs-timestamp rule: if the probability mass of the timestamp tokens exceeds the best text token (or a
    timestamp is forced) every text token is masked to -inf; when `is_initial`, timestamps above beg+100 (1.00 s)
    are masked and only beg..beg+100 are summed. Then the best of the top-4 that is not sot / solm / not wins.
    (std::partial_sort compares probabilities only, so exact ties are resolved arbitrarily by the reference.)"""
    p = np.asarray(probs, np.float64).copy()
    n = len(p)
    max_tx = max(-1.0, p[:token_beg].max())
    i1 = token_beg + 101 if is_initial else n
    if is_initial:
        p[token_beg + 101:] = -np.inf
    ts = p[token_beg:i1]
    sum_ts = float(ts.sum())
    max_ts, tid = -1.0, 0
    if len(ts) and ts.max() > -1.0:
        max_ts = float(ts.max())
        tid = token_beg + int(np.argmax(ts))
    if sum_ts > max_tx or force_timestamp:
        p[:token_beg] = -np.inf
    pt = np.float32(max_ts / (sum_ts + 1e-10))
    order = np.argsort(-p, kind="stable")[:4]
    res = 0
    while order[res] in (token_sot, token_solm, token_not) and res < 3:
        res += 1
    i = int(order[res])
    return dict(id=i, tid=tid, p=float(np.float32(p[i])), pt=float(pt), ptsum=float(np.float32(sum_ts)))
