"""ggml `.bin` model files: reader, writer and synthetic-model generator (host-side tooling).

File layout follows the two loaders of the reference, which both consume the same bytes:
  * GPU model   Whisper/Whisper/WhisperModel.cpp:434-492 (header), :257-340 (tensors), Vocabulary.cpp:64-143
  * CPU model   Whisper/source/whisper.cpp:451-1075
(SURVEY.md appendix A).  Little-endian, unaligned:

    u32 magic 0x67676d6c | 11 x i32 hparams | i32 n_mel, i32 n_fft, f32 filters[n_mel][n_fft]
    i32 n_words, n_words x { u32 len, bytes } | repeat { i32 n_dims, i32 name_len, i32 ftype, i32 ne[n_dims],
    name, payload }  with ne[0] the contiguous dimension and ftype 0 = f32, otherwise f16.

numpy arrays are C-ordered, so a ggml tensor with ne = [ne0, ne1, ne2] is a numpy array of shape (ne2, ne1, ne0).
No real weights exist on this machine (no network), so `synth_model` builds random-weight models of the exact
real shapes for perf work and small ones for parity tests.
"""
from __future__ import annotations

import struct
from dataclasses import dataclass, field, asdict
from typing import Dict, List, Optional

import numpy as np

GGML_MAGIC = 0x67676D6C

HPARAM_FIELDS = ("n_vocab", "n_audio_ctx", "n_audio_state", "n_audio_head", "n_audio_layer",
                 "n_text_ctx", "n_text_state", "n_text_head", "n_text_layer", "n_mels", "f16")


@dataclass
class HParams:
    n_vocab: int = 51864
    n_audio_ctx: int = 1500
    n_audio_state: int = 384
    n_audio_head: int = 6
    n_audio_layer: int = 4
    n_text_ctx: int = 448
    n_text_state: int = 384
    n_text_head: int = 6
    n_text_layer: int = 4
    n_mels: int = 80
    f16: int = 1

    def as_list(self) -> List[int]:
        return [int(getattr(self, k)) for k in HPARAM_FIELDS]

    @property
    def is_multilingual(self) -> bool:
        return self.n_vocab >= 51865


# name -> (n_vocab, d, heads, layers); encoder and decoder widths/depths are equal for every released size.
MODEL_SHAPES = {
    "tiny.en": (51864, 384, 6, 4),
    "tiny": (51865, 384, 6, 4),
    "base": (51865, 512, 8, 6),
    "small": (51865, 768, 12, 12),
    "medium": (51865, 1024, 16, 24),
    "large-v2": (51865, 1280, 20, 32),
    "large": (51865, 1280, 20, 32),
    # large-v3 shape: 128 mel bins and one more language token. NOT loadable by the reference (N_MEL is a constexpr 80,
    # Whisper/Whisper/audioConstants.h:13; special ids keyed on 51865, Vocabulary.h:38-41): an extension without an oracle.
    "large-v3": (51866, 1280, 20, 32),
    # test-size models; layer counts must be one of {4,6,12,24,32} for the CPU reference (whisper.cpp:491-509)
    "test-d128": (51864, 128, 2, 4),
    "test-d128-ml": (51865, 128, 2, 4),
    "test-d192": (51865, 192, 3, 4),
    "test-d128-v3": (51866, 128, 2, 4),
}
N_MELS = {"large-v3": 128, "test-d128-v3": 128}


def hparams_for(kind: str, n_audio_ctx: int = 1500, n_text_ctx: int = 448) -> HParams:
    v, d, h, l = MODEL_SHAPES[kind]
    return HParams(n_vocab=v, n_audio_ctx=n_audio_ctx, n_audio_state=d, n_audio_head=h, n_audio_layer=l,
                   n_text_ctx=n_text_ctx, n_text_state=d, n_text_head=h, n_text_layer=l, n_mels=N_MELS.get(kind, 80), f16=1)


# ----------------------------------------------------------------------------------------------------------------------
# mel filterbank (the real files carry librosa's slaney-normalised 80x201 bank; rebuilt here from its definition)
# ----------------------------------------------------------------------------------------------------------------------
def _hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep, mels)


def _mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_filterbank(n_mels: int = 80, n_fft: int = 400, sr: int = 16000) -> np.ndarray:
    """Slaney-style triangular filters, shape [n_mels][n_fft/2+1] float32 (row j = mel bin j over rFFT bins)."""
    n_bins = n_fft // 2 + 1
    fft_freqs = np.linspace(0, sr / 2, n_bins)
    mel_pts = _mel_to_hz(np.linspace(_hz_to_mel(0.0), _hz_to_mel(sr / 2), n_mels + 2))
    fdiff = np.diff(mel_pts)
    ramps = mel_pts[:, None] - fft_freqs[None, :]
    lower = -ramps[:-2] / fdiff[:-1, None]
    upper = ramps[2:] / fdiff[1:, None]
    w = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_pts[2:n_mels + 2] - mel_pts[:n_mels])
    return (w * enorm[:, None]).astype(np.float32)


# ----------------------------------------------------------------------------------------------------------------------
# tensor inventory (names and ggml shapes: Whisper/Whisper/WhisperModel.cpp:63-162, whisper.cpp:774-940)
# ----------------------------------------------------------------------------------------------------------------------
def tensor_specs(hp: HParams):
    """Yield (name, numpy_shape, is_f16) for every tensor a loader expects, in the reference's creation order."""
    d, v, m = hp.n_audio_state, hp.n_vocab, hp.n_mels
    out = []
    out.append(("encoder.positional_embedding", (hp.n_audio_ctx, d), False))
    out.append(("encoder.conv1.weight", (d, m, 3), True))
    out.append(("encoder.conv1.bias", (d, 1), False))
    out.append(("encoder.conv2.weight", (d, d, 3), True))
    out.append(("encoder.conv2.bias", (d, 1), False))
    out.append(("encoder.ln_post.weight", (d,), False))
    out.append(("encoder.ln_post.bias", (d,), False))

    def attn(prefix):
        return [
            (prefix + ".query.weight", (d, d), True), (prefix + ".query.bias", (d,), False),
            (prefix + ".key.weight", (d, d), True),
            (prefix + ".value.weight", (d, d), True), (prefix + ".value.bias", (d,), False),
            (prefix + ".out.weight", (d, d), True), (prefix + ".out.bias", (d,), False),
        ]

    def mlp(prefix):
        return [
            (prefix + ".mlp_ln.weight", (d,), False), (prefix + ".mlp_ln.bias", (d,), False),
            (prefix + ".mlp.0.weight", (4 * d, d), True), (prefix + ".mlp.0.bias", (4 * d,), False),
            (prefix + ".mlp.2.weight", (d, 4 * d), True), (prefix + ".mlp.2.bias", (d,), False),
        ]

    for i in range(hp.n_audio_layer):
        p = f"encoder.blocks.{i}"
        out += mlp(p)
        out += [(p + ".attn_ln.weight", (d,), False), (p + ".attn_ln.bias", (d,), False)]
        out += attn(p + ".attn")
    out.append(("decoder.positional_embedding", (hp.n_text_ctx, d), False))
    out.append(("decoder.token_embedding.weight", (v, d), True))
    out.append(("decoder.ln.weight", (d,), False))
    out.append(("decoder.ln.bias", (d,), False))
    for i in range(hp.n_text_layer):
        p = f"decoder.blocks.{i}"
        out += mlp(p)
        out += [(p + ".attn_ln.weight", (d,), False), (p + ".attn_ln.bias", (d,), False)]
        out += attn(p + ".attn")
        out += [(p + ".cross_attn_ln.weight", (d,), False), (p + ".cross_attn_ln.bias", (d,), False)]
        out += attn(p + ".cross_attn")
    return out


@dataclass
class GgmlModel:
    hparams: HParams
    filters: np.ndarray                     # [n_mel][n_fft] float32
    vocab: List[bytes]                      # the n_words strings stored in the file
    tensors: Dict[str, np.ndarray] = field(default_factory=dict)


def default_vocab_words(hp: HParams) -> List[bytes]:
    """Stand-in vocabulary: the file stores 50257 (multilingual) or 50256 (.en) byte strings (appendix A)."""
    n_words = 50257 if hp.is_multilingual else 50256      # eot is the last stored word
    words = []
    for i in range(n_words):
        if i < 256:
            words.append(bytes([i]) if i not in (0,) else b"")
        else:
            words.append((" w%d" % i).encode())
    return words


def synth_model(kind: str = "test-d128", seed: int = 1234, *, n_audio_ctx: int = 1500, n_text_ctx: int = 448,
                w_std: float = 0.05, attn_sharpness: float = 1.0, hp: Optional[HParams] = None) -> GgmlModel:
    """Random-weight model (N(0, w_std) f16 weights, LN gains N(1, 0.02)).

    attn_sharpness multiplies the query/key projection weights; >1 makes attention peaky, which is how real
    models behave and shrinks the CPU reference's own thread-count noise in the decoder.
    """
    hp = hp or hparams_for(kind, n_audio_ctx, n_text_ctx)
    rng = np.random.default_rng(seed)
    tensors: Dict[str, np.ndarray] = {}
    for name, shape, is_f16 in tensor_specs(hp):
        n = int(np.prod(shape))
        if name.endswith("_ln.weight") or name.endswith("ln_post.weight") or name == "decoder.ln.weight":
            a = 1.0 + 0.02 * rng.standard_normal(n, dtype=np.float32)
        elif name.endswith(".bias"):
            a = 0.02 * rng.standard_normal(n, dtype=np.float32)
        elif name.endswith("positional_embedding"):
            a = 0.02 * rng.standard_normal(n, dtype=np.float32)
        else:
            a = w_std * rng.standard_normal(n, dtype=np.float32)
            if (".query.weight" in name or ".key.weight" in name) and attn_sharpness != 1.0:
                a *= attn_sharpness
        a = a.reshape(shape)
        tensors[name] = a.astype(np.float16) if is_f16 else a.astype(np.float32)
    return GgmlModel(hp, mel_filterbank(hp.n_mels), default_vocab_words(hp), tensors)


def write_model(path: str, model: GgmlModel) -> int:
    """Serialise `model`; returns the byte count."""
    hp = model.hparams
    with open(path, "wb") as f:
        f.write(struct.pack("<I", GGML_MAGIC))
        f.write(struct.pack("<11i", *hp.as_list()))
        filt = np.ascontiguousarray(model.filters, dtype=np.float32)
        f.write(struct.pack("<2i", filt.shape[0], filt.shape[1]))
        f.write(filt.tobytes())
        f.write(struct.pack("<i", len(model.vocab)))
        buf = bytearray()
        for w in model.vocab:
            buf += struct.pack("<I", len(w)) + w
        f.write(bytes(buf))
        for name, shape, is_f16 in tensor_specs(hp):
            a = model.tensors[name]
            assert tuple(a.shape) == tuple(shape), (name, a.shape, shape)
            a = np.ascontiguousarray(a, dtype=np.float16 if is_f16 else np.float32)
            ne = list(reversed(a.shape))
            nm = name.encode()
            f.write(struct.pack("<3i", len(ne), len(nm), 1 if is_f16 else 0))
            f.write(struct.pack("<%di" % len(ne), *ne))
            f.write(nm)
            f.write(a.tobytes())
        return f.tell()


def read_model(path: str, load_tensors: bool = True) -> GgmlModel:
    with open(path, "rb") as f:
        data = memoryview(f.read())
    off = 0

    def take(fmt):
        nonlocal off
        vals = struct.unpack_from(fmt, data, off)
        off += struct.calcsize(fmt)
        return vals

    (magic,) = take("<I")
    if magic != GGML_MAGIC:
        raise ValueError("bad magic 0x%08x" % magic)
    hp = HParams(*take("<11i"))
    n_mel, n_fft = take("<2i")
    filt = np.frombuffer(data, dtype="<f4", count=n_mel * n_fft, offset=off).reshape(n_mel, n_fft).copy()
    off += 4 * n_mel * n_fft
    (n_words,) = take("<i")
    vocab = []
    for _ in range(n_words):
        (ln,) = take("<I")
        vocab.append(bytes(data[off:off + ln]))
        off += ln
    tensors: Dict[str, np.ndarray] = {}
    while off < len(data):
        n_dims, name_len, ftype = take("<3i")
        ne = take("<%di" % n_dims)
        name = bytes(data[off:off + name_len]).decode()
        off += name_len
        count = int(np.prod(ne))
        dt = "<f4" if ftype == 0 else "<f2"
        if load_tensors:
            tensors[name] = np.frombuffer(data, dtype=dt, count=count, offset=off).reshape(tuple(reversed(ne))).copy()
        off += count * (4 if ftype == 0 else 2)
    return GgmlModel(hp, filt, vocab, tensors)


def special_tokens(hp: HParams) -> Dict[str, int]:
    """Hard-coded ids (Whisper/Whisper/Vocabulary.h:27-41; whisper.cpp:198-221) for 51864 / 51865; for larger
    vocabularies (the large-v3 shape, 51866) every extra language token moves the ids behind the language block up by one."""
    extra = max(0, hp.n_vocab - 51864)
    ml = 1 if extra > 0 else 0
    t = dict(eot=50256 + ml, sot=50257 + ml, prev=50360 + extra, solm=50361 + extra, not_=50362 + extra, beg=50363 + extra)
    t["translate"] = 50358 + max(0, extra - 1)
    t["transcribe"] = 50359 + max(0, extra - 1)
    return t


def scripted_model(script: List[int], prompt_len: int, kind: str = "test-d128-ml", seed: int = 5, gain: float = 0.12) -> GgmlModel:
    """A model whose greedy transcript is known in advance, for testing the HOST loop (windows, stop rules, segments).

    The decoder's attention and MLP output projections are zeroed, so the residual stream stays token + position
    embedding; every position p carries a random +-1 code c_p, and the token scripted for that position gets +gain*c_p
    added to its (tied) embedding row, which makes it win the logits by a wide margin (about 15 vs 4). Token i of a
    window is predicted at position prompt_len - 1 + i. The audio plays no role (cross-attention output is zeroed too),
    so every window of a NoContext run produces the same script, the way a real model repeats on repetitive audio.
    """
    return scripted_model_at({prompt_len - 1 + i: tok for i, tok in enumerate(script)}, kind, seed, gain)


def scripted_model_at(position_tokens: Dict[int, int], kind: str = "test-d128-ml", seed: int = 5, gain: float = 0.12) -> GgmlModel:
    """scripted_model with the script given per decoder position (position -> the token predicted there), which is what a run
    WITH prompt carry-over needs: the prompt, hence the position of a window's first token, grows from window to window."""
    m = synth_model(kind, seed=seed, w_std=0.02)
    hp = m.hparams
    d = hp.n_text_state
    rng = np.random.default_rng(seed + 1)
    codes = rng.choice(np.array([-1.0, 1.0], np.float32), size=(hp.n_text_ctx, d))
    m.tensors["decoder.positional_embedding"] = codes.astype(np.float32)
    te = (0.02 * rng.standard_normal((hp.n_vocab, d))).astype(np.float32)
    for pos, tok in position_tokens.items():
        te[tok] += gain * codes[pos]
    m.tensors["decoder.token_embedding.weight"] = te.astype(np.float16)
    m.tensors["decoder.ln.weight"] = np.ones(d, np.float32)
    m.tensors["decoder.ln.bias"] = np.zeros(d, np.float32)
    for il in range(hp.n_text_layer):
        p = "decoder.blocks.%d" % il
        for nm in (".attn.out", ".cross_attn.out", ".mlp.2"):
            m.tensors[p + nm + ".weight"] = np.zeros_like(m.tensors[p + nm + ".weight"])
            m.tensors[p + nm + ".bias"] = np.zeros_like(m.tensors[p + nm + ".bias"])
    return m


def conditioned_model(layout, prompt_len: int, kind: str = "test-d128-ml", seed: int = 6, gain: float = 0.3, beta: float = 2.0,
                      alpha: float = 10.0, n_cand: int = 4) -> GgmlModel:
    """A model whose transcript has a known STRUCTURE but whose tokens and timestamps depend on the audio, for end-to-end runs of
    the host loop in which numerics matter (scripted_model's tokens win by ~15 vs 4 logits whatever the audio).

    layout[i] says what token i of a window is: ("tok", id) one scripted token, ("text", None) one of n_cand text tokens,
    ("ts", (lo, hi)) one of n_cand timestamps beg + linspace(lo, hi). All candidates of a position get the same +gain * c_p on
    their embedding row (c_p = the position's +-1 code, as in scripted_model), so they tie on the scripted term, plus
    beta * u_j / sqrt(d) with u_j random and orthogonal to c_p: the winner is decided by u_j . (previous token + cross-attention
    output). The cross-attention output projections are kept and scaled by alpha (self-attention and MLP outputs stay zero), so
    the winner -- including WHICH timestamp, hence the next window's seek -- depends on the encoder output through the real
    cross-attention arithmetic. tests/golden/make_golden_runfull.py asserts that the reference's transcript on these models does
    not depend on its own thread count (1, 4 and 8 threads identical) before it commits it."""
    m = synth_model(kind, seed=seed, w_std=0.02, attn_sharpness=4.0)
    hp = m.hparams
    d = hp.n_text_state
    sp = special_tokens(hp)
    rng = np.random.default_rng(seed + 1)
    codes = rng.choice(np.array([-1.0, 1.0], np.float32), size=(hp.n_text_ctx, d))
    m.tensors["decoder.positional_embedding"] = codes.astype(np.float32)
    te = (0.02 * rng.standard_normal((hp.n_vocab, d))).astype(np.float32)
    for i, (cls, arg) in enumerate(layout):
        c = codes[prompt_len - 1 + i]
        if cls == "tok":
            cands = [int(arg)]
        elif cls == "text":
            cands = [2000 + 64 * i + j for j in range(n_cand)]
        elif cls == "ts":
            cands = [sp["beg"] + int(v) for v in np.linspace(arg[0], arg[1], n_cand)]
        else:
            raise ValueError(cls)
        for t in cands:
            u = rng.standard_normal(d).astype(np.float32)
            u -= (u @ c) / d * c
            te[t] += gain * c + (beta * u / np.sqrt(d) if len(cands) > 1 else 0.0)
    m.tensors["decoder.token_embedding.weight"] = te.astype(np.float16)
    m.tensors["decoder.ln.weight"] = np.ones(d, np.float32)
    m.tensors["decoder.ln.bias"] = np.zeros(d, np.float32)
    for il in range(hp.n_text_layer):
        p = "decoder.blocks.%d" % il
        for nm in (".attn.out", ".mlp.2"):
            m.tensors[p + nm + ".weight"] = np.zeros_like(m.tensors[p + nm + ".weight"])
            m.tensors[p + nm + ".bias"] = np.zeros_like(m.tensors[p + nm + ".bias"])
        w = m.tensors[p + ".cross_attn.out.weight"].astype(np.float32) * alpha
        m.tensors[p + ".cross_attn.out.weight"] = w.astype(np.float16)
    return m


def conditioned_layout(hp: HParams):
    """[timestamp <= 0.8 s, 5 text, two timestamps 6-10 s, 6 text, a timestamp 14-22 s, EOT]: two segments per window, the next
    window seeks to the last timestamp."""
    sp = special_tokens(hp)
    return ([("ts", (0, 40))] + [("text", None)] * 5 + [("ts", (300, 500)), ("ts", (300, 500))] + [("text", None)] * 6 +
            [("ts", (700, 1100)), ("tok", sp["eot"])])


def carry_over_script(hp: HParams, n_windows: int, text_per_window: int, n_max_text_ctx: int):
    """Positions -> tokens for a sequential run with prompt carry-over in which EVERY window transcribes as
    [timestamp <= 1 s (forced by the sampler), text_per_window text tokens, the 30.00 s timestamp, EOT]:
    window w's prompt is [prev] + the last min(n_max_text_ctx, n_text_ctx/2, past) tokens + the task tokens
    (ContextImpl.cpp:565-576), so its token i sits at position len(prompt) - 1 + i. Returns (positions, tokens kept per window)."""
    sp = special_tokens(hp)
    n_init = 3 if hp.is_multilingual else 1
    kept = 1 + text_per_window + 1            # first timestamp, text, closing timestamp (EOT is dropped by result_len)
    positions: Dict[int, int] = {}
    past = 0
    for w in range(n_windows):
        take = min(n_max_text_ctx, hp.n_text_ctx // 2, past)
        plen = (1 + take if past else 0) + n_init
        for i in range(text_per_window + 3):
            if i == 0 and w > 0:
                continue                      # the sampler forces SOME timestamp <= 1 s there; scripting it would collide with the previous EOT's position
            pos = plen - 1 + i
            # text tokens are keyed by position: once the carried prompt has reached its cap every window sits at the same positions
            t = sp["beg"] if i == 0 else (1000 + pos if i <= text_per_window else (sp["beg"] + 1500 if i == text_per_window + 1 else sp["eot"]))
            assert pos < hp.n_text_ctx and positions.get(pos, t) == t, "script positions collide: change n_max_text_ctx"
            positions[pos] = t
        past += kept
    return positions, kept
