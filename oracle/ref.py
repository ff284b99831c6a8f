"""ctypes wrapper over oracle/_ref/libwhisper_ref.so -- TEST INFRASTRUCTURE ONLY.

libwhisper_ref.so is the reference's own CPU path (Whisper/source/whisper.cpp + ggml.c, compiled unmodified by
oracle/Makefile) plus the flat C harness of oracle/ref_harness.cpp.  Only tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py may import this module; the product path never does.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_ref", "libwhisper_ref.so")

_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")


def available() -> bool:
    return os.path.exists(LIB_PATH)


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(LIB_PATH)
        L.ref_init.restype = C.c_void_p
        L.ref_init.argtypes = [C.c_char_p]
        L.ref_free.argtypes = [C.c_void_p]
        L.ref_hparams.argtypes = [C.c_void_p, _i32p]
        L.ref_pcm_to_mel.argtypes = [C.c_void_p, _f32p, C.c_int, C.c_int]
        L.ref_set_mel.argtypes = [C.c_void_p, _f32p, C.c_int, C.c_int]
        L.ref_mel_len.argtypes = [C.c_void_p]
        L.ref_set_mel_any.argtypes = [C.c_void_p, _f32p, C.c_int, C.c_int]
        L.ref_get_mel.argtypes = [C.c_void_p, _f32p]
        L.ref_encode.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.ref_set_audio_ctx.argtypes = [C.c_void_p, C.c_int]
        L.ref_set_audio_ctx.restype = None
        L.ref_decode.argtypes = [C.c_void_p, _i32p, C.c_int, C.c_int, C.c_int]
        L.ref_logits_size.restype = C.c_size_t
        L.ref_logits_size.argtypes = [C.c_void_p]
        L.ref_get_logits.argtypes = [C.c_void_p, _f32p]
        L.ref_get_probs.argtypes = [C.c_void_p, _f32p]
        L.ref_get_cross_kv.argtypes = [C.c_void_p, C.c_int, _f32p, _f32p]
        L.ref_get_self_kv.argtypes = [C.c_void_p, C.c_int, C.c_int, _f32p, _f32p]
        L.ref_trace_enable.argtypes = [C.c_int]
        L.ref_trace_name.argtypes = [C.c_int, C.c_char_p, C.c_int]
        L.ref_trace_get.restype = C.c_long
        L.ref_trace_get.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_long]
        L.ref_sample_best.argtypes = [C.c_void_p] + [C.c_void_p] * 5
        L.ref_sample_timestamp.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 5
        L.ref_token_special.argtypes = [C.c_void_p, C.c_int]
        L.ref_token_to_str.restype = C.c_char_p
        L.ref_token_to_str.argtypes = [C.c_void_p, C.c_int]
        L.ref_tokenize.argtypes = [C.c_void_p, C.c_char_p, _i32p, C.c_int]
        L.ref_is_multilingual.argtypes = [C.c_void_p]
        L.ref_full.argtypes = [C.c_void_p, _f32p, C.c_int, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_int,
                               C.c_void_p, C.c_int, C.c_int]
        L.ref_full_range.argtypes = [C.c_void_p, _f32p, C.c_int, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.ref_full_token_timestamps.argtypes = [C.c_void_p, _f32p, C.c_int, C.c_int, C.c_char_p, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                                C.c_float, C.c_float, C.c_int]
        L.ref_full_token_data.argtypes = [C.c_void_p, C.c_int, C.c_int, np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS"),
                                          np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")]
        L.ref_full_n_segments.argtypes = [C.c_void_p]
        for n in ("ref_full_segment_t0", "ref_full_segment_t1"):
            getattr(L, n).restype = C.c_int64
            getattr(L, n).argtypes = [C.c_void_p, C.c_int]
        L.ref_full_segment_text.restype = C.c_char_p
        L.ref_full_segment_text.argtypes = [C.c_void_p, C.c_int]
        L.ref_full_n_tokens.argtypes = [C.c_void_p, C.c_int]
        L.ref_full_token_id.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.ref_full_token_tid.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.ref_full_token_p.restype = C.c_float
        L.ref_full_token_p.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.ref_timings.argtypes = [C.c_void_p, np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")]
        L.ref_reset_timings.argtypes = [C.c_void_p]
        L.ref_system_info.restype = C.c_char_p
        L.ref_lookup_tables.argtypes = [np.ctypeslib.ndpointer(dtype=np.uint16, flags="C_CONTIGUOUS")] * 2
        _lib = L
    return _lib


class RefWhisper:
    """One whisper_context of the reference CPU implementation."""

    def __init__(self, model_path: str, n_threads: int = 1, log_level: int = 1):
        self.L = lib()
        self.L.ref_set_log_level(log_level)
        self.ctx = self.L.ref_init(model_path.encode())
        if not self.ctx:
            raise RuntimeError("reference whisper_init failed for " + model_path)
        self.n_threads = n_threads
        hp = np.zeros(11, np.int32)
        self.L.ref_hparams(self.ctx, hp)
        (self.n_vocab, self.n_audio_ctx, self.n_audio_state, self.n_audio_head, self.n_audio_layer,
         self.n_text_ctx, self.n_text_state, self.n_text_head, self.n_text_layer, self.n_mels, self.f16) = map(int, hp)

    def close(self):
        if self.ctx:
            self.L.ref_free(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- mel ----
    def pcm_to_mel(self, pcm: np.ndarray) -> np.ndarray:
        pcm = np.ascontiguousarray(pcm, np.float32)
        rc = self.L.ref_pcm_to_mel(self.ctx, pcm, len(pcm), self.n_threads)
        assert rc == 0
        return self.get_mel()

    def set_mel(self, mel: np.ndarray):
        """mel: [n_mel][n_len] float32 (row = mel bin), whisper.cpp:2332-2347."""
        mel = np.ascontiguousarray(mel, np.float32)
        assert mel.shape[0] == self.n_mels
        rc = self.L.ref_set_mel(self.ctx, mel, mel.shape[1], mel.shape[0])
        assert rc == 0

    def set_mel_any(self, mel: np.ndarray):
        """The context's spectrogram written directly: any number of mel bins (whisper_set_mel insists on 80); the encoder takes the count
        from the model file, so a model of the large-v3 shape (128 bins) runs."""
        mel = np.ascontiguousarray(mel, np.float32)
        self.L.ref_set_mel_any(self.ctx, mel, mel.shape[1], mel.shape[0])

    def get_mel(self) -> np.ndarray:
        n = self.L.ref_mel_len(self.ctx)
        out = np.zeros((self.n_mels, n), np.float32)
        self.L.ref_get_mel(self.ctx, out)
        return out

    # ---- encoder / decoder ----
    def encode(self, mel_offset: int = 0):
        rc = self.L.ref_encode(self.ctx, mel_offset, self.n_threads)
        assert rc == 0

    def set_audio_ctx(self, n: int):
        """whisper_context::exp_n_audio_ctx (what whisper_full sets from params.audio_ctx, whisper.cpp:2800): encoder positions / cross-attention keys; 0 = the model's."""
        self.L.ref_set_audio_ctx(self.ctx, int(n))

    def cross_kv(self, layer: int):
        k = np.zeros((self.n_audio_ctx, self.n_audio_state), np.float32)
        v = np.zeros_like(k)
        self.L.ref_get_cross_kv(self.ctx, layer, k, v)
        return k, v

    def self_kv(self, layer: int, rows: int):
        k = np.zeros((rows, self.n_text_state), np.float32)
        v = np.zeros_like(k)
        self.L.ref_get_self_kv(self.ctx, layer, rows, k, v)
        return k, v

    def decode(self, tokens: Sequence[int], n_past: int):
        """Returns (logits, probs), each [n_tokens][n_vocab]."""
        t = np.ascontiguousarray(tokens, np.int32)
        rc = self.L.ref_decode(self.ctx, t, len(t), n_past, self.n_threads)
        assert rc == 0
        n = self.L.ref_logits_size(self.ctx)
        logits = np.zeros(n, np.float32)
        probs = np.zeros(n, np.float32)
        self.L.ref_get_logits(self.ctx, logits)
        self.L.ref_get_probs(self.ctx, probs)
        return logits.reshape(len(t), -1), probs.reshape(len(t), -1)

    def sample_best(self):
        vals = [C.c_int32(), C.c_int32(), C.c_float(), C.c_float(), C.c_float()]
        self.L.ref_sample_best(self.ctx, *[C.byref(v) for v in vals])
        return dict(id=vals[0].value, tid=vals[1].value, p=vals[2].value, pt=vals[3].value, ptsum=vals[4].value)

    def sample_timestamp(self, is_initial: bool):
        vals = [C.c_int32(), C.c_int32(), C.c_float(), C.c_float(), C.c_float()]
        self.L.ref_sample_timestamp(self.ctx, int(is_initial), *[C.byref(v) for v in vals])
        return dict(id=vals[0].value, tid=vals[1].value, p=vals[2].value, pt=vals[3].value, ptsum=vals[4].value)

    # ---- probe points ----
    def trace(self, on: bool = True):
        self.L.ref_trace_enable(int(on))

    def traced(self) -> Dict[str, np.ndarray]:
        out = {}
        buf = C.create_string_buffer(128)
        for i in range(self.L.ref_trace_count()):
            self.L.ref_trace_name(i, buf, 128)
            name = buf.value
            ne = (C.c_int32 * 4)()
            n = self.L.ref_trace_get(name, ne, None, 0)
            a = np.zeros(n, np.float32)
            self.L.ref_trace_get(name, ne, a.ctypes.data_as(C.c_void_p), n)
            out[name.decode()] = a.reshape(ne[3], ne[2], ne[1], ne[0]).squeeze()
        return out

    # ---- whisper_full ----
    def full(self, pcm: np.ndarray, lang: str = "en", no_context: bool = True, single_segment: bool = False,
             translate: bool = False, max_tokens: int = 0, audio_ctx: int = 0, prompt: Optional[Sequence[int]] = None,
             n_max_text_ctx: int = -1):
        pcm = np.ascontiguousarray(pcm, np.float32)
        flags = int(no_context) | (int(single_segment) << 1) | (int(translate) << 2)
        pt = np.ascontiguousarray(prompt if prompt is not None else [], np.int32)
        rc = self.L.ref_full(self.ctx, pcm, len(pcm), self.n_threads, lang.encode(), flags, max_tokens, audio_ctx,
                             pt.ctypes.data_as(C.c_void_p) if len(pt) else None, len(pt), n_max_text_ctx)
        if rc != 0:
            raise RuntimeError("whisper_full rc=%d" % rc)
        segs = []
        for i in range(self.L.ref_full_n_segments(self.ctx)):
            nt = self.L.ref_full_n_tokens(self.ctx, i)
            segs.append(dict(
                t0=self.L.ref_full_segment_t0(self.ctx, i), t1=self.L.ref_full_segment_t1(self.ctx, i),
                text=self.L.ref_full_segment_text(self.ctx, i),
                tokens=[self.L.ref_full_token_id(self.ctx, i, j) for j in range(nt)],
                probs=[self.L.ref_full_token_p(self.ctx, i, j) for j in range(nt)]))
        return segs

    def full_range(self, pcm: np.ndarray, lang: str = "en", flags: int = 1, max_tokens: int = 0, prompt: Optional[Sequence[int]] = None,
                   n_max_text_ctx: int = -1, offset_ms: int = 0, duration_ms: int = 0):
        """whisper_full on a range of the audio. flags: 1 no_context, 2 single_segment, 4 translate, 8 print_special."""
        pcm = np.ascontiguousarray(pcm, np.float32)
        pt = np.ascontiguousarray(prompt if prompt is not None else [], np.int32)
        rc = self.L.ref_full_range(self.ctx, pcm if len(pcm) else np.zeros(1, np.float32), len(pcm), self.n_threads, lang.encode(), flags, max_tokens,
                                   pt.ctypes.data_as(C.c_void_p) if len(pt) else None, len(pt), n_max_text_ctx, offset_ms, duration_ms)
        segs = []
        if rc != 0:
            return rc, segs
        for i in range(self.L.ref_full_n_segments(self.ctx)):
            nt = self.L.ref_full_n_tokens(self.ctx, i)
            segs.append(dict(t0=self.L.ref_full_segment_t0(self.ctx, i), t1=self.L.ref_full_segment_t1(self.ctx, i),
                             text=self.L.ref_full_segment_text(self.ctx, i).decode(errors="replace"),
                             tokens=[self.L.ref_full_token_id(self.ctx, i, j) for j in range(nt)]))
        return rc, segs

    def full_token_timestamps(self, pcm: np.ndarray, lang: str = "en", no_context: bool = True, prompt: Optional[Sequence[int]] = None,
                              n_max_text_ctx: int = -1, thold_pt: float = 0.01, thold_ptsum: float = 0.01, max_len: int = 0):
        """whisper_full with token_timestamps = true; every token comes back with t0 / t1 / vlen next to id / tid / p / pt / ptsum."""
        pcm = np.ascontiguousarray(pcm, np.float32)
        pt = np.ascontiguousarray(prompt if prompt is not None else [], np.int32)
        rc = self.L.ref_full_token_timestamps(self.ctx, pcm, len(pcm), self.n_threads, lang.encode(), int(no_context),
                                              pt.ctypes.data_as(C.c_void_p) if len(pt) else None, len(pt), n_max_text_ctx,
                                              thold_pt, thold_ptsum, max_len)
        if rc != 0:
            raise RuntimeError("whisper_full rc=%d" % rc)
        segs = []
        t, f = np.zeros(2, np.int64), np.zeros(4, np.float32)
        for i in range(self.L.ref_full_n_segments(self.ctx)):
            toks = []
            for j in range(self.L.ref_full_n_tokens(self.ctx, i)):
                self.L.ref_full_token_data(self.ctx, i, j, t, f)
                toks.append(dict(id=self.L.ref_full_token_id(self.ctx, i, j), tid=self.L.ref_full_token_tid(self.ctx, i, j),
                                 t0=int(t[0]), t1=int(t[1]), p=float(f[0]), pt=float(f[1]), ptsum=float(f[2]), vlen=float(f[3])))
            segs.append(dict(t0=self.L.ref_full_segment_t0(self.ctx, i), t1=self.L.ref_full_segment_t1(self.ctx, i),
                             text=self.L.ref_full_segment_text(self.ctx, i).decode(), tokens=toks))
        return segs

    def timings_us(self):
        out = np.zeros(5, np.int64)
        self.L.ref_timings(self.ctx, out)
        return dict(load=int(out[0]), mel=int(out[1]), sample=int(out[2]), encode=int(out[3]), decode=int(out[4]))

    def reset_timings(self):
        self.L.ref_reset_timings(self.ctx)


def lookup_tables():
    """(gelu, exp) uint16[65536] bit patterns of the FP16 tables ggml builds at init (ggml.c:1375-1385)."""
    g = np.zeros(65536, np.uint16)
    e = np.zeros(65536, np.uint16)
    lib().ref_lookup_tables(g, e)
    return g, e


def read_wav_mono16(path: str) -> np.ndarray:
    """PCM16 mono WAV -> float32 in [-1, 1) (the conversion the reference's examples use, /32768)."""
    import wave
    with wave.open(path, "rb") as w:
        assert w.getsampwidth() == 2
        n = w.getnframes()
        a = np.frombuffer(w.readframes(n), dtype="<i2").astype(np.float32)
        if w.getnchannels() == 2:
            a = a.reshape(-1, 2).mean(axis=1)
    return a / 32768.0


# ---- the reference's streaming spectrogram (oracle/_ref/libmelstreamer_ref.so: Whisper/Whisper/MelStreamer.cpp + melSpectrogram.cpp
# compiled unmodified, oracle/melstreamer_harness.cpp) ----
MELSTREAMER_LIB_PATH = os.path.join(_HERE, "_ref", "libmelstreamer_ref.so")
_ms_lib = None


def melstreamer_available() -> bool:
    return os.path.exists(MELSTREAMER_LIB_PATH)


def _melstreamer_lib():
    global _ms_lib
    if _ms_lib is None:
        L = C.CDLL(MELSTREAMER_LIB_PATH)
        L.ms_create.restype = C.c_void_p
        L.ms_create.argtypes = [_f32p, C.c_int, C.c_int, _f32p, C.c_longlong, C.c_int, C.c_int]
        L.ms_length.restype = C.c_longlong
        L.ms_length.argtypes = [C.c_void_p]
        L.ms_make_buffer.argtypes = [C.c_void_p, C.c_longlong, C.c_longlong, _f32p]
        L.ms_destroy.argtypes = [C.c_void_p]
        L.ms_pcm_to_mel.argtypes = [_f32p, C.c_int, C.c_int, _f32p, C.c_longlong, C.c_int, _f32p]
        _ms_lib = L
    return _ms_lib


def spectrogram_pcm_to_mel(pcm: np.ndarray, filters: np.ndarray, threads: int = 1) -> np.ndarray:
    """Spectrogram::pcmToMel of the reference's GPU model (Whisper/Whisper/Spectrogram.cpp:64-122): [80][len(pcm) // 160]."""
    pcm = np.ascontiguousarray(pcm, np.float32)
    flt = np.ascontiguousarray(filters, np.float32)
    out = np.zeros((80, len(pcm) // 160), np.float32)
    hr = _melstreamer_lib().ms_pcm_to_mel(flt.reshape(-1), 80, 201, pcm, len(pcm), threads, out)
    if hr < 0:
        raise RuntimeError("Spectrogram::pcmToMel failed: HRESULT 0x%08x" % (hr & 0xFFFFFFFF))
    return out


class RefMelStreamer:
    """The reference's iSpectrogram of iContext::runStreamed over PCM in memory: threads <= 1 = MelStreamerSimple (FFTs on demand),
    threads >= 2 = MelStreamerThread (a background thread keeps a queue of frames full; what runStreamed picks for cpuThreads > 1,
    ContextImpl.misc.cpp:404-413). make_buffer(off, len) = makeBuffer + makeTransposedBuffer: [80][len] frames of the stream
    normalised on the window's own maximum. `block` = samples per delivery of the in-memory "source reader"; with anything but one
    chunk per delivery the reference reads stale memory at the end of a stream whose length is not a multiple of 160 samples
    (oracle/melstreamer_harness.cpp, "the end of a stream")."""

    def __init__(self, pcm: np.ndarray, filters: np.ndarray, threads: int = 1, block: int = 160):
        pcm = np.ascontiguousarray(pcm, np.float32)
        flt = np.ascontiguousarray(filters, np.float32)
        assert flt.shape == (80, 201)
        self._h = _melstreamer_lib().ms_create(flt.reshape(-1), 80, 201, pcm if len(pcm) else np.zeros(1, np.float32), len(pcm), threads, block)
        if not self._h:
            raise RuntimeError("ms_create failed")

    @property
    def length(self) -> int:
        """PcmReader::getLength(): whole 160-sample chunks."""
        return int(_melstreamer_lib().ms_length(self._h))

    def make_buffer(self, off: int, length: int) -> np.ndarray:
        out = np.zeros((80, length), np.float32)
        hr = _melstreamer_lib().ms_make_buffer(self._h, off, length, out)
        if hr < 0:
            raise RuntimeError("MelStreamer::makeBuffer failed: HRESULT 0x%08x" % (hr & 0xFFFFFFFF))
        return out

    def close(self):
        if self._h:
            _melstreamer_lib().ms_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- the reference's GPU-model iContext over its own CPU arithmetic (oracle/_ref/libcontextimpl_ref.so: Whisper/Whisper/ContextImpl.cpp +
# ContextImpl.misc.cpp + Languages.cpp compiled unmodified, the D3D compute context replaced by whisper.cpp; oracle/contextimpl_harness.cpp) ----
CONTEXTIMPL_LIB_PATH = os.path.join(_HERE, "_ref", "libcontextimpl_ref.so")
_ci_lib = None

FLAG_TRANSLATE, FLAG_NO_CONTEXT, FLAG_SINGLE_SEGMENT, FLAG_PRINT_SPECIAL, FLAG_TOKEN_TIMESTAMPS = 1, 2, 4, 8, 0x100    # eFullParamsFlags (API/sFullParams.h:22-35)


def contextimpl_available() -> bool:
    return os.path.exists(CONTEXTIMPL_LIB_PATH) and available()


class _CiParams(C.Structure):
    _fields_ = [("flags", C.c_uint32), ("language", C.c_uint32), ("cpuThreads", C.c_int32), ("n_max_text_ctx", C.c_int32), ("offset_ms", C.c_int32),
                ("duration_ms", C.c_int32), ("max_tokens", C.c_int32), ("max_len", C.c_int32), ("thold_pt", C.c_float), ("thold_ptsum", C.c_float),
                ("prompt_tokens", C.POINTER(C.c_int32)), ("prompt_n_tokens", C.c_int32), ("mediaTime", C.c_int64)]


def _contextimpl_lib():
    global _ci_lib
    if _ci_lib is None:
        C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        L = C.CDLL(CONTEXTIMPL_LIB_PATH)
        L.ci_create.restype = C.c_void_p
        L.ci_create.argtypes = [C.c_char_p, _f32p, C.c_int, C.c_int, C.c_int]
        L.ci_destroy.argtypes = [C.c_void_p]
        L.ci_last_error.restype = C.c_char_p
        for n in ("ci_run_full", "ci_run_streamed"):
            getattr(L, n).argtypes = [C.c_void_p, C.POINTER(_CiParams), _f32p, C.c_int]
        L.ci_counts.argtypes = [C.c_void_p, _i32p]
        L.ci_progress.argtypes = [C.c_void_p, np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")]
        L.ci_segment.restype = C.c_char_p
        L.ci_segment.argtypes = [C.c_void_p, C.c_int, np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS"),
                                 np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")]
        L.ci_token.restype = C.c_char_p
        L.ci_token.argtypes = [C.c_void_p, C.c_int, np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS"), _f32p, _i32p]
        L.ci_language_id.argtypes = [C.c_char_p]
        L.ci_get_results.argtypes = [C.c_void_p, C.c_uint32]
        _ci_lib = L
    return _ci_lib


def language_key(code: str) -> int:
    """makeLanguageKey (API/sFullParams.h:117-132)."""
    k = 0
    for i, ch in enumerate(code.encode()[:4]):
        k |= ch << (8 * i)
    return k


class RefContextImpl:
    """The reference's ContextImpl (runFull / runStreamed / getResults of its GPU model) computing with the reference's CPU model.
    run_full / run_streamed return (HRESULT, segments); a segment = dict(t0, t1 [100 ns ticks incl. the media time offset], text, tokens=[dict(id,
    text, t0, t1, p, pt, ptsum, vlen, flags)]). self.progress = what the progress sink saw (runStreamed), self.new_segment = (calls, sum of n_new)."""

    def __init__(self, model_path: str, filters: np.ndarray, encoder_threads: int = 4):
        flt = np.ascontiguousarray(filters, np.float32)
        assert flt.shape == (80, 201)
        lib().ref_set_log_level(0)
        self._h = _contextimpl_lib().ci_create(model_path.encode(), flt.reshape(-1), 80, 201, encoder_threads)
        if not self._h:
            raise RuntimeError("ci_create failed for " + model_path)
        self.progress, self.new_segment = [], (0, 0)

    def _params(self, lang, flags, cpu_threads, n_max_text_ctx, offset_ms, duration_ms, max_tokens, max_len, thold_pt, thold_ptsum, prompt, media_time):
        p = _CiParams()
        p.flags, p.language, p.cpuThreads, p.n_max_text_ctx = flags, language_key(lang), cpu_threads, n_max_text_ctx
        p.offset_ms, p.duration_ms, p.max_tokens, p.max_len, p.thold_pt, p.thold_ptsum = offset_ms, duration_ms, max_tokens, max_len, thold_pt, thold_ptsum
        self._prompt = (C.c_int32 * max(1, len(prompt or [])))(*(prompt or [0]))
        p.prompt_tokens = C.cast(self._prompt, C.POINTER(C.c_int32)) if prompt else None
        p.prompt_n_tokens = len(prompt or [])
        p.mediaTime = media_time
        return p

    def results(self, flags: int):
        """iContext::getResults with these eResultFlags (Tokens = 1, Timestamps = 2) on the transcript of the last run."""
        hr = _contextimpl_lib().ci_get_results(self._h, flags)
        if hr < 0:
            raise RuntimeError("getResults failed: 0x%08x" % (hr & 0xFFFFFFFF))
        return self._results(with_tokens=bool(flags & 1))

    def _results(self, with_tokens=True):
        L = _contextimpl_lib()
        cnt = np.zeros(4, np.int32)
        n_prog = L.ci_counts(self._h, cnt)
        prog = np.zeros(max(1, n_prog), np.float64)
        L.ci_progress(self._h, prog)
        self.progress, self.new_segment = [float(x) for x in prog[:n_prog]], (int(cnt[2]), int(cnt[3]))
        segs = []
        for i in range(int(cnt[0])):
            t, ft = np.zeros(2, np.int64), np.zeros(2, np.uint32)
            text = L.ci_segment(self._h, i, t, ft)
            toks = []
            for j in range(int(ft[0]), int(ft[0] + ft[1])) if with_tokens else []:
                tt, pr, idf = np.zeros(2, np.int64), np.zeros(4, np.float32), np.zeros(2, np.int32)
                s = L.ci_token(self._h, j, tt, pr, idf)
                toks.append(dict(id=int(idf[0]), flags=int(idf[1]), text=(s or b"").decode(errors="replace"), t0=int(tt[0]), t1=int(tt[1]),
                                 p=float(pr[0]), pt=float(pr[1]), ptsum=float(pr[2]), vlen=float(pr[3])))
            segs.append(dict(t0=int(t[0]), t1=int(t[1]), text=(text or b"").decode(errors="replace"), tokens=toks, first_token=int(ft[0]), count_tokens=int(ft[1])))
        return segs

    def _run(self, fn, pcm, lang="en", flags=0, cpu_threads=4, n_max_text_ctx=-1, offset_ms=0, duration_ms=0, max_tokens=0, max_len=0,
             thold_pt=-1.0, thold_ptsum=-1.0, prompt=None, media_time=0):
        pcm = np.ascontiguousarray(pcm, np.float32)
        p = self._params(lang, flags, cpu_threads, n_max_text_ctx, offset_ms, duration_ms, max_tokens, max_len, thold_pt, thold_ptsum, prompt, media_time)
        hr = fn(self._h, C.byref(p), pcm if len(pcm) else np.zeros(1, np.float32), len(pcm))
        if hr < 0:
            return hr & 0xFFFFFFFF, []
        return hr, self._results()

    def run_full(self, pcm, **kw):
        return self._run(_contextimpl_lib().ci_run_full, pcm, **kw)

    def run_streamed(self, pcm, **kw):
        return self._run(_contextimpl_lib().ci_run_streamed, pcm, **kw)

    def last_error(self) -> str:
        return (_contextimpl_lib().ci_last_error() or b"").decode(errors="replace")

    def close(self):
        if self._h:
            _contextimpl_lib().ci_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
