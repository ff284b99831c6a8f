"""Python face of libWhisper.so, the COM-style host API (include/whisperApi.h) through its flat C mirror (include/whisper_c.h).

Mirrors how the reference's own callers drive it (Examples/main/main.cpp:174-330): loadModel -> createContext ->
fullDefaultParams -> runFull -> getResults. No fallback: without the shared libraries and a GPU every call raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
HOST_LIB_PATH = os.path.join(_HERE, "lib", "libWhisper.so")

# eFullParamsFlags (Whisper/API/sFullParams.h:21-35)
TRANSLATE, NO_CONTEXT, SINGLE_SEGMENT, PRINT_SPECIAL = 1, 2, 4, 8
TOKEN_TIMESTAMPS = 0x100

CPP_EXPORTS = ["setupLogger", "loadModel", "initMediaFoundation", "findLanguageKeyW", "findLanguageKeyA", "getSupportedLanguages", "listGPUs"]
# extensions next to the seven names of whisper.def: one process per GPU, and K streams in lock step on one GPU
CPP_EXTENSIONS = ["loadModelShared", "createBatchRunner", "runFullBatch"]

_lib = None


class WhisperError(RuntimeError):
    def __init__(self, hr: int, what: str):
        super().__init__("%s failed: HRESULT 0x%08x" % (what, hr & 0xFFFFFFFF))
        self.hr = hr & 0xFFFFFFFF


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(HOST_LIB_PATH):
            raise RuntimeError("libWhisper.so is missing: run `python -m whisper_amd.build`")
        L = C.CDLL(HOST_LIB_PATH)
        vp = C.c_void_p
        L.whisperc_load_model.argtypes = [C.c_char_p, C.c_int, C.POINTER(vp)]
        L.whisperc_release.argtypes = [vp]
        L.whisperc_release.restype = None
        L.whisperc_create_context.argtypes = [vp, C.POINTER(vp)]
        L.whisperc_special_tokens.argtypes = [vp, C.POINTER(C.c_int32)]
        L.whisperc_token_string.argtypes = [vp, C.c_int]
        L.whisperc_token_string.restype = C.c_char_p
        L.whisperc_is_multilingual.argtypes = [vp]
        L.whisperc_tokenize.argtypes = [vp, C.c_char_p, C.POINTER(C.c_int32), C.c_int]
        L.whisperc_run_full.argtypes = [vp, vp, C.c_uint32, C.c_char_p, C.c_uint32, C.c_int, vp, C.c_int, C.c_int]
        L.whisperc_run_full_audio_ctx.argtypes = [vp, vp, C.c_uint32, C.c_char_p, C.c_uint32, C.c_int, vp, C.c_int, C.c_int, C.c_int]
        L.whisperc_run_full_beam.argtypes = [vp, vp, C.c_uint32, C.c_char_p, C.c_uint32, C.c_int, vp, C.c_int, C.c_int, C.c_int]
        L.whisperc_result_counts.argtypes = [vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.whisperc_result_segment.argtypes = [vp, C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32),
                                              C.POINTER(C.c_uint32), C.c_char_p, C.c_uint32]
        L.whisperc_run_full_tt.argtypes = [vp, vp, C.c_uint32, C.c_char_p, C.c_uint32, C.c_int, vp, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int]
        L.whisperc_result_token_times.argtypes = [vp, C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_float)]
        L.whisperc_result_token.argtypes = [vp, C.c_uint32, C.POINTER(C.c_int32), C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.whisperc_timings_print.argtypes = [vp]
        L.whisperc_run_streamed.argtypes = [vp, vp, C.c_uint64, C.c_char_p, C.c_uint32, C.c_int, vp, C.c_int, C.c_int,
                                            C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_int)]
        L.whisperc_batch_create.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(vp)]
        L.whisperc_batch_run.argtypes = [vp, C.c_uint32, vp, vp, vp, vp, C.c_char_p, C.c_uint32, C.c_int, vp, C.c_int, C.c_int, vp, vp]
        L.whisperc_tr_counts.argtypes = [vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.whisperc_tr_segment.argtypes = [vp, C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32),
                                          C.POINTER(C.c_uint32), C.c_char_p, C.c_uint32]
        L.whisperc_tr_token.argtypes = [vp, C.c_uint32, C.POINTER(C.c_int32), C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float),
                                        C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_float)]
        _lib = L
    return _lib


def set_host_loop_rules(mode: int):
    """0 = the reference CPU model's whisper_full rules (default), 1 = its GPU model's ContextImpl rules."""
    _check(lib().whisperc_set_host_loop_rules(mode), "set_host_loop_rules")


def set_beam_ranking(on_host: bool):
    """eSamplingStrategy::BeamSearch: rank every step's candidates on the host (round 4's decoder, the checker) instead of on the device (default)."""
    _check(lib().whisperc_set_beam_ranking(1 if on_host else 0), "set_beam_ranking")


def _check(hr: int, what: str) -> int:
    if hr < 0:
        raise WhisperError(hr, what)
    return hr


class Model:
    """iModel."""

    def __init__(self, path: str, device: int = 0):
        self.h = C.c_void_p()
        _check(lib().whisperc_load_model(path.encode(), device, C.byref(self.h)), "loadModel")

    def close(self):
        if self.h:
            lib().whisperc_release(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def create_context(self) -> "Context":
        return Context(self)

    def create_batch_runner(self, max_slots: int = 0, groups: int = 0, greedy_chunk: int = 0, flags: int = 0) -> "BatchRunner":
        return BatchRunner(self, max_slots, groups, greedy_chunk, flags)

    def special_tokens(self):
        a = (C.c_int32 * 8)()
        _check(lib().whisperc_special_tokens(self.h, a), "getSpecialTokens")
        names = ("eot", "sot", "prev", "solm", "not_", "beg", "translate", "transcribe")
        return dict(zip(names, list(a)))

    def token_string(self, token: int) -> Optional[bytes]:
        return lib().whisperc_token_string(self.h, token)

    def is_multilingual(self) -> bool:
        return lib().whisperc_is_multilingual(self.h) == 0

    def tokenize(self, text: str) -> List[int]:
        buf = (C.c_int32 * 4096)()
        n = _check(lib().whisperc_tokenize(self.h, text.encode(), buf, 4096), "tokenize")
        return list(buf[:n])


class Context:
    """iContext."""

    def __init__(self, model: Model):
        self.model = model
        self.h = C.c_void_p()
        _check(lib().whisperc_create_context(model.h, C.byref(self.h)), "createContext")

    def close(self):
        if self.h:
            lib().whisperc_release(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def run_full(self, pcm: np.ndarray, language: str = "en", flags: int = 0, max_tokens: int = 0,
                 prompt: Optional[Sequence[int]] = None, n_max_text_ctx: int = -1, max_len: int = 0, thold_pt: float = 0.01,
                 thold_ptsum: float = 0.01, beam_width: int = 0, audio_ctx: int = 0) -> int:
        """runFull on mono float32 16 kHz PCM. Returns the HRESULT (0 = S_OK, 1 = S_FALSE: less than 1 s of audio).
        With TOKEN_TIMESTAMPS in flags the tokens of results() carry t0 / t1 / vlen and max_len > 0 wraps the segments."""
        pcm = np.ascontiguousarray(pcm, np.float32)
        pt = np.ascontiguousarray(prompt if prompt is not None else [], np.int32)
        if audio_ctx:               # sFullParams::audio_ctx (ContextImpl.cpp:488-489)
            return _check(lib().whisperc_run_full_audio_ctx(self.h, pcm.ctypes.data_as(C.c_void_p), len(pcm), language.encode(), flags, max_tokens,
                                                            pt.ctypes.data_as(C.c_void_p) if len(pt) else None, len(pt), n_max_text_ctx, audio_ctx), "runFull")
        if beam_width > 0:          # eSamplingStrategy::BeamSearch with this beam_width (extension: the reference only declares it)
            return _check(lib().whisperc_run_full_beam(self.h, pcm.ctypes.data_as(C.c_void_p), len(pcm), language.encode(), flags, max_tokens,
                                                       pt.ctypes.data_as(C.c_void_p) if len(pt) else None, len(pt), n_max_text_ctx, beam_width), "runFull")
        if flags & TOKEN_TIMESTAMPS:
            return _check(lib().whisperc_run_full_tt(self.h, pcm.ctypes.data_as(C.c_void_p), len(pcm), language.encode(), flags, max_tokens,
                                                     pt.ctypes.data_as(C.c_void_p) if len(pt) else None, len(pt), n_max_text_ctx,
                                                     thold_pt, thold_ptsum, max_len), "runFull")
        return _check(lib().whisperc_run_full(self.h, pcm.ctypes.data_as(C.c_void_p), len(pcm), language.encode(), flags, max_tokens,
                                              pt.ctypes.data_as(C.c_void_p) if len(pt) else None, len(pt), n_max_text_ctx), "runFull")

    def run_streamed(self, pcm: np.ndarray, language: str = "en", flags: int = 0, max_tokens: int = 0,
                     prompt: Optional[Sequence[int]] = None, n_max_text_ctx: int = -1):
        """iMediaFoundation::loadAudioFileData (the PCM wrapped as a float32 WAV image) + iContext::runStreamed.
        Returns (HRESULT, [progress values the sink received])."""
        wav = wav_bytes(pcm)
        pt = np.ascontiguousarray(prompt if prompt is not None else [], np.int32)
        prog = (C.c_double * 4096)()
        n = C.c_int()
        hr = _check(lib().whisperc_run_streamed(self.h, wav, len(wav), language.encode(), flags, max_tokens,
                                                pt.ctypes.data_as(C.c_void_p) if len(pt) else None, len(pt), n_max_text_ctx,
                                                prog, 4096, C.byref(n)), "runStreamed")
        return hr, list(prog[:min(n.value, 4096)])

    def results(self):
        """getResults(Tokens | Timestamps): list of segments {t0, t1 (100 ns ticks), text, tokens[{id, p, pt, ptsum}]}."""
        ns, nt = C.c_uint32(), C.c_uint32()
        _check(lib().whisperc_result_counts(self.h, C.byref(ns), C.byref(nt)), "getResults")
        out = []
        for i in range(ns.value):
            t0, t1, ft, ct = C.c_uint64(), C.c_uint64(), C.c_uint32(), C.c_uint32()
            text = C.create_string_buffer(4096)
            _check(lib().whisperc_result_segment(self.h, i, C.byref(t0), C.byref(t1), C.byref(ft), C.byref(ct), text, 4096), "getSegments")
            toks = []
            for j in range(ft.value, ft.value + ct.value):
                tid, p, pt, ps = C.c_int32(), C.c_float(), C.c_float(), C.c_float()
                _check(lib().whisperc_result_token(self.h, j, C.byref(tid), C.byref(p), C.byref(pt), C.byref(ps)), "getTokens")
                k0, k1, vl = C.c_uint64(), C.c_uint64(), C.c_float()
                _check(lib().whisperc_result_token_times(self.h, j, C.byref(k0), C.byref(k1), C.byref(vl)), "getTokens")
                toks.append(dict(id=tid.value, p=p.value, pt=pt.value, ptsum=ps.value, t0=k0.value, t1=k1.value, vlen=vl.value))
            out.append(dict(t0=t0.value, t1=t1.value, text=text.value, tokens=toks))
        return out

    def timings_print(self):
        _check(lib().whisperc_timings_print(self.h), "timingsPrint")


def read_result(h) -> list:
    """iTranscribeResult -> list of segments {t0, t1 (100 ns ticks), text, tokens[{id, p, pt, ptsum, t0, t1, vlen}]}."""
    L = lib()
    ns, nt = C.c_uint32(), C.c_uint32()
    _check(L.whisperc_tr_counts(h, C.byref(ns), C.byref(nt)), "getSize")
    out = []
    for i in range(ns.value):
        t0, t1, ft, ct = C.c_uint64(), C.c_uint64(), C.c_uint32(), C.c_uint32()
        text = C.create_string_buffer(4096)
        _check(L.whisperc_tr_segment(h, i, C.byref(t0), C.byref(t1), C.byref(ft), C.byref(ct), text, 4096), "getSegments")
        toks = []
        for j in range(ft.value, ft.value + ct.value):
            tid, p, pt, ps = C.c_int32(), C.c_float(), C.c_float(), C.c_float()
            k0, k1, vl = C.c_uint64(), C.c_uint64(), C.c_float()
            _check(L.whisperc_tr_token(h, j, C.byref(tid), C.byref(p), C.byref(pt), C.byref(ps), C.byref(k0), C.byref(k1), C.byref(vl)), "getTokens")
            toks.append(dict(id=tid.value, p=p.value, pt=pt.value, ptsum=ps.value, t0=k0.value, t1=k1.value, vlen=vl.value))
        out.append(dict(t0=t0.value, t1=t1.value, text=text.value, tokens=toks))
    return out


class BatchRunner:
    """iBatchRunner (Whisper::createBatchRunner): K streams in lock step; each keeps the semantics of iContext::runFull."""

    def __init__(self, model: Model, max_slots: int = 0, groups: int = 0, greedy_chunk: int = 0, flags: int = 0):
        self.model = model
        self.h = C.c_void_p()
        _check(lib().whisperc_batch_create(model.h, max_slots, groups, greedy_chunk, flags, C.byref(self.h)), "createBatchRunner")

    def close(self):
        if self.h:
            lib().whisperc_release(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def run(self, streams, language: str = "en", flags: int = 0, max_tokens: int = 0, prompt: Optional[Sequence[int]] = None,
            n_max_text_ctx: int = -1, want_results: bool = True):
        """streams: list of float32 PCM arrays, or of (pcm, first_sample, count_samples) -- pieces of a recording share the array.
        Returns (HRESULT, [segments per stream or None], [per-stream HRESULT])."""
        n = len(streams)
        keep, ptrs, lens, first, cnt = [], (C.c_void_p * n)(), (C.c_uint32 * n)(), (C.c_int64 * n)(), (C.c_int64 * n)()
        for i, s in enumerate(streams):
            pcm, f, c = (s, 0, 0) if not isinstance(s, tuple) else s
            assert pcm.dtype == np.float32 and pcm.flags["C_CONTIGUOUS"]
            keep.append(pcm)
            ptrs[i], lens[i], first[i], cnt[i] = pcm.ctypes.data, len(pcm), f, c
        pt = np.ascontiguousarray(prompt if prompt is not None else [], np.int32)
        res = (C.c_void_p * n)()
        per = (C.c_int32 * n)()
        import time
        t0 = time.perf_counter()
        hr = lib().whisperc_batch_run(self.h, n, ptrs, lens, first, cnt, language.encode(), flags, max_tokens,
                                      pt.ctypes.data_as(C.c_void_p) if len(pt) else None, len(pt), n_max_text_ctx, res, per)
        self.last_run_seconds = time.perf_counter() - t0          # the library call alone (bench.py)
        out = []
        for i in range(n):
            out.append(read_result(res[i]) if (res[i] and want_results) else None)
            if res[i]:
                lib().whisperc_release(res[i])
        _check(hr, "runFullBatch")
        return hr, out, [int(x) & 0xFFFFFFFF for x in per]


def wav_bytes(pcm: np.ndarray, rate: int = 16000) -> bytes:
    """Mono float32 PCM -> the bytes of a RIFF/WAVE file (format 3 = IEEE float), what loadAudioFileData / openAudioFile read."""
    import struct
    data = np.ascontiguousarray(pcm, "<f4").tobytes()
    fmt = struct.pack("<HHIIHH", 3, 1, rate, rate * 4, 4, 32)
    return b"RIFF" + struct.pack("<I", 4 + 8 + len(fmt) + 8 + len(data)) + b"WAVE" + b"fmt " + struct.pack("<I", len(fmt)) + fmt + \
        b"data" + struct.pack("<I", len(data)) + data
