"""ctypes binding of include/whisper_hip.h (libwhisper_hip.so) -- the only way Python reaches the GPU path.

There is no CPU fallback here by design: if the shared library is missing or no HIP device is visible every entry point
raises. torch is used only as plumbing (device buffers for inputs, streams, torch.distributed for the weight broadcast).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import numpy as np

from . import ggml_format as gf

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libwhisper_hip.so")

WH_FLAG_PARITY_PV = 1
WH_FLAG_NO_GRAPH = 2
WH_FLAG_DEBUG_CAPTURE = 4
WH_FLAG_PARITY_EXACT = 8      # the reference CPU path's arithmetic in its own summation order (exact.hip); never timed
# kernel-variant switches (whisper_amd/csrc/kernels.h eTuning); TUNE_DEFAULT is what the library starts with
TUNE_GEMM_8WAVE = 16
TUNE_GEMM_4WAVE = 4
TUNE_GEMM_FAST_EPI = 1
TUNE_GEMV_ROWS4, TUNE_GEMM_BIG, TUNE_GEMV_SMALLREG, TUNE_GEMM_GL, TUNE_LN_SEPARATE_BIGM, TUNE_ATTN_XCD = 2, 8, 32, 64, 128, 256
TUNE_ATTN_DEC_G, TUNE_FUSE_CROSS_Q, TUNE_GEMM_GROUP_M, TUNE_FUSE_SELF_BLOCK, TUNE_GEMV_LN_BLOCK, TUNE_GEMV_K8 = 512, 1024, 2048, 4096, 8192, 16384
TUNE_SPLIT_STREAMS, TUNE_ATTN_ENC_F, TUNE_GEMM_WIDE_EPI, TUNE_GEMM_FRAGPF, TUNE_ATTN_ENC_2SWEEP = 32768, 65536, 131072, 262144, 524288
TUNE_GEMV_ALLROWS, TUNE_GEMV_ROWGROUPS, TUNE_SELF_MFMA, TUNE_MEL_MFMA = 1048576, 2097152, 4194304, 8388608
TUNE_DECODE_SMALL, TUNE_DECODE_PREFETCH, TUNE_ATTN_DEC_NT, TUNE_ENC_SERIAL = 16777216, 33554432, 67108864, 134217728
TUNE_ATTN_ENC_TABLE = 268435456
TUNE_GEMV_MT8 = 536870912
TUNE_ATTN_ENC_TABLE_ANY = 1073741824
TUNE_SAMPLE_SPREAD = 2147483648
TUNE_DEFAULT = (TUNE_GEMM_FAST_EPI | TUNE_GEMM_8WAVE | TUNE_GEMV_ROWS4 | TUNE_GEMV_SMALLREG | TUNE_GEMM_BIG | TUNE_GEMM_GL | TUNE_LN_SEPARATE_BIGM | TUNE_ATTN_XCD |
                TUNE_ATTN_DEC_G | TUNE_FUSE_CROSS_Q | TUNE_GEMM_GROUP_M | TUNE_FUSE_SELF_BLOCK | TUNE_GEMV_K8 | TUNE_ATTN_ENC_F | TUNE_GEMM_WIDE_EPI |
                TUNE_GEMM_FRAGPF | TUNE_ATTN_ENC_2SWEEP | TUNE_GEMV_ALLROWS |
                TUNE_GEMV_ROWGROUPS | TUNE_SELF_MFMA | TUNE_MEL_MFMA | TUNE_DECODE_SMALL | TUNE_ATTN_DEC_NT | TUNE_ATTN_ENC_TABLE)

# every symbol include/whisper_hip.h declares (checked by tests/test_abi.py)
EXPORTS = [
    "wh_last_error", "wh_device_count", "wh_device_info", "wh_device_set",
    "wh_model_arena_bytes", "wh_model_create", "wh_model_destroy", "wh_model_set_tensor", "wh_model_set_filters",
    "wh_model_finalize", "wh_model_arena", "wh_model_hparams",
    "wh_comm_runtime_check", "wh_comm_unique_id", "wh_comm_create", "wh_comm_destroy", "wh_comm_info", "wh_comm_barrier", "wh_comm_create_timeout", "wh_comm_set_timeout", "wh_comm_broadcast_i32", "wh_model_broadcast",
    "wh_context_create", "wh_context_create_hyp", "wh_context_destroy", "wh_context_bind", "wh_context_set_flags", "wh_context_set_audio_ctx", "wh_context_synchronize", "wh_context_memory",
    "wh_buffer_alloc", "wh_buffer_free", "wh_buffer_upload", "wh_buffer_upload_async", "wh_buffer_download",
    "wh_mel_spectrogram", "wh_encode", "wh_encode_windows", "wh_decode", "wh_sample_best", "wh_beam_candidates", "wh_reorder_self_cache", "wh_beam_window_start", "wh_beam_window_continue", "wh_beam_window_status", "wh_beam_window_records", "wh_decode_greedy", "wh_decode_window_start", "wh_decode_window_finish", "wh_decode_window_continue", "wh_decode_window_fetch", "wh_decode_window_start_ragged", "wh_decode_window_ready", "wh_mel_spectrogram_window", "wh_mel_spectrogram_batch", "wh_profile_enable", "wh_profile_read", "wh_debug_read", "wh_debug_probe", "wh_debug_set_tuning", "wh_debug_set_option", "wh_debug_get_option",
    "wh_op_mul_mat", "wh_op_mul_mat_gelu", "wh_op_layer_norm", "wh_op_flash_attention", "wh_op_soft_max", "wh_op_decoder_attention", "wh_op_decoder_cross_attention",
]


class HParamsC(C.Structure):
    _fields_ = [(k, C.c_int32) for k in gf.HPARAM_FIELDS]


class TokenDataC(C.Structure):
    _fields_ = [("id", C.c_int32), ("tid", C.c_int32), ("p", C.c_float), ("pt", C.c_float), ("ptsum", C.c_float)]


class BeamRulesC(C.Structure):
    _fields_ = [(k, C.c_int32) for k in ("seek", "seekEnd", "nMax", "maxTokens", "singleSegment", "tokenBeg", "tokenEot", "forced")]


class BeamHypC(C.Structure):
    _fields_ = [("sum", C.c_double)] + [(k, C.c_int32) for k in ("i", "hasTs", "seekDelta", "resultLen", "failed", "over", "nTok", "rec")]


class BeamWindowC(C.Structure):
    _fields_ = [(k, C.c_int32) for k in ("step", "nLive", "nFinished", "done", "nPrompt", "nTextCtx", "reserved0", "reserved1")] + [("live", BeamHypC * 8), ("finished", BeamHypC * 24)]


class BeamRecordC(C.Structure):
    _fields_ = [("id", C.c_int32), ("tid", C.c_int32), ("p", C.c_float), ("pt", C.c_float), ("ptsum", C.c_float), ("parent", C.c_int32), ("finished", C.c_int32), ("reserved", C.c_int32)]


class MelWindowC(C.Structure):
    _fields_ = [("mel", C.c_void_p), ("len", C.c_int64), ("offset", C.c_int32), ("reserved", C.c_int32)]


class ProfileEntryC(C.Structure):
    _fields_ = [("name", C.c_char * 32), ("calls", C.c_int64), ("ms", C.c_double), ("flops", C.c_double), ("bytes", C.c_double)]


class WhisperHipError(RuntimeError):
    pass


_lib = None


def lib():
    """Loads libwhisper_hip.so; raises if it has not been built (python -m whisper_amd.build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise WhisperHipError("libwhisper_hip.so is missing (%s): run `python -m whisper_amd.build`; "
                                  "there is no CPU fallback" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        vp, i32, i64, f32p = C.c_void_p, C.c_int, C.c_int64, C.POINTER(C.c_float)
        L.wh_last_error.restype = C.c_char_p
        L.wh_debug_set_tuning.argtypes = [C.c_uint32]
        L.wh_debug_set_option.argtypes = [C.c_char_p, C.c_int]
        L.wh_debug_get_option.argtypes = [C.c_char_p, C.POINTER(C.c_int)]
        L.wh_beam_window_start.argtypes = [vp, i32, vp, i32, i32, C.POINTER(BeamRulesC), i32]
        L.wh_beam_window_continue.argtypes = [vp, i32]
        L.wh_beam_window_status.argtypes = [vp, C.POINTER(BeamWindowC)]
        L.wh_beam_window_records.argtypes = [vp, i32, i32, C.POINTER(BeamRecordC)]
        if os.environ.get("WH_TUNING"):          # kernel-variant mask for A/B runs and for testing a candidate variant
            L.wh_debug_set_tuning(int(os.environ["WH_TUNING"], 0))
        L.wh_device_info.argtypes = [i32, C.c_char_p, C.c_size_t, C.POINTER(C.c_uint64), C.POINTER(C.c_int)]
        L.wh_model_arena_bytes.restype = i64
        L.wh_model_arena_bytes.argtypes = [C.POINTER(HParamsC)]
        L.wh_model_create.argtypes = [C.POINTER(HParamsC), vp, i32, C.POINTER(vp)]
        L.wh_model_destroy.argtypes = [vp]
        L.wh_model_destroy.restype = None
        L.wh_model_set_tensor.argtypes = [vp, C.c_char_p, i32, C.POINTER(C.c_int32), i32, vp]
        L.wh_model_set_filters.argtypes = [vp, i32, i32, vp]
        L.wh_model_finalize.argtypes = [vp]
        L.wh_model_arena.argtypes = [vp, C.POINTER(vp), C.POINTER(i64)]
        L.wh_model_hparams.argtypes = [vp, C.POINTER(HParamsC)]
        L.wh_context_create.argtypes = [vp, i32, vp, C.POINTER(vp)]
        L.wh_context_create_hyp.argtypes = [vp, i32, i32, vp, C.POINTER(vp)]
        L.wh_context_bind.argtypes = [vp]
        L.wh_context_destroy.argtypes = [vp]
        L.wh_context_destroy.restype = None
        L.wh_context_set_flags.argtypes = [vp, C.c_uint32, i32]
        L.wh_context_set_audio_ctx.argtypes = [vp, i32]
        L.wh_context_memory.argtypes = [vp, C.POINTER(i64)]
        L.wh_context_synchronize.argtypes = [vp]
        L.wh_buffer_upload_async.argtypes = [vp, vp, vp, i64]
        L.wh_mel_spectrogram.argtypes = [vp, vp, i64, vp, C.POINTER(i64)]
        L.wh_encode.argtypes = [vp, vp, i32, i64, i64, vp]
        L.wh_encode_windows.argtypes = [vp, vp, i32]
        L.wh_decode.argtypes = [vp, vp, i32, i32, i32, vp, vp]
        L.wh_sample_best.argtypes = [vp, i32, i32, i32, C.POINTER(TokenDataC)]
        L.wh_beam_candidates.argtypes = [vp, i32, i32, i32, i32, C.POINTER(TokenDataC)]
        L.wh_reorder_self_cache.argtypes = [vp, i32, vp, i32]
        L.wh_debug_read.argtypes = [vp, C.c_char_p, i32, i32, vp, i64]
        L.wh_decode_greedy.argtypes = [vp, i32, vp, i32, i32, i32, i32, C.POINTER(TokenDataC)]
        L.wh_decode_window_start.argtypes = [vp, i32, vp, i32, i32, i32, i32]
        L.wh_decode_window_finish.argtypes = [vp, C.POINTER(TokenDataC)]
        L.wh_decode_window_start_ragged.argtypes = [vp, i32, vp, vp, i32, i32, i32, i32]
        L.wh_decode_window_ready.argtypes = [vp, i32, i32]
        L.wh_decode_window_continue.argtypes = [vp, i32]
        L.wh_decode_window_fetch.argtypes = [vp, i32, i32, C.POINTER(TokenDataC)]
        L.wh_mel_spectrogram_window.argtypes = [vp, vp, i64, i64, i64, i64, i32, vp]
        L.wh_mel_spectrogram_batch.argtypes = [vp, vp, i64, i64, i32, vp, i64]
        L.wh_debug_probe.argtypes = [vp, i32, i32, i32, i32, i32, i32, C.POINTER(C.c_float)]
        L.wh_profile_enable.argtypes = [vp, i32]
        L.wh_profile_read.argtypes = [vp, C.POINTER(ProfileEntryC), i32, C.POINTER(i32)]
        L.wh_op_mul_mat.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, i32]
        L.wh_op_mul_mat_gelu.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32]
        L.wh_op_layer_norm.argtypes = [vp, vp, vp, vp, vp, i32, i32]
        L.wh_op_flash_attention.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32]
        L.wh_op_soft_max.argtypes = [vp, vp, i32, i32]
        L.wh_op_decoder_attention.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32]
        L.wh_op_decoder_cross_attention.argtypes = [vp, vp, vp, vp, vp, vp, C.c_float, vp, vp, vp, i32, i32, i32, i32, i32]
        _lib = L
    return _lib


def check(rc: int):
    if rc != 0:
        raise WhisperHipError("libwhisper_hip rc=%d: %s" % (rc, lib().wh_last_error().decode(errors="replace")))


def set_option(name: str, value: int):
    """Integer knobs beyond the 32 tuning bits (csrc/kernels.h struct Options): dec_tile, vocab_decrows, enc_chunk, self_fuse_max_rows, self_nq."""
    check(lib().wh_debug_set_option(name.encode(), int(value)))


# the defaults of csrc/kernels.h struct Options (what a test restores an option to)
OPTION_DEFAULTS = {"dec_tile": 0, "dec_depth": 0, "dec_wide_rows": 1, "dec_deep_rows": 0, "vocab_decrows": 0, "enc_chunk": 128, "self_fuse_max_rows": 32, "self_nq": 0, "self_wave_min_rows": 32, "enc_exp": 5, "exact_enc_layers": -1, "exact_alt_order": 0, "gemm_mf16": 1, "dec_lds": 1, "dec_lds_ks": 2, "dec_split": 1, "cross_mfma": 1, "vocab_lds": 1, "beam_regs": 1, "reorder_group": 1, "gemm_big_min_rows": 8192}


def get_option(name: str) -> int:
    """The library's current value of an option (host only)."""
    v = C.c_int(0)
    check(lib().wh_debug_get_option(name.encode(), C.byref(v)))
    return v.value


def get_option_default(name: str) -> int:
    return OPTION_DEFAULTS[name]


def device_count() -> int:
    return lib().wh_device_count()


def device_info(device: int = 0):
    name = C.create_string_buffer(256)
    mem = C.c_uint64()
    cus = C.c_int()
    check(lib().wh_device_info(device, name, 256, C.byref(mem), C.byref(cus)))
    return dict(name=name.value.decode(), total_mem=mem.value, compute_units=cus.value)


def _hp_c(hp: gf.HParams) -> HParamsC:
    return HParamsC(*hp.as_list())


def arena_bytes(hp: gf.HParams) -> int:
    n = lib().wh_model_arena_bytes(C.byref(_hp_c(hp)))
    if n < 0:
        check(-1)
    return int(n)


class HipModel:
    """Weights resident in one packed device arena (ModelBuffers of the reference)."""

    def __init__(self, hp: gf.HParams, arena_ptr: int = 0, already_filled: bool = False, keepalive=None):
        self.hp = hp
        self.handle = C.c_void_p()
        self._keepalive = keepalive
        check(lib().wh_model_create(C.byref(_hp_c(hp)), C.c_void_p(arena_ptr) if arena_ptr else None, int(already_filled),
                                    C.byref(self.handle)))

    @classmethod
    def from_ggml(cls, model: gf.GgmlModel, arena_ptr: int = 0, keepalive=None) -> "HipModel":
        m = cls(model.hparams, arena_ptr, False, keepalive)
        L = lib()
        filt = np.ascontiguousarray(model.filters, np.float32)
        check(L.wh_model_set_filters(m.handle, filt.shape[0], filt.shape[1], filt.ctypes.data_as(C.c_void_p)))
        for name, a in model.tensors.items():
            a = np.ascontiguousarray(a)
            ne = (C.c_int32 * a.ndim)(*reversed(a.shape))
            check(L.wh_model_set_tensor(m.handle, name.encode(), a.ndim, ne, int(a.dtype == np.float16), a.ctypes.data_as(C.c_void_p)))
        check(L.wh_model_finalize(m.handle))
        return m

    @classmethod
    def from_file(cls, path: str) -> "HipModel":
        return cls.from_ggml(gf.read_model(path))

    def arena(self):
        p = C.c_void_p()
        n = C.c_int64()
        check(lib().wh_model_arena(self.handle, C.byref(p), C.byref(n)))
        return p.value, n.value

    def close(self):
        if self.handle:
            lib().wh_model_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HipContext:
    """Activations + KV caches for up to max_batch 30 s windows processed in lock step (WhisperContext of the reference)."""

    def __init__(self, model: HipModel, max_batch: int = 1, stream: int = 0, hypotheses: int = 1):
        """max_batch windows; `hypotheses` decoder sequences per window share the window's cross-attention K/V (then every
        decode entry point counts sequences = windows * hypotheses, window-major)."""
        self.model = model
        self.hp = model.hp
        self.max_batch = max_batch
        self.hypotheses = hypotheses
        self.handle = C.c_void_p()
        check(lib().wh_context_create_hyp(model.handle, max_batch, hypotheses, C.c_void_p(stream) if stream else None, C.byref(self.handle)))
        self.batch = 0

    def close(self):
        if self.handle:
            lib().wh_context_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_parity(self, n_threads: int):
        """n_threads > 0: emulate the CPU reference's FP16 thread-partitioned P.V accumulation; 0: FP32 fast path."""
        check(lib().wh_context_set_flags(self.handle, WH_FLAG_PARITY_PV if n_threads > 0 else 0, max(n_threads, 1)))

    def upload_async(self, dst_dev, src_host_pinned):
        """Enqueue a host -> device copy on the context's stream (torch tensors; the host one should be pinned)."""
        assert dst_dev.numel() * dst_dev.element_size() == src_host_pinned.numel() * src_host_pinned.element_size()
        check(lib().wh_buffer_upload_async(self.handle, C.c_void_p(dst_dev.data_ptr()), C.c_void_p(src_host_pinned.data_ptr()),
                                           dst_dev.numel() * dst_dev.element_size()))

    def vram_bytes(self) -> int:
        n = C.c_int64()
        check(lib().wh_context_memory(self.handle, C.byref(n)))
        return n.value

    def synchronize(self):
        check(lib().wh_context_synchronize(self.handle))

    @staticmethod
    def _wait_for_torch():
        """Inputs come from torch's current stream; the context runs on its own (capturable) stream."""
        import torch
        torch.cuda.current_stream().synchronize()

    def mel_spectrogram(self, pcm_dev, out_dev=None, sync: bool = True):
        """pcm_dev: torch float32 CUDA tensor [n]. Returns torch float32 [n_mel][n//160] on the device.
        sync=False skips the ordering with torch's stream (caller guarantees it, e.g. bench.py inside its timed region)."""
        import torch
        if sync:
            self._wait_for_torch()
        n = pcm_dev.numel()
        n_len = n // 160
        if out_dev is None:
            out_dev = torch.empty((self.hp.n_mels, n_len), dtype=torch.float32, device=pcm_dev.device)
        got = C.c_int64()
        check(lib().wh_mel_spectrogram(self.handle, C.c_void_p(pcm_dev.data_ptr()), n, C.c_void_p(out_dev.data_ptr()), C.byref(got)))
        if sync:
            self.synchronize()
        return out_dev

    def mel_spectrogram_batch(self, pcm_dev, out_dev=None, sync: bool = True):
        """pcm_dev: torch float32 CUDA tensor [batch][n] (contiguous rows): `batch` independent buffers, each normalised by its own maximum, in three
        launches. Returns torch float32 [batch][n_mel][n//160]."""
        import torch
        if sync:
            self._wait_for_torch()
        assert pcm_dev.dim() == 2 and pcm_dev.stride(1) == 1
        batch, n = pcm_dev.shape
        n_len = n // 160
        if out_dev is None:
            out_dev = torch.empty((batch, self.hp.n_mels, n_len), dtype=torch.float32, device=pcm_dev.device)
        assert out_dev.is_contiguous()
        check(lib().wh_mel_spectrogram_batch(self.handle, C.c_void_p(pcm_dev.data_ptr()), n, pcm_dev.stride(0), batch, C.c_void_p(out_dev.data_ptr()), self.hp.n_mels * n_len))
        if sync:
            self.synchronize()
        return out_dev

    def encode(self, mel_dev, offsets: Optional[Sequence[int]] = None, sync: bool = True):
        """mel_dev: torch float32 CUDA tensor [batch][n_mel][mel_len] (or [n_mel][mel_len])."""
        if sync:
            self._wait_for_torch()
        if mel_dev.dim() == 2:
            mel_dev = mel_dev.unsqueeze(0)
        assert mel_dev.is_contiguous() and mel_dev.shape[1] == self.hp.n_mels
        b, _, ln = mel_dev.shape
        offs = (C.c_int32 * b)(*(offsets if offsets is not None else [0] * b))
        check(lib().wh_encode(self.handle, C.c_void_p(mel_dev.data_ptr()), b, ln, self.hp.n_mels * ln, offs))
        self.batch = b

    def encode_windows(self, windows):
        """windows: list of (mel_dev [n_mel][len] torch float32 CUDA tensor or None, offset): one window per spectrogram."""
        self._wait_for_torch()
        arr = (MelWindowC * len(windows))()
        for i, (mel, off) in enumerate(windows):
            if mel is not None:
                assert mel.is_contiguous() and mel.shape[0] == self.hp.n_mels
                arr[i] = MelWindowC(mel.data_ptr(), mel.shape[1], off, 0)
        check(lib().wh_encode_windows(self.handle, arr, len(windows)))
        self.batch = len(windows)

    def decode(self, tokens, n_past: int, want_logits: bool = True, want_probs: bool = True):
        """tokens: int array [batch][n_tokens]. Returns (logits, probs) of the LAST token, each [batch][n_vocab] or None."""
        t = np.ascontiguousarray(tokens, np.int32)
        if t.ndim == 1:
            t = t[None, :]
        b, n = t.shape
        logits = np.empty((b, self.hp.n_vocab), np.float32) if want_logits else None
        probs = np.empty((b, self.hp.n_vocab), np.float32) if want_probs else None
        check(lib().wh_decode(self.handle, t.ctypes.data_as(C.c_void_p), b, n, n_past,
                              logits.ctypes.data_as(C.c_void_p) if want_logits else None,
                              probs.ctypes.data_as(C.c_void_p) if want_probs else None))
        return logits, probs

    def sample_best(self, batch: int, force_timestamp: bool = False, is_initial: bool = False):
        out = (TokenDataC * batch)()
        check(lib().wh_sample_best(self.handle, batch, int(force_timestamp), int(is_initial), out))
        return [dict(id=o.id, tid=o.tid, p=o.p, pt=o.pt, ptsum=o.ptsum) for o in out]

    def beam_candidates(self, batch: int, width: int, force_timestamp: bool = False, is_initial: bool = False):
        """The `width` best continuations of every sequence (candidate 0 = sample_best's token): dict of arrays [batch][width]."""
        out = (TokenDataC * (batch * width))()
        check(lib().wh_beam_candidates(self.handle, batch, width, int(force_timestamp), int(is_initial), out))
        a = np.frombuffer(out, dtype=np.dtype([("id", "<i4"), ("tid", "<i4"), ("p", "<f4"), ("pt", "<f4"), ("ptsum", "<f4")])).reshape(batch, width)
        return {k: a[k].copy() for k in a.dtype.names}

    def reorder_self_cache(self, parents, rows: int):
        """Sequence j continues sequence parents[j]: its self-attention cache rows [0, rows) are replaced by the parent's."""
        p = np.ascontiguousarray(parents, np.int32)
        check(lib().wh_reorder_self_cache(self.handle, len(p), p.ctypes.data_as(C.c_void_p), rows))

    # ---- beam search with the ranking on the device ----
    def beam_window_start(self, prompts, width: int, n_steps: int, rules=None):
        """prompts: int array [windows][n_prompt]; rules: list of dicts (seek, seekEnd, nMax, maxTokens, singleSegment, tokenBeg, tokenEot, forced) per window,
        or None = forced steps (no stop rules). Non-blocking."""
        t = np.ascontiguousarray(prompts, np.int32)
        if t.ndim == 1:
            t = t[None, :]
        w = t.shape[0]
        arr = (BeamRulesC * w)()
        for i in range(w):
            r = rules[i] if rules is not None else dict(forced=1)
            arr[i] = BeamRulesC(*[int(r.get(k, 0)) for k in ("seek", "seekEnd", "nMax", "maxTokens", "singleSegment", "tokenBeg", "tokenEot", "forced")])
        self._beam = (w, width)
        check(lib().wh_beam_window_start(self.handle, w, t.ctypes.data_as(C.c_void_p), t.shape[1], width, arr, n_steps))

    def beam_window_continue(self, n_steps: int):
        check(lib().wh_beam_window_continue(self.handle, n_steps))

    def beam_window_status(self):
        """Blocks; the search state per window: list of dicts (step, nLive, nFinished, done, live [..], finished [..])."""
        w, _ = self._beam
        arr = (BeamWindowC * w)()
        check(lib().wh_beam_window_status(self.handle, arr))
        hyp = lambda h: {k: getattr(h, k) for k in ("sum", "i", "hasTs", "seekDelta", "resultLen", "failed", "over", "nTok", "rec")}
        return [dict(step=a.step, nLive=a.nLive, nFinished=a.nFinished, done=a.done, live=[hyp(a.live[j]) for j in range(a.nLive)],
                     finished=[hyp(a.finished[j]) for j in range(a.nFinished)]) for a in arr]

    def beam_window_records(self, first: int, count: int):
        """Structured array [count][windows][width] with fields id, tid, p, pt, ptsum, parent, finished."""
        w, width = self._beam
        out = (BeamRecordC * (count * w * width))()
        check(lib().wh_beam_window_records(self.handle, first, count, out))
        dt = np.dtype([("id", "<i4"), ("tid", "<i4"), ("p", "<f4"), ("pt", "<f4"), ("ptsum", "<f4"), ("parent", "<i4"), ("finished", "<i4"), ("reserved", "<i4")])
        return np.frombuffer(out, dtype=dt).reshape(count, w, width).copy()

    @staticmethod
    def beam_chain(records, window: int, rec: int):
        """The token ids of the hypothesis whose last record is `rec` (= step * width + index), oldest first."""
        width = records.shape[2]
        ids = []
        while rec >= 0:
            r = records[rec // width, window, rec % width]
            ids.append(int(r["id"]))
            rec = int(r["parent"])
        return ids[::-1]

    def decode_greedy(self, first_tokens, n_past: int, n_steps: int, force_first_timestamp: bool = False,
                      first_is_initial: bool = False):
        """Device-side greedy loop: returns (ids [n_steps][batch] int32, token data list per step)."""
        t = np.ascontiguousarray(first_tokens, np.int32).reshape(-1)
        b = len(t)
        out = (TokenDataC * (b * n_steps))()
        check(lib().wh_decode_greedy(self.handle, b, t.ctypes.data_as(C.c_void_p), n_past, n_steps, int(force_first_timestamp),
                                     int(first_is_initial), out))
        ids = np.array([o.id for o in out], np.int32).reshape(n_steps, b)
        data = [[dict(id=out[s * b + i].id, tid=out[s * b + i].tid, p=out[s * b + i].p, pt=out[s * b + i].pt, ptsum=out[s * b + i].ptsum)
                 for i in range(b)] for s in range(n_steps)]
        return ids, data

    def decode_window_start(self, prompt_tokens, n_steps: int, force_first_timestamp: bool = True, first_is_initial: bool = True):
        """Non-blocking: prompt step + first sample + n_steps greedy steps are enqueued on the context's stream."""
        t = np.ascontiguousarray(prompt_tokens, np.int32)
        if t.ndim == 1:
            t = t[None, :]
        self._win = (t.shape[0], 1 + n_steps)
        check(lib().wh_decode_window_start(self.handle, t.shape[0], t.ctypes.data_as(C.c_void_p), t.shape[1], n_steps,
                                           int(force_first_timestamp), int(first_is_initial)))

    def decode_window_start_ragged(self, prompts, n_steps: int, force_first_timestamp: bool = True, first_is_initial: bool = True):
        """prompts: one token list per sequence, of different lengths (the streams of a batch scheduler). Non-blocking."""
        lens = np.asarray([len(p) for p in prompts], np.int32)
        n_max = int(lens.max())
        t = np.zeros((len(prompts), n_max), np.int32)
        for b, p in enumerate(prompts):
            t[b, :len(p)] = p
        self._win = (t.shape[0], 1 + n_steps)
        check(lib().wh_decode_window_start_ragged(self.handle, t.shape[0], t.ctypes.data_as(C.c_void_p), lens.ctypes.data_as(C.c_void_p), n_max, n_steps,
                                                  int(force_first_timestamp), int(first_is_initial)))

    def decode_window_ready(self, first: int, count: int) -> bool:
        rc = lib().wh_decode_window_ready(self.handle, first, count)
        if rc < 0:
            check(rc)
        return rc == 1

    def decode_window_finish(self):
        """Blocks; returns (ids [1 + n_steps][batch], probabilities of the chosen tokens, same shape)."""
        b, n = self._win
        out = (TokenDataC * (b * n))()
        check(lib().wh_decode_window_finish(self.handle, out))
        ids = np.array([o.id for o in out], np.int32).reshape(n, b)
        ps = np.array([o.p for o in out], np.float32).reshape(n, b)
        return ids, ps

    def decode_window_continue(self, n_steps: int):
        b, n = self._win
        check(lib().wh_decode_window_continue(self.handle, n_steps))
        self._win = (b, n + n_steps)

    def decode_window_fetch(self, first: int, count: int):
        """Blocks until samples [first, first + count) exist; returns their ids [count][batch]."""
        b, _ = self._win
        out = (TokenDataC * (b * count))()
        check(lib().wh_decode_window_fetch(self.handle, first, count, out))
        return np.array([o.id for o in out], np.int32).reshape(count, b)

    def decode_window_fetch_data(self, first: int, count: int):
        """Like decode_window_fetch, all five fields: dict of arrays [count][batch] (id, tid int32; p, pt, ptsum float32)."""
        b, _ = self._win
        out = (TokenDataC * (b * count))()
        check(lib().wh_decode_window_fetch(self.handle, first, count, out))
        a = np.frombuffer(out, dtype=np.dtype([("id", "<i4"), ("tid", "<i4"), ("p", "<f4"), ("pt", "<f4"), ("ptsum", "<f4")])).reshape(count, b)
        return {k: a[k].copy() for k in a.dtype.names}

    def mel_spectrogram_window(self, pcm_dev, frame0: int, n_frames: int, n_chunks: Optional[int] = None, reuse_previous_max: bool = False):
        """One window of a streamed spectrogram (MelStreamer semantics): torch float32 [n_mel][n_frames] on the device."""
        import torch
        self._wait_for_torch()
        n = pcm_dev.numel()
        out = torch.empty((self.hp.n_mels, n_frames), dtype=torch.float32, device=pcm_dev.device)
        check(lib().wh_mel_spectrogram_window(self.handle, C.c_void_p(pcm_dev.data_ptr()), n, frame0, n_frames,
                                              (n + 159) // 160 if n_chunks is None else n_chunks, int(reuse_previous_max), C.c_void_p(out.data_ptr())))
        self.synchronize()
        return out

    def set_audio_ctx(self, audio_ctx: int):
        """sFullParams::audio_ctx: encoder positions / cross-attention keys per window (0 = the model's n_audio_ctx)."""
        check(lib().wh_context_set_audio_ctx(self.handle, audio_ctx))

    def set_flags(self, flags: int, parity_threads: int = 1):
        check(lib().wh_context_set_flags(self.handle, flags, parity_threads))

    def probe(self, kind: int, variant: int, M: int = 0, N: int = 0, K: int = 0, iters: int = 100) -> float:
        ms = C.c_float()
        check(lib().wh_debug_probe(self.handle, kind, variant, M, N, K, iters, C.byref(ms)))
        return ms.value

    def profile(self, on: bool):
        check(lib().wh_profile_enable(self.handle, int(on)))

    def profile_read(self):
        """Per kernel class: calls, total GPU ms, algorithmic flops and bytes since profile(True)."""
        buf = (ProfileEntryC * 32)()
        n = C.c_int()
        check(lib().wh_profile_read(self.handle, buf, 32, C.byref(n)))
        return {buf[i].name.decode(): dict(calls=buf[i].calls, ms=buf[i].ms, flops=buf[i].flops, bytes=buf[i].bytes)
                for i in range(n.value)}

    def debug_read(self, what: str, layer: int = 0, rows: int = 0) -> np.ndarray:
        d = self.hp.n_audio_state
        b = self.batch
        if what == "exp-table":
            out = np.empty(0x5000, np.float32)
            check(lib().wh_debug_read(self.handle, what.encode(), 0, 0, out.ctypes.data_as(C.c_void_p), out.size))
            return out
        if what in ("exact-gelu-table", "exact-exp-table"):
            out = np.empty(65536, np.float32)
            check(lib().wh_debug_read(self.handle, what.encode(), 0, 0, out.ctypes.data_as(C.c_void_p), out.size))
            return out
        if what.startswith("exact:"):
            mult = 4 if what == "exact:h" else 1
            shape = (b, 2 * self.hp.n_audio_ctx, d) if what == "exact:conv1" else (b, self.hp.n_audio_ctx, d * mult)
            out = np.empty(shape, np.float32)
            check(lib().wh_debug_read(self.handle, what.encode(), 0, 0, out.ctypes.data_as(C.c_void_p), out.size))
            return out
        if what in ("logits", "probs"):
            shape = (rows, self.hp.n_vocab)         # rows = sequences of the last decode step
            rows = 0
        elif what in ("cross-k1", "cross-v1"):
            shape = (self.hp.n_audio_ctx, d)        # rows = window index
        elif what == "enc.temp1":
            shape = (b, 2 * self.hp.n_audio_ctx, d)
        elif what.startswith("dec-KQV"):
            shape = (rows, d)
        elif what in ("encode-out", "enc.layer0.in", "enc-KQV") or what.startswith("cross"):
            shape = (b, self.hp.n_audio_ctx, d)
        else:
            shape = (b * self.hypotheses, rows, d)
        out = np.empty(shape, np.float32)
        check(lib().wh_debug_read(self.handle, what.encode(), layer, rows, out.ctypes.data_as(C.c_void_p), out.size))
        return out
