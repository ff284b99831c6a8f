"""TEST INFRASTRUCTURE: the reference's graph (whisper.cpp:1084-1496, :1508-1872) driven from Python over the CPU build of
whisper_amd/csrc/exact_ops.h (tests/exact_cpu/ops.cpp -> tests/_build/libexact_cpu.so).

This is how the arithmetic of WH_FLAG_PARITY_EXACT is held against the reference's own code WITHOUT a GPU: the primitives the HIP kernels
are built from (dot order, FP16 multiply-add, double LayerNorm sums, table softmax) must reproduce oracle/_ref bit for bit on the CPU first
(tests/test_exact_cpu.py). The graph is oracle/whisper_np.py's, with every op replaced by its exact-order counterpart.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "tests", "_build")
SRC = os.path.join(ROOT, "tests", "exact_cpu", "ops.cpp")
HDR = os.path.join(ROOT, "whisper_amd", "csrc", "exact_ops.h")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"          # g++ 11 has no _Float16 in C++; the ROCm clang compiles the shared header for the host

F32, F16 = np.float32, np.float16
_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    os.makedirs(BUILD, exist_ok=True)
    so = os.path.join(BUILD, "libexact_cpu.so")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(SRC), os.path.getmtime(HDR)):
        subprocess.check_call([CLANG, "-O2", "-ffp-contract=off", "-mf16c", "-mfma", "-shared", "-fPIC", "-o", so, SRC])
    L = C.CDLL(so)
    _lib = L
    return L


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def tables():
    g = np.zeros(65536, np.uint16)
    e = np.zeros(65536, np.uint16)
    lib().x_tables(_p(g), _p(e))
    return g, e


class WhisperExact:
    """Encoder + decoder of one window with the exact-order ops. Same interface as oracle.whisper_np.WhisperNP where the tests need it."""

    def __init__(self, model):
        self.hp = model.hparams
        self.t = model.tensors
        self.gelu_t, self.exp_t = tables()
        self.cross_k, self.cross_v, self.self_k, self.self_v = [], [], [], []

    # ---- ops ----
    def mul_mat(self, w16, x):
        w16 = np.ascontiguousarray(w16, F16)
        x = np.ascontiguousarray(x, F32)
        out = np.zeros((x.shape[0], w16.shape[0]), F32)
        lib().x_mul_mat(_p(w16), w16.shape[0], w16.shape[1], _p(x), x.shape[0], _p(out))
        return out

    def layer_norm(self, x, w, b):
        x = np.ascontiguousarray(x, F32)
        out = np.zeros_like(x)
        lib().x_norm(_p(x), _p(np.ascontiguousarray(w, F32)), _p(np.ascontiguousarray(b, F32)), _p(out), x.shape[0], x.shape[1])
        return out

    def gelu(self, x):
        x = np.ascontiguousarray(x, F32)
        out = np.zeros_like(x)
        lib().x_gelu(_p(self.gelu_t), _p(x), _p(out), C.c_int64(x.size))
        return out

    def softmax(self, s):
        s = np.ascontiguousarray(s, F32).copy()
        lib().x_softmax(_p(self.exp_t), _p(s), s.shape[0], s.shape[1])
        return s

    def conv(self, w16, x, stride):
        w16 = np.ascontiguousarray(w16, F16)
        x = np.ascontiguousarray(x, F32)
        out = np.zeros((w16.shape[0], x.shape[1] // stride), F32)
        lib().x_conv(_p(w16), w16.shape[0], w16.shape[1], _p(x), x.shape[1], stride, _p(out))
        return out

    # ---- encoder ----
    def encode(self, mel, mel_offset=0, trace=None, n_layers=None):
        hp, t = self.hp, self.t
        n_ctx, d, H = hp.n_audio_ctx, hp.n_audio_state, hp.n_audio_head
        inp = np.zeros((hp.n_mels, 2 * n_ctx), F32)
        i0, i1 = min(mel_offset, mel.shape[1]), min(mel_offset + 2 * n_ctx, mel.shape[1])
        inp[:, :i1 - i0] = mel[:, i0:i1]
        cur = self.conv(t["encoder.conv1.weight"], inp, 1)
        cur = self.gelu((t["encoder.conv1.bias"].reshape(-1, 1).astype(F32) + cur).astype(F32))
        if trace is not None:
            trace["conv1"] = cur.T.copy()
        cur = self.conv(t["encoder.conv2.weight"], cur, 2)
        cur = self.gelu((t["encoder.conv2.bias"].reshape(-1, 1).astype(F32) + cur).astype(F32))
        x = (t["encoder.positional_embedding"][:n_ctx].astype(F32) + cur.T).astype(F32)
        for il in range(hp.n_audio_layer if n_layers is None else n_layers):
            p = f"encoder.blocks.{il}"
            if trace is not None:
                trace["x_in"] = x.copy()
            cur = self.layer_norm(x, t[p + ".attn_ln.weight"], t[p + ".attn_ln.bias"])
            if trace is not None:
                trace["ln1"] = cur.copy()
            q = (self.mul_mat(t[p + ".attn.query.weight"], cur) + t[p + ".attn.query.bias"].astype(F32)).astype(F32)
            k = self.mul_mat(t[p + ".attn.key.weight"], cur)
            v = (self.mul_mat(t[p + ".attn.value.weight"], cur) + t[p + ".attn.value.bias"].astype(F32)).astype(F32)
            q16, k16, v16 = (np.ascontiguousarray(a.astype(F16)) for a in (q, k, v))
            kqv = np.zeros((n_ctx, d), F32)
            for h in range(H):
                o = h * 64
                lib().x_flash_attn(C.c_void_p(q16.ctypes.data + 2 * o), C.c_void_p(k16.ctypes.data + 2 * o), C.c_void_p(v16.ctypes.data + 2 * o),
                                   d, n_ctx, _p(self.exp_t), C.c_void_p(kqv.ctypes.data + 4 * o), d)
            if trace is not None:
                trace.update(q=q.copy(), k=k.copy(), v=v.copy(), kqv=kqv.copy())
            cur = (self.mul_mat(t[p + ".attn.out.weight"], kqv) + t[p + ".attn.out.bias"].astype(F32)).astype(F32)
            x = (cur + x).astype(F32)
            if trace is not None:
                trace["x_attn"] = x.copy()
            cur = self.layer_norm(x, t[p + ".mlp_ln.weight"], t[p + ".mlp_ln.bias"])
            if trace is not None:
                trace["ln2"] = cur.copy()
            cur = self.gelu((self.mul_mat(t[p + ".mlp.0.weight"], cur) + t[p + ".mlp.0.bias"].astype(F32)).astype(F32))
            if trace is not None:
                trace["h"] = cur.copy()
            cur = (self.mul_mat(t[p + ".mlp.2.weight"], cur) + t[p + ".mlp.2.bias"].astype(F32)).astype(F32)
            x = (cur + x).astype(F32)
        if trace is not None:
            trace["x"] = x.copy()
        if n_layers is not None:
            return x
        out = self.layer_norm(x, t["encoder.ln_post.weight"], t["encoder.ln_post.bias"])
        ks = F32(np.power(np.float64(F32(d) / F32(H)), -0.25))
        self.cross_k, self.cross_v = [], []
        for il in range(hp.n_text_layer):
            p = f"decoder.blocks.{il}.cross_attn"
            kc = (self.mul_mat(t[p + ".key.weight"], out) * ks).astype(F32)
            vc = (self.mul_mat(t[p + ".value.weight"], out) + t[p + ".value.bias"].astype(F32)).astype(F32)
            self.cross_k.append(np.ascontiguousarray(kc.astype(F16)))
            self.cross_v.append(np.ascontiguousarray(vc.astype(F16)))
        self.self_k = [np.zeros((hp.n_text_ctx, d), F16) for _ in range(hp.n_text_layer)]
        self.self_v = [np.zeros((hp.n_text_ctx, d), F16) for _ in range(hp.n_text_layer)]
        return out

    # ---- decoder ----
    def _attention(self, q, K16, V16, n_keys, mask_past, n_threads):
        H = self.hp.n_text_head
        d = H * 64
        N = q.shape[0]
        q = np.ascontiguousarray(q, F32)
        out = np.zeros((N, d), F32)
        for h in range(H):
            o = h * 64
            S = np.zeros((N, n_keys), F32)
            lib().x_kq(C.c_void_p(K16.ctypes.data + 2 * o), d, n_keys, C.c_void_p(q.ctypes.data + 4 * o), d, N, _p(S))
            if mask_past is not None:
                j = np.arange(n_keys)[None, :]
                i = np.arange(N)[:, None]
                S = np.where(j > mask_past + i, F32(-np.inf), S).astype(F32)
            P = self.softmax(S)
            lib().x_pv_mad(_p(P), C.c_void_p(V16.ctypes.data + 2 * o), d, n_keys, N, n_threads, C.c_void_p(out.ctypes.data + 4 * o), d)
        return out

    def decode(self, tokens, n_past, n_threads=1):
        hp, t = self.hp, self.t
        d, H = hp.n_text_state, hp.n_text_head
        N = len(tokens)
        tok = np.asarray(tokens, np.int64)
        x = (t["decoder.token_embedding.weight"][tok].astype(F32) + t["decoder.positional_embedding"][n_past:n_past + N].astype(F32)).astype(F32)
        s = F32(np.power(np.float64(F32(d) / F32(H)), -0.25))
        for il in range(hp.n_text_layer):
            p = f"decoder.blocks.{il}"
            cur = self.layer_norm(x, t[p + ".attn_ln.weight"], t[p + ".attn_ln.bias"])
            q = ((self.mul_mat(t[p + ".attn.query.weight"], cur) + t[p + ".attn.query.bias"].astype(F32)).astype(F32) * s).astype(F32)
            k = (self.mul_mat(t[p + ".attn.key.weight"], cur) * s).astype(F32)
            v = (self.mul_mat(t[p + ".attn.value.weight"], cur) + t[p + ".attn.value.bias"].astype(F32)).astype(F32)
            self.self_k[il][n_past:n_past + N] = k.astype(F16)
            self.self_v[il][n_past:n_past + N] = v.astype(F16)
            a = self._attention(q, self.self_k[il], self.self_v[il], n_past + N, n_past, n_threads)
            cur = (self.mul_mat(t[p + ".attn.out.weight"], a) + t[p + ".attn.out.bias"].astype(F32)).astype(F32)
            x = (cur + x).astype(F32)
            cur = self.layer_norm(x, t[p + ".cross_attn_ln.weight"], t[p + ".cross_attn_ln.bias"])
            q = ((self.mul_mat(t[p + ".cross_attn.query.weight"], cur) + t[p + ".cross_attn.query.bias"].astype(F32)).astype(F32) * s).astype(F32)
            a = self._attention(q, self.cross_k[il], self.cross_v[il], hp.n_audio_ctx, None, n_threads)
            cur = (self.mul_mat(t[p + ".cross_attn.out.weight"], a) + t[p + ".cross_attn.out.bias"].astype(F32)).astype(F32)
            x = (cur + x).astype(F32)
            cur = self.layer_norm(x, t[p + ".mlp_ln.weight"], t[p + ".mlp_ln.bias"])
            cur = self.gelu((self.mul_mat(t[p + ".mlp.0.weight"], cur) + t[p + ".mlp.0.bias"].astype(F32)).astype(F32))
            cur = (self.mul_mat(t[p + ".mlp.2.weight"], cur) + t[p + ".mlp.2.bias"].astype(F32)).astype(F32)
            x = (cur + x).astype(F32)
        cur = self.layer_norm(x, t["decoder.ln.weight"], t["decoder.ln.bias"])
        logits = self.mul_mat(t["decoder.token_embedding.weight"], cur)
        probs = self.softmax(logits)
        return logits, probs
