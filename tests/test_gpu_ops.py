"""GPU parity tests, op level: each HIP kernel against the oracle restatement on the same seeded inputs.

These are the counterparts of the reference's op-level A/B tests (Whisper/ML/tensorOpsTests.cpp:10-183: testMulMat,
testFlashAttention, testConvolution against live ggml tensors). Every call goes through the C ABI (include/whisper_hip.h).
Inputs are teacher-forced (taken from the oracle), so differences do not compound and the bounds are tight.
"""
import ctypes as C

import numpy as np
import pytest

torch = pytest.importorskip("torch")

from oracle import whisper_np as wn  # noqa: E402
from whisper_amd import binding  # noqa: E402

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def ptr(t):
    return C.c_void_p(t.data_ptr())


def report(name, got, want):
    d = np.abs(got.astype(np.float64) - want.astype(np.float64))
    print("%-34s max|want|=%9.4f maxdiff=%.3e meandiff=%.3e" % (name, np.abs(want).max(), d.max(), d.mean()))
    return d


@pytest.mark.parametrize("M,N,K", [(1500, 128, 128), (300, 384, 512), (128, 128, 64), (257, 130, 192), (7, 128, 128),
                                   (32, 1024, 1024), (3, 51864, 128), (64, 4096, 1024), (1, 96, 4096),
                                   # gemv (decode): one and two MFMA column tiles, 4-row variant (K >= 2048), odd N, ragged M
                                   (16, 1024, 1024), (17, 1024, 1024), (21, 3072, 1024), (28, 1024, 4096), (21, 51865, 384),
                                   # skinny kernel (K not a multiple of 128)
                                   (21, 512, 192), (5, 130, 64)])
def test_mul_mat(M, N, K):
    """out = fp16(a) . w^T + bias + residual, FP32 accumulate (ggml_mul_mat with an FP16 weight, ggml.c:4588-4687)."""
    rng = np.random.default_rng(M * 7 + N)
    a = rng.standard_normal((M, K)).astype(np.float16)
    w = (0.05 * rng.standard_normal((N, K))).astype(np.float16)
    bias = rng.standard_normal(N).astype(np.float32)
    res = rng.standard_normal((M, N)).astype(np.float32)
    want = (a.astype(np.float64) @ w.astype(np.float64).T + bias + res).astype(np.float32)
    ad, wd, bd, rd = dev(a), dev(w), dev(bias), dev(res)
    out = torch.full((M, N), float("nan"), dtype=torch.float32, device="cuda")
    binding.check(binding.lib().wh_op_mul_mat(None, ptr(ad), ptr(wd), ptr(bd), ptr(rd), ptr(out), M, N, K))
    torch.cuda.synchronize()
    d = report("mul_mat %dx%dx%d" % (M, N, K), out.cpu().numpy(), want)
    # FP32 accumulation of K products of magnitude ~0.05: round-off only
    assert d.max() < 2e-5 * max(1.0, np.sqrt(K / 128))


def test_mul_mat_is_transpose_correct():
    """A = identity against an asymmetric W catches a swapped C layout."""
    K = 128
    a = np.eye(K, dtype=np.float16)
    w = (np.arange(K * K, dtype=np.float32).reshape(K, K) % 97 / 97.0).astype(np.float16)
    out = torch.zeros((K, K), dtype=torch.float32, device="cuda")
    ad, wd = dev(a), dev(w)
    binding.check(binding.lib().wh_op_mul_mat(None, ptr(ad), ptr(wd), None, None, ptr(out), K, K, K))
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), w.astype(np.float32).T)


@pytest.mark.parametrize("M,N,K", [(1500, 512, 128), (5, 512, 128), (21, 4096, 1024), (21, 512, 192)])
def test_mul_mat_gelu(M, N, K, golden):
    """mulMat + addRepeatGelu: fp16 table GELU of (acc + bias) (ggml.c:1003-1021)."""
    rng = np.random.default_rng(5)
    a = rng.standard_normal((M, K)).astype(np.float16)
    w = (0.2 * rng.standard_normal((N, K))).astype(np.float16)
    bias = rng.standard_normal(N).astype(np.float32)
    pre = (a.astype(np.float32) @ w.astype(np.float32).T + bias).astype(np.float32)
    table = golden["table_gelu"].view(np.float16)
    want = table[pre.astype(np.float16).view(np.uint16)].astype(np.float32)      # the reference's own table
    out = torch.zeros((M, N), dtype=torch.float16, device="cuda")
    ad, wd, bd = dev(a), dev(w), dev(bias)
    binding.check(binding.lib().wh_op_mul_mat_gelu(None, ptr(ad), ptr(wd), ptr(bd), ptr(out), M, N, K))
    torch.cuda.synchronize()
    got = out.cpu().numpy().astype(np.float32)
    d = report("mul_mat_gelu", got, want)
    # differences only where the FP32 accumulation order flips the FP16 rounding of the GELU argument, or 1 ulp of the table
    assert (d > 0).mean() < 0.02 and (d <= np.maximum(4e-3, np.abs(want) * 2.0 ** -10)).all()     # at most one FP16 ulp


def test_gelu_table_exhaustive(golden):
    """Every finite FP16 input through the GEMM epilogue's GELU against the reference's 65536-entry table."""
    x = np.arange(65536, dtype=np.uint16).view(np.float16)
    fin = np.isfinite(x)
    xs = x[fin]
    M = len(xs)
    K = 64
    a = np.zeros((M, K), np.float16)
    a[:, 0] = xs
    w = np.zeros((64, K), np.float16)
    w[:, 0] = 1.0
    bias = np.zeros(64, np.float32)
    out = torch.zeros((M, 64), dtype=torch.float16, device="cuda")
    ad, wd, bd = dev(a), dev(w), dev(bias)
    binding.check(binding.lib().wh_op_mul_mat_gelu(None, ptr(ad), ptr(wd), ptr(bd), ptr(out), M, 64, K))
    torch.cuda.synchronize()
    got = out.cpu().numpy()[:, 3].view(np.uint16).astype(np.int32)
    want = golden["table_gelu"][fin].astype(np.int32)
    diff = np.abs(got - want)
    diff[(got & 0x7fff) + (want & 0x7fff) == 0] = 0      # x = -0.0 reaches the epilogue as +0.0 (the FP32 accumulator of -0*1 + 0*0 is +0)
    print("gelu table: %d of %d entries differ, max ulp %d" % ((diff > 0).sum(), M, diff.max()))
    for i in np.nonzero(diff)[0][:12]:
        print("   x=%r (0x%04x) got 0x%04x want 0x%04x" % (float(xs[i]), int(xs[i:i + 1].view(np.uint16)[0]), got[i], want[i]))
    assert diff.max() <= 1 and (diff > 0).mean() < 2e-3


@pytest.mark.parametrize("rows,d", [(1500, 128), (7, 1024), (33, 1280), (1, 384)])
def test_layer_norm(rows, d):
    rng = np.random.default_rng(rows + d)
    x = (rng.standard_normal((rows, d)) * 3 + 0.5).astype(np.float32)
    w = (1 + 0.1 * rng.standard_normal(d)).astype(np.float32)
    b = (0.1 * rng.standard_normal(d)).astype(np.float32)
    want = wn.r16(wn.layer_norm(x, w, b))
    out = torch.zeros((rows, d), dtype=torch.float16, device="cuda")
    xd, wd, bd = dev(x), dev(w), dev(b)
    binding.check(binding.lib().wh_op_layer_norm(None, ptr(xd), ptr(wd), ptr(bd), ptr(out), rows, d))
    torch.cuda.synchronize()
    got = out.cpu().numpy().astype(np.float32)
    d_ = report("layer_norm %dx%d" % (rows, d), got, want)
    assert (d_ > 0).mean() < 5e-3 and d_.max() < 5e-3          # only FP16 rounding flips (1 ulp at |x| <= 4 is 3.9e-3)


@pytest.mark.parametrize("rows,cols", [(4, 51864), (3, 1500), (2, 7)])
def test_soft_max(rows, cols):
    rng = np.random.default_rng(cols)
    x = (rng.standard_normal((rows, cols)) * 3).astype(np.float32)
    x[0, cols // 2] = -np.inf
    want = wn.softmax_table(x)
    xd = dev(x)
    binding.check(binding.lib().wh_op_soft_max(None, ptr(xd), rows, cols))
    torch.cuda.synchronize()
    got = xd.cpu().numpy()
    d = report("soft_max %dx%d" % (rows, cols), got, want)
    assert got[0, cols // 2] == 0.0
    # exp16 of identical FP16 arguments: at most 1 FP16 ulp apart where expf and glibc exp round differently
    assert d.max() < 1e-3 * want.max() and (d > 0).mean() < 0.01


@pytest.mark.parametrize("batch,heads,T", [(1, 2, 1500), (2, 3, 200), (1, 1, 64), (1, 2, 777),
                                           (2, 4, 300), (4, 6, 130)])        # batch x heads a multiple of 8: XCD-grouped block order
def test_flash_attention(batch, heads, T):
    """Unmasked encoder attention against ggml_flash_attn_f16 semantics (ggml.c:5912-6097)."""
    rng = np.random.default_rng(T)
    D = 64
    q = (rng.standard_normal((batch * heads, T, D)) * 1.5).astype(np.float16)
    k = (rng.standard_normal((batch * heads, T, D)) * 1.5).astype(np.float16)
    v = rng.standard_normal((batch * heads, T, D)).astype(np.float16)
    Tpad = (T + 255) // 256 * 256
    # V in the kernel's operand order (whisper_hip.h, gemm.hip vFragIndex): blocks of 16 keys x 32 dims, lane-major 8-half fragments
    key, dd = np.meshgrid(np.arange(T), np.arange(D), indexing="ij")
    idx = (((key >> 4) * 2 + (dd >> 5)) * 64 + ((key >> 2) & 1) * 32 + (dd & 31)) * 8 + ((key >> 3) & 1) * 4 + (key & 3)
    vT = np.zeros((batch * heads, D * Tpad), np.float16)
    vT[:, idx.ravel()] = v.reshape(batch * heads, T * D)
    want = np.zeros((batch, T, heads * D), np.float32)
    for bh in range(batch * heads):
        S = ((q[bh].astype(np.float32) @ k[bh].astype(np.float32).T) * np.float32(0.125)).astype(np.float32)
        P = wn.softmax_table(S)
        o = (wn.r16(P) @ v[bh].astype(np.float32)).astype(np.float32)
        want[bh // heads, :, (bh % heads) * D:(bh % heads + 1) * D] = o
    out = torch.full((batch, T, heads * D), float("nan"), dtype=torch.float16, device="cuda")
    qd, kd, vd = dev(q), dev(k), dev(vT)
    binding.check(binding.lib().wh_op_flash_attention(None, ptr(qd), ptr(kd), ptr(vd), ptr(out), batch, heads, T))
    torch.cuda.synchronize()
    got = out.cpu().numpy().astype(np.float32)
    assert np.isfinite(got).all()
    d = report("flash_attention b%d h%d T%d" % (batch, heads, T), got, wn.r16(want))
    # S differs by FP32 summation order only; that can flip the FP16 rounding of (S - max) for a few keys, each worth
    # <= 1.6 % of that key's probability, plus the final FP16 rounding of the output
    assert d.max() < 6e-3 and d.mean() < 2e-4


def test_exp_table_exhaustive(golden):
    """exp16 (the FP16 exp table semantics, ggml.c:1382) for EVERY non-positive finite FP16 input, through the row softmax:
    with max = 0 the arguments are exactly the FP16 inputs and p = e * float(1 / double sum)."""
    bits = np.arange(0x8000, 0xFC00, dtype=np.uint16)                  # -0.0 ... -65504
    x = np.concatenate([[np.float32(0.0)], bits.view(np.float16).astype(np.float32)])
    table = golden["table_exp"].view(np.float16)
    e = np.concatenate([[np.float32(1.0)], table[bits].astype(np.float32)])
    inv = np.float32(1.0 / e.astype(np.float64).sum())
    want = (e * inv).astype(np.float32)
    xd = dev(x[None, :].copy())
    binding.check(binding.lib().wh_op_soft_max(None, ptr(xd), 1, len(x)))
    torch.cuda.synchronize()
    got = xd.cpu().numpy()[0]
    rel = np.abs(got - want) / np.maximum(want, 1e-30)
    bad = np.nonzero(got != want)[0]
    print("exp table: %d of %d entries differ, worst relative difference %.3e" % (len(bad), len(x), rel.max() if len(bad) else 0.0))
    for i in bad[:8]:
        print("   x=%r got %r want %r" % (float(x[i]), float(got[i]), float(want[i])))
    # a differing entry is one FP16 ulp of e (<= 2^-10 relative); the sum (hence inv) may move by one FP32 ulp with it
    assert len(bad) <= 0.002 * len(x) or rel.max() < 2e-7
    assert rel.max() < 1.1e-3
