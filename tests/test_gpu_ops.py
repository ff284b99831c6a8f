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
from whisper_amd import binding, ggml_format as gf  # noqa: E402

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


@pytest.mark.parametrize("batch,heads,T", [(1, 2, 1500), (2, 3, 200), (1, 1, 64), (1, 2, 777), (1, 1, 1), (1, 2, 129), (1, 1, 1536),
                                           (2, 4, 300), (4, 6, 130), (2, 16, 1500)])        # batch x heads a multiple of 8: XCD-grouped block order
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
    qd, kd, vd = dev(q), dev(k), dev(vT)
    L = binding.lib()
    # every kernel variant: two sweeps (default: unnormalised e into P.V, O / sum at the end), three sweeps (the reference's
    # fp16(e / sum) operand), and the scores kept in registers
    results = {}
    for name, mask in (("two-sweep", (binding.TUNE_DEFAULT | binding.TUNE_ATTN_ENC_2SWEEP) & ~binding.TUNE_ATTN_ENC_TABLE),
                       ("table", binding.TUNE_DEFAULT | binding.TUNE_ATTN_ENC_2SWEEP | binding.TUNE_ATTN_ENC_TABLE | binding.TUNE_ATTN_ENC_TABLE_ANY),
                       ("valu-exp", binding.TUNE_DEFAULT | binding.TUNE_ATTN_ENC_2SWEEP | binding.TUNE_ATTN_ENC_TABLE | binding.TUNE_ATTN_ENC_TABLE_ANY),
                       ("wide", binding.TUNE_DEFAULT | binding.TUNE_ATTN_ENC_2SWEEP | binding.TUNE_ATTN_ENC_TABLE | binding.TUNE_ATTN_ENC_TABLE_ANY),
                       ("wide-online", binding.TUNE_DEFAULT | binding.TUNE_ATTN_ENC_2SWEEP | binding.TUNE_ATTN_ENC_TABLE | binding.TUNE_ATTN_ENC_TABLE_ANY),
                       ("wide-online-fma", binding.TUNE_DEFAULT | binding.TUNE_ATTN_ENC_2SWEEP | binding.TUNE_ATTN_ENC_TABLE | binding.TUNE_ATTN_ENC_TABLE_ANY),
                       ("three-sweep", binding.TUNE_DEFAULT & ~binding.TUNE_ATTN_ENC_2SWEEP),
                       ("scores-in-registers", binding.TUNE_DEFAULT & ~binding.TUNE_ATTN_ENC_F)):
        L.wh_debug_set_tuning(mask)
        # attentionEncT<0> (the table in LDS) / <1> (v_exp_f32) / attentionEncW<false> (64 query rows per wave) / <true> (one sweep, lazily raised running maximum)
        binding.set_option("enc_exp", {"valu-exp": 1, "wide": 2, "wide-online": 3, "wide-online-fma": 5}.get(name, 0))
        try:
            out = torch.full((batch, T, heads * D), float("nan"), dtype=torch.float16, device="cuda")
            binding.check(L.wh_op_flash_attention(None, ptr(qd), ptr(kd), ptr(vd), ptr(out), batch, heads, T))
            torch.cuda.synchronize()
        finally:
            L.wh_debug_set_tuning(binding.TUNE_DEFAULT)
            binding.set_option("enc_exp", binding.get_option_default("enc_exp"))
        got = out.cpu().numpy().astype(np.float32)
        assert np.isfinite(got).all()
        d = report("flash_attention %s b%d h%d T%d" % (name, batch, heads, T), got, wn.r16(want))
        # S differs by FP32 summation order only; that can flip the FP16 rounding of (S - max) for a few keys, each worth
        # <= 1.6 % of that key's probability, plus the final FP16 rounding of the output
        assert d.max() < 6e-3 and d.mean() < 2e-4
        results[name] = got
    # attentionEncT = the two-sweep kernel with the exponential looked up in the reference's own table (LDS) instead of computed by exp16:
    # the scores are the same instruction sequence, so the outputs differ only where exp16 and the table differ (~1e-4 of the inputs,
    # one FP16 ulp of e there) -- almost every output element is identical
    same = float((results["table"] == results["two-sweep"]).mean())
    dt = np.abs(results["table"] - results["two-sweep"]).max()
    print("table kernel vs exp16 kernel: %.4f of the outputs identical, max difference %.2e" % (same, dt))
    assert same > 0.9 and dt < 2e-3
    # attentionEncT<1> (the timed path since round 6): the same sweeps with e = fp16( 2^( fp16( s - max ) * log2 e ) ) from v_exp_f32 instead of the table: the
    # FP32 product and the instruction's last bit move e by one FP16 ulp in ~0.3 % of the entries; an output sums hundreds of them
    same_v = float((results["valu-exp"] == results["table"]).mean())
    dv = np.abs(results["valu-exp"] - results["table"])
    print("VALU-exponential kernel vs table kernel: %.4f of the outputs identical, max difference %.2e, mean %.2e" % (same_v, dv.max(), dv.mean()))
    assert dv.max() < 2e-3 and dv.mean() < 2e-5
    # attentionEncW: the same two sweeps with two query groups per wave -- the same MFMAs on the same operands per query row, so it must equal attentionEncT<1> exactly;
    # its one-sweep form rounds the exponentials' arguments against the RUNNING maximum: each e moves by an FP16 rounding, an output sums hundreds of them
    assert np.array_equal(results["wide"], results["valu-exp"])
    do = np.abs(results["wide-online"] - results["wide"])
    print("one-sweep (lazy running maximum) vs two-sweep: max difference %.2e, mean %.2e; against the reference's softmax: max %.2e mean %.2e"
          % (do.max(), do.mean(), np.abs(results["wide-online"] - wn.r16(want)).max(), np.abs(results["wide-online"] - wn.r16(want)).mean()))
    assert do.max() < 4e-3 and do.mean() < 1e-4
    # ... and the timed default (round 6): the exponential's argument is not rounded to FP16 first (one FMA per score instead of two subtractions, a packed
    # conversion and two mixed FMAs per pair): each e moves by at most the reference's own argument-rounding error, |x| 2^-11 relative
    df = np.abs(results["wide-online-fma"] - results["wide-online"])
    dr = np.abs(results["wide-online-fma"] - wn.r16(want))
    print("argument not rounded to FP16: vs the rounded form max %.2e mean %.2e; against the reference's softmax max %.2e mean %.2e" % (df.max(), df.mean(), dr.max(), dr.mean()))
    assert df.max() < 4e-3 and df.mean() < 1.5e-4 and dr.max() < 6e-3 and dr.mean() < 2e-4


def test_exp_table_in_the_arena(golden):
    """The model's copy of the reference's exponential table (ggml.c:1375-1385, what attentionEncT looks e up in): entry i must be the
    reference's table_exp_f16 entry of the FP16 number -|bits i|, all 0x5000 of them, bit for bit; from 0x4C56 on they are 0."""
    m = binding.HipModel.from_ggml(gf.synth_model("test-d128", seed=3))
    ctx = binding.HipContext(m, 1)
    got = ctx.debug_read("exp-table")
    want = golden["table_exp"].view(np.float16)[0x8000:0x8000 + 0x5000].astype(np.float32)
    assert np.array_equal(got, want)
    assert got[0] == 1.0 and got[0x4C55] > 0.0 and not got[0x4C56:].any()
    # and fp16( expf( x ) ) of everything the table does not hold is 0 as well: the clamp of the index is exact
    assert not golden["table_exp"].view(np.float16)[0x8000 + 0x5000:0xFC01].astype(np.float32).any()
    ctx.close()
    m.close()


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


# ----------------------------------------------------------------------------------------------------------------------
# round 2: the tile configurations the bench actually runs, the 128-row decode kernel, decoder attention on its own
# ----------------------------------------------------------------------------------------------------------------------
def _torch_ref_mul_mat(a16, w16, bias=None, res=None):
    """float64 reference computed on the GPU (numpy would take minutes at these sizes); test plumbing only."""
    r = a16.double() @ w16.double().T
    if bias is not None:
        r = r + bias.double()
    if res is not None:
        r = r + res.double()
    return r


@pytest.mark.parametrize("M,N,K", [(16500, 4608, 1024), (16400, 4096, 1024), (17000, 5120, 192), (16384 + 77, 4672, 256)])
def test_mul_mat_big_tiles(M, N, K):
    """M >= 16384 rows and >= 300 256x256 tiles: the 256x256x64 direct-to-LDS instance with the banded block walk
    (gemm.hip CfgGlBig, launchGemm) that the encoder of a 28-window batch runs -- plain FP32 epilogue with bias + residual,
    ragged M and N (clamped edge tiles) included."""
    g = torch.Generator(device="cuda").manual_seed(M + N)
    a = torch.randn((M, K), generator=g, device="cuda").half()
    w = (0.05 * torch.randn((N, K), generator=g, device="cuda")).half()
    bias = torch.randn(N, generator=g, device="cuda")
    res = torch.randn((M, N), generator=g, device="cuda")
    want = _torch_ref_mul_mat(a, w, bias, res)
    out = torch.full((M, N), float("nan"), dtype=torch.float32, device="cuda")
    binding.check(binding.lib().wh_op_mul_mat(None, ptr(a), ptr(w), ptr(bias), ptr(res), ptr(out), M, N, K))
    torch.cuda.synchronize()
    d = (out.double() - want).abs()
    print("mul_mat big %dx%dx%d maxdiff %.3e meandiff %.3e" % (M, N, K, float(d.max()), float(d.mean())))
    assert bool(torch.isfinite(out).all())
    assert float(d.max()) < 2e-5 * max(1.0, np.sqrt(K / 128))


@pytest.mark.parametrize("M,N,K", [(16500, 4608, 1024), (1500, 1024, 1024), (3000, 384, 1536), (16384 + 77, 4672, 256)])
def test_mul_mat_fragment_prefetch_is_bit_identical(M, N, K):
    """TUNE_GEMM_FRAGPF only changes WHEN the MFMA operands are read from LDS (two register sets, counted waits, the
    direct-to-LDS loads issued as assembly): the products and their order are the same, so both K loops must give the same
    bits -- on the 128x128x32 and the 256x256x64 instance, ragged edges included. A missing wait shows up here."""
    g = torch.Generator(device="cuda").manual_seed(M + N + 1)
    a = torch.randn((M, K), generator=g, device="cuda").half()
    w = (0.05 * torch.randn((N, K), generator=g, device="cuda")).half()
    bias = torch.randn(N, generator=g, device="cuda")
    L = binding.lib()
    outs = []
    try:
        for mask in (binding.TUNE_DEFAULT | binding.TUNE_GEMM_FRAGPF, binding.TUNE_DEFAULT & ~binding.TUNE_GEMM_FRAGPF):
            L.wh_debug_set_tuning(mask)
            for rep in range(3):        # races are intermittent: a few launches each
                out = torch.full((M, N), float("nan"), dtype=torch.float32, device="cuda")
                binding.check(L.wh_op_mul_mat(None, ptr(a), ptr(w), ptr(bias), None, ptr(out), M, N, K))
                outs.append(out)
    finally:
        L.wh_debug_set_tuning(binding.TUNE_DEFAULT)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(outs[0]).all())
    for o in outs[1:]:
        assert torch.equal(o, outs[0])


@pytest.mark.parametrize("M,N,K", [(16500, 4608, 1024), (17000, 5120, 192), (16384 + 77, 4672, 256), (16400, 1256, 256), (16400, 1250, 320),
                                   (33000, 1024, 4096)])
def test_mul_mat_4wave_kernel_equals_8wave(M, N, K):
    """gemmTiled4 (one wave per SIMD, 128 x 128 outputs per wave, a hand-pipelined K loop with one barrier per K tile) and gemmTiled8
    with the lean interior-tile epilogue (TUNE_GEMM_FAST_EPI) against gemmTiled8 with the general block epilogues: the same MFMAs in the
    same order per output and the same arithmetic per element, so FP32 (bias + residual) and FP16 GELU outputs must agree bit for
    bit -- whole and ragged tiles, the shortest K loop (two K tiles), N % 8 != 0 (the element-wise epilogue), several launches
    (a missing wait or a buffer overwritten too early is intermittent), and against the float64 product."""
    g = torch.Generator(device="cuda").manual_seed(M + N + 7)
    a = torch.randn((M, K), generator=g, device="cuda").half()
    w = (0.05 * torch.randn((N, K), generator=g, device="cuda")).half()
    bias = torch.randn(N, generator=g, device="cuda")
    res = torch.randn((M, N), generator=g, device="cuda")
    want = _torch_ref_mul_mat(a, w, bias, res)
    L = binding.lib()
    outs, gelus = [], []
    try:
        base = binding.TUNE_DEFAULT & ~(binding.TUNE_GEMM_4WAVE | binding.TUNE_GEMM_FAST_EPI)
        for mask in (base, base | binding.TUNE_GEMM_4WAVE, base | binding.TUNE_GEMM_FAST_EPI):
            L.wh_debug_set_tuning(mask)
            for rep in range(3):
                out = torch.full((M, N), float("nan"), dtype=torch.float32, device="cuda")
                binding.check(L.wh_op_mul_mat(None, ptr(a), ptr(w), ptr(bias), ptr(res), ptr(out), M, N, K))
                outs.append(out)
                o16 = torch.zeros((M, N), dtype=torch.float16, device="cuda")
                binding.check(L.wh_op_mul_mat_gelu(None, ptr(a), ptr(w), ptr(bias), ptr(o16), M, N, K))
                gelus.append(o16)
    finally:
        L.wh_debug_set_tuning(binding.TUNE_DEFAULT)
    torch.cuda.synchronize()
    d = (outs[-1].double() - want).abs()
    print("mul_mat 4-wave %dx%dx%d maxdiff %.3e" % (M, N, K, float(d.max())))
    assert bool(torch.isfinite(outs[-1]).all())
    assert float(d.max()) < 2e-5 * max(1.0, np.sqrt(K / 128))
    for o in outs[1:]:
        assert torch.equal(o, outs[0])
    for o in gelus[1:]:
        assert torch.equal(o, gelus[0])


def test_mul_mat_gelu_big_tiles(golden):
    """The same instance with the FP16 GELU epilogue (EPI_F16_GELU, the encoder's MLP up-projection)."""
    M, N, K = 16390, 4608, 512
    g = torch.Generator(device="cuda").manual_seed(3)
    a = torch.randn((M, K), generator=g, device="cuda").half()
    w = (0.1 * torch.randn((N, K), generator=g, device="cuda")).half()
    bias = torch.randn(N, generator=g, device="cuda")
    pre = (_torch_ref_mul_mat(a, w, bias)).float()
    table = torch.from_numpy(golden["table_gelu"].astype(np.int32)).cuda()
    idx = pre.half().view(torch.int16).to(torch.int32) & 0xFFFF
    want = table[idx.long()].to(torch.int16).view(torch.float16).float()
    out = torch.zeros((M, N), dtype=torch.float16, device="cuda")
    binding.check(binding.lib().wh_op_mul_mat_gelu(None, ptr(a), ptr(w), ptr(bias), ptr(out), M, N, K))
    torch.cuda.synchronize()
    d = (out.float() - want).abs()
    frac = float((d > 0).float().mean())
    print("mul_mat_gelu big: %.4f %% of entries differ, max %.3e" % (100 * frac, float(d.max())))
    assert frac < 0.02 and bool((d <= torch.maximum(torch.tensor(4e-3, device="cuda"), want.abs() * 2.0 ** -10)).all())


@pytest.mark.parametrize("M,N,K", [(33, 1024, 1024), (40, 3840, 1280), (64, 1024, 4096), (65, 4096, 1024), (84, 1024, 1024),
                                   (112, 5120, 1280), (128, 1024, 1024), (100, 51865, 1024), (112, 1024, 4096), (112, 1280, 5120),
                                   (49, 1000, 384), (96, 52, 2048), (40, 20000, 256)])
def test_mul_mat_decode_rows(M, N, K):
    """33 .. 128 activation rows through the decode kernels -- what a lock-step batch of up to 128 sequences runs every token:
    gemvFused with 32 or 64 rows per workgroup (whichever gives >= 256 workgroups) and, for the vocabulary-sized N,
    gemmAllRows (32 columns x all rows per workgroup); with both tuning bits off, the 64-row gemvFused for everything.
    Repeated launches are bit-identical (fixed summation order)."""
    rng = np.random.default_rng(M * 3 + N)
    a = rng.standard_normal((M, K)).astype(np.float16)
    w = (0.05 * rng.standard_normal((N, K))).astype(np.float16)
    bias = rng.standard_normal(N).astype(np.float32)
    res = rng.standard_normal((M, N)).astype(np.float32)
    want = (a.astype(np.float64) @ w.astype(np.float64).T + bias + res).astype(np.float32)
    ad, wd, bd, rd = dev(a), dev(w), dev(bias), dev(res)
    L = binding.lib()
    for name, mask in (("default", binding.TUNE_DEFAULT), ("gemv64", binding.TUNE_DEFAULT & ~(binding.TUNE_GEMV_ALLROWS | binding.TUNE_GEMV_ROWGROUPS))):
        outs = []
        try:
            L.wh_debug_set_tuning(mask)
            for rep in range(4):
                out = torch.full((M, N), float("nan"), dtype=torch.float32, device="cuda")
                binding.check(L.wh_op_mul_mat(None, ptr(ad), ptr(wd), ptr(bd), ptr(rd), ptr(out), M, N, K))
                outs.append(out)
            torch.cuda.synchronize()
        finally:
            L.wh_debug_set_tuning(binding.TUNE_DEFAULT)
        d = report("mul_mat decode rows %s %dx%dx%d" % (name, M, N, K), outs[0].cpu().numpy(), want)
        assert d.max() < 2e-5 * max(1.0, np.sqrt(K / 128))
        for o in outs[1:]:
            assert torch.equal(o, outs[0])


@pytest.mark.parametrize("M,N,K", [(112, 4096, 1024), (48, 5120, 1280), (128, 2000, 512)])
def test_mul_mat_gelu_decode_rows(M, N, K, golden):
    """The MLP up-projection of a 33 .. 128-row decode step: FP16 GELU-table epilogue, both row groupings."""
    g = torch.Generator(device="cuda").manual_seed(M + N)
    a = torch.randn((M, K), generator=g, device="cuda").half()
    w = (0.1 * torch.randn((N, K), generator=g, device="cuda")).half()
    bias = torch.randn(N, generator=g, device="cuda")
    pre = (_torch_ref_mul_mat(a, w, bias)).float()
    table = torch.from_numpy(golden["table_gelu"].astype(np.int32)).cuda()
    idx = pre.half().view(torch.int16).to(torch.int32) & 0xFFFF
    want = table[idx.long()].to(torch.int16).view(torch.float16).float()
    L = binding.lib()
    for name, mask in (("default", binding.TUNE_DEFAULT), ("gemv64", binding.TUNE_DEFAULT & ~(binding.TUNE_GEMV_ALLROWS | binding.TUNE_GEMV_ROWGROUPS))):
        try:
            L.wh_debug_set_tuning(mask)
            out = torch.zeros((M, N), dtype=torch.float16, device="cuda")
            binding.check(L.wh_op_mul_mat_gelu(None, ptr(a), ptr(w), ptr(bias), ptr(out), M, N, K))
            torch.cuda.synchronize()
        finally:
            L.wh_debug_set_tuning(binding.TUNE_DEFAULT)
        d = (out.float() - want).abs()
        frac = float((d > 0).float().mean())
        print("mul_mat_gelu decode rows %s %dx%dx%d: %.4f %% of entries differ, max %.3e" % (name, M, N, K, 100 * frac, float(d.max())))
        assert frac < 0.02 and bool((d <= torch.maximum(torch.tensor(4e-3, device="cuda"), want.abs() * 2.0 ** -10)).all())


def _np_decoder_attention(q, K, V, n_tok, n_keys, causal, n_past, group, n_threads):
    """WhisperNP._attention_dec semantics on explicit caches: q [seq*n_tok][H*64] (already scaled, FP16 values),
    K, V [blocks][H][stride][64]. n_threads = 0: FP32 P.V, else the reference's FP16 thread-partitioned accumulation."""
    S_, d = q.shape
    H = d // 64
    seqs = S_ // n_tok
    out = np.zeros((S_, d), np.float32)
    for s in range(seqs):
        blk = s // group
        for h in range(H):
            sl = slice(h * 64, (h + 1) * 64)
            qq = q[s * n_tok:(s + 1) * n_tok, sl].astype(np.float32)
            Kh = K[blk, h, :n_keys].astype(np.float32)
            Vh = V[blk, h, :n_keys].astype(np.float32)
            sc = (qq @ Kh.T).astype(np.float32)
            if causal:
                j = np.arange(n_keys)[None, :]
                i = np.arange(n_tok)[:, None]
                sc = np.where(j > n_past + i, np.float32(-np.inf), sc)
            P = wn.softmax_table(sc)
            if n_threads > 0:
                out[s * n_tok:(s + 1) * n_tok, sl] = wn.WhisperNP.pv_f16_accumulate(P, Vh, n_threads)
            else:
                out[s * n_tok:(s + 1) * n_tok, sl] = (P.astype(np.float64) @ Vh.astype(np.float64)).astype(np.float32)
    return out


@pytest.mark.parametrize("seqs,heads,n_tok,n_keys,stride,causal,n_past,group,par", [
    (3, 2, 1, 1500, 1500, 0, 0, 1, 0),        # cross-attention, one query row per window
    (2, 2, 1, 1500, 1500, 0, 0, 1, 1),        # the reference's FP16 P.V, one thread
    (2, 2, 1, 1500, 1500, 0, 0, 1, 8),        # ... eight threads
    (10, 2, 1, 1500, 1500, 0, 0, 5, 0),       # 5 hypotheses per window share one pass over K/V
    (8, 1, 1, 777, 1500, 0, 0, 8, 0),
    (6, 3, 1, 130, 448, 0, 0, 2, 1),
    (4, 2, 1, 37, 448, 1, 36, 1, 0),          # self-attention at position 36
    (2, 2, 3, 3, 448, 1, 0, 1, 0),            # 3-token prompt step, causal
    (2, 2, 5, 70, 448, 1, 65, 1, 1),          # crosses the 64-row group boundary
    (1, 2, 1, 1, 448, 1, 0, 1, 0),            # a single key
])
def test_decoder_attention(seqs, heads, n_tok, n_keys, stride, causal, n_past, group, par):
    """attentionDecG (8 lanes per K/V row, hypothesis groups) and the first kernel (tuning bit off) against the restatement
    of the decoder's mulMat(K,Q) -> diagMaskInf -> softMax -> mulMat(V,.) chain (whisper.cpp:1618-1660, 1715-1748)."""
    rng = np.random.default_rng(seqs * 100 + n_keys)
    d = heads * 64
    q = (rng.standard_normal((seqs * n_tok, d)) * 0.8).astype(np.float16)
    blocks = seqs // group
    K = (rng.standard_normal((blocks, heads, stride, 64)) * 0.8).astype(np.float16)
    V = rng.standard_normal((blocks, heads, stride, 64)).astype(np.float16)
    want = _np_decoder_attention(q, K, V, n_tok, n_keys, causal, n_past, group, par)
    qd, kd, vd = dev(q), dev(K), dev(V)
    L = binding.lib()
    results = {}
    for name, mask in (("grouped", None), ("first", 0)):
        if mask == 0 and group > 1:
            continue
        if mask is not None:
            L.wh_debug_set_tuning(binding.TUNE_DEFAULT & ~binding.TUNE_ATTN_DEC_G)
        try:
            out = torch.full((seqs * n_tok, d), float("nan"), dtype=torch.float16, device="cuda")
            binding.check(L.wh_op_decoder_attention(None, ptr(qd), ptr(kd), ptr(vd), ptr(out), seqs, heads, n_tok, n_keys, stride, causal, n_past,
                                                    group, par))
            torch.cuda.synchronize()
        finally:
            L.wh_debug_set_tuning(binding.TUNE_DEFAULT)
        got = out.cpu().numpy().astype(np.float32)
        assert np.isfinite(got).all()
        dd = report("decoder_attention %s keys=%d group=%d par=%d" % (name, n_keys, group, par), got, wn.r16(want))
        # scores differ by FP32 summation order only: a flipped FP16 rounding of (s - max) moves one key's probability by
        # <= 1.6 %; with the FP16 P.V emulation a flip can also move a partial sum by one FP16 ulp
        assert dd.max() < (4e-3 if par else 2e-3) and dd.mean() < 2e-4
        results[name] = got


@pytest.mark.parametrize("seqs,heads,group", [(3, 2, 1), (10, 16, 5), (4, 20, 1)])
def test_decoder_cross_attention_fused_query(seqs, heads, group):
    """LayerNorm + query projection inside the attention kernel == the three separate steps of WhisperContext.cpp:489-519."""
    rng = np.random.default_rng(seqs + heads)
    d = heads * 64
    n_keys = 1500
    x = (rng.standard_normal((seqs, d)) * 2 + 0.3).astype(np.float32)
    lnw = (1 + 0.1 * rng.standard_normal(d)).astype(np.float32)
    lnb = (0.1 * rng.standard_normal(d)).astype(np.float32)
    wq = (rng.standard_normal((d, d)) * (1.0 / np.sqrt(d))).astype(np.float16)
    bq = (0.1 * rng.standard_normal(d)).astype(np.float32)
    scale = np.float32(64.0 ** -0.25)
    blocks = seqs // group
    K = (rng.standard_normal((blocks, heads, n_keys, 64)) * 0.8).astype(np.float16)
    V = rng.standard_normal((blocks, heads, n_keys, 64)).astype(np.float16)
    xn = wn.layer_norm(x, lnw, lnb)
    q = wn.r16(((wn.mul_mat_w(wq, xn) + bq).astype(np.float32) * scale).astype(np.float32))
    want = _np_decoder_attention(q.astype(np.float16), K, V, 1, n_keys, 0, 0, group, 0)
    xd, lw, lb, wd, bd, kd, vd = dev(x), dev(lnw), dev(lnb), dev(wq), dev(bq), dev(K), dev(V)
    out = torch.full((seqs, d), float("nan"), dtype=torch.float16, device="cuda")
    binding.check(binding.lib().wh_op_decoder_cross_attention(None, ptr(xd), ptr(lw), ptr(lb), ptr(wd), ptr(bd), C.c_float(float(scale)), ptr(kd), ptr(vd),
                                                              ptr(out), seqs, heads, n_keys, n_keys, group))
    torch.cuda.synchronize()
    got = out.cpu().numpy().astype(np.float32)
    assert np.isfinite(got).all()
    dd = report("fused cross attention seqs=%d heads=%d group=%d" % (seqs, heads, group), got, wn.r16(want))
    # one more rounding point than the plain attention test: an FP16 flip of a LayerNorm output or of q moves a score by ~1e-3
    assert dd.max() < 4e-3 and dd.mean() < 3e-4


@pytest.mark.parametrize("blocks,heads,group,n_keys,stride", [(8, 20, 5, 1500, 1500), (3, 16, 5, 1500, 1500), (2, 2, 2, 1500, 1500), (2, 6, 3, 1500, 1536), (2, 8, 4, 700, 1500),
                                                              (1, 16, 8, 1500, 1500), (2, 20, 5, 37, 1500), (2, 16, 5, 1, 1500), (1, 2, 5, 129, 1500), (1, 16, 5, 1489, 1500)])
def test_decoder_cross_attention_of_hypothesis_groups_on_the_matrix_cores(blocks, heads, group, n_keys, stride):
    """attentionDecM (option cross_mfma, round 6's default for hypothesis groups: LayerNorm by a wave per row, query projection, K.Q^T and V^T.P^T as 16x16x32 MFMAs,
    V transposed through per-wave LDS) against the numpy restatement of WhisperContext.cpp:489-519 and against attentionDecG<NQ, true> (cross_mfma 0): groups of
    2 / 3 / 4 / 5 / 8 rows, key counts that end inside a tile, inside the first tile, on one key (audio_ctx overrides), d = 128 .. 1280."""
    rng = np.random.default_rng(blocks * 7 + heads + group + n_keys)
    seqs = blocks * group
    d = heads * 64
    x = (rng.standard_normal((seqs, d)) * 2 + 0.3).astype(np.float32)
    lnw = (1 + 0.1 * rng.standard_normal(d)).astype(np.float32)
    lnb = (0.1 * rng.standard_normal(d)).astype(np.float32)
    wq = (rng.standard_normal((d, d)) * (1.0 / np.sqrt(d))).astype(np.float16)
    bq = (0.1 * rng.standard_normal(d)).astype(np.float32)
    scale = np.float32(64.0 ** -0.25)
    K = (rng.standard_normal((blocks, heads, stride, 64)) * 0.8).astype(np.float16)
    V = rng.standard_normal((blocks, heads, stride, 64)).astype(np.float16)
    V[:, :, n_keys:, :] = np.float16(np.nan)          # rows beyond the key count must never reach a sum
    K[:, :, n_keys:, :] = np.float16(np.nan)
    xn = wn.layer_norm(x, lnw, lnb)
    q = wn.r16(((wn.mul_mat_w(wq, xn) + bq).astype(np.float32) * scale).astype(np.float32))
    want = _np_decoder_attention(q.astype(np.float16), K[:, :, :n_keys], V[:, :, :n_keys], 1, n_keys, 0, 0, group, 0)
    xd, lw, lb, wd, bd, kd, vd = dev(x), dev(lnw), dev(lnb), dev(wq), dev(bq), dev(K), dev(V)
    got = {}
    try:
        for mode in (0, 1):
            binding.set_option("cross_mfma", mode)
            out = torch.full((seqs, d), float("nan"), dtype=torch.float16, device="cuda")
            binding.check(binding.lib().wh_op_decoder_cross_attention(None, ptr(xd), ptr(lw), ptr(lb), ptr(wd), ptr(bd), C.c_float(float(scale)), ptr(kd), ptr(vd),
                                                                      ptr(out), seqs, heads, n_keys, stride, group))
            torch.cuda.synchronize()
            got[mode] = out.cpu().numpy().astype(np.float32)
    finally:
        binding.set_option("cross_mfma", binding.get_option_default("cross_mfma"))
    assert np.isfinite(got[1]).all()
    dd = report("cross attention on the matrix cores blocks=%d heads=%d group=%d keys=%d" % (blocks, heads, group, n_keys), got[1], wn.r16(want))
    d0 = np.abs(got[0] - wn.r16(want))
    print("  attentionDecG against the same restatement: max %.2e mean %.2e; the two kernels: max %.2e" % (d0.max(), d0.mean(), np.abs(got[0] - got[1]).max()))
    assert dd.max() < 4e-3 and dd.mean() < 3e-4
    assert np.abs(got[0] - got[1]).max() < 4e-3
