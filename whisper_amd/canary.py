"""GPU canary: the smallest possible uses of the device, in escalating order, so that a fault can be attributed.

A `Memory access fault by GPU` kills the process, so attribution has to come from ORDER: a torch-only op runs first (no
code of ours involved: if that faults the lease is bad and the word BAD_BOX is printed before it), then the library's own
smallest entry points. tests/test_00_canary.py and the first lines of __graft_entry__.smoke() both run this sequence.
"""
from __future__ import annotations

import ctypes as C
import sys

import numpy as np


def _say(msg: str):
    sys.stdout.write("[canary] " + msg + "\n")
    sys.stdout.flush()


def torch_only():
    """No code of ours: allocate, multiply, reduce, copy back. A failure here is the box, not the product."""
    import torch
    _say("BAD_BOX if the process dies or this check fails before 'torch ok' is printed")
    try:
        assert torch.cuda.is_available(), "no HIP device visible"
        x = torch.ones(1 << 22, device="cuda") * 2
        s = float(x.sum())
        y = (torch.arange(1 << 20, device="cuda", dtype=torch.float32) % 7).cpu().numpy()
        ok = s == float(1 << 23) and np.array_equal(y, np.arange(1 << 20, dtype=np.float32) % 7)
    except Exception as e:  # noqa: BLE001
        _say("BAD_BOX: torch-only op raised %r" % (e,))
        raise AssertionError("BAD_BOX: torch-only GPU op failed: %r" % (e,))
    if not ok:
        _say("BAD_BOX: torch-only op returned wrong data")
        raise AssertionError("BAD_BOX: torch-only GPU op returned wrong data")
    p = torch.cuda.get_device_properties(0)
    _say("torch ok: %s, %d CUs, %.0f GB, torch %s hip %s" % (p.name, p.multi_processor_count, p.total_memory / 2**30,
                                                             torch.__version__, torch.version.hip))


def device():
    from . import binding
    info = binding.device_info(0)
    _say("library ok: %s, %d CUs, %.0f GB" % (info["name"], info["compute_units"], info["total_mem"] / 2**30))
    return info


def trivial_mul_mat():
    """128 x 128 x 128 through wh_op_mul_mat against numpy: one kernel of ours, no model, no context."""
    import torch
    from . import binding
    rng = np.random.default_rng(1)
    a = rng.standard_normal((128, 128)).astype(np.float16)
    w = (0.05 * rng.standard_normal((128, 128))).astype(np.float16)
    ad, wd = torch.from_numpy(a).cuda(), torch.from_numpy(w).cuda()
    out = torch.full((128, 128), float("nan"), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    binding.check(binding.lib().wh_op_mul_mat(None, C.c_void_p(ad.data_ptr()), C.c_void_p(wd.data_ptr()), None, None,
                                              C.c_void_p(out.data_ptr()), 128, 128, 128))
    torch.cuda.synchronize()
    want = a.astype(np.float64) @ w.astype(np.float64).T
    d = float(np.abs(out.cpu().numpy() - want).max())
    assert d < 2e-5, ("trivial mul_mat mismatch", d)
    _say("mul_mat ok: 128^3 max diff %.2e" % d)


def one_frame_mel():
    """Smallest model (d = 128, 4 + 4 layers) loaded into an arena, a context, and a ONE-frame spectrogram (160 samples)."""
    import torch
    from . import binding, ggml_format as gf
    model = gf.synth_model("test-d128", seed=11)
    hm = binding.HipModel.from_ggml(model)
    ctx = binding.HipContext(hm, 1)
    try:
        pcm = (0.1 * np.random.default_rng(2).standard_normal(160)).astype(np.float32)
        mel = ctx.mel_spectrogram(torch.from_numpy(pcm).cuda()).cpu().numpy()
        assert mel.shape == (80, 1) and np.isfinite(mel).all(), mel.shape
        # one frame: after the global clamp to (max - 8) and (x + 4) / 4 the largest bin is exactly (max + 4) / 4
        assert mel.max() - mel.min() <= 2.0 + 1e-6
    finally:
        ctx.close()
        hm.close()
    _say("mel ok: one frame, range [%.3f, %.3f]" % (mel.min(), mel.max()))


def run_all():
    torch_only()
    device()
    trivial_mul_mat()
    one_frame_mel()
