"""Collected FIRST (file name): tells a bad GPU lease from a product fault before anything large runs.

Order of escalation, each step in its own test so that `-x` stops at the smallest failing surface:
  1. a torch-only op (no code of ours)      -> fails with the word BAD_BOX: the lease is broken, not the product
  2. wh_device_info                         -> the library loads and sees the device
  3. a 128^3 wh_op_mul_mat                  -> one trivial kernel of ours
  4. a one-frame wh_mel_spectrogram         -> model arena + context allocation + the spectrogram kernel
The same sequence is the first thing __graft_entry__.smoke() does (whisper_amd/canary.py).
Reference analogue: the op-level A/B tests of Whisper/ML/tensorOpsTests.cpp:10-183 run before any graph-level test.
"""
import pytest

torch = pytest.importorskip("torch")

from whisper_amd import canary  # noqa: E402

pytestmark = pytest.mark.gpu


def test_canary_1_torch_only_op():
    canary.torch_only()


def test_canary_2_device_info():
    info = canary.device()
    assert info["compute_units"] > 0 and "gfx950" in info["name"], info


def test_canary_3_trivial_mul_mat():
    canary.trivial_mul_mat()


def test_canary_4_one_frame_spectrogram():
    canary.one_frame_mel()


def test_results_do_not_depend_on_stale_device_memory():
    """WH_DEBUG_POISON fills every buffer the library does not REQUIRE to be zero with the given byte (0xFF: NaN as FP16 /
    FP32, -1 as an index; 0x7F: NaN as FP16, 3.4e38 as FP32, 2139062143 as an index) instead of zeros, puts guard regions
    around every allocation and verifies them when the context is destroyed. The hot path must give bit-identical results
    either way: a kernel that reads memory it did not write (or writes memory that is not its own) shows up here instead of
    as a placement- or lease-dependent `Memory access fault`."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lines = {}
    for poison in (None, "0xFF", "0x7F"):
        env = dict(os.environ, PYTHONPATH=root)
        env.pop("WH_DEBUG_POISON", None)
        if poison:
            env["WH_DEBUG_POISON"] = poison
        r = subprocess.run([sys.executable, os.path.join(root, "tests", "poison_probe.py")], env=env, cwd=root,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        err = r.stderr.decode(errors="replace")
        assert r.returncode == 0, "poison=%s rc=%d\n%s" % (poison, r.returncode, err[-3000:])
        assert "WH_GUARD_VIOLATION" not in err, err[-3000:]
        probe = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("PROBE ")]
        assert len(probe) == 1
        lines[poison] = json.loads(probe[0][6:])
    for poison in ("0xFF", "0x7F"):
        diff = {k: (lines[None][k], lines[poison][k]) for k in lines[None] if lines[None][k] != lines[poison][k]}
        assert not diff, "results change when stale memory is %s: %s" % (poison, diff)
