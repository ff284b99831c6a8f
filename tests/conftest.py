import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden", "ref_test_d128.npz")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    """Outputs of the reference's own CPU path (oracle/_ref) on seeded inputs; see tests/golden/make_golden.py."""
    return dict(np.load(GOLDEN))


@pytest.fixture(scope="session")
def golden_e2e():
    """Exact-arithmetic and 8-thread reference logits + the reference's greedy ids on jfk.wav; tests/golden/make_golden_e2e.py."""
    return dict(np.load(os.path.join(ROOT, "tests", "golden", "ref_e2e_d128.npz")))


@pytest.fixture(scope="session")
def tiny_model():
    """The synthetic model the golden fixtures were produced with (regenerated from its seed)."""
    from whisper_amd import ggml_format as gf
    return gf.synth_model("test-d128", seed=1234, attn_sharpness=2.0)


@pytest.fixture(scope="session")
def ref_lib_available():
    from oracle import ref
    return ref.available()
