"""CPU tests of the drop-in boundary: the C-ABI library loads and exports every symbol include/whisper_hip.h declares.
No compute is called here (there is no GPU in the build container)."""
import ctypes
import os
import re

import pytest

from whisper_amd import binding, ggml_format as gf

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "whisper_hip.h")).read()
    return sorted(set(re.findall(r"WH_API\s+[\w\s\*]+?\b(wh_\w+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    if not os.path.exists(binding.LIB_PATH):
        from whisper_amd import build
        build.build_hip()
    lib = ctypes.CDLL(binding.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 25
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    assert sorted(binding.EXPORTS) == names


def test_arena_layout_is_pure_function_of_hparams():
    """Ranks that only receive the RCCL broadcast rely on this: same hparams -> same byte count (no GPU needed)."""
    hp = gf.hparams_for("medium")
    n1 = binding.arena_bytes(hp)
    n2 = binding.arena_bytes(gf.hparams_for("medium"))
    assert n1 == n2
    # FP16 matrices dominate: ~ the size of the real ggml-medium.bin (1.53 GB) minus FP32 leftovers
    assert 1.4e9 < n1 < 1.7e9
    assert binding.arena_bytes(gf.hparams_for("large-v2")) > 2.9e9


def test_invalid_hparams_are_rejected():
    hp = gf.hparams_for("tiny")
    hp.n_audio_head = 5                     # 384 / 5 != 64
    with pytest.raises(binding.WhisperHipError):
        binding.arena_bytes(hp)


def test_no_silent_cpu_fallback():
    """Without a GPU, creating a model must fail loudly (WH_E_NO_DEVICE), never compute on the host."""
    if binding.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(binding.WhisperHipError):
        binding.HipModel(gf.hparams_for("test-d128"))


def test_product_never_touches_the_oracle():
    """oracle/ is the checker: nothing under whisper_amd/ (Python, C++ host, HIP) may import, include, link or execute it, and
    the bench uses it only in the cpu_baseline leg."""
    import re
    pkg = os.path.join(ROOT, "whisper_amd")
    offenders = []
    for base, _, files in os.walk(pkg):
        if os.sep + "lib" in base or "__pycache__" in base:
            continue
        for f in files:
            if not f.endswith((".py", ".cpp", ".h", ".hip")):
                continue
            text = open(os.path.join(base, f), errors="ignore").read()
            if f == "build.py":
                # the build script may BUILD the checker (make -C oracle); it must not import it
                text = re.sub(r'os\.path\.join\(ROOT, "oracle"\)|"oracle"', "", text)
            for m in re.finditer(r"^\s*(from|import)\s+oracle\b|#include\s+[\"<][^\">]*oracle|dlopen\([^)]*oracle|CDLL\([^)]*oracle", text, re.M):
                offenders.append((f, m.group(0)))
    assert not offenders, offenders
    bench = open(os.path.join(ROOT, "bench.py")).read()
    uses = [m.start() for m in re.finditer(r"from oracle import|import oracle", bench)]
    assert uses, "bench.py times the reference CPU path in its cpu_baseline leg"
    for u in uses:
        # every import sits inside cpu_baseline_worker / cpu_baseline
        head = bench[:u]
        assert head.rfind("def cpu_baseline") > head.rfind("def main") and head.rfind("def cpu_baseline") > head.rfind("def run_passes")


def test_tuning_bits_of_the_binding_match_the_library_header():
    """whisper_amd/binding.py mirrors csrc/kernels.h eTuning by hand; tests restore binding.TUNE_DEFAULT after an A/B, so a bit that is in one default
    and not in the other would silently change what the rest of a test session measures."""
    import re
    from whisper_amd import binding
    text = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "whisper_amd", "csrc", "kernels.h")).read()
    body = text[text.index("enum eTuning"):]
    body = body[:body.index("};")]
    values = {}
    for name, expr in re.findall(r"^\s*(TUNE_[A-Z0-9_]+)\s*=\s*([^,/\n]+?)\s*,?\s*(?://.*)?$", body, re.M):
        expr = re.sub(r"(\d+)u\b", r"\1", expr)
        values[name] = eval(expr, {}, values)
    assert "TUNE_DEFAULT" in values and len(values) > 25
    for name, v in values.items():
        assert hasattr(binding, name), name + " is missing in binding.py"
        assert getattr(binding, name) == v, (name, getattr(binding, name), v)
