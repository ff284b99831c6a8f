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


def test_option_defaults_of_the_binding_match_the_library():
    """binding.OPTION_DEFAULTS mirrors the initialisers of csrc/kernels.h `struct Options` by hand, and tests restore an option from it after an A/B: a value that differs
    from the library's compiled-in default would silently change what the rest of a session runs. Read back from the library itself (wh_debug_get_option is host code) in
    a child process without WH_OPT_* variables, so that neither this session's environment nor an earlier test's set_option is in the way."""
    import subprocess
    import sys
    code = ("import json; from whisper_amd import binding; "
            "print(json.dumps({k: binding.get_option(k) for k in binding.OPTION_DEFAULTS}))")
    env = {k: v for k, v in os.environ.items() if not k.startswith("WH_OPT_")}
    env["PYTHONPATH"] = ROOT
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    got = json.loads(out.stdout.strip().splitlines()[-1])
    assert got == binding.OPTION_DEFAULTS, {k: (got[k], v) for k, v in binding.OPTION_DEFAULTS.items() if got[k] != v}
    # every option the library knows is mirrored (the names are listed in the header's comment of wh_debug_set_option)
    text = open(os.path.join(ROOT, "include", "whisper_hip.h")).read()
    at = text.index("WH_API int wh_debug_set_option")
    listed = set(re.findall(r'"([a-z0-9_]+)"', text[at - 2000:at]))
    assert set(binding.OPTION_DEFAULTS) <= listed | {"enc_ablate"}, sorted(set(binding.OPTION_DEFAULTS) - listed)
    with pytest.raises(RuntimeError):
        binding.get_option("no_such_option")
