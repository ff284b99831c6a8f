"""The drop-in claim, executed: a caller compiled from the REFERENCE's own public headers links and runs against libWhisper.so.

tests/abi_caller/caller.cpp is built twice by whisper_amd/build.py -- against /root/reference's Whisper/API/whisperComLight.h +
ComLightLib/comLightClient.h (the headers Examples/main/main.cpp:1-30 includes; found through -I/root/reference where that
tree exists, i.e. in the build container -- the GPU box receives only the binary) and against include/whisperApi.h.
  * CPU: both binaries print sizeof / offsetof of every POD structure crossing the boundary (sFullParams, sSegment, sToken,
    sModelSetup ...), the six interface ids and the HRESULT constants; the two printouts must be identical.
  * GPU: both run loadModel -> createContext -> fullDefaultParams -> initMediaFoundation -> loadAudioFile -> runFull ->
    getResults on a scripted model (tests/golden/ref_hostloop.json, produced by the reference's whisper_full) and must print
    the transcript the reference produced, through the reference-header binary's vtable calls.
Found by this test when it was written: iMediaFoundation paths are LPCTSTR = const char* off Windows
(ComLightLib/comLightCommon.h:5-9), not wchar_t*; sFullParams::resetFlag was missing from include/whisperApi.h.
"""
import json
import os
import subprocess
import wave

import numpy as np
import pytest

from whisper_amd import build, ggml_format as gf

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden", "ref_hostloop.json")


@pytest.fixture(scope="module")
def callers():
    if not os.path.exists(build.HOST_LIB):
        pytest.skip("libWhisper.so not built")
    build.build_abi_callers()
    if not os.path.exists(build.ABI_REF_BIN):
        pytest.skip("no caller built from the reference's headers (/root/reference absent and no prebuilt binary)")
    return build.ABI_REF_BIN, build.ABI_OUR_BIN


def test_reference_headers_and_ours_describe_the_same_binary_contract(callers):
    ref, our = callers
    a = subprocess.run([ref, "layout"], stdout=subprocess.PIPE, check=True).stdout.decode()
    b = subprocess.run([our, "layout"], stdout=subprocess.PIPE, check=True).stdout.decode()
    la, lb = json.loads(a), json.loads(b)
    assert la == lb, {k: (la.get(k), lb.get(k)) for k in set(la) | set(lb) if la.get(k) != lb.get(k)}
    assert la["sizeof sFullParams"] == 112 and la["sizeof sSegment"] == 32 and la["sizeof sToken"] == 48
    # ComLight GUID bytes of {b9956374-3b18-4943-90f2-2ab18a404537} (Whisper/API/iContext.cl.h:25), little-endian first three fields
    assert la["iid iContext"] == "746395b9183b434990f22ab18a404537"
    assert la["iid iModel"] == "c9b4efabd8e8a34687475afbadef1adb"


def test_the_reference_header_caller_resolves_every_export(callers):
    """The seven names of Whisper/whisper.def, as the reference's headers mangle them, are what the binary imports."""
    ref, _ = callers
    syms = subprocess.run(["nm", "-D", "--undefined-only", ref], stdout=subprocess.PIPE, check=True).stdout.decode()
    for name in ("loadModel", "setupLogger", "initMediaFoundation", "findLanguageKeyA"):
        assert any(name in ln and "Whisper" in ln for ln in syms.splitlines()), name
    ldd = subprocess.run(["ldd", ref], stdout=subprocess.PIPE).stdout.decode()
    assert "libWhisper.so" in ldd and "not found" not in ldd


@pytest.mark.gpu
@pytest.mark.parametrize("case_name", ["first_window_no_prompt", "multi_window"])
def test_reference_header_caller_transcribes(callers, tmp_path, case_name):
    ref, our = callers
    case = [c for c in json.load(open(GOLDEN))["cases"] if c["name"] == case_name][0]
    model = str(tmp_path / "m.bin")
    gf.write_model(model, gf.scripted_model(case["script"], case["prompt_len"]))
    rng = np.random.default_rng(case["pcm_seed"])
    pcm = (0.05 * rng.standard_normal(case["n_samples"])).astype(np.float32)
    wav = str(tmp_path / "clip.wav")
    with wave.open(wav, "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(16000)
        w.writeframes(np.clip(np.round(pcm * 32768.0), -32768, 32767).astype("<i2").tobytes())
    outs = []
    for exe in (ref, our):
        r = subprocess.run([exe, "run", model, wav, case["lang"], ",".join(str(t) for t in (case["prompt"] or [])), str(case["n_max_text_ctx"])], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        print(os.path.basename(exe), r.stdout.decode()[:600], r.stderr.decode()[-800:])
        assert r.returncode == 0
        outs.append(json.loads(r.stdout.decode().strip().splitlines()[-1]))
    assert outs[0] == outs[1]
    got = outs[0]
    want = case["segments"]
    assert got["callback_segments"] == len(got["segments"]) == len(want)
    assert [s["ids"] for s in got["segments"]] == [s["tokens"] for s in want]
    assert [(s["begin"], s["end"]) for s in got["segments"]] == [(s["t0"] * 100000, s["t1"] * 100000) for s in want]
