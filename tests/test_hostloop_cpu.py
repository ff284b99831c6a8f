"""CPU tests of the product's HOST LOGIC: whisper_amd/host/hostLoop.h (StreamRun / WindowScan -- the objects behind iContext::runFull,
runStreamed and the batch runner), tokenTimestamps.cpp and support.cpp (vocabulary, languages) compiled into a small test library
(tests/hostloop_cpu/driver.cpp) whose "device" is the reference's own CPU model (oracle/_ref/libwhisper_ref.so). No GPU, no
libWhisper.so: the same source files, fed with the reference's tokens, must reproduce
  * rules 0: the reference's whisper_full transcripts (tests/golden/ref_hostloop.json, make_golden_hostloop.py), and
  * rules 1: the transcripts of the reference's GPU-model host code, ContextImpl::runFullImpl compiled unmodified
    (tests/golden/ref_hostloop_contextimpl.json, make_golden_hostloop_contextimpl.py) -- where the two differ: the past prompt is kept
    to the last window and through failed windows, which are skipped at once.
The GPU twins of these tests run the same cases through libWhisper.so (tests/test_host_api.py)."""
import ctypes as C
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

from whisper_amd import ggml_format as gf

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "tests", "_build")
LIB = os.path.join(BUILD, "libhostloop_cpu.so")
REF_DIR = os.path.join(ROOT, "oracle", "_ref")
HIP_DIR = os.path.join(ROOT, "whisper_amd", "lib")
SOURCES = [os.path.join(ROOT, "tests", "hostloop_cpu", "driver.cpp"), os.path.join(ROOT, "whisper_amd", "host", "support.cpp"),
           os.path.join(ROOT, "whisper_amd", "host", "tokenTimestamps.cpp")]
HEADERS = [os.path.join(ROOT, "whisper_amd", "host", h) for h in ("hostLoop.h", "hostCommon.h", "results.h")] + \
          [os.path.join(ROOT, "include", h) for h in ("whisperApi.h", "whisper_hip.h")]

FLAG_TRANSLATE, FLAG_NO_CONTEXT, FLAG_SINGLE_SEGMENT, FLAG_TOKEN_TIMESTAMPS = 1, 2, 4, 0x100


class HlParams(C.Structure):
    _fields_ = [("flags", C.c_uint32), ("language", C.c_uint32), ("n_max_text_ctx", C.c_int32), ("offset_ms", C.c_int32), ("duration_ms", C.c_int32),
                ("max_tokens", C.c_int32), ("max_len", C.c_int32), ("thold_pt", C.c_float), ("thold_ptsum", C.c_float),
                ("prompt_tokens", C.POINTER(C.c_int32)), ("prompt_n_tokens", C.c_int32), ("withProgress", C.c_int32), ("resultFlags", C.c_uint32),
                ("mediaTime", C.c_int64)]


@pytest.fixture(scope="module")
def driver():
    if not os.path.exists(os.path.join(REF_DIR, "libwhisper_ref.so")):
        pytest.skip("oracle/_ref/libwhisper_ref.so not built (needs /root/reference)")
    if not os.path.exists(os.path.join(HIP_DIR, "libwhisper_hip.so")):
        pytest.skip("libwhisper_hip.so not built: support.cpp's loader links against it")
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    os.makedirs(BUILD, exist_ok=True)
    deps = SOURCES + HEADERS + [os.path.join(REF_DIR, "libwhisper_ref.so")]
    if not os.path.exists(LIB) or any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in deps):
        cmd = ["g++", "-std=c++17", "-O1", "-fPIC", "-shared", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "whisper_amd", "host")] + SOURCES + \
              ["-o", LIB, "-L" + HIP_DIR, "-lwhisper_hip", "-L" + REF_DIR, "-lwhisper_ref", "-Wl,-rpath," + HIP_DIR, "-Wl,-rpath," + REF_DIR, "-lpthread"]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert r.returncode == 0, r.stdout
    L = C.CDLL(LIB)
    L.hl_run.argtypes = [C.c_char_p, C.c_int, C.POINTER(HlParams), np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS"), C.c_int, C.c_int]
    L.hl_result.restype = C.c_char_p
    return L


def language_key(code):
    k = 0
    for i, ch in enumerate(code.encode()[:4]):
        k |= ch << (8 * i)
    return k


def run_case(L, tmp_path, c, pcm, rules, with_progress=False, result_flags=3, media_time=0, model=None):
    model = model if model is not None else gf.scripted_model(c["script"], c["prompt_len"])
    path = str(tmp_path / (c["name"] + ".bin"))
    gf.write_model(path, model)
    fl = c.get("flags", dict(no_context=True))
    p = HlParams()
    p.flags = (FLAG_NO_CONTEXT if fl.get("no_context") else 0) | (FLAG_SINGLE_SEGMENT if fl.get("single_segment") else 0) | \
              (FLAG_TRANSLATE if fl.get("translate") else 0) | (FLAG_TOKEN_TIMESTAMPS if fl.get("token_timestamps") else 0)
    p.language = language_key(c.get("lang", "en"))
    p.n_max_text_ctx, p.max_tokens, p.max_len = c["n_max_text_ctx"], fl.get("max_tokens", 0), fl.get("max_len", 0)
    p.thold_pt = p.thold_ptsum = -1.0
    prompt = c["prompt"] or []
    arr = (C.c_int32 * max(1, len(prompt)))(*(prompt or [0]))
    p.prompt_tokens = C.cast(arr, C.POINTER(C.c_int32)) if prompt else None
    p.prompt_n_tokens = len(prompt)
    p.withProgress = int(with_progress)
    p.resultFlags, p.mediaTime = result_flags, media_time
    pcm = np.ascontiguousarray(pcm, np.float32)
    hr = L.hl_run(path.encode(), rules, C.byref(p), pcm, len(pcm), 4)
    assert hr >= 0, "hl_run failed: 0x%08x" % (hr & 0xFFFFFFFF)
    return hr, json.loads(L.hl_result().decode()) if hr == 0 else None


def test_whisper_full_rules(driver, tmp_path):
    """rules 0 against the reference's whisper_full (all cases of ref_hostloop.json: ids, texts, 10 ms times)."""
    G = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_hostloop.json")))
    rng = np.random.default_rng(G["cases"][0]["pcm_seed"])
    for c in G["cases"]:
        if c.get("pcm") == "jfk":
            pcm = np.load(os.path.join(ROOT, "tests", "golden", "ref_test_d128.npz"))["pcm16"].astype(np.float32) / 32768.0
        else:
            pcm = (0.05 * rng.standard_normal(c["n_samples"])).astype(np.float32)
        hr, got = run_case(driver, tmp_path, c, pcm, rules=0)
        if c["name"] == "too_short":
            assert hr == 1
            continue
        want = c["segments"]
        assert hr == 0 and len(got["segments"]) == len(want), (c["name"], len(got["segments"]), len(want))
        for g, w in zip(got["segments"], want):
            assert (g["t0"], g["t1"], g["text"]) == (w["t0"], w["t1"], w["text"]), (c["name"], g, w)
            assert [t["id"] for t in g["tokens"]] == w["tokens"]


def test_contextimpl_rules(driver, tmp_path):
    """rules 1 against the reference's GPU-model host code (ContextImpl.cpp compiled unmodified): ids, texts, times in 100 ns ticks, the
    token-level timestamps and max_len wrap of ContextImpl's port, new-segment callbacks, and the progress values runStreamed reports."""
    G = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_hostloop_contextimpl.json")))
    rng = np.random.default_rng(G["cases"][0]["pcm_seed"])
    differ = 0
    for c in G["cases"]:
        pcm = (0.05 * rng.standard_normal(c["n_samples"])).astype(np.float32)
        hr, got = run_case(driver, tmp_path, c, pcm, rules=1, with_progress=True)
        assert hr == c["hr"], (c["name"], hr)
        if hr == 1:
            continue
        want = c["segments"]
        assert len(got["segments"]) == len(want), (c["name"], [(g["t0"], g["t1"]) for g in got["segments"]], [(w["t0"] // 100000, w["t1"] // 100000) for w in want])
        for g, w in zip(got["segments"], want):
            assert (g["t0"] * 100000, g["t1"] * 100000, g["text"]) == (w["t0"], w["t1"], w["text"]), (c["name"], g, w)
            assert [t["id"] for t in g["tokens"]] == [t["id"] for t in w["tokens"]]
            if c["flags"].get("token_timestamps"):
                assert [(t["t0"] * 100000, t["t1"] * 100000) for t in g["tokens"]] == [(t["t0"], t["t1"]) for t in w["tokens"]], c["name"]
                assert all(abs(a["vlen"] - b["vlen"]) < 1e-4 for a, b in zip(g["tokens"], w["tokens"]))
            assert all(abs(a["p"] - b["p"]) < 1e-5 for a, b in zip(g["tokens"], w["tokens"]))     # the same arithmetic produced both
        assert got["new_segment"] == [c["new_segment_calls"], c["new_segments"]], c["name"]
        if c["streamed"]:
            assert got["progress"] == pytest.approx(c["streamed"]["progress"]), c["name"]
        # where the rule sets part, rules 0 must NOT give this transcript (the case would prove nothing otherwise)
        if c["name"].startswith("carry_one"):
            _, other = run_case(driver, tmp_path, c, pcm, rules=0)
            key = lambda r: [(s["t0"], s["t1"], [t["id"] for t in s["tokens"]]) for s in r["segments"]]      # noqa: E731
            differ += key(other) != key(got)
    assert differ == 2


def test_result_pods_under_every_flag(driver, tmp_path):
    """results.h fillResultData (what iContext::getResults hands out) against the reference's makeResults (ContextImpl.misc.cpp:196-300) for
    every combination of eResultFlags, with a media time that is not zero: tick scaling + media time, zero times without Timestamps, no token
    array without Tokens but firstToken / countTokens all the same, eTokenFlags::Special from token_eot on, the token's text."""
    G = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_hostloop_contextimpl.json")))
    r = G["results"]
    c = [x for x in G["cases"] if x["name"] == r["case"]][0]
    pcm = (0.05 * np.random.default_rng(r["pcm_seed"]).standard_normal(r["n_samples"])).astype(np.float32)
    for rf, want in r["by_flags"].items():
        hr, got = run_case(driver, tmp_path, c, pcm, rules=1, result_flags=int(rf), media_time=r["media_time"])
        assert hr == 0 and len(got["pods"]) == len(want)
        for g, w in zip(got["pods"], want):
            assert {k: g[k] for k in ("t0", "t1", "text", "first_token", "count_tokens")} == {k: w[k] for k in ("t0", "t1", "text", "first_token", "count_tokens")}, (rf, g, w)
            assert g["tokens"] == w["tokens"], (rf, g["tokens"][:3], w["tokens"][:3])
    # whisper_full's rules: a token whose times were never computed reports 0, not the media time
    hr, got = run_case(driver, tmp_path, c, pcm, rules=0, result_flags=3, media_time=r["media_time"])
    assert hr == 0 and all(t["t0"] == 0 and t["t1"] == 0 for s in got["pods"] for t in s["tokens"])
    assert got["pods"][0]["t0"] == r["by_flags"]["3"][0]["t0"]


@pytest.mark.parametrize("rules", [0, 1])
def test_audio_conditioned_models(driver, tmp_path, rules):
    """Numerics in the loop: models whose tokens AND timestamps depend on the audio (ggml_format.conditioned_model; a wrong logit moves a
    timestamp, the timestamp moves the next window). rules 0 against whisper_full (ref_runfull_conditioned.json), rules 1 against the
    reference's ContextImpl::runFull over the same arithmetic (ref_hostloop_contextimpl.json "conditioned": thread-count independent, and on
    these cases the same transcript as whisper_full's). Three of the six cases (the GPU tests run all of them)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden_runfull", os.path.join(ROOT, "tests", "golden", "make_golden_runfull.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    if rules == 0:
        cases = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_runfull_conditioned.json")))["cases"]
        scale = 100000
    else:
        cases = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_hostloop_contextimpl.json")))["conditioned"]
        scale = 1
    ran = 0
    for c in cases:
        if c["name"] not in ("jfk_s10", "mixed_s10", "jfk_s11"):
            continue
        hr, got = run_case(driver, tmp_path, c, mod.pcm_for(c["pcm"]), rules=rules, model=mod.model_for(c["seed"]))
        want = c["segments"]
        assert hr == 0 and len(got["pods"]) == len(want), (c["name"], len(got["pods"]), len(want))
        for g, w in zip(got["pods"], want):
            assert (g["t0"], g["t1"], g["text"]) == (w["t0"] * scale, w["t1"] * scale, w["text"]), (c["name"], g, w)
            assert [t["id"] for t in g["tokens"]] == w["tokens"]
        ran += 1
    assert ran == 3
