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
                ("mediaTime", C.c_int64), ("mel", C.POINTER(C.c_float)), ("melLen", C.c_int32)]


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


def run_case(L, tmp_path, c, pcm, rules, with_progress=False, result_flags=3, media_time=0, model=None, mel=None):
    model = model if model is not None else gf.scripted_model(c["script"], c["prompt_len"])
    path = str(tmp_path / (c["name"] + ".bin"))
    gf.write_model(path, model)
    fl = c.get("flags", dict(no_context=True))
    p = HlParams()
    p.flags = (FLAG_NO_CONTEXT if fl.get("no_context") else 0) | (FLAG_SINGLE_SEGMENT if fl.get("single_segment") else 0) | \
              (FLAG_TRANSLATE if fl.get("translate") else 0) | (FLAG_TOKEN_TIMESTAMPS if fl.get("token_timestamps") else 0) | (8 if fl.get("print_special") else 0)
    p.offset_ms, p.duration_ms = c.get("offset_ms", 0), c.get("duration_ms", 0)
    if mel is not None:
        mel = np.ascontiguousarray(mel, np.float32)
        p.mel, p.melLen = mel.ctypes.data_as(C.POINTER(C.c_float)), mel.shape[1]
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


def _fuzz_cases(n):
    """Seeded parameter combinations: model kind x flags x prompt x text context x range of the audio x audio length."""
    rng = np.random.default_rng(2026)
    hp = gf.hparams_for("test-d128-ml")
    sp = gf.special_tokens(hp)
    beg, eot = sp["beg"], sp["eot"]
    script = [beg, 300, 301, 302, beg + 120, beg + 120, 400, 401, 402, beg + 250, beg + 250, 500, 501, beg + 360, eot]
    out = []
    for i in range(n):
        kind = ("scripted", "conditioned", "scripted", "conditioned", "random-ml", "scripted", "conditioned", "random-en")[i % 8]
        seconds = 0.6 if i == 5 else float(rng.choice([3.0, 8.0, 11.0, 20.0, 33.0, 45.0]))
        c = dict(name="fuzz%d" % i, kind=kind, seed=int(rng.integers(1, 1000)), seconds=seconds,
                 lang="en" if kind == "random-en" else str(rng.choice(["en", "de", "ja"])),
                 flags=dict(no_context=bool(rng.integers(0, 2)), single_segment=bool(rng.integers(0, 5) == 0), translate=bool(rng.integers(0, 3) == 0),
                            print_special=bool(rng.integers(0, 3) == 0), max_tokens=int(rng.choice([0, 0, 0, 7, 40]))),
                 prompt=[None, [1000], [1000], [1000, 1001, 1002, 1003, 1004]][int(rng.integers(0, 4))],
                 n_max_text_ctx=int(rng.choice([-1, 0, 0, 1, 5, 64])),
                 offset_ms=int(rng.choice([0, 0, 0, 1500, 7000])), duration_ms=int(rng.choice([0, 0, 0, 9000, 15000])), script=script, prompt_len=4)
        out.append(c)
    return out


def _fuzz_model(c):
    if c["kind"] == "scripted":
        return gf.scripted_model(c["script"], c["prompt_len"])
    if c["kind"] == "conditioned":
        return gf.conditioned_model(gf.conditioned_layout(gf.hparams_for("test-d128-ml")), 4, kind="test-d128-ml", seed=c["seed"])
    return gf.synth_model("test-d128-ml" if c["kind"] == "random-ml" else "test-d128", seed=c["seed"])


def test_differential_against_the_reference_host_loops(driver, tmp_path):
    """Differential test of the host logic against the reference's two host loops RUN LIVE on the same arithmetic: 16 seeded combinations of
    model (scripted, audio-conditioned, random weights: whatever tokens come out, both sides see the same ones) x flags (no_context,
    single_segment, translate, print_special, max_tokens) x initial prompt x n_max_text_ctx (prompts of varying length carried over) x
    offset_ms / duration_ms x audio length (incl. < 1 s) x language. rules 0 = whisper_full (Whisper/source/whisper.cpp), rules 1 =
    ContextImpl::runFull (Whisper/Whisper/ContextImpl.cpp compiled unmodified) fed the GPU model's own spectrogram (Spectrogram.cpp).
    Same CPU model, same thread count on both sides, so the transcripts must be IDENTICAL: ids, texts, times.
    (Offline, the same comparison over 150 more random combinations -- prompts of 300 tokens, text contexts of 300, offsets past the end of the
    audio, max_tokens 1 -- found one difference, and it is deliberate: an UNKNOWN LANGUAGE fails the run with E_INVALIDARG under both rule sets,
    like the GPU model's iContext (ContextImpl.cpp:497-505); whisper_full logs the error and decodes on with language token sot + 1 + (-1)
    (whisper.cpp whisper_lang_id / whisper_token_lang). Languages here are known ones.)"""
    from oracle import ref
    if not (ref.contextimpl_available() and ref.melstreamer_available()):
        pytest.skip("oracle/_ref is not complete (needs /root/reference)")
    jfk = np.load(os.path.join(ROOT, "tests", "golden", "ref_test_d128.npz"))["pcm16"].astype(np.float32) / 32768.0
    seen = dict(segments=0, empty=0, short=0, rules_differ=0)
    for c in _fuzz_cases(16):
        model = _fuzz_model(c)
        n = int(16000 * c["seconds"])
        pcm = np.resize(jfk, n).astype(np.float32) * (0.3 + 0.7 * (c["seed"] % 7) / 7.0)
        path = str(tmp_path / (c["name"] + ".bin"))
        gf.write_model(path, model)
        fl = c["flags"]
        # ---- rules 0: whisper_full
        w = ref.RefWhisper(path, n_threads=4, log_level=0)
        rc, want0 = w.full_range(pcm, lang=c["lang"], flags=int(fl["no_context"]) | (int(fl["single_segment"]) << 1) | (int(fl["translate"]) << 2) | (int(fl["print_special"]) << 3),
                                 max_tokens=fl["max_tokens"], prompt=c["prompt"], n_max_text_ctx=c["n_max_text_ctx"], offset_ms=c["offset_ms"], duration_ms=c["duration_ms"])
        w.close()
        hr0, got0 = run_case(driver, tmp_path, c, pcm, rules=0, model=model)
        if rc != 0:
            assert hr0 < 0 or hr0 == 1, (c, rc, hr0)      # whisper_full refuses (unknown language ...): so must the host loop
            key0 = None
        else:
            key0 = [(s["t0"], s["t1"], s["text"], [t["id"] for t in s["tokens"]]) for s in (got0["segments"] if got0 else [])]
            assert key0 == [(s["t0"], s["t1"], s["text"], s["tokens"]) for s in want0], (c, "rules 0")
        # ---- rules 1: ContextImpl::runFull on Spectrogram::pcmToMel's spectrogram
        ci = ref.RefContextImpl(path, model.filters, encoder_threads=4)
        hr_ref, want1 = ci.run_full(pcm, cpu_threads=4, lang=c["lang"],
                                    flags=(ref.FLAG_NO_CONTEXT if fl["no_context"] else 0) | (ref.FLAG_SINGLE_SEGMENT if fl["single_segment"] else 0) |
                                    (ref.FLAG_TRANSLATE if fl["translate"] else 0) | (ref.FLAG_PRINT_SPECIAL if fl["print_special"] else 0),
                                    max_tokens=fl["max_tokens"], prompt=c["prompt"], n_max_text_ctx=c["n_max_text_ctx"], offset_ms=c["offset_ms"], duration_ms=c["duration_ms"])
        ci.close()
        mel = ref.spectrogram_pcm_to_mel(pcm, model.filters, threads=2)
        hr1, got1 = run_case(driver, tmp_path, c, pcm, rules=1, model=model, mel=mel)
        if hr_ref > 1:
            assert hr1 == hr_ref - (1 << 32) or hr1 < 0, (c, hex(hr_ref), hr1)
            continue
        assert hr1 == hr_ref, (c, hr1, hr_ref)
        key1 = [(s["t0"] * 100000, s["t1"] * 100000, s["text"], [t["id"] for t in s["tokens"]]) for s in (got1["segments"] if got1 else [])]
        assert key1 == [(s["t0"], s["t1"], s["text"], [t["id"] for t in s["tokens"]]) for s in want1], (c, "rules 1")
        seen["segments"] += len(key1)
        seen["empty"] += not key1
        seen["short"] += hr_ref == 1
        if key0 is not None:
            seen["rules_differ"] += [(a, b, i) for a, b, _, i in key0] != [(a // 100000, b // 100000, i) for a, b, _, i in key1]
    print("differential:", seen)
    assert seen["segments"] >= 30 and seen["short"] >= 1 and seen["rules_differ"] >= 1


def test_token_timestamps_differential(driver, tmp_path):
    """tokenTimestamps.cpp (TokenTimestamper: the energy-based token times and the max_len wrap) against BOTH reference implementations run
    live on the same tokens: whisper.cpp's whisper_exp_compute_token_level_timestamps / whisper_wrap_segment (rules 0) and ContextImpl's port
    (rules 1: times start from 0, the proportional split never runs -- hostLoop.h) -- on speech (jfk.wav) and on a speech / silence / noise mix,
    thresholds 0.01 and 0.5, max_len 0 / 6 / 20, scripted and audio-conditioned models."""
    from oracle import ref
    if not (ref.contextimpl_available() and ref.melstreamer_available()):
        pytest.skip("oracle/_ref is not complete (needs /root/reference)")
    jfk = np.load(os.path.join(ROOT, "tests", "golden", "ref_test_d128.npz"))["pcm16"].astype(np.float32) / 32768.0
    rng = np.random.default_rng(5)
    mix = np.concatenate([jfk[:48000], np.zeros(24000, np.float32), 0.2 * rng.standard_normal(32000).astype(np.float32), jfk[60000:150000]]).astype(np.float32)
    hp = gf.hparams_for("test-d128-ml")
    sp = gf.special_tokens(hp)
    beg, eot = sp["beg"], sp["eot"]
    script = [beg, 300, 301, 302, beg + 120, beg + 120, 400, 401, 402, beg + 250, beg + 250, 500, 501, beg + 360, eot]
    combos = [("scripted", jfk, 0.01, 0), ("scripted", mix, 0.01, 6), ("conditioned", jfk, 0.5, 20), ("conditioned", mix, 0.01, 0), ("scripted", mix, 0.5, 20)]
    checked = 0
    for i, (kind, pcm, thold, max_len) in enumerate(combos):
        model = gf.scripted_model(script, 4) if kind == "scripted" else gf.conditioned_model(gf.conditioned_layout(hp), 4, kind="test-d128-ml", seed=10 + i)
        c = dict(name="tt%d" % i, lang="en", flags=dict(no_context=True, token_timestamps=True, max_len=max_len), prompt=[1000], n_max_text_ctx=0)
        path = str(tmp_path / (c["name"] + ".bin"))
        gf.write_model(path, model)

        w = ref.RefWhisper(path, n_threads=4, log_level=0)
        want0 = w.full_token_timestamps(pcm, lang="en", no_context=True, prompt=[1000], n_max_text_ctx=0, thold_pt=thold, thold_ptsum=thold, max_len=max_len)
        w.close()
        got0 = run_case_tt(driver, tmp_path, c, pcm, 0, model, thold, None)
        k0 = [(s["t0"], s["t1"], s["text"], [(t["id"], t["t0"], t["t1"]) for t in s["tokens"]]) for s in got0["segments"]]
        assert k0 == [(s["t0"], s["t1"], s["text"], [(t["id"], t["t0"], t["t1"]) for t in s["tokens"]]) for s in want0], (c["name"], "rules 0")
        assert all(abs(a["vlen"] - b["vlen"]) < 1e-4 for sa, sb in zip(got0["segments"], want0) for a, b in zip(sa["tokens"], sb["tokens"]))
        ci = ref.RefContextImpl(path, model.filters, encoder_threads=4)
        hr, want1 = ci.run_full(pcm, cpu_threads=4, lang="en", flags=ref.FLAG_NO_CONTEXT | ref.FLAG_TOKEN_TIMESTAMPS, prompt=[1000], n_max_text_ctx=0,
                                thold_pt=thold, thold_ptsum=thold, max_len=max_len)
        ci.close()
        assert hr == 0
        got1 = run_case_tt(driver, tmp_path, c, pcm, 1, model, thold, ref.spectrogram_pcm_to_mel(pcm, model.filters, threads=2))
        k1 = [(s["t0"] * 100000, s["t1"] * 100000, s["text"], [(t["id"], t["t0"] * 100000, t["t1"] * 100000) for t in s["tokens"]]) for s in got1["segments"]]
        assert k1 == [(s["t0"], s["t1"], s["text"], [(t["id"], t["t0"], t["t1"]) for t in s["tokens"]]) for s in want1], (c["name"], "rules 1")
        checked += sum(len(s["tokens"]) for s in want0) + sum(len(s["tokens"]) for s in want1)
    assert checked > 150


def run_case_tt(L, tmp_path, c, pcm, rules, model, thold, mel):
    """run_case with the two probability thresholds of the token-level timestamps set"""
    path = str(tmp_path / (c["name"] + ".bin"))
    fl = c["flags"]
    p = HlParams()
    p.flags = FLAG_NO_CONTEXT | FLAG_TOKEN_TIMESTAMPS
    p.language = language_key("en")
    p.n_max_text_ctx, p.max_tokens, p.max_len = c["n_max_text_ctx"], 0, fl.get("max_len", 0)
    p.thold_pt = p.thold_ptsum = thold
    arr = (C.c_int32 * 1)(*c["prompt"])
    p.prompt_tokens, p.prompt_n_tokens = C.cast(arr, C.POINTER(C.c_int32)), 1
    p.resultFlags = 3
    if mel is not None:
        mel = np.ascontiguousarray(mel, np.float32)
        p.mel, p.melLen = mel.ctypes.data_as(C.POINTER(C.c_float)), mel.shape[1]
    pcm = np.ascontiguousarray(pcm, np.float32)
    hr = L.hl_run(path.encode(), rules, C.byref(p), pcm, len(pcm), 4)
    assert hr == 0, hr
    return json.loads(L.hl_result().decode())


def test_language_table_against_the_reference(driver):
    """support.cpp's language table against the reference's Languages.cpp + languageCodez.inl (compiled unmodified into
    oracle/_ref/libcontextimpl_ref.so): every code of one to three lower-case letters gives the same id (the language token is sot + 1 + id,
    ContextImpl.cpp:508) or is unknown on both sides."""
    from oracle import ref
    if not ref.contextimpl_available():
        pytest.skip("oracle/_ref/libcontextimpl_ref.so not built (needs /root/reference)")
    R = ref._contextimpl_lib()
    driver.hl_language_id.argtypes = [C.c_uint32]
    import itertools
    import string
    known = 0
    for n in (1, 2, 3):
        for t in itertools.product(string.ascii_lowercase, repeat=n):
            code = "".join(t)
            want = R.ci_language_id(code.encode())
            assert driver.hl_language_id(language_key(code)) == want, code
            known += want >= 0
    assert known >= 99


@pytest.mark.parametrize("kind", ["test-d128", "test-d128-ml"])
def test_vocabulary_against_the_reference(driver, tmp_path, kind):
    """support.cpp's vocabulary of a ggml file -- the stored tokens, the ones the loader synthesises ([_EOT_], [_SOT_], [_TT_n], [_extra_token_n] ...)
    and the special ids of an English-only and a multilingual model -- against the reference CPU model's vocabulary of the same file, id by id."""
    path = str(tmp_path / "m.bin")
    gf.write_model(path, gf.synth_model(kind, seed=3))
    n = C.c_int(0)
    driver.hl_vocabulary_differences.argtypes = [C.c_char_p, C.POINTER(C.c_int)]
    assert driver.hl_vocabulary_differences(path.encode(), C.byref(n)) == 0
    assert n.value == gf.hparams_for(kind).n_vocab


def test_tokenizer_against_the_reference(driver, tmp_path):
    """Vocabulary::tokenize (iModel::tokenize; the CLI's --prompt) against the reference's whisper_tokenize (whisper.cpp:2186-2248) on the same
    vocabulary: 400 seeded texts of words the vocabulary holds, fragments of them, digits, punctuation, contractions, runs of blanks and tabs, and
    characters it does not hold (skipped by both)."""
    path = str(tmp_path / "m.bin")
    model = gf.synth_model("test-d128-ml", seed=3)
    gf.write_model(path, model)
    rng = np.random.default_rng(12)
    pieces = [" w%d" % i for i in range(300, 340)] + ["w301w302", " w", "w", "3", "42", " 1234", "'s", "'t", "'re", "'ve", "'m", "'ll", "'d", ",", ".", "!", "?", " -", "--",
              " ", "  ", "\t", " \t ", "it", "It's", " don't", "\u00e9", "\u00fc", "\u4e2d", "#", "@", "(", ")", "a", "b", " the", "W300", " W300"]
    texts = []
    for _ in range(400):
        k = int(rng.integers(1, 12))
        texts.append("".join(pieces[int(i)] for i in rng.integers(0, len(pieces), k)))
    n_texts, n_tokens = C.c_int(0), C.c_int(0)
    driver.hl_tokenize_differences.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    assert driver.hl_tokenize_differences(path.encode(), "\n".join(texts).encode(), C.byref(n_texts), C.byref(n_tokens)) == 0
    assert n_texts.value == 400 and n_tokens.value > 800
