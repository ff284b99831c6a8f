"""Token-level timestamps + max_len wrapping (eFullParamsFlags::TokenTimestamps): host-only post-processing
in libWhisper.so (whisper_amd/host/tokenTimestamps.cpp) against outputs of the reference's CPU model
(whisper_exp_compute_token_level_timestamps / whisper_wrap_segment, whisper.cpp:3374-3575, 2711-2760), committed as
tests/golden/ref_token_timestamps.json by tests/golden/make_golden_token_timestamps.py.

CPU test: the reference's own unsplit segments and token data (id, tid, p, pt, ptsum) go through the post-processing alone;
token times, voice lengths and the wrapped segments must come back exactly.
GPU test: iContext::runFull with the flag on the same scripted model and audio reproduces the reference's wrapped transcript."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from whisper_amd import api, ggml_format as gf

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_token_timestamps.json")))["cases"]


def bursty_pcm(n, seed):
    """Same generator as tests/golden/make_golden_token_timestamps.py."""
    rng = np.random.default_rng(seed)
    t = np.arange(n, dtype=np.float64) / 16000.0
    env = 0.004 + 0.25 * (np.sin(2 * np.pi * t / 0.83) > 0.2) * (0.6 + 0.4 * np.sin(2 * np.pi * t / 0.31) ** 2)
    return (rng.standard_normal(n) * env).astype(np.float32)


def post_process(lib, model_path, pcm, segments, thold_pt, thold_ptsum, max_len):
    toks = [t for s in segments for t in s["tokens"]]
    seg_times = np.array([[s["t0"], s["t1"]] for s in segments], np.int64)
    counts = np.array([len(s["tokens"]) for s in segments], np.int32)
    ids = np.array([t["id"] for t in toks], np.int32)
    tids = np.array([t["tid"] for t in toks], np.int32)
    p, pt, ptsum = (np.array([t[k] for t in toks], np.float32) for k in ("p", "pt", "ptsum"))
    seg_cap, tok_cap = 512, 4096
    n_seg = C.c_int32()
    o_times = np.zeros((seg_cap, 2), np.int64)
    o_counts = np.zeros(seg_cap, np.int32)
    o_text = C.create_string_buffer(1 << 16)
    o_tok = np.zeros((tok_cap, 2), np.int64)
    o_vlen = np.zeros(tok_cap, np.float32)
    ptr = lambda a: a.ctypes.data_as(C.c_void_p)      # noqa: E731
    rc = lib.whisperc_debug_token_timestamps(
        model_path.encode(), ptr(pcm), C.c_uint64(len(pcm)), len(segments), ptr(seg_times), ptr(counts), ptr(ids), ptr(tids), ptr(p), ptr(pt),
        ptr(ptsum), C.c_float(thold_pt), C.c_float(thold_ptsum), max_len, seg_cap, tok_cap, C.byref(n_seg), ptr(o_times), ptr(o_counts),
        o_text, len(o_text), ptr(o_tok), ptr(o_vlen))
    assert rc == 0
    texts = o_text.raw.split(b"\0")[:n_seg.value]
    out, k = [], 0
    for i in range(n_seg.value):
        n = int(o_counts[i])
        out.append(dict(t0=int(o_times[i, 0]), t1=int(o_times[i, 1]), text=texts[i].decode(),
                        tokens=[dict(t0=int(o_tok[k + j, 0]), t1=int(o_tok[k + j, 1]), vlen=float(o_vlen[k + j])) for j in range(n)]))
        k += n
    return out


def same(got, want):
    assert len(got) == len(want), (len(got), len(want))
    for g, w in zip(got, want):
        assert (g["t0"], g["t1"], g["text"]) == (w["t0"], w["t1"], w["text"]), (g, w["t0"], w["t1"], w["text"])
        assert len(g["tokens"]) == len(w["tokens"])
        for a, b in zip(g["tokens"], w["tokens"]):
            assert (a["t0"], a["t1"]) == (b["t0"], b["t1"]) and a["vlen"] == np.float32(b["vlen"]), (a, b)


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_post_processing_matches_the_reference(case, tmp_path):
    if not os.path.exists(api.HOST_LIB_PATH):
        from whisper_amd import build
        build.build_all()
    lib = C.CDLL(api.HOST_LIB_PATH)
    model = str(tmp_path / "m.bin")
    gf.write_model(model, gf.scripted_model(case["script"], case["prompt_len"]))
    pcm = bursty_pcm(case["n_samples"], case["pcm_seed"])
    assert len(case["wrapped"]) > len(case["plain"]) > 0
    same(post_process(lib, model, pcm, case["plain"], case["thold_pt"], case["thold_ptsum"], 0), case["plain"])
    same(post_process(lib, model, pcm, case["plain"], case["thold_pt"], case["thold_ptsum"], case["max_len"]), case["wrapped"])


@pytest.mark.gpu
def test_run_full_with_token_timestamps(tmp_path):
    case = CASES[0]
    model = str(tmp_path / "m.bin")
    gf.write_model(model, gf.scripted_model(case["script"], case["prompt_len"]))
    pcm = bursty_pcm(case["n_samples"], case["pcm_seed"])
    m = api.Model(model)
    ctx = m.create_context()
    for max_len, want in ((0, case["plain"]), (case["max_len"], case["wrapped"])):
        hr = ctx.run_full(pcm, language="en", flags=api.NO_CONTEXT | api.TOKEN_TIMESTAMPS, max_len=max_len,
                          thold_pt=case["thold_pt"], thold_ptsum=case["thold_ptsum"])
        assert hr == 0
        got = ctx.results()
        assert len(got) == len(want)
        for g, w in zip(got, want):
            assert g["t0"] == w["t0"] * 100000 and g["t1"] == w["t1"] * 100000 and g["text"].decode() == w["text"]
            assert [t["id"] for t in g["tokens"]] == [t["id"] for t in w["tokens"]]
            assert [(t["t0"], t["t1"]) for t in g["tokens"]] == [(t["t0"] * 100000, t["t1"] * 100000) for t in w["tokens"]]
    ctx.close()
    m.close()
