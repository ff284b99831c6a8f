"""Whisper::createBatchRunner / runFullBatch (libWhisper.so): K streams in lock step behind the drop-in boundary.

The contract under test: every stream keeps the semantics of iContext::runFull (Whisper/Whisper/ContextImpl.cpp:452-793) -- its transcript
is the transcript runFull gives for the same samples -- whatever its neighbours in the batch do: other recordings, other lengths, windows
that end at different tokens, streams that finish early and are replaced, prompts of different lengths (prompt carry-over), idle slots.
Oracles: the reference's whisper_full transcripts committed under tests/golden (scripted models: ref_hostloop.json; models whose tokens and
timestamps depend on the audio: ref_runfull_conditioned.json) and K sequential runFull calls through the same library.
"""
import ctypes
import importlib.util
import json
import os
import subprocess

import numpy as np
import pytest

from whisper_amd import api, build as wbuild, ggml_format as gf

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOSTLOOP = os.path.join(ROOT, "tests", "golden", "ref_hostloop.json")
CONDITIONED = os.path.join(ROOT, "tests", "golden", "ref_runfull_conditioned.json")


def _generator():
    spec = importlib.util.spec_from_file_location("make_golden_runfull", os.path.join(ROOT, "tests", "golden", "make_golden_runfull.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def strip(segs):
    return [(s["t0"], s["t1"], s["text"], [t["id"] for t in s["tokens"]]) for s in segs]


def test_exports_and_flat_mirror():
    """CPU: the extension entry points are exported next to the seven names of whisper.def, and every whisperc_batch_* / whisperc_tr_*
    declaration of include/whisper_c.h resolves."""
    wbuild.build_all()
    lib = ctypes.CDLL(api.HOST_LIB_PATH)
    out = subprocess.run(["nm", "-D", "--defined-only", api.HOST_LIB_PATH], stdout=subprocess.PIPE, text=True, check=True).stdout
    for name in api.CPP_EXTENSIONS:
        assert any(("Whisper" in ln and name in ln) for ln in out.splitlines()), name
    for name in ("whisperc_batch_create", "whisperc_batch_run", "whisperc_tr_counts", "whisperc_tr_segment", "whisperc_tr_token"):
        getattr(lib, name)
    # a null model is refused before anything touches a device
    r = ctypes.c_void_p()
    assert lib.whisperc_batch_create(None, 0, 0, 0, 0, ctypes.byref(r)) & 0xFFFFFFFF == 0x80004003


@pytest.mark.gpu
def test_batch_of_scripted_streams_equals_run_full_and_the_reference(tmp_path):
    """One scripted model, streams of different lengths (40 s = three windows, 20 s, 12 s, 31 s, 0.9 s = too short, 6 s): the batch's
    transcripts are those of sequential runFull calls, and the 40 s stream's is the reference's whisper_full transcript of the
    multi_window case. Two slots for six streams: finished streams are replaced while their neighbours are mid-recording."""
    cases = {c["name"]: c for c in json.load(open(HOSTLOOP))["cases"]}
    c = cases["multi_window"]
    rng = np.random.default_rng(c["pcm_seed"])
    first = (0.05 * rng.standard_normal(c["n_samples"])).astype(np.float32)        # the golden case's own PCM (first draw of the seed)
    rng2 = np.random.default_rng(99)
    pcms = [first] + [(0.05 * rng2.standard_normal(int(16000 * s))).astype(np.float32) for s in (20.0, 12.0, 31.0, 0.9, 6.0)]
    path = str(tmp_path / "scripted.bin")
    gf.write_model(path, gf.scripted_model(c["script"], c["prompt_len"]))
    m = api.Model(path)
    ctx = m.create_context()
    want = []
    for pcm in pcms:
        hr = ctx.run_full(pcm, flags=api.NO_CONTEXT, prompt=c["prompt"], n_max_text_ctx=c["n_max_text_ctx"])
        want.append((hr, strip(ctx.results())))
    assert [w["tokens"] for w in c["segments"]] == [ids for (_, _, _, ids) in want[0][1]]
    for slots, groups in ((2, 1), (2, 2), (6, 1), (3, 2)):
        runner = m.create_batch_runner(max_slots=slots, groups=groups)
        hr, got, per = runner.run(pcms, flags=api.NO_CONTEXT, prompt=c["prompt"], n_max_text_ctx=c["n_max_text_ctx"])
        assert hr == 0
        for i, (w_hr, w) in enumerate(want):
            assert per[i] == w_hr, (slots, groups, i, per)
            assert strip(got[i]) == w, (slots, groups, i)
        # the runner is reusable: the same call again, same transcripts
        hr2, got2, _ = runner.run(pcms[:3], flags=api.NO_CONTEXT, prompt=c["prompt"], n_max_text_ctx=c["n_max_text_ctx"])
        assert hr2 == 0 and [strip(g) for g in got2] == [w for (_, w) in want[:3]]
        runner.close()
    ctx.close()
    m.close()


@pytest.mark.gpu
def test_batch_on_audio_conditioned_models_matches_whisper_full(tmp_path):
    """Numerics in the loop: models whose tokens AND timestamps depend on the audio (ggml_format.conditioned_model). The three recordings of
    one model run as ONE lock-step batch (their windows seek differently, end at different tokens, and the 11 s recording finishes while
    the 60 s one has four windows to go) and must reproduce the reference's whisper_full transcripts (tests/golden/ref_runfull_conditioned.json):
    ids, segment boundaries, tick times."""
    mg = _generator()
    fx = json.load(open(CONDITIONED))
    by_seed = {}
    for c in fx["cases"]:
        by_seed.setdefault(c["seed"], []).append(c)
    for seed, cases in by_seed.items():
        path = str(tmp_path / ("cond%d.bin" % seed))
        gf.write_model(path, mg.model_for(seed))
        m = api.Model(path)
        pcms = [mg.pcm_for(c["pcm"]) for c in cases]
        for slots, groups in ((3, 1), (2, 1), (1, 2)):
            runner = m.create_batch_runner(max_slots=slots, groups=groups)
            hr, got, per = runner.run(pcms + pcms[::-1], flags=api.NO_CONTEXT, prompt=cases[0]["prompt"], n_max_text_ctx=cases[0]["n_max_text_ctx"])
            assert hr == 0 and all(p == 0 for p in per)
            for c, g in list(zip(cases, got[:len(cases)])) + list(zip(cases[::-1], got[len(cases):])):
                want = [(s["t0"] * 100000, s["t1"] * 100000, s["text"], s["tokens"]) for s in c["segments"]]
                have = [(s["t0"], s["t1"], s["text"].decode(), [t["id"] for t in s["tokens"]]) for s in g]
                assert have == want, (seed, slots, groups, c["name"])
                worst = max(abs(t["p"] - p) for s, w in zip(g, c["segments"]) for t, p in zip(s["tokens"], w["probs"]))
                assert worst < 2e-2
            runner.close()
        m.close()


@pytest.mark.gpu
def test_batch_with_prompt_carry_over(tmp_path):
    """Streams that carry their past text into the next window's prompt (no NoContext between windows, n_max_text_ctx = cap): the prompts of
    a round differ in length from slot to slot (a stream in its first window has 3 tokens, one in its third 3 + 1 + cap), i.e. the
    sequences of the lock-step batch stand at different decoder positions. Every stream must transcribe what runFull transcribes."""
    hp = gf.hparams_for("test-d128-ml")
    cap = 24
    positions, kept = gf.carry_over_script(hp, 4, 10, cap)      # prompts of 3, 16, 28, 28 tokens from window to window
    path = str(tmp_path / "carry.bin")
    gf.write_model(path, gf.scripted_model_at(positions))
    m = api.Model(path)
    rng = np.random.default_rng(5)
    pcms = [(0.05 * rng.standard_normal(int(16000 * s))).astype(np.float32) for s in (95.0, 31.0, 61.0, 29.0, 118.0)]
    ctx = m.create_context()
    want = []
    for pcm in pcms:
        assert ctx.run_full(pcm, flags=api.NO_CONTEXT, n_max_text_ctx=cap) == 0
        want.append(strip(ctx.results()))
    assert len(want[0]) >= 3 and len(want[4]) >= 4
    for slots, groups in ((5, 1), (2, 1), (2, 2)):
        runner = m.create_batch_runner(max_slots=slots, groups=groups, greedy_chunk=3)
        hr, got, _ = runner.run(pcms, flags=api.NO_CONTEXT, n_max_text_ctx=cap)
        assert hr == 0
        for i in range(len(pcms)):
            assert strip(got[i]) == want[i], (slots, groups, i)
        runner.close()
    ctx.close()
    m.close()


@pytest.mark.gpu
def test_chunked_streams_are_recordings_of_their_own(tmp_path):
    """sBatchStream::firstSample / countSamples: independent 30 s chunks of ONE recording (north_star's sharding unit) declared as streams.
    Chunk k must transcribe exactly like runFull on a buffer holding only its samples, with its times shifted by k * 30 s."""
    mg = _generator()
    path = str(tmp_path / "cond.bin")
    gf.write_model(path, mg.model_for(10))
    m = api.Model(path)
    pcm = mg.pcm_for("long")                         # 60.5 s
    n = len(pcm)
    win = 30 * 16000
    pieces = [(k * win, min(win, n - k * win)) for k in range((n + win - 1) // win)]
    ctx = m.create_context()
    want = []
    for f, c in pieces:
        hr = ctx.run_full(np.ascontiguousarray(pcm[f:f + c]), flags=api.NO_CONTEXT, prompt=[1000], n_max_text_ctx=0)
        shift = f * 10000000 // 16000
        want.append((hr, [(t0 + shift, t1 + shift, text, ids) for (t0, t1, text, ids) in strip(ctx.results())]))
    runner = m.create_batch_runner(max_slots=4, groups=1)
    hr, got, per = runner.run([(pcm, f, c) for f, c in pieces], flags=api.NO_CONTEXT, prompt=[1000], n_max_text_ctx=0)
    assert hr == 0
    for i, (w_hr, w) in enumerate(want):
        assert per[i] == w_hr and strip(got[i]) == w, i
    assert sum(len(w) for _, w in want) >= 4
    # a stream that names samples outside its buffer fails alone; its neighbours are transcribed
    with pytest.raises(api.WhisperError):
        runner.run([(pcm, 0, win), (pcm, n - 10, 100)], flags=api.NO_CONTEXT, prompt=[1000], n_max_text_ctx=0)
    runner.close()
    ctx.close()
    m.close()


@pytest.mark.gpu
def test_batch_caller_cpp(tmp_path):
    """The C++ face with callbacks (tests/abi_caller/batch_caller.cpp, compiled against include/whisperApi.h only): runFullBatch with
    new_segment / encoder_begin callbacks on every stream vs K sequential iContext::runFull calls in the same program; it exits 0
    only when transcripts, callback counts and the callbacks' views of getResults agree."""
    exe = wbuild.build_batch_caller()
    mg = _generator()
    path = str(tmp_path / "cond.bin")
    gf.write_model(path, mg.model_for(11))
    wavs = []
    for name in ("jfk", "long", "mixed"):
        p = str(tmp_path / (name + ".wav"))
        with open(p, "wb") as f:
            f.write(api.wav_bytes(mg.pcm_for(name)))
        wavs.append(p)
    r = subprocess.run([exe, path] + wavs, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    print(r.stdout[-3000:])
    assert r.returncode == 0
    assert "BATCH_CALLER_OK" in r.stdout


@pytest.mark.gpu
def test_batch_stress_random_lengths_and_flags(tmp_path):
    """Many streams of random lengths (0.4 s .. 70 s; some too short, some exactly at window boundaries) on an audio-conditioned model, few slots,
    small chunks, with and without the extra queued chunk (sBatchSetup.flags bit 0): slots are refilled all the time, windows of a round end at
    different tokens, a group runs half empty at the tail. Every stream must equal its sequential runFull; the same under TokenTimestamps (token
    times and vlen are host-side post-processing of the same tokens) and SingleSegment."""
    mg = _generator()
    path = str(tmp_path / "cond.bin")
    gf.write_model(path, mg.model_for(10))
    m = api.Model(path)
    base = np.concatenate([mg.pcm_for("long"), mg.pcm_for("mixed")])             # 99 s of material
    rng = np.random.default_rng(21)
    pcms = []
    for s in [0.4, 30.0, 60.0, 0.99, 1.5] + [float(x) for x in rng.uniform(2.0, 70.0, 18)]:
        n = int(s * 16000)
        off = int(rng.integers(0, len(base) - n))
        pcms.append(np.ascontiguousarray(base[off:off + n]))
    ctx = m.create_context()

    def sequential(flags):
        out = []
        for pcm in pcms:
            hr = ctx.run_full(pcm, flags=flags, prompt=[1000], n_max_text_ctx=0)
            segs = ctx.results()
            out.append((hr, [(s["t0"], s["t1"], s["text"], [(t["id"], t["t0"], t["t1"], round(t["vlen"], 4)) for t in s["tokens"]]) for s in segs]))
        return out

    def batched(runner, flags):
        hr, got, per = runner.run(pcms, flags=flags, prompt=[1000], n_max_text_ctx=0)
        assert hr == 0
        return [(per[i], [(s["t0"], s["t1"], s["text"], [(t["id"], t["t0"], t["t1"], round(t["vlen"], 4)) for t in s["tokens"]]) for s in got[i]]) for i in range(len(pcms))]

    for flags in (api.NO_CONTEXT, api.NO_CONTEXT | api.TOKEN_TIMESTAMPS, api.NO_CONTEXT | api.SINGLE_SEGMENT):
        want = sequential(flags)
        assert sum(len(w) for _, w in want) > 20
        for slots, groups, chunk, fl in ((4, 2, 2, 0), (3, 1, 5, 1), (16, 2, 0, 0)):
            runner = m.create_batch_runner(max_slots=slots, groups=groups, greedy_chunk=chunk, flags=fl)
            got = batched(runner, flags)
            for i in range(len(pcms)):
                assert got[i] == want[i], (flags, slots, groups, chunk, fl, i, len(pcms[i]) / 16000.0)
            runner.close()
    ctx.close()
    m.close()
