"""Golden transcripts of the reference's GPU-MODEL host code -- ContextImpl::runFullImpl / runFull / runStreamed / getResults
(Whisper/Whisper/ContextImpl.cpp:452-793, ContextImpl.misc.cpp) compiled unmodified, computing with the reference's own CPU model
(oracle/_ref/libcontextimpl_ref.so, oracle/contextimpl_harness.cpp) -- on "scripted" models (whisper_amd.ggml_format.scripted_model).
The companion of make_golden_hostloop.py (whisper_full): the two host loops differ in two rules, and the cases below are built so
that every window decodes its scripted tokens by a wide margin under the GPU model's rules:
  * whisper_full drops the past prompt when less than 5 s of audio remain (whisper.cpp:2874-2878); runFullImpl does not;
  * whisper_full retries a failed window once without the past prompt (whisper.cpp:3006-3016); runFullImpl skips a second at once.
Prompt lengths are held constant so that the position-coded script stays aligned: prompt = [1000] with n_max_text_ctx = 0 gives
[prev, sot, lang, task] in every window (4 tokens); n_max_text_ctx = 1 gives [prev, last token, sot, lang, task] (5 tokens) -- with
a past prompt that is NOT empty, which is where the two rule sets part.
Run in the build container: make -C oracle && python tests/golden/make_golden_hostloop_contextimpl.py"""
import json
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from whisper_amd import ggml_format as gf  # noqa: E402
from oracle import ref  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
PCM_SEED = 23


def cases():
    hp = gf.hparams_for("test-d128-ml")
    sp = gf.special_tokens(hp)
    beg, eot = sp["beg"], sp["eot"]
    script_a = [beg, 300, 301, 302, beg + 120, beg + 120, 400, 401, 402, beg + 250, beg + 250, 500, 501, beg + 360, eot]
    script_b = [beg + 10, 600, 601, beg + 200, beg + 200, 610, 611, 612, eot]           # text after the last timestamp pair, no closing timestamp
    script_c = [beg, 700, 701, 702, eot]                                                # never a timestamp: "failed to generate timestamp token"
    four = dict(prompt=[1000], n_max_text_ctx=0, prompt_len=4)
    five = dict(prompt=[1000], n_max_text_ctx=1, prompt_len=5)
    return [
        dict(name="multi_window", script=script_a, seconds=40.0, flags=dict(no_context=True), **four),
        dict(name="single_segment", script=script_a, seconds=20.0, flags=dict(no_context=True, single_segment=True), **four),
        dict(name="max_tokens", script=script_a, seconds=20.0, flags=dict(no_context=True, max_tokens=6), **four),
        dict(name="open_ended", script=script_b, seconds=31.0, flags=dict(no_context=True), **four),
        dict(name="no_timestamp", script=script_c, seconds=6.0, flags=dict(no_context=True), **four),
        dict(name="translate_de", script=script_a, seconds=12.0, flags=dict(no_context=True, translate=True), lang="de", **four),
        dict(name="too_short", script=script_a, seconds=0.9, flags=dict(no_context=True), **four),
        # a past prompt that is not empty: kept to the last window (whisper_full would drop it 5 s before the end) ...
        dict(name="carry_one_multi_window", script=script_a, seconds=40.0, flags=dict(), **five),
        # ... and kept through failed windows (36 s: the windows at 0 .. 4 s end without a timestamp more than a second before the end of
        # the audio), each skipped by a second at once (whisper_full would retry it without the prompt). With n_max_text_ctx = 0 the past
        # prompt is empty after the first window and the second window's prompt is one token shorter: not a robust script, not a case
        dict(name="carry_one_no_timestamp", script=script_c, seconds=36.0, flags=dict(), **five),
        # ContextImpl's port of the token-level timestamps and of the max_len wrap (ContextImpl.cpp:219-419, ContextImpl.misc.cpp:302-352)
        dict(name="token_timestamps", script=script_a, seconds=20.0, flags=dict(no_context=True, token_timestamps=True), **four),
        dict(name="token_timestamps_max_len", script=script_a, seconds=20.0, flags=dict(no_context=True, token_timestamps=True, max_len=10), **four),
    ]


def flags_of(fl):
    return (ref.FLAG_NO_CONTEXT if fl.get("no_context") else 0) | (ref.FLAG_SINGLE_SEGMENT if fl.get("single_segment") else 0) | \
           (ref.FLAG_TRANSLATE if fl.get("translate") else 0) | (ref.FLAG_TOKEN_TIMESTAMPS if fl.get("token_timestamps") else 0)


def main():
    out = []
    rng = np.random.default_rng(PCM_SEED)
    for c in cases():
        model = gf.scripted_model(c["script"], c["prompt_len"])
        n = int(16000 * c["seconds"])
        pcm = (0.05 * rng.standard_normal(n)).astype(np.float32)
        with tempfile.TemporaryDirectory() as td:
            path = os.path.join(td, "m.bin")
            gf.write_model(path, model)
            ci = ref.RefContextImpl(path, model.filters, encoder_threads=4)
            fl = c["flags"]
            kw = dict(lang=c.get("lang", "en"), flags=flags_of(fl), max_tokens=fl.get("max_tokens", 0), max_len=fl.get("max_len", 0), prompt=c["prompt"],
                      n_max_text_ctx=c["n_max_text_ctx"])
            hr, segs = ci.run_full(pcm, cpu_threads=4, **kw)
            new_segment = ci.new_segment
            streamed = None
            if not fl.get("token_timestamps"):          # runStreamed refuses the flag (ContextImpl.misc.cpp:393-397)
                hr2, segs2 = ci.run_streamed(pcm, cpu_threads=4, **kw)          # cpuThreads > 1: MelStreamerThread
                progress_thread = ci.progress
                hr3, segs3 = ci.run_streamed(pcm, cpu_threads=1, **kw)          # MelStreamerSimple
                key = lambda ss: [(s["t0"], s["t1"], s["text"], [t["id"] for t in s["tokens"]]) for s in ss]        # noqa: E731
                assert hr2 == hr3 == hr and key(segs2) == key(segs3) == key(segs), c["name"]
                assert progress_thread == ci.progress
                streamed = dict(progress=ci.progress)
            else:
                hr2, _ = ci.run_streamed(pcm, cpu_threads=1, **kw)
                assert hr2 == 0x80004001, hex(hr2)              # E_NOTIMPL
            ci.close()
        # every token the host loop kept is the scripted one, by a wide margin -- the case is robust against FP32 summation order
        worst = min([t["p"] for s in segs for t in s["tokens"]], default=1.0)
        assert worst > 0.9, (c["name"], worst)
        rec = dict(name=c["name"], script=c["script"], prompt_len=c["prompt_len"], n_samples=n, pcm_seed=PCM_SEED, lang=c.get("lang", "en"),
                   prompt=c["prompt"], n_max_text_ctx=c["n_max_text_ctx"], flags=fl, hr=hr, new_segment_calls=new_segment[0], new_segments=new_segment[1],
                   streamed=streamed,
                   segments=[dict(t0=s["t0"], t1=s["t1"], text=s["text"],
                                  tokens=[dict(id=t["id"], flags=t["flags"], t0=t["t0"], t1=t["t1"], p=round(t["p"], 6), vlen=round(t["vlen"], 6)) for t in s["tokens"]])
                             for s in segs])
        print(c["name"], "hr", hr, "->", len(segs), "segments", [(s["t0"] // 100000, s["t1"] // 100000) for s in segs][:10], "min p %.3f" % worst,
              "progress", (streamed or {}).get("progress"))
        out.append(rec)
    # Numerics through the GPU model's host loop: the audio-CONDITIONED models of make_golden_runfull.py (tokens and timestamps depend on the
    # audio: a wrong logit moves a timestamp, the timestamp moves the next window's seek) through ContextImpl::runFull at 1, 4 and 8 decoder
    # threads -- the transcript must not depend on the thread count, as there -- next to whisper_full's transcript of the same case.
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden_runfull", os.path.join(HERE, "make_golden_runfull.py"))
    rf_mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rf_mod)
    wf = {c["name"]: c for c in json.load(open(os.path.join(HERE, "ref_runfull_conditioned.json")))["cases"]}
    conditioned = []
    for name, c in wf.items():
        pcm = rf_mod.pcm_for(c["pcm"])
        model = rf_mod.model_for(c["seed"])
        res = {}
        with tempfile.TemporaryDirectory() as td:
            path = os.path.join(td, "m.bin")
            gf.write_model(path, model)
            for nt in (1, 4, 8):
                ci = ref.RefContextImpl(path, model.filters, encoder_threads=nt)
                hr, segs = ci.run_full(pcm, cpu_threads=nt, lang="en", flags=ref.FLAG_NO_CONTEXT, prompt=c["prompt"], n_max_text_ctx=c["n_max_text_ctx"])
                assert hr == 0
                res[nt] = [dict(t0=s["t0"], t1=s["t1"], text=s["text"], tokens=[t["id"] for t in s["tokens"]], probs=[round(t["p"], 5) for t in s["tokens"]]) for s in segs]
                ci.close()
        strip = lambda r: [(s["t0"], s["t1"], s["tokens"]) for s in r]          # noqa: E731
        assert strip(res[1]) == strip(res[4]) == strip(res[8]), name
        # iContext::runStreamed on the same recording (row f1 end to end): the spectrogram comes window by window from the reference's MelStreamer,
        # normalised on the window's own maximum, so the logits are NOT runFull's. Both streamers (cpuThreads 1: on demand, 4 / 8: background
        # thread) and all thread counts must agree among themselves.
        streamed = {}
        with tempfile.TemporaryDirectory() as td:
            path = os.path.join(td, "m.bin")
            gf.write_model(path, model)
            for nt in (1, 4, 8):
                ci = ref.RefContextImpl(path, model.filters, encoder_threads=nt)
                hr, segs = ci.run_streamed(pcm, cpu_threads=nt, lang="en", flags=ref.FLAG_NO_CONTEXT, prompt=c["prompt"], n_max_text_ctx=c["n_max_text_ctx"])
                assert hr == 0
                streamed[nt] = [dict(t0=s["t0"], t1=s["t1"], text=s["text"], tokens=[t["id"] for t in s["tokens"]], probs=[round(t["p"], 5) for t in s["tokens"]]) for s in segs]
                progress = ci.progress
                ci.close()
        assert strip(streamed[1]) == strip(streamed[4]) == strip(streamed[8]), (name, "runStreamed")
        print("conditioned", name, "runStreamed ->", len(streamed[4]), "segments;", "the same transcript as runFull" if strip(streamed[4]) == strip(res[4]) else "DIFFERS from runFull (window-local normalisation)")
        same = strip(res[4]) == [(s["t0"] * 100000, s["t1"] * 100000, s["tokens"]) for s in c["segments"]]
        print("conditioned", name, "->", len(res[4]), "segments;", "the same transcript as whisper_full" if same else "DIFFERS from whisper_full (rules)")
        conditioned.append(dict(name=name, pcm=c["pcm"], seed=c["seed"], n_samples=len(pcm), prompt=c["prompt"], n_max_text_ctx=c["n_max_text_ctx"],
                                same_as_whisper_full=same, min_logit_margin=c["min_logit_margin"], segments=res[4],
                                streamed=dict(segments=streamed[4], progress=progress, same_as_run_full=strip(streamed[4]) == strip(res[4]))))

    # iContext::getResults / makeResults (ContextImpl.misc.cpp:196-300) under every combination of eResultFlags, with a buffer whose media time
    # is not zero: what the POD structures of API/TranscribeStructs.h carry then (times scaled to 100 ns ticks + the media time, zero without
    # Timestamps; no token array without Tokens, firstToken / countTokens all the same; eTokenFlags::Special from token_eot on)
    c = [x for x in cases() if x["name"] == "translate_de"][0]
    model = gf.scripted_model(c["script"], c["prompt_len"])
    n = int(16000 * c["seconds"])
    pcm = (0.05 * np.random.default_rng(PCM_SEED + 1).standard_normal(n)).astype(np.float32)
    media_time = 76543210
    variants = {}
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "m.bin")
        gf.write_model(path, model)
        ci = ref.RefContextImpl(path, model.filters, encoder_threads=4)
        hr, _ = ci.run_full(pcm, cpu_threads=4, lang="de", flags=flags_of(c["flags"]), prompt=c["prompt"], n_max_text_ctx=c["n_max_text_ctx"], media_time=media_time)
        assert hr == 0
        for rf in (0, 1, 2, 3):
            variants[str(rf)] = [dict(t0=s["t0"], t1=s["t1"], text=s["text"], first_token=s["first_token"], count_tokens=s["count_tokens"],
                                      tokens=[dict(id=t["id"], flags=t["flags"], t0=t["t0"], t1=t["t1"], text=t["text"]) for t in s["tokens"]])
                                 for s in ci.results(rf)]
        ci.close()
    results = dict(case="translate_de", pcm_seed=PCM_SEED + 1, n_samples=n, media_time=media_time, by_flags=variants)
    with open(os.path.join(HERE, "ref_hostloop_contextimpl.json"), "w") as f:
        json.dump(dict(cases=out, conditioned=conditioned, results=results), f, indent=1)


if __name__ == "__main__":
    main()
