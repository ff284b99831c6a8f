"""Golden transcripts of the reference's complete host loop (whisper_full, the code ContextImpl::runFullImpl ports) on
"scripted" models whose greedy token sequence is known in advance (whisper_amd.ggml_format.scripted_model).
Run in the build container: make -C oracle && python tests/golden/make_golden_hostloop.py"""
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


def cases():
    hp = gf.hparams_for("test-d128-ml")
    sp = gf.special_tokens(hp)
    beg, eot = sp["beg"], sp["eot"]
    # every window's prompt is [prev, sot, lang, task] when one dummy prompt token is given and n_max_text_ctx = 0
    script_a = [beg, 300, 301, 302, beg + 120, beg + 120, 400, 401, 402, beg + 250, beg + 250, 500, 501, beg + 360, eot]
    # no closing timestamp before EOT, text after the last timestamp pair
    script_b = [beg + 10, 600, 601, beg + 200, beg + 200, 610, 611, 612, eot]
    # never produces a timestamp after the first token: the "failed to generate timestamp token" path
    script_c = [beg, 700, 701, 702, eot]
    common = dict(prompt=[1000], n_max_text_ctx=0)
    return [
        dict(name="multi_window", script=script_a, prompt_len=4, seconds=40.0, flags=dict(no_context=True), **common),
        dict(name="first_window_no_prompt", script=script_a, prompt_len=3, seconds=9.0, flags=dict(no_context=True), prompt=None, n_max_text_ctx=-1),
        dict(name="single_segment", script=script_a, prompt_len=4, seconds=20.0, flags=dict(no_context=True, single_segment=True), **common),
        dict(name="max_tokens", script=script_a, prompt_len=4, seconds=20.0, flags=dict(no_context=True, max_tokens=6), **common),
        dict(name="open_ended", script=script_b, prompt_len=4, seconds=31.0, flags=dict(no_context=True), **common),
        dict(name="no_timestamp", script=script_c, prompt_len=4, seconds=6.0, flags=dict(no_context=True), **common),
        dict(name="translate_de", script=script_a, prompt_len=4, seconds=12.0, flags=dict(no_context=True, translate=True), lang="de", **common),
        dict(name="too_short", script=script_a, prompt_len=4, seconds=0.9, flags=dict(no_context=True), **common),
        # the reference's own sample clip (SampleClips/jfk.wav, 11 s; its PCM is stored in ref_test_d128.npz) with the default
        # parameters of the CLI: two windows, the second one seeks to the last timestamp of the first
        dict(name="jfk_wav", script=script_a, prompt_len=3, seconds=11.0, pcm="jfk", flags=dict(no_context=True), prompt=None, n_max_text_ctx=-1),
    ]


def main():
    out = []
    rng = np.random.default_rng(11)
    for c in cases():
        model = gf.scripted_model(c["script"], c["prompt_len"])
        if c.get("pcm") == "jfk":
            pcm = np.load(os.path.join(HERE, "ref_test_d128.npz"))["pcm16"].astype(np.float32) / 32768.0
            n = len(pcm)
        else:
            n = int(16000 * c["seconds"])
            pcm = (0.05 * rng.standard_normal(n)).astype(np.float32)
        with tempfile.TemporaryDirectory() as td:
            path = os.path.join(td, "m.bin")
            gf.write_model(path, model)
            w = ref.RefWhisper(path, n_threads=4, log_level=0)
            segs = w.full(pcm, lang=c.get("lang", "en"), prompt=c["prompt"], n_max_text_ctx=c["n_max_text_ctx"], **c["flags"])
            w.close()
        rec = dict(name=c["name"], script=c["script"], prompt_len=c["prompt_len"], n_samples=n, pcm_seed=11, pcm=c.get("pcm", "noise"), lang=c.get("lang", "en"),
                   prompt=c["prompt"], n_max_text_ctx=c["n_max_text_ctx"], flags=c["flags"],
                   segments=[dict(t0=s["t0"], t1=s["t1"], text=s["text"].decode(), tokens=s["tokens"]) for s in segs])
        print(c["name"], "->", len(segs), "segments", [(s["t0"], s["t1"]) for s in segs][:8])
        out.append(rec)
    # tokenizer: whisper_tokenize of the reference (whisper.cpp:2186-2248) on the stand-in vocabulary
    model = gf.synth_model("test-d128-ml")
    tok = {}
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "m.bin")
        gf.write_model(path, model)
        w = ref.RefWhisper(path, n_threads=1, log_level=0)
        for text in [" w300 w4242", "hello, world! 123", "it's  a\ttest", "", "\u00e9t\u00e9 na\u00efve"]:
            buf = np.zeros(256, np.int32)
            n = w.L.ref_tokenize(w.ctx, text.encode(), buf, 256)
            tok[text] = [int(x) for x in buf[:n]]
        w.close()
    with open(os.path.join(HERE, "ref_hostloop.json"), "w") as f:
        json.dump(dict(cases=out, tokenize=tok), f, indent=1)


if __name__ == "__main__":
    main()
