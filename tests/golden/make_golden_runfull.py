"""Generates tests/golden/ref_runfull_conditioned.json: transcripts of the reference's whisper_full (Whisper/source/whisper.cpp:2765,
the loop ContextImpl::runFullImpl ports) on models whose tokens and timestamps DEPEND ON THE AUDIO
(whisper_amd.ggml_format.conditioned_model) -- north_star's "identical token ids on jfk.wav", through the whole host loop
(seek by the last timestamp, stop rules, segment cutting), with numerics in the loop: a wrong logit moves a timestamp, the
timestamp moves the next window.

Run in the build container: make -C oracle && python tests/golden/make_golden_runfull.py

Every case is run with 1, 4 and 8 reference threads and must give the same transcript (the reference's own FP16 P.V noise,
ggml.c:4689-4735, must not decide a token), otherwise the case is rejected here."""
import json
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from whisper_amd import ggml_format as gf  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
KIND, PROMPT_LEN = "test-d128-ml", 4           # prompt = [prev, sot, lang, task] in every window (one dummy prompt token, n_max_text_ctx = 0)
CASES = [dict(name="%s_s%d" % (n, sd), pcm=n, seed=sd) for sd in (10, 11) for n in ("jfk", "long", "mixed")]


def jfk_pcm():
    """SampleClips/jfk.wav (11 s) as stored in ref_test_d128.npz"""
    return np.load(os.path.join(HERE, "ref_test_d128.npz"))["pcm16"].astype(np.float32) / 32768.0


def pcm_for(name: str) -> np.ndarray:
    """The recordings of the cases, rebuilt from jfk.wav and seeded noise (the tests call this too)."""
    jfk = jfk_pcm()
    rng = np.random.default_rng(3)
    noisy = (jfk * 0.5 + 0.01 * rng.standard_normal(len(jfk))).astype(np.float32)
    if name == "jfk":
        return jfk
    if name == "long":          # 60.5 s: five windows whose seeks differ with the audio
        return np.concatenate([jfk, 0.3 * jfk[::-1], noisy, jfk, jfk[::2], jfk]).astype(np.float32)
    if name == "mixed":         # 38.5 s
        return np.concatenate([jfk[::-1], noisy, jfk[::2], jfk]).astype(np.float32)
    if name == "quiet":         # 27.5 s
        return np.concatenate([0.05 * jfk, noisy[::-1], 0.3 * jfk[::2]]).astype(np.float32)
    raise KeyError(name)


def model_for(seed: int):
    hp = gf.hparams_for(KIND)
    return gf.conditioned_model(gf.conditioned_layout(hp), PROMPT_LEN, kind=KIND, seed=seed)


def replay_margins(w, pcm, sp, want_tokens):
    """The same windows once more through whisper_decode step by step (a plain restatement of the loop for THIS layout: every
    window ends on EOT after its last timestamp), to learn how far the runner-up was: returns the smallest top-1 / top-2 logit
    margin along the transcript. Asserts that the replay chooses whisper_full's tokens."""
    w.pcm_to_mel(pcm)
    n_frames = len(pcm) // 160
    seek, got, margin = 0, [], 1e9
    prompt = [sp["prev"], sp["sot"], sp["sot"] + 1, sp["transcribe"]]
    while seek + 100 < n_frames:
        w.encode(seek)
        toks, n_past, delta, cur = list(prompt), 0, 3000, []
        for i in range(220):
            logits, _ = w.decode(toks, n_past)
            sb = w.sample_timestamp(True) if i == 0 else w.sample_best()
            top = np.sort(logits[-1][sp["beg"]:] if sb["id"] >= sp["beg"] else logits[-1])[-2:]
            margin = min(margin, float(top[1] - top[0]))
            n_past += len(toks)
            toks = [sb["id"]]
            if sb["id"] == sp["eot"]:
                break
            cur.append(sb["id"])
            if sb["id"] > sp["beg"]:
                delta = 2 * (sb["id"] - sp["beg"])
        got += cur
        seek += delta
    # segments hold the text tokens and their closing timestamp, not the window's leading / repeated timestamps: compare the text
    text = lambda ids: [t for t in ids if t < sp["eot"]]
    assert text(got) == text(want_tokens), (got, want_tokens)
    return margin


MIN_MARGIN = 0.04      # logits, between candidates ~2 apart (seeds were chosen for this: 20 tried, 6 .. 25); the reference at 1, 4 and 8 threads agrees


def main():
    from oracle import ref
    out = []
    for c in CASES:
        pcm = pcm_for(c.get("pcm", c["name"]))
        with tempfile.TemporaryDirectory() as td:
            path = os.path.join(td, "m.bin")
            gf.write_model(path, model_for(c["seed"]))
            res = {}
            for nt in (1, 4, 8):
                w = ref.RefWhisper(path, n_threads=nt, log_level=0)
                segs = w.full(pcm, no_context=True, prompt=[1000], n_max_text_ctx=0)
                res[nt] = [dict(t0=s["t0"], t1=s["t1"], text=s["text"].decode(), tokens=s["tokens"], probs=[round(float(p), 5) for p in s["probs"]]) for s in segs]
                w.close()
        strip = lambda r: [(s["t0"], s["t1"], s["tokens"]) for s in r]
        ok = strip(res[1]) == strip(res[4]) == strip(res[8])
        margin = None
        if ok:
            with tempfile.TemporaryDirectory() as td:
                path = os.path.join(td, "m.bin")
                gf.write_model(path, model_for(c["seed"]))
                w = ref.RefWhisper(path, n_threads=4, log_level=0)
                margin = replay_margins(w, pcm, gf.special_tokens(gf.hparams_for(KIND)), [t for s in res[4] for t in s["tokens"]])
                w.close()
            print("    smallest top-1 / top-2 logit margin along the transcript: %.3f" % margin)
            ok = margin >= MIN_MARGIN
        print(c["name"], "seed", c["seed"], "%.1f s" % (len(pcm) / 16000.0), "->", len(res[4]), "segments", "" if ok else "REJECTED: depends on the reference's thread count")
        for s in res[4]:
            print("    ", s["t0"], s["t1"], s["tokens"])
        if ok:
            out.append(dict(name=c["name"], pcm=c.get("pcm", c["name"]), seed=c["seed"], n_samples=len(pcm), prompt=[1000], n_max_text_ctx=0, min_logit_margin=round(margin, 4), segments=res[4]))
    assert len(out) >= 4
    with open(os.path.join(HERE, "ref_runfull_conditioned.json"), "w") as f:
        json.dump(dict(kind=KIND, prompt_len=PROMPT_LEN, cases=out), f, indent=1)


if __name__ == "__main__":
    main()
