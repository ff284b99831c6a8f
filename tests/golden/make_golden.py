"""Generates tests/golden/*.npz from the reference's own CPU path (oracle/_ref = Whisper/source/{whisper.cpp,ggml.c}
compiled unmodified by oracle/Makefile). Run in the build container, where /root/reference exists:

    make -C oracle && python tests/golden/make_golden.py

The reference tree ships no golden vectors for this path, so these fixtures -- outputs of the
reference itself on seeded inputs -- are what pins both the numpy restatement (oracle/whisper_np.py) and the HIP path
on machines where /root/reference is absent (the GPU box). Inputs: the synthetic model `test-d128` (regenerated from its
seed by whisper_amd.ggml_format.synth_model, never stored) and the first 11 s of SampleClips/jfk.wav.
"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from whisper_amd import ggml_format as gf  # noqa: E402
from oracle import ref  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
MODEL_KIND, MODEL_SEED, SHARPNESS = "test-d128", 1234, 2.0
N_THREADS = 1   # pinned: the reference's decoder output depends on the thread count


def main():
    model = gf.synth_model(MODEL_KIND, seed=MODEL_SEED, attn_sharpness=SHARPNESS)
    sp = gf.special_tokens(model.hparams)
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "m.bin")
        gf.write_model(path, model)
        w = ref.RefWhisper(path, n_threads=N_THREADS, log_level=0)
        pcm = ref.read_wav_mono16("/root/reference/SampleClips/jfk.wav")
        mel = w.pcm_to_mel(pcm)
        w.trace(True)
        w.encode(0)
        tr = w.traced()
        out = dict(pcm16=(pcm * 32768.0).astype(np.int16), mel=mel,
                   enc_kqv0=tr["enc-KQV"].astype(np.float16),
                   encode_out=tr["encode-out"].astype(np.float32))
        for il in (0, model.hparams.n_text_layer - 1):
            k, v = w.cross_kv(il)
            out["cross_k%d" % il] = k.astype(np.float16)
            out["cross_v%d" % il] = v.astype(np.float16)
        # forced-token decode: a 3-token prompt then 4 single-token steps
        prompt = [sp["sot"], sp["transcribe"], sp["not_"]]
        steps = [prompt, [sp["beg"]], [1234], [40000], [220]]
        n_past = 0
        for i, toks in enumerate(steps):
            w.trace(False)
            w.trace(True)
            logits, probs = w.decode(toks, n_past)
            tr = w.traced()
            out["logits%d" % i] = logits[-1].astype(np.float32)
            sb = w.sample_best()
            st = w.sample_timestamp(i == 0)
            out["sample%d" % i] = np.array([sb["id"], sb["tid"], st["id"], st["tid"]], np.int32)
            out["samplep%d" % i] = np.array([sb["p"], sb["pt"], sb["ptsum"], st["p"], st["pt"], st["ptsum"]], np.float32)
            if i == 0:
                out["dec_kqv_self0"] = tr["dec-KQV"].astype(np.float32)
                out["dec_kqv_cross0"] = tr["dec-KQV#2"].astype(np.float32)
                out["probs0_sum"] = np.array([probs[-1].astype(np.float64).sum()])
            n_past += len(toks)
        k, v = w.self_kv(0, n_past)
        out["self_k0"] = k.astype(np.float16)
        out["self_v0"] = v.astype(np.float16)
        out["steps"] = np.array([t for s in steps for t in s], np.int32)
        out["step_lens"] = np.array([len(s) for s in steps], np.int32)
        # greedy whisper_full on the clip (token ids of the reference's complete host loop)
        segs = w.full(pcm, lang="en", no_context=True, max_tokens=0)
        ids = [t for s in segs for t in s["tokens"]]
        out["full_tokens"] = np.array(ids, np.int32)
        out["full_seg_t"] = np.array([[s["t0"], s["t1"]] for s in segs], np.int64)
        out["full_seg_ntok"] = np.array([len(s["tokens"]) for s in segs], np.int32)
        g, e = ref.lookup_tables()
        out["table_gelu"] = g
        out["table_exp"] = e
    np.savez_compressed(os.path.join(HERE, "ref_test_d128.npz"), **out)
    sz = os.path.getsize(os.path.join(HERE, "ref_test_d128.npz"))
    print("wrote ref_test_d128.npz: %.2f MB, %d arrays; full() produced %d tokens in %d segments" % (sz / 1e6, len(out), len(ids), len(segs)))


if __name__ == "__main__":
    main()
