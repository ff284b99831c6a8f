"""How wide is the reference's own band?  ref(1 thread) vs ref(n threads) on logits, as a function of attn_sharpness.

TEST INFRASTRUCTURE (uses oracle/_ref): sweeps `synth_model(kind, attn_sharpness=s)` and reports, per s, the max / mean
|logits(1 thread) - logits(n threads)| over a 3-token prompt + `steps` teacher-forced steps (tokens = the 1-thread reference's greedy
choice).  VERDICT r5 item 1(a): find s where the band is <= 5e-4 so north_star's 1e-3 becomes testable at full shape.
Usage: python tools/ref_band_sweep.py medium 1,2,3,4 [steps] [threads]
"""
import os, sys, time, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from whisper_amd import ggml_format as gf
from oracle import ref


def main():
    kind = sys.argv[1]
    sharps = [float(x) for x in sys.argv[2].split(",")]
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    nth = int(sys.argv[4]) if len(sys.argv) > 4 else 8
    import bench
    pcm = bench.synth_pcm(1, seed=100)[0]
    for s in sharps:
        t0 = time.time()
        model = gf.synth_model(kind, seed=1, attn_sharpness=s)
        sp = gf.special_tokens(model.hparams)
        d = tempfile.mkdtemp(dir="/tmp")
        path = os.path.join(d, "m.bin")
        gf.write_model(path, model)
        del model
        out = {}
        toks_seq = None
        for n in (1, nth):
            w = ref.RefWhisper(path, n_threads=n, log_level=0)
            mel = w.pcm_to_mel(pcm)
            w.set_mel(mel)
            w.encode(0)
            kv = [w.cross_kv(0), w.cross_kv(w_layers - 1) if False else None] if False else None
            prompt = [sp["sot"], sp["sot"] + 1, sp["transcribe"]]
            logs = []
            toks = prompt
            n_past = 0
            seq = []
            for st in range(steps + 1):
                rl, _ = w.decode([int(t) for t in toks], n_past)
                logs.append(rl[-1].copy())
                n_past += len(toks)
                if toks_seq is None:
                    nxt = int(np.argmax(rl[-1]))
                    seq.append(nxt)
                else:
                    nxt = toks_seq[st]
                toks = [nxt]
            if toks_seq is None:
                toks_seq = seq
            out[n] = np.stack(logs)
            w.close()
        os.remove(path)
        os.rmdir(d)
        diff = np.abs(out[1] - out[nth])
        span = out[1].max(axis=1) - out[1].min(axis=1)
        top2 = np.sort(out[1], axis=1)[:, -2:]
        print("kind %s sharpness %.2f: band max %.3e mean %.3e | per step max %s | logit span %.2f | min top-2 margin %.3e | top-1 equal %d/%d | %.0f s"
              % (kind, s, diff.max(), diff.mean(), " ".join("%.1e" % x for x in diff.max(axis=1)), span.mean(),
                 (top2[:, 1] - top2[:, 0]).min(), int((out[1].argmax(1) == out[nth].argmax(1)).sum()), len(span), time.time() - t0), flush=True)


if __name__ == "__main__":
    main()
