"""The reference CPU path at 8 / 16 / 32 / 64 threads on the GPU box's host cores (the cpu_baseline uses 16 of the host's threads: say once
what the other counts give). One 30 s window of the medium shape: encoder + 3-token prompt + 8 decode steps, timed; audio-s/s for the bench's 52 decode steps per
window extrapolated from the measured per-step time. TEST INFRASTRUCTURE (oracle/_ref). Usage: python tools/cpu_sweep.py [medium] [threads ...]"""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from whisper_amd import ggml_format as gf
from oracle import ref
import bench

kind = sys.argv[1] if len(sys.argv) > 1 else "medium"
threads = [int(x) for x in sys.argv[2:]] or [8, 16, 32, 64]
model = gf.synth_model(kind, seed=1)
sp = gf.special_tokens(model.hparams)
d = tempfile.mkdtemp(dir="/tmp")
path = os.path.join(d, "m.bin")
gf.write_model(path, model)
del model
pcm = bench.synth_pcm(1, seed=100)[0]
print("host: %d logical CPUs" % os.cpu_count(), flush=True)
for n in threads:
    w = ref.RefWhisper(path, n_threads=n, log_level=0)
    mel = w.pcm_to_mel(pcm)
    w.set_mel(mel)
    t0 = time.time(); w.encode(0); t_enc = time.time() - t0
    toks, n_past = [sp["sot"], sp["sot"] + 1, sp["transcribe"]], 0
    t0 = time.time(); rl, _ = w.decode(toks, n_past); t_prompt = time.time() - t0
    n_past += 3
    t0 = time.time()
    for _ in range(8):
        rl, _ = w.decode([int(np.argmax(rl[-1]))], n_past); n_past += 1
    t_step = (time.time() - t0) / 8
    total = t_enc + t_prompt + 51 * t_step
    print("%s, %3d threads: encoder %.2f s, prompt step %.3f s, decode step %.4f s -> %.2f s per 30 s window (1 + 51 steps) = %.2f audio-s/s" % (kind, n, t_enc, t_prompt, t_step, total, 30.0 / total), flush=True)
    w.close()
os.remove(path); os.rmdir(d)
