"""The sequential single-stream scenario of bench.py's `single_stream` object alone (one iContext, libWhisper.so runFull over the
scripted medium-shape model), for rocprofv3:
    cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ss -- python $REPO/tools/single_stream_prof.py
Prints seconds per run; RUNS (default 3) timed runs after one warm-up."""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from whisper_amd import api, ggml_format as gf  # noqa: E402


def main():
    kind = os.environ.get("SS_MODEL", "medium")
    runs = int(os.environ.get("RUNS", "3"))
    hp = gf.hparams_for(kind)
    cap = 102
    positions, kept = gf.carry_over_script(hp, 7, 49, cap)
    cached = os.environ.get("SS_MODEL_FILE")
    model = None if cached and os.path.exists(cached) else gf.scripted_model_at(positions, kind=kind, seed=7)
    pcm = bench.synth_pcm(7, seed=100).reshape(-1)[:int(bench.CLIP_SECONDS * 16000)]
    with tempfile.TemporaryDirectory() as td:
        path = os.environ.get("SS_MODEL_FILE") or os.path.join(td, "scripted.bin")
        if not os.path.exists(path):
            gf.write_model(path, model)
        del model
        m = api.Model(path)
        ctx = m.create_context()
        ctx.run_full(pcm, n_max_text_ctx=cap)
        for _ in range(runs):
            t0 = time.perf_counter()
            ctx.run_full(pcm, n_max_text_ctx=cap)
            dt = time.perf_counter() - t0
            print("run_full %.4f s = %.1f audio-s/s, %d decode steps" % (dt, bench.CLIP_SECONDS / dt, 7 * (kept + 1)), flush=True)
        ctx.timings_print()
        ctx.close()
        m.close()


if __name__ == "__main__":
    main()
