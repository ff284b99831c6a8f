#!/usr/bin/env python3
"""bench.py -- audio-seconds/sec of the Whisper hot path (PCM -> mel -> encoder -> KV-cached greedy decoder -> token ids).

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched under
torch.distributed.run with one rank per GPU. Prints ONE JSON line on rank 0.

Default workload at N = 1 = BASELINE.json configs[1]: a ggml-medium-shaped model (random FP16 weights of the exact real
shapes -- no real weights exist offline) on a clip of the length of the reference's columbia sample (198.762 s,
Tools/PerfSummary/Summary.cs:50) = 7 windows of 30 s, processed as one lock-step batch of independent windows
(NoContext semantics, ContextImpl.cpp:476-477): pinned host PCM -> H2D -> GPU mel -> encoder -> 3-token prompt step +
51 greedy steps per window (the reference's observed 511 steps / 10 windows, columbia-medium-1080ti.txt:8-10; random
weights never emit EOT sensibly, so the step count is forced while the sampled token IS fed back). One "step" of the
bench = one pass of the hot path over ONE LOCK-STEP BATCH: --clips-per-step (64) passes over the clip = 448 windows encoded and decoded in lock step on one
context (round 5: a context takes up to 512 windows -- encoder in chunks of <= 128, decode products of > 128 rows on gemmDecRows); the K steps alternate over
--inflight (2) contexts, so two batches are in flight. The span is first H2D byte to last token id on the host. value = audio seconds / wall seconds; weak
scaling for N > 1 (every rank its own clips; the weight arena is broadcast once over RCCL before the timed region). Rounds 1-4 called one CLIP pass a step
and dealt the K passes into batches -- at the driver's K = 20 two batches of 70 windows, a job so small that the latency-bound decode chain is a third of it;
that measurement is still in every line: `small_job`.

Objects next to the contract fields:
  roofline      algorithmic bytes (or flops) per launch / average launch duration, hipEvent pairs on the launch stream (one batch of the size the timed
                region ran, one context at a time) minus the calibrated cost of an empty bracket; `traffic` = HBM bytes per launch from the committed PMC
                pass (profiles/r06_pmc.json, collected at the timed batch size of 448 windows; counters cannot be collected inside a timed run). The two classes with the most kernel time are both in
                every line -- `mfma_kernel` (the encoder's matrix-core product) and `hbm_kernel` (the decode step's cross-attention) -- and the TOP LEVEL
                repeats whichever of them sits LOWER against its roofline; `encoder_attention`; `decode_chain` = every launch of a decode step outside the
                cross-attention as one class; `end_to_end` = (sum flops / 2.5 PF + sum bytes / 8 TB/s) / measured, per batch of the size that ran
  cpu_baseline  the reference's own CPU path (oracle/_ref, kind "reference") on a bounded sample, same run
  parity        ggml-medium shape, window 0: the measured (FP32 P.V) GPU path against the reference CPU path -- cross-KV,
                logits of the prompt and of teacher-forced greedy steps, top-1 agreement; `timed_ids` = the ids the TIMED region produced for clip 0
                against the same windows decoded one at a time (contexts of one window), greedily and teacher-forced
  small_job     20 CLIP passes as two lock-step batches of 70 windows in flight: the timed region of rounds 1-4's driver invocation
  through_boundary  the SAME workload (two batches) driven by the plain C++ host code: libWhisper.so createBatchRunner / iBatchRunner::run (lock-step
                scheduler, the reference's host loop per stream) on a scripted model -- what a caller of the drop-in library gets
  single_stream the SAME clip through the drop-in boundary, sequentially: libWhisper.so iContext::runFull with prompt
                carry-over on a scripted medium-shape model (7 windows x 52 steps) -- the like-for-like figure against the
                reference's published single-clip number (`vs_baseline` lives here; `roofline_frac` = its byte / FLOP floor over
                the measured time), plus T host threads x their own iContext
  large_v2      the batched pipeline once more on the ggml-large-v2 shape (BASELINE names both models), with its own roofline / parity / cpu_baseline and
                `beam5` = BASELINE configs[2] (8 x 30 s chunks, beam_size 5, the ranking on the device)

Other workloads (BASELINE configs 3-5): --workload shard256 | beam5 | v3stream, see --help.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

T_START = time.time()
CLIP_SECONDS = 198.762
WINDOW_SAMPLES = 480000
N_PROMPT = 3
N_GREEDY = 51
# the reference's own published audio-s/s for this clip (its D3D11 backend on a GTX 1080Ti, SampleClips/summary.tsv:10, :14)
PUBLISHED_AUDIO_S_PER_S = {"medium": 13.30, "large-v2": 7.22, "large": 7.22}
MAX_LOCKSTEP_WINDOWS = 512   # rows of the decode kernels (csrc/kernels.h GEMV_MAX_ROWS): one context decodes up to this many windows in lock step
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s
MFMA_PEAK_TFLOPS = 2500.0    # dense FP16/BF16 MFMA
MFMA_CLASSES = ("gemmTiled", "attentionEnc")
PMC_JSON = next((p for p in (os.path.join(ROOT, "profiles", "r%02d_pmc.json" % r) for r in (6, 5, 4, 3)) if os.path.exists(p)), os.path.join(ROOT, "profiles", "r04_pmc.json"))
BCAST_NOTE = {}        # model kind -> what the weight broadcast of this run was (ranks, bytes, seconds, GB/s)
EMPTY_KERNEL_US = 1.9
METRIC = "audio-seconds/sec (real-time factor), ggml-medium & large, 30s chunks @1/2/4/8 GPU"


def synth_pcm(n_windows: int, seed: int) -> np.ndarray:
    """Seeded band-limited noise, uniform +-0.1 envelope."""
    rng = np.random.default_rng(seed)
    n = n_windows * WINDOW_SAMPLES
    x = rng.uniform(-1, 1, n).astype(np.float32)
    k = np.hanning(33).astype(np.float32)
    x = np.convolve(x, k / k.sum(), mode="same")
    env = 0.1 * (0.6 + 0.4 * np.sin(np.arange(n, dtype=np.float32) * (2 * np.pi / 16000 / 2.7)))
    return (x * env * 3).astype(np.float32).reshape(n_windows, WINDOW_SAMPLES)


def clip_start(group, prompt, n_greedy):
    """Enqueue one whole pass on the group's context: H2D of the PCM, mel, encoder, prompt step, greedy steps; no host sync.
    group = (ctx, pcm_host_pinned [k][480000] or None, pcm_dev [k][480000], mel_dev [k][n_mels][3000])."""
    ctx, pcm_host, pcm_dev, mel_dev = group
    k = pcm_dev.shape[0]
    if pcm_host is not None:
        ctx.upload_async(pcm_dev, pcm_host)
    # the k windows are independent buffers (NoContext chunks): one batched spectrogram call = three launches instead of 3 k
    ctx.mel_spectrogram_batch(pcm_dev, mel_dev, sync=False)
    ctx.encode(mel_dev, sync=False)
    ctx.decode_window_start(np.tile(np.asarray(prompt, np.int32), (k, 1)), n_greedy, force_first_timestamp=True, first_is_initial=True)


def clip_finish(group):
    ids, _ = group[0].decode_window_finish()
    return ids.T


def run_passes(sequence, prompt, n_greedy, max_in_flight):
    """Runs the slots of `sequence` in order with up to max_in_flight of them enqueued at once, each on its own context /
    HIP stream: the MFMA-bound encoder of one pass runs under the latency-bound decode chains of its neighbours. Every
    pass does the full work; results come back in order. Returns the token ids of the last pass."""
    pending, toks = [], None
    for g in sequence:
        while len(pending) >= max_in_flight or any(p is g for p in pending):
            toks = clip_finish(pending.pop(0))
        clip_start(g, prompt, n_greedy)
        pending.append(g)
    while pending:
        toks = clip_finish(pending.pop(0))
    return toks


def log(msg):
    sys.stderr.write("[bench %7.1fs] %s\n" % (time.time() - T_START, msg))
    sys.stderr.flush()


CPU_BASELINE_THREADS_MAX = 16      # ggml's spin-wait thread pool stops scaling (and can collapse) far below a big host's core count
CPU_BASELINE_TIMEOUT_S = 300
PARITY_STEPS = 8


def cpu_baseline_worker(model_path, model_kind, pcm_path, prompt, n_threads, parity_path, out_path, n_win=3):
    """Runs in a child process (so a slow host cannot stall the bench): the reference's own CPU path.
    1. cpu_baseline: n_win times the same 30 s window end to end, timed. 2. parity: the GPU's spectrogram of that window and
    the GPU's own greedy ids are fed to the reference; its cross-KV and logits are compared with what the GPU computed."""
    from oracle import ref
    w = ref.RefWhisper(model_path, n_threads=n_threads, log_level=0)
    pcm = np.load(pcm_path)
    t_mel = t_enc = t_prompt = t_dec = 0.0
    for _ in range(n_win):
        t0 = time.time()
        w.pcm_to_mel(pcm)
        t1 = time.time()
        w.encode(0)
        t2 = time.time()
        w.decode(prompt, 0)
        t3 = time.time()
        for i in range(N_GREEDY):
            w.decode([1000 + i], len(prompt) + i)
        t4 = time.time()
        t_mel += t1 - t0; t_enc += t2 - t1; t_prompt += t3 - t2; t_dec += t4 - t3
    total = t_mel + t_enc + t_prompt + t_dec
    res = {"cpu_baseline": {
        "value": round(30.0 * n_win / total, 4), "unit": "audio-seconds/sec", "cores": n_threads, "kind": "reference",
        "sample": "%s-shape model, %d x the same 30 s window end to end on the reference's CPU path (Whisper/source compiled into "
                  "oracle/_ref): %.1f s of CPU time = mel %.2f s + encode %.2f s + %d-token prompt %.2f s + %d greedy steps %.2f s "
                  "(%.1f ms/token); n_threads=%d of %d host cpus" % (model_kind, n_win, total, t_mel, t_enc, len(prompt), t_prompt,
                                                                    N_GREEDY * n_win, t_dec, 1e3 * t_dec / (N_GREEDY * n_win),
                                                                    n_threads, os.cpu_count() or 1)}}
    if parity_path:
        g = np.load(parity_path)
        w.set_mel(g["mel"])
        w.encode(0)
        par = {"model": "ggml-%s shape, window 0 of the bench clip" % model_kind, "reference_threads": n_threads, "gpu_path": "FP32 P.V (the timed path)"}
        for il, name in ((0, "first"), (w.n_text_layer - 1, "last")):
            k, v = w.cross_kv(il)
            par["cross_k_%s_layer_max" % name] = float(np.abs(k - g["cross_k_%s" % name]).max())
            par["cross_v_%s_layer_max" % name] = float(np.abs(v - g["cross_v_%s" % name]).max())
        ids = [int(x) for x in g["ids"]]
        worst_max = worst_mean = 0.0
        agree = 0
        spans = []
        n_past = 0
        toks = list(prompt)
        refN_logits = []
        for s in range(len(ids)):
            rl = w.decode(toks, n_past)[0][-1]
            refN_logits.append(rl.astype(np.float64))
            gl = g["logits"][s]
            d = np.abs(rl.astype(np.float64) - gl)
            worst_max, worst_mean = max(worst_max, float(d.max())), max(worst_mean, float(d.mean()))
            spans.append(float(rl.max() - rl.min()))
            agree += int(np.argmax(rl) == np.argmax(gl))
            n_past += len(toks)
            toks = [ids[s]]
        # the reference's own band at this shape: the same teacher-forced steps replayed with ONE thread on the same encoder
        # state (the encoder and the cross-K/V are thread-count invariant; the decoder's FP16 P.V partition is not)
        w.n_threads = 1
        n_past = 0
        toks = list(prompt)
        b_max = b_mean = g1_max = 0.0
        for s in range(len(ids)):
            r1 = w.decode(toks, n_past)[0][-1]
            d = np.abs(r1.astype(np.float64) - refN_logits[s])
            b_max, b_mean = max(b_max, float(d.max())), max(b_mean, float(d.mean()))
            g1_max = max(g1_max, float(np.abs(r1.astype(np.float64) - g["logits"][s]).max()))
            n_past += len(toks)
            toks = [ids[s]]
        w.n_threads = n_threads
        par.update({"ref1_vs_ref%d_max" % n_threads: b_max, "ref1_vs_ref%d_mean" % n_threads: b_mean, "gpu_vs_ref1_max": g1_max})
        if "truth" in g.files:
            par["truth_d128"] = json.loads(str(g["truth"]))
            if "medium_shape" in par["truth_d128"]:
                par["truth_medium"] = par["truth_d128"].pop("medium_shape")
        par.update({"steps": len(ids), "logits_max_abs_diff": worst_max, "logits_mean_abs_diff": worst_mean,
                    "logit_span_min": min(spans), "top1_agreement": "%d/%d" % (agree, len(ids)),
                    "note": "teacher-forced by the GPU's own greedy ids; the reference's decoder result itself moves by ~5e-2 between 1 and "
                            "8 threads on random weights (FP16 P.V accumulation, ggml.c:4689-4735): see tests/test_gpu_model.py::test_decoder_fast_path "
                            "for the exact-arithmetic yardstick"})
        res["parity"] = par
    with open(out_path, "w") as f:
        json.dump(res, f)


def gpu_parity_record(hip_model, hp, pcm_window, prompt, path):
    """Window 0 through a lone 1-window context on the measured path; what the reference is compared with."""
    import torch
    from whisper_amd import binding
    ctx = binding.HipContext(hip_model, 1)
    mel = ctx.mel_spectrogram(torch.from_numpy(pcm_window).cuda())
    ctx.encode(mel)
    rec = {"mel": mel.cpu().numpy()}
    for il, name in ((0, "first"), (hp.n_text_layer - 1, "last")):
        rec["cross_k_%s" % name] = ctx.debug_read("cross-k", il)[0]
        rec["cross_v_%s" % name] = ctx.debug_read("cross-v", il)[0]
    logits, ids = [], []
    toks = np.asarray([prompt], np.int32)
    n_past = 0
    for s in range(PARITY_STEPS + 1):
        gl, _ = ctx.decode(toks, n_past)
        logits.append(gl[0])
        n_past += toks.shape[1]
        t = ctx.sample_best(1, s == 0, s == 0)[0]["id"]
        ids.append(t)
        toks = np.asarray([[t]], np.int32)
    rec["logits"] = np.stack(logits)
    rec["ids"] = np.asarray(ids, np.int32)
    ctx.close()
    t = truth_yardstick()
    if t is not None:
        if hp.n_text_state == 1024 and hp.n_text_layer == 24:
            tm = truth_medium(hip_model)
            if tm is not None:
                t["medium_shape"] = tm
        rec["truth"] = np.asarray(json.dumps(t))
    np.savez(path, **rec)


def truth_medium(hip_model):
    """The same yardstick at the MEASURED shape: tests/golden/truth_medium.npz holds the float64 no-rounding logits of this very model (seed 1) on
    the bench's window for the prompt and three teacher-forced steps, and how far the reference's CPU path (8 threads) is from them
    (tests/golden/make_golden_truth_medium.py). Returns the HIP path's distance next to the reference's."""
    import torch
    from whisper_amd import binding
    try:
        g = np.load(os.path.join(ROOT, "tests", "golden", "truth_medium.npz"))
    except OSError:
        return None
    stats = json.loads(str(g["stats"]))
    ctx = binding.HipContext(hip_model, 1)
    ctx.encode(torch.from_numpy(g["mel"]).cuda())
    steps = [[int(t) for t in g["prompt"]]] + [[int(t)] for t in g["extra"]]
    out = {"model": "ggml-medium shape (the bench's model), window 0, prompt + 3 teacher-forced steps", "gpu_vs_truth_max": 0.0, "gpu_vs_truth_mean": 0.0,
           "ref8_vs_truth_max": max(s["ref8_vs_truth_max"] for s in stats), "ref8_vs_truth_mean": max(s["ref8_vs_truth_mean"] for s in stats), "top1_equal_exact": 0}
    n_past = 0
    for i, toks in enumerate(steps):
        gl, _ = ctx.decode(np.asarray([toks], np.int32), n_past)
        n_past += len(toks)
        d = np.abs(gl[0].astype(np.float64) - g["truth_logits%d" % i].astype(np.float64))
        out["gpu_vs_truth_max"] = max(out["gpu_vs_truth_max"], float(d.max()))
        out["gpu_vs_truth_mean"] = max(out["gpu_vs_truth_mean"], float(d.mean()))
        out["top1_equal_exact"] += int(int(np.argmax(gl[0])) == stats[i]["truth_top1"])
    out["top1_equal_exact"] = "%d/%d" % (out["top1_equal_exact"], len(steps))
    ctx.close()
    return out


def truth_yardstick():
    """north_star's 1e-3 next to the oracle's own band, against EXACT arithmetic: the d = 128 test model of the committed
    fixtures (tests/golden/ref_test_d128.npz = the reference at 1 thread, ref_e2e_d128.npz = the reference at 8 threads and the
    float64 no-rounding restatement, tests/golden/make_golden_e2e.py), 5 teacher-forced steps through the measured GPU path."""
    import torch
    from whisper_amd import binding, ggml_format as gf
    try:
        g1 = np.load(os.path.join(ROOT, "tests", "golden", "ref_test_d128.npz"))
        g8 = np.load(os.path.join(ROOT, "tests", "golden", "ref_e2e_d128.npz"))
    except OSError:
        return None
    model = gf.synth_model("test-d128", seed=1234, attn_sharpness=2.0)
    hm = binding.HipModel.from_ggml(model)
    ctx = binding.HipContext(hm, 1)
    ctx.encode(torch.from_numpy(g1["mel"]).cuda())
    out = {"model": "test-d128 (seed 1234), 5 teacher-forced steps", "gpu_vs_truth_max": 0.0, "gpu_vs_truth_mean": 0.0,
           "ref8_vs_truth_max": 0.0, "ref1_vs_truth_max": 0.0, "ref1_vs_ref8_max": 0.0, "gpu_vs_ref8_max": 0.0}
    pos = n_past = 0
    for i, ln in enumerate(g1["step_lens"]):
        ln = int(ln)
        gl, _ = ctx.decode(g1["steps"][pos:pos + ln][None, :], n_past)
        truth, r1, r8 = g8["truth_logits%d" % i].astype(np.float64), g1["logits%d" % i].astype(np.float64), g8["ref8_logits%d" % i].astype(np.float64)
        d = np.abs(gl[0] - truth)
        out["gpu_vs_truth_max"] = max(out["gpu_vs_truth_max"], float(d.max()))
        out["gpu_vs_truth_mean"] = max(out["gpu_vs_truth_mean"], float(d.mean()))
        out["ref8_vs_truth_max"] = max(out["ref8_vs_truth_max"], float(np.abs(r8 - truth).max()))
        out["ref1_vs_truth_max"] = max(out["ref1_vs_truth_max"], float(np.abs(r1 - truth).max()))
        out["ref1_vs_ref8_max"] = max(out["ref1_vs_ref8_max"], float(np.abs(r1 - r8).max()))
        out["gpu_vs_ref8_max"] = max(out["gpu_vs_ref8_max"], float(np.abs(gl[0] - r8).max()))
        pos += ln
        n_past += ln
    ctx.close()
    hm.close()
    return out


def cpu_baseline(model, model_kind, pcm_one_window, prompt, hip_model=None, want_parity=True, n_win=3, timeout_s=None):
    """The reference's own CPU path (compiled unmodified into oracle/_ref) timed on this host, bounded by a timeout."""
    null = {"value": None, "unit": "audio-seconds/sec", "cores": 0, "kind": "reference"}
    try:
        from oracle import ref
        if not ref.available():
            return dict(null, sample="oracle/_ref/libwhisper_ref.so not present"), None
    except Exception as e:      # pragma: no cover
        return dict(null, sample="oracle unavailable: %s" % e), None
    import subprocess
    import tempfile
    from whisper_amd import ggml_format as gf
    n_threads = max(1, min(os.cpu_count() or 1, CPU_BASELINE_THREADS_MAX))
    with tempfile.TemporaryDirectory() as td:
        mp, pp, op = os.path.join(td, "m.bin"), os.path.join(td, "pcm.npy"), os.path.join(td, "out.json")
        gp = os.path.join(td, "gpu.npz") if (want_parity and hip_model is not None) else ""
        gf.write_model(mp, model)
        np.save(pp, pcm_one_window)
        if gp:
            gpu_parity_record(hip_model, model.hparams, pcm_one_window, prompt, gp)
        code = ("import sys; sys.path.insert(0, %r); import bench; bench.cpu_baseline_worker(%r, %r, %r, %r, %d, %r, %r, %d)"
                % (ROOT, mp, model_kind, pp, list(map(int, prompt)), n_threads, gp, op, n_win))
        try:
            subprocess.run([sys.executable, "-c", code], timeout=timeout_s or CPU_BASELINE_TIMEOUT_S, check=True,
                           stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
            with open(op) as f:
                r = json.load(f)
            return r["cpu_baseline"], r.get("parity")
        except subprocess.TimeoutExpired:
            return dict(null, cores=n_threads, sample="reference CPU path did not finish within %d s on %d threads" % (timeout_s or CPU_BASELINE_TIMEOUT_S, n_threads)), None
        except Exception as e:
            return dict(null, cores=n_threads, sample="reference CPU run failed: %s" % str(e)[:200]), None


def timed_ids_check(hip_model, hp, pcm_dev_clip, toks_clip, prompt, margin_band=2.5e-2):
    """What the timed region produced against the same windows decoded ONE AT A TIME (a context of one window: decode1.hip's kernels, nothing in common
    with the lock-step batch's decode kernels but the arithmetic they implement). toks_clip: ids [windows][1 + N_GREEDY] of one clip of the timed pass.
    (a) every window alone through the captured greedy graph: its own ids, their checksum next to the timed pass's; (b) every window alone TEACHER-FORCED
    with the timed pass's ids: where the lone context's argmax is another token, its own margin between the two candidates -- a disagreement is a
    near-tie decided by FP32 summation order (random weights have many) when that margin is inside the band, and a defect when it is not."""
    import torch
    from whisper_amd import binding
    n_win = int(toks_clip.shape[0])
    ctx = binding.HipContext(hip_model, 1)
    own, equal_windows, first_div = [], 0, []
    disagreements, worst_margin, compared = 0, 0.0, 0
    for w in range(n_win):
        mel = ctx.mel_spectrogram(pcm_dev_clip[w])
        ctx.encode(mel)
        ctx.decode_window_start(np.asarray([prompt], np.int32), N_GREEDY, force_first_timestamp=True, first_is_initial=True)
        ids, _ = ctx.decode_window_finish()
        own.append(ids[:, 0])
        same = ids[:, 0] == toks_clip[w]
        equal_windows += int(same.all())
        first_div.append(None if same.all() else int(np.argmin(same)))
        # teacher-forced with the timed ids
        ctx.encode(mel)
        toks = np.asarray([prompt], np.int32)
        n_past = 0
        for s_ in range(N_GREEDY + 1):
            gl, _ = ctx.decode(toks, n_past, want_probs=False)
            tok = ctx.sample_best(1, s_ == 0, s_ == 0)[0]["id"]
            timed = int(toks_clip[w][s_])
            compared += 1
            if tok != timed:
                disagreements += 1
                worst_margin = max(worst_margin, abs(float(gl[0][tok]) - float(gl[0][timed])))
            n_past += toks.shape[1]
            toks = np.asarray([[timed]], np.int32)
    ctx.close()
    own = np.stack(own)
    return {"windows": n_win, "samples_compared": compared,
            "tokens_checksum_timed": int(np.asarray(toks_clip, np.int64).sum() % 1000003),
            "tokens_checksum_one_window_at_a_time": int(own.astype(np.int64).sum() % 1000003),
            "windows_with_identical_ids": "%d/%d" % (equal_windows, n_win), "first_divergence_step": first_div,
            "teacher_forced_disagreements": disagreements, "largest_margin_at_a_disagreement": round(worst_margin, 5), "margin_band": margin_band,
            "consistent": bool(disagreements == 0 or worst_margin < margin_band),
            "what": "clip 0 of the timed pass: each of its windows decoded alone (context of one window, decode1.hip kernels) -- greedily, and teacher-forced with the "
                    "timed ids; a teacher-forced disagreement is admissible when the lone context's own logit margin between the two tokens is inside the band "
                    "(a near-tie of random weights decided by FP32 summation order), a greedy divergence follows from the first such tie"}


# ----------------------------------------------------------------------------------------------------------------------
# the batched pipeline (default workload; also the large_v2 sub-object)
# ----------------------------------------------------------------------------------------------------------------------
def plan_batches(steps, C, inflight):
    """How `steps` clip passes are dealt into lock-step batches of at most C clips for `inflight` contexts: as few batches as C
    allows, rounded up to a multiple of the contexts in flight (every context then runs the same number of batches and none
    idles through the tail), sizes equal to within one clip. 20 passes, C = 16, two contexts: 10 + 10, not 16 + 4 -- a batch
    costs about the same decode chain whatever its size, so the straggler would run for as long as the big one."""
    steps, C, inflight = int(steps), max(1, int(C)), max(1, int(inflight))
    if steps <= 0:
        return []
    nb = -(-steps // C)
    if steps >= inflight and nb % inflight:
        nb += inflight - nb % inflight
    nb = min(nb, steps)
    base, extra = divmod(steps, nb)
    return [base + 1] * extra + [base] * (nb - extra)


def measure_batched(hip_model, hp, prompt, steps, warmup, B, C, inflight, rank, world, dist, want_kernels, h2d=True, plan=None, single_clip=True, warmup_batches=None):
    """`steps` CLIP passes dealt into lock-step batches (plan, or plan_batches(steps, C, inflight)). warmup = passes over every distinct context, or -- with
    warmup_batches -- that many untimed batch passes alternating over the contexts (at least one per context: the graphs are captured there)."""
    import torch
    from whisper_amd import binding
    n_frames = WINDOW_SAMPLES // 160

    def make_slot(n_clips):
        """A context for n_clips clip passes in lock step + its inputs; clip j of every slot is the same seeded clip."""
        host = torch.from_numpy(np.concatenate([synth_pcm(B, seed=100 + rank + 1000 * j) for j in range(n_clips)])).pin_memory()
        dev = host.cuda()
        mel = torch.empty((B * n_clips, hp.n_mels, n_frames), dtype=torch.float32, device="cuda")
        return (binding.HipContext(hip_model, B * n_clips), host if h2d else None, dev, mel)

    inflight = max(1, inflight)
    sizes = list(plan) if plan else plan_batches(steps, C, inflight)
    assert sum(sizes) == steps and all(0 < n <= MAX_LOCKSTEP_WINDOWS // B for n in sizes), sizes
    # contexts: for every batch size of the plan as many as are ever in flight at once (at most `inflight`); a context is
    # re-used, in order, by the later batches of its size
    pool, sequence = {}, []
    for n in sizes:
        have = pool.setdefault(n, [])
        if len(have) < min(inflight, sizes.count(n)):
            have.append(make_slot(n))
    counters = {n: 0 for n in pool}
    for n in sizes:
        sequence.append(pool[n][counters[n] % len(pool[n])])
        counters[n] += 1
    distinct = [sl for n in sorted(pool, reverse=True) for sl in pool[n]]
    # the per-kernel tables and the lone-batch latency are taken from the contexts of the plan's LARGEST batch: what ran in the timed region
    slots = distinct[:inflight]
    torch.cuda.synchronize()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if warmup_batches is not None:
        n_warm = max(int(warmup_batches), len(distinct))
        run_passes([distinct[i % len(distinct)] for i in range(n_warm)], prompt, N_GREEDY, inflight)
    else:
        for _ in range(warmup):
            run_passes(distinct, prompt, N_GREEDY, inflight)
    barrier()
    t0 = time.perf_counter()
    toks = run_passes(sequence, prompt, N_GREEDY, inflight)
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed_local = elapsed
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    out = {"elapsed": elapsed, "elapsed_local": elapsed_local, "toks": toks, "slots": slots, "last_slot": sequence[-1], "single_clip_ms": None, "kernels": {}, "lone_batch_ms": None, "plan": sizes,
           "kernel_clips": int(slots[0][2].shape[0]) // B}
    if want_kernels and rank == 0:
        grp = slots[0]
        # (a) one lone batch pass from the captured graph: latency of a batch with nothing else on the GPU
        lone = 1e9
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run_passes([grp], prompt, N_GREEDY, 1)
            lone = min(lone, 1e3 * (time.perf_counter() - t0))
        out["lone_batch_ms"] = lone
        if not single_clip:
            slots_k = [slots[0]] if len({int(sl[2].shape[0]) for sl in slots}) > 1 else slots
            return finish_kernels(out, slots_k, prompt)
        # (b) a lone SINGLE clip (7 windows): the latency a caller with one recording sees from this path
        one = make_slot(1)
        run_passes([one], prompt, N_GREEDY, 1)
        sc = 1e9
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run_passes([one], prompt, N_GREEDY, 1)
            sc = min(sc, 1e3 * (time.perf_counter() - t0))
        out["single_clip_ms"] = sc
        one[0].close()
        # (c) one pass per slot with hipEvent pairs around every launch (eager: a captured graph cannot hold them), ONE slot at a
        # time: a bracket also counts the time a launch waits for a CU, and with the persistent encoder product of the other
        # batch holding (all but 32 of) them that wait depends on how the host happens to interleave two eager passes, not on
        # the kernel (round 3, r3G: cross-attention 134.6 us in brackets with both slots eager, 116.4 us in rocprofv3's trace of
        # the timed region). Alone, the bracket is the launch; rocprofv3 of the timed region (both batches in flight, captured
        # graphs) is committed next to it and tests/test_profiles.py holds the two together.
        slots_k = [sl for sl in slots if sl[2].shape[0] == slots[0][2].shape[0]]      # equal batch sizes only: per-launch averages of ONE shape
        return finish_kernels(out, slots_k, prompt)
    return out


def finish_kernels(out, slots, prompt):
    """One eager pass per slot with hipEvent pairs around every launch, one slot at a time -> out["kernels"] (summed over the slots)."""
    for sl in slots:
        sl[0].profile(True)
    for sl in slots:
        run_passes([sl], prompt, N_GREEDY, 1)
    acc = {}
    for sl in slots:
        for k, v in sl[0].profile_read().items():
            a = acc.setdefault(k, dict(calls=0, ms=0.0, flops=0.0, bytes=0.0))
            for f in a:
                a[f] += v[f]
        sl[0].profile(False)
    out["kernels"] = acc
    out["kernel_batches"] = len(slots)
    return out


DECODE_CHAIN_CLASSES = ("selfBlockDec", "gemvFused", "layerNormDec", "attentionDec", "gemmDecode", "gemmSkinny", "embed", "softMaxSample", "vocabSoftMax")


def roofline_from(kernels, batch_ms_timed, lone_ms, n_batches=1, batch_windows=None, n_layers=None):
    """Per-launch figures of the two headline kernel classes (the encoder's matrix-core product, the decode step's HBM-bound cross-attention), the
    decode chain outside the cross-attention as ONE class, and the whole-path floor -- all from the eager pass over the batch size the timed region
    ran. The event bracket's own cost (class "eventPair": an empty kernel between the same two records) is subtracted from every launch -- a
    per-launch constant, not a proportional rescale. The TOP LEVEL repeats whichever of the two headline classes sits LOWER against its roofline."""
    pair = kernels.get("eventPair")
    # the bracket around an EMPTY kernel measures the bracket plus the empty kernel's own run time, 1.9 us in
    # rocprofv3's kernel trace (profiles/r02_kernel_stats_ab.csv, probeEmpty): only the rest is bracket
    calib_us = max(0.0, 1e3 * pair["ms"] / pair["calls"] - EMPTY_KERNEL_US) if pair and pair["calls"] else 0.0
    classes = {}
    for k, v in kernels.items():
        if k == "eventPair" or not v["calls"]:
            continue
        ms = max(v["ms"] - v["calls"] * calib_us * 1e-3, 0.05 * v["ms"])
        classes[k] = dict(v, ms_net=ms)
    total = sum(c["ms_net"] for c in classes.values())
    try:
        with open(PMC_JSON) as f:
            pmc = json.load(f)
    except (OSError, ValueError):
        pmc = {}

    def entry(name):
        """achieved / peak / frac / traffic of one kernel class: algorithmic work per launch over the average launch duration."""
        c = classes[name]
        if name in MFMA_CLASSES:
            ach = c["flops"] / (c["ms_net"] * 1e-3) / 1e12
            e = {"kernel": name, "bound": "mfma", "achieved": round(ach, 2), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / MFMA_PEAK_TFLOPS, 4)}
        else:
            ach = c["bytes"] / (c["ms_net"] * 1e-3) / 1e9
            e = {"kernel": name, "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4)}
        e["traffic"] = None
        try:
            if name in pmc.get("kernels", {}):
                k = pmc["kernels"][name]
                counted = k["hbm_read_bytes_per_launch"] + k["hbm_write_bytes_per_launch"]
                algo_pmc = max(k.get("algorithmic_bytes_per_launch", c["bytes"] / c["calls"]), 1.0)
                algo_now = c["bytes"] / c["calls"]
                e["traffic_over_algorithmic"] = round(counted / algo_pmc, 3)
                # the counter pass ran a batch of its own size: per launch of THIS run's size = the counted ratio x this run's algorithmic bytes
                same = abs(algo_now - algo_pmc) / algo_pmc < 0.02
                e["traffic"] = counted if same else int(round(counted / algo_pmc * algo_now))
                e["traffic_source"] = "committed counters, NOT measured in this run%s -- %s: %s" % (
                    "" if same else " (counted bytes / algorithmic bytes of the counter pass x the algorithmic bytes of this run's launches)",
                    os.path.relpath(PMC_JSON, ROOT), pmc.get("note", ""))
        except (ValueError, KeyError):
            pass
        e.update({"avg_launch_us": round(1e3 * c["ms_net"] / c["calls"], 2), "launches_per_batch_pass": c["calls"] // n_batches,
                  "share_of_kernel_time": round(c["ms_net"] / total, 3),
                  "algorithmic_per_launch": round((c["flops"] if e["bound"] == "mfma" else c["bytes"]) / c["calls"], 1)})
        return e

    heads = [entry(n) for n in ("gemmTiled", "attentionDecCross") if n in classes]
    if not heads:
        heads = [entry(max(classes.items(), key=lambda kv: kv[1]["ms_net"])[0])]
    r = dict(min(heads, key=lambda e: e["frac"]))
    r["which"] = ("the LOWER roofline fraction of the two kernel classes with the most time: mfma_kernel (the encoder's matrix-core product) and "
                  "hbm_kernel (the decode step's cross-attention); both follow, and decode_chain is everything of a decode step outside the cross-attention")
    if "gemmTiled" in classes:
        r["mfma_kernel"] = entry("gemmTiled")
    if "attentionDecCross" in classes:
        r["hbm_kernel"] = entry("attentionDecCross")
    if "attentionEnc" in classes:
        r["encoder_attention"] = entry("attentionEnc")
    chain = [classes[k] for k in DECODE_CHAIN_CLASSES if k in classes]
    if chain:
        ms = sum(c["ms_net"] for c in chain)
        by = sum(c["bytes"] for c in chain)
        calls = sum(c["calls"] for c in chain)
        ach = by / (ms * 1e-3) / 1e9
        r["decode_chain"] = {"kernel": "+".join(k for k in DECODE_CHAIN_CLASSES if k in classes), "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                             "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None, "ms_per_batch": round(ms / n_batches, 2),
                             "launches_per_batch_pass": calls // n_batches, "avg_launch_us": round(1e3 * ms / calls, 2), "share_of_kernel_time": round(ms / total, 3),
                             "us_per_window_step_layer": round(1e3 * ms / n_batches / ((N_GREEDY + 1) * n_layers * batch_windows), 4) if (n_layers and batch_windows) else None,
                             "what": "every launch of a decode step except the cross-attention: LayerNorm + per-head QKV + cache append + self-attention (selfBlockDec), the four "
                                     "products per layer (class gemvFused: gemvFused up to 128 sequences, gemmDecRows beyond), LayerNorm, the vocabulary product, sampler, "
                                     "embedding, and the prompt step's tiles; algorithmic bytes = weights once per launch + activations"}
    floor_ms = sum(1e3 * (c["flops"] / (MFMA_PEAK_TFLOPS * 1e12) if k in MFMA_CLASSES else c["bytes"] / (HBM_PEAK_GBS * 1e9)) for k, c in classes.items()) / n_batches
    r.update({
        "event_pair_us": round(calib_us, 2),
        "batch_windows": batch_windows,
        "timing": "hipEvent pairs around every launch (eager) on the launch stream, one batch pass per slot, one slot at a time, "
                  "minus %.2f us per launch = the same bracket around an empty kernel less that kernel's own 1.9 us" % calib_us,
        "end_to_end": {"floor_ms_per_batch": round(floor_ms, 2), "measured_ms_per_batch": round(batch_ms_timed, 2),
                       "frac": round(floor_ms / batch_ms_timed, 4), "lone_batch_ms": round(lone_ms, 2) if lone_ms else None,
                       "batch_windows": batch_windows,
                       "definition": "sum over kernel classes of algorithmic flops / 2.5 PFLOP/s (gemmTiled, attentionEnc) or "
                                     "algorithmic bytes / 8 TB/s (all others) of ONE lock-step batch of the size the timed region ran, over the time the timed region "
                                     "took per such batch (elapsed / batches x batches in flight... = elapsed x batch clips / clip passes)"}})
    table = {k: {"calls": c["calls"], "ms": round(c["ms_net"], 3), "avg_us": round(1e3 * c["ms_net"] / c["calls"], 2),
                 "tflops": round(c["flops"] / c["ms_net"] / 1e9, 2), "gbs": round(c["bytes"] / c["ms_net"] / 1e6, 1)} for k, c in classes.items()}
    return r, table


# ----------------------------------------------------------------------------------------------------------------------
# the same clip through the drop-in boundary (libWhisper.so, iContext::runFull), sequentially
# ----------------------------------------------------------------------------------------------------------------------
def single_stream(model_kind, n_threads_multi=4):
    import tempfile
    import threading
    from whisper_amd import api, ggml_format as gf
    hp = gf.hparams_for(model_kind)
    cap = 102
    positions, kept = gf.carry_over_script(hp, 7, 49, cap)
    model = gf.scripted_model_at(positions, kind=model_kind, seed=7)
    pcm = synth_pcm(7, seed=100).reshape(-1)[:int(CLIP_SECONDS * 16000)]
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "scripted.bin")
        gf.write_model(path, model)
        del model
        m = api.Model(path)
        ctx = m.create_context()
        hr = ctx.run_full(pcm, n_max_text_ctx=cap)            # warm-up: graph capture, buffers
        segs = ctx.results()
        n_tok = sum(len(s["tokens"]) for s in segs)
        best = 1e9
        for _ in range(2):
            t0 = time.perf_counter()
            ctx.run_full(pcm, n_max_text_ctx=cap)
            best = min(best, time.perf_counter() - t0)
        res = {"value": round(CLIP_SECONDS / best, 2), "unit": "audio-seconds/sec", "seconds": round(best, 4), "hr": hr,
               "windows": 7, "tokens_transcribed": n_tok, "decode_steps": 7 * (kept + 1),
               "workload": "libWhisper.so loadModel -> createContext -> runFull (COM-style iContext, batch 1, sequential windows with prompt "
                           "carry-over capped at %d tokens) on a scripted ggml-%s-shape model that transcribes 51 tokens + EOT per window "
                           "(the reference's run: 10 windows, 511 steps, SampleClips/columbia-medium-1080ti.txt)" % (cap, model_kind),
               "vs_baseline": round(CLIP_SECONDS / best / PUBLISHED_AUDIO_S_PER_S[model_kind], 2) if model_kind in PUBLISHED_AUDIO_S_PER_S else None}
        # floor of this scenario on this chip: per window the encoder's FLOP at the dense FP16 peak + per decode step every decoder
        # weight, the vocabulary matrix and the window's cross-attention keys / values once at the HBM peak
        d, L, T = hp.n_text_state, hp.n_text_layer, hp.n_audio_ctx
        step_bytes = L * 14 * d * d * 2 + hp.n_vocab * d * 2 + L * 2 * T * d * 2
        da, La = hp.n_audio_state, hp.n_audio_layer
        enc_flops = La * (2.0 * T * da * da * 12 + 4.0 * T * T * da) + 2.0 * T * da * (L * 2 * d) + 2.0 * 2 * T * 3 * hp.n_mels * da + 2.0 * T * 3 * da * da
        floor_s = 7 * enc_flops / (MFMA_PEAK_TFLOPS * 1e12) + 7 * (kept + 1) * step_bytes / (HBM_PEAK_GBS * 1e9)
        res["floor_seconds"] = round(floor_s, 5)
        res["roofline_frac"] = round(floor_s / best, 4)
        res["floor_definition"] = ("7 windows x (encoder FLOP / 2.5 PFLOP/s) + %d decode steps x %.3f GB (decoder weights + vocabulary matrix + the window's "
                                   "cross-attention K/V) / 8 TB/s; the prompt step and the spectrogram are not counted" % (7 * (kept + 1), step_bytes / 1e9))
        # T host threads, each with its own iContext on the shared model (what iModel::clone is for in the reference)
        ctxs = [m.create_context() for _ in range(n_threads_multi)]
        for c in ctxs:
            c.run_full(pcm[:WINDOW_SAMPLES * 2], n_max_text_ctx=cap)
        t0 = time.perf_counter()
        th = [threading.Thread(target=lambda c=c: c.run_full(pcm, n_max_text_ctx=cap)) for c in ctxs]
        for t in th:
            t.start()
        for t in th:
            t.join()
        dt = time.perf_counter() - t0
        res["multi_stream"] = {"streams": n_threads_multi, "value": round(n_threads_multi * CLIP_SECONDS / dt, 2), "seconds": round(dt, 4)}
        for c in ctxs + [ctx]:
            c.close()
        m.close()
    return res


# ----------------------------------------------------------------------------------------------------------------------
# the batched pipeline THROUGH the drop-in boundary: libWhisper.so createBatchRunner / iBatchRunner::run (plain C++ scheduler)
# ----------------------------------------------------------------------------------------------------------------------
def through_boundary(model_kind, steps, C, inflight, B=7):
    """The headline's workload driven by the C++ host code instead of this script: `steps` passes over the 198.762 s clip, every pass
    declared as 7 independent 30 s streams (sBatchStream::firstSample / countSamples), all of them handed to ONE iBatchRunner::run call.
    The runner keeps `inflight` lock-step groups in flight and applies the reference's host loop to every stream (stop rules on the
    sampled tokens, segments, timestamps) -- so the model is scripted to transcribe 49 text tokens between two timestamps and EOT, the
    reference's observed ~51 steps per window. Timed: the run() call, PCM in host memory to transcripts in host memory."""
    import tempfile
    from whisper_amd import api, ggml_format as gf
    hp = gf.hparams_for(model_kind)
    sp = gf.special_tokens(hp)
    script = [sp["beg"]] + [1000 + i for i in range(49)] + [sp["beg"] + 1500, sp["eot"]]
    n_prompt = 3 if hp.is_multilingual else 1
    model = gf.scripted_model(script, n_prompt, kind=model_kind, seed=7)
    n_clip = int(CLIP_SECONDS * 16000)
    clips = [np.ascontiguousarray(synth_pcm(B, seed=100 + j).reshape(-1)[:n_clip]) for j in range(min(steps, 4))]
    streams = []
    for j in range(steps):
        pcm = clips[j % len(clips)]
        for k in range(B):
            first = k * WINDOW_SAMPLES
            if first < n_clip:
                streams.append((pcm, first, min(WINDOW_SAMPLES, n_clip - first)))
    sizes = plan_batches(steps, C, inflight)
    inflight = min(inflight, len(sizes))
    slots = min(MAX_LOCKSTEP_WINDOWS, max(sizes) * B)
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "scripted.bin")
        gf.write_model(path, model)
        del model
        m = api.Model(path)
        runner = m.create_batch_runner(max_slots=slots, groups=inflight)
        runner.run(streams, flags=api.NO_CONTEXT, want_results=False)            # warm-up: contexts, graph capture
        best, out = 1e9, None
        for _ in range(2):
            t0 = time.perf_counter()
            hr, out, per = runner.run(streams, flags=api.NO_CONTEXT)
            dt = time.perf_counter() - t0
            # the conversion of the result objects into Python dicts (ctypes, ~35 k calls) is this script's, not the library's: timed apart
            t_run = runner.last_run_seconds
            best = min(best, t_run)
        n_tok = sum(len(s["tokens"]) for r in out for s in r)
        n_seg = sum(len(r) for r in out)
        ok = all(p == 0 for p in per) and all(len(r) >= 1 for r in out)
        res = {"value": round(steps * CLIP_SECONDS / best, 2), "unit": "audio-seconds/sec", "seconds": round(best, 4), "ms_per_step": round(1e3 * best / steps, 3),
               "hr": hr, "streams": len(streams), "slots_per_group": slots, "groups": inflight, "segments": n_seg, "tokens_transcribed": n_tok,
               "all_streams_ok": bool(ok), "seconds_with_python_result_conversion": round(dt, 4),
               "api": "libWhisper.so: loadModel -> createBatchRunner( { maxSlots %d, groups %d } ) -> iBatchRunner::run( %d sBatchStream = %d clip passes x %d chunks "
                      "of 30 s ) -> iTranscribeResult per stream; plain C++ scheduler (whisper_amd/host/batchScheduler.cpp): the reference's host loop per "
                      "stream, lock-step rounds, greedy chunks of 4 steps; scripted ggml-%s-shape model (51 samples per window)" % (slots, inflight, len(streams), steps, B, model_kind)}
        runner.close()
        m.close()
    return res


# ----------------------------------------------------------------------------------------------------------------------
# BASELINE configs 3 / 4: N x 30 s synthetic-mel chunks, sharded over the ranks (strong scaling), optional hypotheses
# ----------------------------------------------------------------------------------------------------------------------
def pass_floor(hp, windows, hyp, n_prompt, n_steps, measured_ms):
    """ALGORITHMIC floor of one lock-step pass of `windows` 30 s windows x `hyp` sequences each: the encoder's FLOPs at the dense FP16 MFMA
    peak plus, per decode step, the bytes a step cannot avoid at the HBM peak -- every decoder weight once, the cross-attention K/V of every window once (the
    hypotheses of a window share the pass), the self-attention rows written so far, the vocabulary matrix once. Returned as a roofline object of the pass:
    `frac` = floor / measured (1 = every byte and FLOP at its peak, nothing else)."""
    d, L, T, V = hp.n_text_state, hp.n_text_layer, hp.n_audio_ctx, hp.n_vocab
    de, Le = hp.n_audio_state, hp.n_audio_layer
    enc_flops = windows * (2.0 * 3000 * de * 3 * hp.n_mels + 2.0 * T * de * 3 * de + Le * (2.0 * T * 12 * de * de + 4.0 * T * T * de) + 2.0 * T * 2 * L * d * de)
    seqs = windows * hyp
    weights = L * 14.0 * d * d * 2 + V * d * 2.0
    cross = windows * L * 2.0 * T * d * 2
    steps = n_steps + 1                         # the prompt step reads what a single-token step reads (its extra rows are FLOPs, not bytes)
    self_rows = sum(n_prompt + i for i in range(steps))
    dec_bytes = steps * (weights + cross) + seqs * L * 2.0 * d * 2 * self_rows
    floor_ms = 1e3 * (enc_flops / 2.5e15 + dec_bytes / 8.0e12)
    return {"bound": "hbm", "achieved": round(dec_bytes / (measured_ms * 1e-3) * 1e-9, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(floor_ms / measured_ms, 4), "traffic": None,
            "floor_ms_per_pass": round(floor_ms, 3), "measured_ms_per_pass": round(measured_ms, 3), "encoder_flops_per_pass": enc_flops, "decode_bytes_per_pass": dec_bytes,
            "decode_bytes_per_step": round(dec_bytes / steps), "ms_per_decode_step_if_the_encoder_ran_at_its_peak": round((measured_ms - 1e3 * enc_flops / 2.5e15) / steps, 4),
            "what": "whole pass: `achieved` = the decode steps' algorithmic bytes / the pass's measured time (the encoder's time is inside: a lower bound on the decode "
                    "steps' rate); `frac` = (encoder FLOPs / 2.5 PF + decode bytes / 8 TB/s) / measured"}


def run_chunks(args, hip_model, hp, prompt, rank, world, dist, n_chunks, hyp, n_steps, t_bcast):
    import torch
    from whisper_amd import binding, distributed as wd
    per = args.batch
    b, e = wd.shard_range(n_chunks, rank, world)

    def synth_mel(idx):
        g = torch.Generator(device="cuda").manual_seed(1000 + idx)          # seed = chunk index
        return torch.rand((hp.n_mels, 3000), generator=g, device="cuda") * 2.0 - 1.0

    mels = torch.stack([synth_mel(i) for i in range(b, e)]) if e > b else torch.empty((0, hp.n_mels, 3000), device="cuda")
    ctxs = [binding.HipContext(hip_model, per, hypotheses=hyp) for _ in range(max(1, args.inflight))]
    # hypothesis j of a window starts from its own third prompt token, so the sequences of a window differ
    base = np.asarray(prompt, np.int32)

    def prompts(k):
        p = np.tile(base, (k * hyp, 1))
        for j in range(hyp):
            p[j::hyp, -1] = base[-1] + j
        return p

    def beam_batch(c, mel_batch):
        """BEAM SEARCH proper on one lock-step batch of k windows x hyp hypotheses (wh_beam_candidates / wh_reorder_self_cache): every step each
        live hypothesis proposes its hyp best continuations under sampleBest's rules, a window's pool is ranked by cumulative log-probability and
        its best hyp survive (their parents' self-attention cache rows move with them). Random weights never stop sensibly, so n_steps is
        forced; returns the best hypothesis' ids per window [k][n_steps + 1]."""
        k = mel_batch.shape[0]
        S = k * hyp
        c.encode(mel_batch, sync=False)
        c.decode(np.tile(base, (S, 1)), 0, want_logits=False, want_probs=False)
        cand = c.beam_candidates(S, hyp, True, True)
        ids0, p0 = cand["id"][::hyp], cand["p"][::hyp]                      # every hypothesis of a window holds the same prompt: the first speaks
        score = np.log(np.maximum(p0, 1e-30)).astype(np.float64)           # [k][hyp]
        tok = ids0.astype(np.int32)                                         # [k][hyp]
        parents = (np.arange(k)[:, None] * hyp + np.zeros((1, hyp), np.int64)).astype(np.int32)
        hist = tok[:, :, None]
        for s_ in range(n_steps):
            c.reorder_self_cache(parents.reshape(-1), len(base) + s_)
            c.decode(tok.reshape(-1, 1), len(base) + s_, want_logits=False, want_probs=False)
            cand = c.beam_candidates(S, hyp)
            pool = score[:, :, None] + np.log(np.maximum(cand["p"].reshape(k, hyp, hyp), 1e-30))
            flat = pool.reshape(k, hyp * hyp)
            order = np.argsort(-flat, axis=1, kind="stable")[:, :hyp]
            par_local, cidx = order // hyp, order % hyp
            score = np.take_along_axis(flat, order, axis=1)
            tok = np.take_along_axis(cand["id"].reshape(k, hyp * hyp), order, axis=1).astype(np.int32)
            parents = (np.arange(k)[:, None] * hyp + par_local).astype(np.int32)
            hist = np.concatenate([np.take_along_axis(hist, par_local[:, :, None], axis=1), tok[:, :, None]], axis=2)
        best = np.argmax(score, axis=1)
        return hist[np.arange(k), best]

    def beam_start(c, mel_batch):
        """The same search with the ranking ON THE DEVICE (wh_beam_window_*, round 5): encoder, prompt step, first ranking and n_steps ranked steps are
        enqueued without a host sync -- every step one replay of a captured graph (cache reorder -> decode -> softmax -> candidates -> ranking)."""
        k = mel_batch.shape[0]
        c.encode(mel_batch, sync=False)
        c.beam_window_start(np.tile(base, (k, 1)), hyp, n_steps)          # rules = None: forced steps, like the host-ranked loop above

    def beam_finish(c, k):
        st = c.beam_window_status()
        rec = c.beam_window_records(0, n_steps + 1)
        out = np.zeros((k, n_steps + 1), np.int32)
        for w in range(k):
            best = max(range(st[w]["nLive"]), key=lambda j: (st[w]["live"][j]["sum"], -j))
            out[w] = c.beam_chain(rec, w, st[w]["live"][best]["rec"])
        return out

    def transcribe(lb, le):
        outs, pending = [], []
        idx = list(range(lb - b, le - b, per))
        if hyp > 1 and getattr(args, "beam_host", False):
            # host-ranked steps: one batch at a time per context; the contexts still alternate
            for n, i0 in enumerate(idx):
                k = min(per, le - b - i0)
                best = beam_batch(ctxs[n % len(ctxs)], mels[i0:i0 + k])
                outs.append(np.concatenate([best] * hyp, axis=1))      # one row of hyp * (n_steps + 1) ints per window, like the greedy layout
            ids = np.concatenate(outs, axis=0) if outs else np.zeros((0, hyp * (n_steps + 1)), np.int32)
            return ids.reshape(le - lb, hyp * (n_steps + 1))
        if hyp > 1:
            # device-ranked: nothing comes back between the steps, so the contexts overlap like the greedy ones do
            for n, i0 in enumerate(idx):
                c = ctxs[n % len(ctxs)]
                while len(pending) >= len(ctxs):
                    pc, pk = pending.pop(0)
                    outs.append(np.concatenate([beam_finish(pc, pk)] * hyp, axis=1))
                k = min(per, le - b - i0)
                beam_start(c, mels[i0:i0 + k])
                pending.append((c, k))
            for pc, pk in pending:
                outs.append(np.concatenate([beam_finish(pc, pk)] * hyp, axis=1))
            ids = np.concatenate(outs, axis=0) if outs else np.zeros((0, hyp * (n_steps + 1)), np.int32)
            return ids.reshape(le - lb, hyp * (n_steps + 1))
        for n, i0 in enumerate(idx):
            c = ctxs[n % len(ctxs)]
            while len(pending) >= len(ctxs):
                pc, _ = pending.pop(0)
                outs.append(pc.decode_window_finish()[0].T)
            k = min(per, le - b - i0)
            c.encode(mels[i0:i0 + k], sync=False)
            c.decode_window_start(prompts(k), n_steps, force_first_timestamp=True, first_is_initial=True)
            pending.append((c, k))
        for pc, _ in pending:
            outs.append(pc.decode_window_finish()[0].T)
        ids = np.concatenate(outs, axis=0) if outs else np.zeros((0, n_steps + 1), np.int32)
        return ids.reshape(le - lb, hyp * (n_steps + 1))

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        transcribe(b, min(e, b + per * len(ctxs)))
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        toks = wd.transcribe_sharded(n_chunks, transcribe, hyp * (n_steps + 1))
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank != 0:
        return None
    value = n_chunks * 30.0 * args.steps / elapsed
    return {
        "metric": METRIC, "value": round(value, 2), "unit": "audio-seconds/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f16", "data": "synthetic",
        "config": {"workload": "ggml-%s shape (random weights), %d x 30 s synthetic mel chunks (U(-1,1), seed = chunk index) resident in HBM, contiguous "
                               "shards over %d rank(s), lock-step batches of %d windows x %d hypotheses sharing one pass over the cross-attention K/V%s, "
                               "%d contexts in flight, %d-token prompt + %d greedy steps per sequence; token ids gathered on rank 0"
                               % (args.model, n_chunks, world, per, hyp, " (beam search: every step the pool of hyp x hyp continuations of a window is ranked by cumulative "
                                  "log-probability %s, the best hyp survive, parents' self-attention cache rows move with them)" %
                                  ("on the host after every step (--beam-host)" if getattr(args, "beam_host", False) else
                                   "ON THE DEVICE by a kernel inside the captured step graph: no host round trip between the steps of a window") if hyp > 1 else "",
                                  len(ctxs), N_PROMPT, n_steps),
                   "model": "ggml-" + args.model, "chunks": n_chunks, "hypotheses": hyp, "windows_per_batch": per,
                   "parallelism": "dp%d (independent windows; RCCL weight broadcast outside the timed region: %s; no collective in the step)" % (world, BCAST_NOTE.get(args.model, "%.3f s" % t_bcast))},
        "rtf": round(elapsed / (args.steps * n_chunks * 30.0), 6),
        # one lock-step pass = `per` windows on one context; the passes of a step overlap on the contexts in flight, so the chip's time PER PASS is the step's / passes
        "roofline": pass_floor(hp, min(per, n_chunks), hyp, len(base), n_steps, 1e3 * elapsed / args.steps / max(1, (e - b + per - 1) // per)) if world == 1 else None,
        "cpu_baseline": None,
        "sequences_per_second": round(n_chunks * hyp * args.steps / elapsed, 2),
        "tokens_checksum": int(np.asarray(toks, np.int64).clip(min=0).sum() % 1000003),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4, help="steps in the timed region; a step = one lock-step batch of --clips-per-step clips (default 4 = 256 clip passes, two batches in flight)")
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--model", default=None, help="medium (default), large-v2, large-v3")
    ap.add_argument("--workload", default="clip", choices=["clip", "shard256", "beam5", "v3stream"],
                    help="clip = BASELINE configs[1] (default, the driver's line); shard256 = configs[3]: 256 x 30 s chunks sharded over the ranks "
                         "(large-v2, strong scaling); beam5 = configs[2]: 8 x 30 s chunks x 5 hypotheses per chunk (large-v2, 50 steps); "
                         "v3stream = configs[4]: the clip workload on the large-v3 shape (128 mels, vocabulary 51866), translate task")
    ap.add_argument("--windows", type=int, default=7, help="30 s windows per clip (7 = the 198.762 s columbia clip)")
    ap.add_argument("--clips-per-step", "--clips-per-batch", dest="clips_per_step", type=int, default=64, help="clips (7 windows each) in the lock-step batch that "
                    "ONE STEP encodes and decodes on one context: 64 = 448 windows (the decode kernels take up to 512 rows; the encoder runs in chunks of <= 128 "
                    "windows; 90 GB of KV caches per context). Measured at 20 steps, two contexts in flight: 32 clips 9274-9281 audio-s/s, 64 clips 9478; three "
                    "contexts of 32: 9389-9394, four: 9135. Rounds 1-4 called one CLIP pass a step: that figure is the line's `small_job`")
    ap.add_argument("--inflight", type=int, default=2, help="contexts the steps alternate over = batches in flight, each on its own HIP stream")
    ap.add_argument("--no-small-job", action="store_true", help="skip the small_job sub-object (20 clip passes as two batches of 70 windows: rounds 1-4's driver line)")
    ap.add_argument("--no-ids-check", action="store_true", help="skip parity.timed_ids (the timed pass's ids against the same windows one at a time)")
    ap.add_argument("--plan", default=None, help="explicit batch sizes (clips) of the timed region for experiments, e.g. 16,4 (default: --steps batches of --clips-per-step clips)")
    ap.add_argument("--batch", type=int, default=16, help="shard256 / beam5: windows per lock-step batch")
    ap.add_argument("--beam-host", action="store_true", help="beam5: rank every step's candidates on the host (round 4's data path) instead of on the device")
    ap.add_argument("--no-beam", action="store_true", help="skip the beam5 sub-object of the default line (BASELINE configs[2] on the large-v2 model)")
    ap.add_argument("--no-workloads", action="store_true", help="skip the shard256 (large_v2.shard256) and v3stream sub-objects of the default line (BASELINE configs[3] and [4] on one rank)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-single-stream", action="store_true")
    ap.add_argument("--no-large", action="store_true", help="skip the large_v2 sub-object of the default line")
    ap.add_argument("--no-boundary", action="store_true", help="skip the through_boundary sub-object (the workload through libWhisper.so's batch runner)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo lets two ranks share one GPU in a dry run)")
    ap.add_argument("--device", type=int, default=-1, help="HIP device for this rank (default LOCAL_RANK)")
    args = ap.parse_args()
    if args.model is None:
        args.model = {"clip": "medium", "shard256": "large-v2", "beam5": "large-v2", "v3stream": "large-v3"}[args.workload]

    import torch
    import torch.distributed as dist
    from whisper_amd import binding, ggml_format as gf

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.device >= 0:
        local = args.device
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    torch.cuda.set_device(local)
    binding.check(binding.lib().wh_device_set(local))
    if world > 1:
        # a rank that never arrives must end the job with an error, not hang the node: collectives time out after 10 minutes
        import datetime
        limit = datetime.timedelta(minutes=10)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local), timeout=limit)
        else:
            dist.init_process_group(args.backend, timeout=limit)
        if dist.get_world_size() != world or dist.get_rank() != rank:
            raise SystemExit("bench.py: the process group (%d of %d) does not match RANK / WORLD_SIZE (%d of %d)" %
                             (dist.get_rank(), dist.get_world_size(), rank, world))
        log("rank %d of %d on device %d (%s backend)" % (rank, world, local, args.backend))

    def load(kind):
        """Rank 0 builds the model and fills its arena; the others receive one RCCL broadcast over xGMI."""
        hp = gf.hparams_for(kind)
        arena = torch.empty(binding.arena_bytes(hp), dtype=torch.uint8, device="cuda")
        model, hm = None, None
        t0 = time.time()
        if rank == 0:
            log("building %s-shape random model ..." % kind)
            model = gf.synth_model(kind, seed=1)
            log("uploading weights ...")
            hm = binding.HipModel.from_ggml(model, arena_ptr=arena.data_ptr(), keepalive=arena)
        t_load = time.time() - t0
        t_bcast = 0.0
        if world > 1:
            torch.cuda.synchronize()
            dist.barrier()
            t0 = time.time()
            dist.broadcast(arena, src=0)
            torch.cuda.synchronize()
            t_bcast = time.time() - t0
            if rank != 0:
                hm = binding.HipModel(hp, arena_ptr=arena.data_ptr(), already_filled=True, keepalive=arena)
        if world > 1 and rank == 0:
            log("RCCL saw %d ranks; arena broadcast %.2f GB in %.3f s = %.1f GB/s" % (dist.get_world_size(), arena.numel() / 1e9, t_bcast,
                                                                                    arena.numel() / 1e9 / max(t_bcast, 1e-9)))
        BCAST_NOTE[kind] = "%d ranks, %.2f GB in %.3f s = %.1f GB/s" % (world, arena.numel() / 1e9, t_bcast, arena.numel() / 1e9 / max(t_bcast, 1e-9)) if world > 1 else "1 rank, none"
        return hp, model, hm, t_load, t_bcast

    hp, model, hip_model, t_load, t_bcast = load(args.model)
    sp = gf.special_tokens(hp)
    task = sp["translate"] if args.workload == "v3stream" else sp["transcribe"]
    prompt = [sp["sot"], sp["sot"] + 1, task] if hp.is_multilingual else [sp["sot"], sp["not_"], sp["beg"]]
    prompt = prompt[:N_PROMPT]

    if args.workload in ("shard256", "beam5"):
        n_chunks, hyp, n_steps = (256, 1, N_GREEDY) if args.workload == "shard256" else (8, 5, 50)
        if args.workload == "beam5":
            args.batch = min(args.batch, 8)
        line = run_chunks(args, hip_model, hp, prompt, rank, world, dist, n_chunks, hyp, n_steps, t_bcast)
        if rank == 0:
            print(json.dumps(line), flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    B = args.windows
    C = max(1, args.clips_per_step)
    if B * C > MAX_LOCKSTEP_WINDOWS:
        raise SystemExit("windows x clips-per-step must not exceed %d (rows of the decode kernels)" % MAX_LOCKSTEP_WINDOWS)
    # A STEP = one pass of the hot path over ONE LOCK-STEP BATCH: C clips (default 64 = 448 windows of 30 s) encoded and decoded in lock step on one
    # context; the K steps of the timed region alternate over `inflight` (2) contexts, so two batches are in flight at any time. (Rounds 1-4 called one
    # CLIP pass a step and dealt the K passes into batches; at the driver's K = 20 that made two batches of 70 windows -- a job so small that the
    # latency-bound decode chain is a third of it. That figure is still in the line: `small_job`.) Measured (profiles/r05_ab_variants.txt section 1):
    # two contexts in flight beat one of twice the size (8821 vs 8400 audio-s/s); 448 windows per context: +2.2 % over 224 (9478 vs 9274-9281, section 14).
    inflight = max(1, args.inflight)
    audio_seconds = CLIP_SECONDS * B / 7.0
    plan = [int(x) for x in args.plan.split(",")] if args.plan else [C] * args.steps
    passes = sum(plan)                      # clip passes in the timed region
    if rank == 0:
        log("warmup + timed region: %d steps of %d clips ..." % (args.steps, C))
    m = measure_batched(hip_model, hp, prompt, passes, args.warmup, B, C, inflight, rank, world, dist,
                        want_kernels=not args.no_roofline, plan=plan, warmup_batches=args.warmup)
    elapsed, toks, batch_plan = m["elapsed"], m["toks"], m.get("plan")
    inflight = min(inflight, len(batch_plan))
    if rank == 0:
        log("timed region done: %.3f s (batches of %s clips, %d in flight)" % (elapsed, batch_plan, inflight))
    per_rank = None
    if world > 1:
        # every rank's own figure next to the job's (value = all ranks' audio / the slowest rank's time)
        # (a sum over one-hot vectors: all_reduce is the one collective both RCCL and the dry run's gloo take on device tensors)
        slots_t = torch.zeros(world, dtype=torch.float64, device="cuda")
        slots_t[rank] = m.get("elapsed_local", elapsed)
        dist.all_reduce(slots_t, op=dist.ReduceOp.SUM)
        per_rank = [round(audio_seconds * passes / float(t), 2) for t in slots_t.cpu().tolist()]

    roofline, kernels = None, {}
    if rank == 0 and m["kernels"]:
        kc = m["kernel_clips"]
        roofline, kernels = roofline_from(m["kernels"], 1e3 * elapsed * kc / passes, m["lone_batch_ms"], m.get("kernel_batches", 1),
                                          batch_windows=kc * B, n_layers=hp.n_text_layer)
        roofline["single_clip"] = {"ms": round(m["single_clip_ms"], 2), "audio_seconds_per_sec": round(audio_seconds / (m["single_clip_ms"] * 1e-3), 1),
                                   "what": "ONE %.0f s clip (7 windows as one lock-step batch) alone on the GPU, H2D to token ids" % audio_seconds}

    cpu = parity = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        log("cpu baseline + parity (reference CPU path, bounded) ...")
        cpu, parity = cpu_baseline(model, args.model, m["slots"][0][2][0].cpu().numpy(), prompt, hip_model)
        log("cpu baseline done: %s" % cpu.get("value"))
    if rank == 0 and world == 1 and not args.no_ids_check:
        # the ids the TIMED region produced (last batch, clip 0) against the same windows one at a time
        try:
            last_slot = m["last_slot"]
            chk = timed_ids_check(hip_model, hp, last_slot[2][:B], np.asarray(toks)[:B], prompt)
            parity = dict(parity or {}, timed_ids=chk)
            log("timed ids vs one window at a time: %s identical, consistent = %s" % (chk["windows_with_identical_ids"], chk["consistent"]))
        except Exception as e:       # the sub-object must not take the line down
            parity = dict(parity or {}, timed_ids={"error": str(e)[:300]})
    for s in m["slots"]:
        s[0].close()
    del m

    small = None
    if rank == 0 and world == 1 and args.workload == "clip" and not args.no_small_job:
        # what rounds 1-4's driver line measured: 20 CLIP passes = two lock-step batches of 70 windows in flight (a step was one clip pass)
        try:
            ms_ = measure_batched(hip_model, hp, prompt, 20, 1, B, 16, 2, 0, 1, dist, want_kernels=False)
            small = {"value": round(audio_seconds * 20 / ms_["elapsed"], 2), "unit": "audio-seconds/sec", "clip_passes": 20, "batch_plan": ms_["plan"],
                     "ms_per_clip_pass": round(1e3 * ms_["elapsed"] / 20, 3),
                     "what": "the timed region of rounds 1-4's driver invocation (--steps 20 when a step was ONE clip pass): two lock-step batches of 70 windows -- "
                             "a job of 140 windows, where the latency-bound decode chain (the same ~60 us per layer and step whatever the batch) is a third of the time; "
                             "BENCH_r04.json: 7582"}
            for s_ in ms_["slots"]:
                s_[0].close()
            del ms_
            log("small job (20 clip passes): %s audio-s/s" % small["value"])
        except Exception as e:
            small = {"error": str(e)[:300]}

    single = large = boundary = v3stream = None
    if rank == 0 and world == 1 and args.workload == "clip" and not args.no_boundary and args.model in ("medium", "large-v2"):
        log("the same workload through libWhisper.so (createBatchRunner) ...")
        try:
            Cb = C                  # the batch runner's groups at the headline's batch size (64 clips = 448 slots each; round 6: 10311 against 9801 at 224 slots on one box)
            boundary = through_boundary(args.model, min(passes, 2 * Cb), Cb, inflight, B)
            log("through the boundary: %s audio-s/s" % boundary["value"])
        except Exception as e:
            boundary = {"error": str(e)[:300]}
    if rank == 0 and world == 1 and args.workload == "clip":
        if not args.no_single_stream and args.model in ("medium", "large-v2"):
            log("single stream through libWhisper.so ...")
            try:
                single = single_stream(args.model)
                log("single stream: %s audio-s/s" % single["value"])
            except Exception as e:       # the sub-object must not take the line down
                single = {"error": str(e)[:300]}
        if not args.no_large and args.model == "medium":
            log("large-v2 shape ...")
            try:
                hip_model.close()
                hp2, model2, hm2, _, _ = load("large-v2")
                sp2 = gf.special_tokens(hp2)
                p2 = [sp2["sot"], sp2["sot"] + 1, sp2["transcribe"]]
                C2 = min(C, 32)         # large-v2: 224 windows per context (71 GB of KV caches each; 448 would be 143 GB x 2)
                n2 = min(passes, 2 * C2)
                m2 = measure_batched(hm2, hp2, p2, n2, 1, B, C2, inflight, 0, 1, dist, want_kernels=not args.no_roofline, single_clip=False)
                large = {"model": "ggml-large-v2", "value": round(audio_seconds * n2 / m2["elapsed"], 2), "unit": "audio-seconds/sec", "steps": len(m2["plan"]),
                         "clip_passes": n2, "ms_per_step": round(1e3 * m2["elapsed"] / len(m2["plan"]), 3), "same_pipeline": True, "batch_plan": m2["plan"],
                         "vs_published_single_clip": "the reference publishes 7.22 audio-s/s for ONE sequential clip on a GTX 1080Ti (BASELINE.md section 1)"}
                if m2["kernels"]:
                    kc2 = m2["kernel_clips"]
                    large["roofline"], large["kernels"] = roofline_from(m2["kernels"], 1e3 * m2["elapsed"] * kc2 / n2, m2["lone_batch_ms"], m2.get("kernel_batches", 1),
                                                                        batch_windows=kc2 * B, n_layers=hp2.n_text_layer)
                    # the counter file holds the medium shape: no traffic figure for this one
                    for e in (large["roofline"], large["roofline"].get("mfma_kernel", {}), large["roofline"].get("hbm_kernel", {}), large["roofline"].get("encoder_attention", {})):
                        for k_ in ("traffic_source", "traffic_over_algorithmic"):
                            e.pop(k_, None)
                        if "traffic" in e:
                            e["traffic"] = None
                pcm2 = m2["slots"][0][2][0].cpu().numpy()
                for s in m2["slots"]:
                    s[0].close()
                log("large-v2: %s audio-s/s" % large["value"])
                if not args.no_beam:
                    # BASELINE configs[2]: large-v2, 8 x 30 s synthetic mel chunks, beam_size = 5 -- ranking on the device
                    try:
                        ba = argparse.Namespace(batch=8, inflight=2, warmup=2, steps=8, model="large-v2", beam_host=False)
                        bl = run_chunks(ba, hm2, hp2, p2, 0, 1, dist, 8, 5, 50, 0.0)
                        large["beam5"] = {"value": bl["value"], "unit": "audio-seconds/sec", "sequences_per_second": bl["sequences_per_second"], "ms_per_step": bl["ms_per_step"],
                                          "steps": bl["steps"], "tokens_checksum": bl["tokens_checksum"], "workload": bl["config"]["workload"], "roofline": bl["roofline"]}
                        log("large-v2 beam5: %s audio-s/s" % bl["value"])
                    except Exception as e:
                        large["beam5"] = {"error": str(e)[:300]}
                if not args.no_workloads:
                    # BASELINE configs[3] on ONE rank: large-v2, 256 x 30 s chunks in lock-step batches of 128 windows, two contexts in flight
                    try:
                        sa = argparse.Namespace(batch=128, inflight=2, warmup=1, steps=2, model="large-v2", beam_host=False)
                        sl = run_chunks(sa, hm2, hp2, p2, 0, 1, dist, 256, 1, N_GREEDY, 0.0)
                        large["shard256"] = {"value": sl["value"], "unit": "audio-seconds/sec", "ms_per_step": sl["ms_per_step"], "steps": sl["steps"], "n_gpus": 1,
                                             "tokens_checksum": sl["tokens_checksum"], "workload": sl["config"]["workload"], "roofline": sl["roofline"]}
                        log("large-v2 shard256 on one rank: %s audio-s/s" % sl["value"])
                    except Exception as e:
                        large["shard256"] = {"error": str(e)[:300]}
                if not args.no_cpu_baseline:
                    # BASELINE names both models: the reference's CPU path beside the large-v2 figure too (one window: its encoder alone
                    # is ~15-20 s on 16 threads), and the same parity object as the headline's, at d = 1280 / 20 heads / 32 layers
                    log("large-v2: cpu baseline + parity (reference CPU path, one window) ...")
                    cpu2, par2 = cpu_baseline(model2, "large-v2", pcm2, p2, hm2, n_win=1, timeout_s=420)
                    large["cpu_baseline"], large["parity"] = cpu2, par2
                    log("large-v2 cpu baseline done: %s" % cpu2.get("value"))
            except Exception as e:
                large = {"error": str(e)[:300]}
        if not args.no_workloads and args.model == "medium":
            # BASELINE configs[4] on ONE rank: the clip workload on the large-v3 shape (128 mel bins, 51866 tokens), translate task
            log("large-v3 shape (configs[4]) ...")
            try:
                try:
                    hm2.close()
                except Exception:
                    pass
                hp3, model3, hm3, _, _ = load("large-v3")
                del model3
                sp3 = gf.special_tokens(hp3)
                p3 = [sp3["sot"], sp3["sot"] + 1, sp3["translate"]]
                C3 = min(C, 16)
                n3 = 2 * C3
                m3 = measure_batched(hm3, hp3, p3, n3, 1, B, C3, inflight, 0, 1, dist, want_kernels=False, single_clip=False)
                ms3 = 1e3 * m3["elapsed"] / len(m3["plan"])
                v3stream = {"model": "ggml-large-v3", "task": "translate", "value": round(audio_seconds * n3 / m3["elapsed"], 2), "unit": "audio-seconds/sec", "steps": len(m3["plan"]),
                            "clip_passes": n3, "ms_per_step": round(ms3, 3), "batch_plan": m3["plan"], "n_gpus": 1,
                            "roofline": pass_floor(hp3, C3 * B, 1, N_PROMPT, N_GREEDY, ms3)}
                for s_ in m3["slots"]:
                    s_[0].close()
                hm3.close()
                log("large-v3: %s audio-s/s" % v3stream["value"])
            except Exception as e:
                v3stream = {"error": str(e)[:300]}

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = world * audio_seconds * passes / elapsed
        line = {
            "metric": METRIC,
            "value": round(value, 2), "unit": "audio-seconds/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak",
            # rounds 1-4 called ONE CLIP PASS (7 windows) a step; since round 5 a step is one lock-step batch of `clips_per_step` clips. The figure of the job the driver's
            # invocation measured then (20 clip passes as two batches of 70 windows) is kept at the top level so that round-over-round readings never mix the two
            "value_r04_definition": (small or {}).get("value") if isinstance(small, dict) else None,
            "step_definition": "one lock-step batch of %d clips = %d windows (rounds 1-4: one clip pass = %d windows; that job's figure is value_r04_definition = small_job.value)" % (C, C * B, B),
            # the published number (13.30 audio-s/s, one clip, sequential, GTX 1080Ti) is not this batched workload:
            # the like-for-like ratio is single_stream.vs_baseline
            "vs_baseline": None,
            "dtype": "f16", "data": "synthetic",
            "config": {"workload": "ggml-%s shape (random weights), %.3f s clip = %d x 30 s independent windows; ONE STEP = one lock-step batch of %d clips = %d windows "
                                   "(%.1f s of audio) on one context: pinned host PCM -> H2D -> GPU mel + encoder + %d-token prompt + %d greedy steps per window, device-side "
                                   "sampling (captured hipGraph per token); the %d steps of the timed region (%d clip passes, plan %s) alternate over %d contexts on separate HIP "
                                   "streams, so %d batches are in flight; span = first H2D byte to last token id on the host"
                                   % (args.model, audio_seconds, B, C, C * B, audio_seconds * C, N_PROMPT, N_GREEDY, args.steps, passes,
                                      batch_plan if len(batch_plan) <= 8 else "%s x %d" % ([batch_plan[0]], len(batch_plan)), inflight, inflight),
                       "model": "ggml-" + args.model, "task": "translate" if args.workload == "v3stream" else "transcribe",
                       "baseline": "BASELINE.md section 1 publishes one sequential clip on a GTX 1080Ti (13.30 audio-s/s medium): compared in single_stream, not here",
                       "windows_per_clip": B, "clips_per_step": C, "clip_passes": passes, "audio_seconds_per_step": round(audio_seconds * passes / args.steps, 3),
                       "clips_per_batch": C, "batch_plan": batch_plan, "batches_in_flight": inflight, "lockstep_windows": max(batch_plan) * B, "decode_steps_per_window": N_GREEDY + 1,
                       "ranks_seen": (dist.get_world_size() if world > 1 else 1), "per_rank_audio_seconds_per_sec": per_rank,
                       "parallelism": "dp%d (independent windows, RCCL weight broadcast outside the timed region: %s)" % (world, BCAST_NOTE.get(args.model, "%.3f s" % t_bcast))},
            "rtf": round(elapsed / (passes * audio_seconds), 6),
            "roofline": roofline,
            "cpu_baseline": cpu,
            "parity": parity,
            "through_boundary": boundary,
            "small_job": small,
            "single_stream": single,
            "large_v2": large,
            "v3stream": v3stream,
            "kernels": kernels,
            "model_build_s": round(t_load, 1),
            "tokens_checksum": int(np.asarray(toks, np.int64)[:B].sum() % 1000003),
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
