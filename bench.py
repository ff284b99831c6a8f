#!/usr/bin/env python3
"""bench.py -- audio-seconds/sec of the Whisper hot path (mel -> encoder -> KV-cached greedy decoder -> token ids).

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched under
torch.distributed.run with one rank per GPU. Prints ONE JSON line on rank 0.

Workload at N = 1 = BASELINE.json configs[1]: a ggml-medium-shaped model (random FP16 weights of the exact real
shapes -- no real weights exist offline) on a clip of the length of the reference's columbia sample (198.762 s,
Tools/PerfSummary/Summary.cs:50) = 7 windows of 30 s, processed as one lock-step batch of independent windows
(NoContext semantics, ContextImpl.cpp:476-477): PCM resident in HBM -> GPU mel -> encoder -> 3-token prompt step +
51 greedy steps per window (the reference's observed 511 steps / 10 windows, columbia-medium-1080ti.txt:8-10; random
weights never emit EOT sensibly, so the step count is forced while the sampled token IS fed back). One "step" of the
bench = one pass over the whole clip. value = audio seconds / wall seconds; weak scaling for N > 1 (every rank
transcribes its own clip; the weight arena is broadcast once over RCCL before the timed region).

Extra objects: `roofline` for the dominant kernel class (per-launch durations from hipEvent pairs on the launch stream,
collected in a separate, identical pass because two event records per launch would perturb the launch-bound decode
loop), and `cpu_baseline` = the reference's own CPU path (oracle/_ref, kind "reference") on a bounded sample.
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
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s
MFMA_PEAK_TFLOPS = 2500.0    # dense FP16/BF16 MFMA


def synth_pcm(n_windows: int, seed: int) -> np.ndarray:
    """Seeded band-limited noise, uniform +-0.1 envelope."""
    rng = np.random.default_rng(seed)
    n = n_windows * WINDOW_SAMPLES
    x = rng.uniform(-1, 1, n).astype(np.float32)
    k = np.hanning(33).astype(np.float32)
    x = np.convolve(x, k / k.sum(), mode="same")
    env = 0.1 * (0.6 + 0.4 * np.sin(np.arange(n, dtype=np.float32) * (2 * np.pi / 16000 / 2.7)))
    return (x * env * 3).astype(np.float32).reshape(n_windows, WINDOW_SAMPLES)


def transcribe_clip(groups, prompt, n_greedy):
    """One pass of the hot path over the clip. `groups` = [(ctx, pcm_dev [k][480000], mel_dev [k][80][3000])]: every group is a
    lock-step batch of windows on its own context (= its own HIP stream and captured decode graph). Everything is enqueued
    without a host sync -- GPU mel, encoder, prompt step, first sample, n_greedy device-side greedy steps -- group after
    group, so the latency-bound decode steps of one group overlap the encoder GEMMs and decode steps of the others; then
    the sampled token ids are collected. Returns [n_windows][n_greedy + 1] token ids."""
    for ctx, pcm_dev, mel_dev in groups:
        k = pcm_dev.shape[0]
        for b in range(k):
            ctx.mel_spectrogram(pcm_dev[b], mel_dev[b], sync=False)
        ctx.encode(mel_dev, sync=False)
        ctx.decode_window_start(np.tile(np.asarray(prompt, np.int32), (k, 1)), n_greedy, force_first_timestamp=True, first_is_initial=True)
    outs = []
    for ctx, _, _ in groups:
        ids, _ = ctx.decode_window_finish()
        outs.append(ids.T)
    return np.concatenate(outs, axis=0)


def clip_start(group, prompt, n_greedy):
    """Enqueue one whole clip pass on the group's context (mel, encoder, prompt step, greedy steps); no host sync."""
    ctx, pcm_dev, mel_dev = group
    k = pcm_dev.shape[0]
    for b in range(k):
        ctx.mel_spectrogram(pcm_dev[b], mel_dev[b], sync=False)
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
CPU_BASELINE_TIMEOUT_S = 240


def cpu_baseline_worker(model_path, model_kind, pcm_path, prompt, n_threads, out_path):
    """Runs in a child process (so a slow host cannot stall the bench): the reference's own CPU path, one 30 s window."""
    from oracle import ref
    w = ref.RefWhisper(model_path, n_threads=n_threads, log_level=0)
    pcm = np.load(pcm_path)
    n_win = 3
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
    res = {"value": round(30.0 * n_win / total, 4), "unit": "audio-seconds/sec", "cores": n_threads, "kind": "reference",
           "sample": "%s-shape model, %d x the same 30 s window end to end on the reference's CPU path (Whisper/source compiled into "
                     "oracle/_ref): %.1f s of CPU time = mel %.2f s + encode %.2f s + %d-token prompt %.2f s + %d greedy steps %.2f s "
                     "(%.1f ms/token); n_threads=%d of %d host cpus" % (model_kind, n_win, total, t_mel, t_enc, len(prompt), t_prompt,
                                                                       N_GREEDY * n_win, t_dec, 1e3 * t_dec / (N_GREEDY * n_win),
                                                                       n_threads, os.cpu_count() or 1)}
    with open(out_path, "w") as f:
        json.dump(res, f)


def cpu_baseline(model, model_kind, pcm_one_window, prompt):
    """The reference's own CPU path (compiled unmodified into oracle/_ref) timed on this host, bounded by a timeout."""
    null = {"value": None, "unit": "audio-seconds/sec", "cores": 0, "kind": "reference"}
    try:
        from oracle import ref
        if not ref.available():
            return dict(null, sample="oracle/_ref/libwhisper_ref.so not present")
    except Exception as e:      # pragma: no cover
        return dict(null, sample="oracle unavailable: %s" % e)
    import subprocess
    import tempfile
    from whisper_amd import ggml_format as gf
    n_threads = max(1, min(os.cpu_count() or 1, CPU_BASELINE_THREADS_MAX))
    with tempfile.TemporaryDirectory() as td:
        mp, pp, op = os.path.join(td, "m.bin"), os.path.join(td, "pcm.npy"), os.path.join(td, "out.json")
        gf.write_model(mp, model)
        np.save(pp, pcm_one_window)
        code = ("import sys; sys.path.insert(0, %r); import bench; bench.cpu_baseline_worker(%r, %r, %r, %r, %d, %r)"
                % (ROOT, mp, model_kind, pp, list(map(int, prompt)), n_threads, op))
        try:
            subprocess.run([sys.executable, "-c", code], timeout=CPU_BASELINE_TIMEOUT_S, check=True,
                           stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
            with open(op) as f:
                return json.load(f)
        except subprocess.TimeoutExpired:
            return dict(null, cores=n_threads, sample="reference CPU path did not finish one 30 s window of the %s-shape model within "
                                                      "%d s on %d threads" % (model_kind, CPU_BASELINE_TIMEOUT_S, n_threads))
        except Exception as e:
            return dict(null, cores=n_threads, sample="reference CPU run failed: %s" % str(e)[:200])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--model", default="medium")
    ap.add_argument("--windows", type=int, default=7, help="30 s windows per clip (7 = the 198.762 s columbia clip)")
    ap.add_argument("--clips-per-batch", type=int, default=4, help="clip passes decoded as ONE lock-step batch (7 windows each, at most 4: the decode "
                    "gemv holds 32 rows); a step stays one clip pass, K steps run as K // C batches plus one batch with the remainder")
    ap.add_argument("--inflight", type=int, default=3, help="clip passes in flight, each on its own context and HIP stream: the decode chain of one "
                    "pass is latency-bound, so the encoder GEMMs and the decode chains of its neighbours run underneath it "
                    "(measured on MI355X, one clip per batch: 107 / 74 / 66 / 65 ms per pass with 1 / 2 / 3 / 6 in flight; four clips "
                    "per batch: 62.6 / 48.8 / 45.6 with 1 / 2 / 3)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo lets two ranks share one GPU in a dry run)")
    ap.add_argument("--device", type=int, default=-1, help="HIP device for this rank (default LOCAL_RANK)")
    ap.add_argument("--no-roofline", action="store_true")
    args = ap.parse_args()

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
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(args.backend)

    hp = gf.hparams_for(args.model)
    sp = gf.special_tokens(hp)
    prompt = [sp["sot"], sp["sot"] + 1, sp["transcribe"]] if hp.is_multilingual else [sp["sot"], sp["not_"], sp["beg"]]
    prompt = prompt[:N_PROMPT]

    # ---- weights: rank 0 builds the model and fills its arena; the others receive one RCCL broadcast over xGMI ----
    arena = torch.empty(binding.arena_bytes(hp), dtype=torch.uint8, device="cuda")
    model = None
    t0 = time.time()
    if rank == 0:
        log("building %s-shape random model ..." % args.model)
        model = gf.synth_model(args.model, seed=1)
        log("uploading weights ...")
        hip_model = binding.HipModel.from_ggml(model, arena_ptr=arena.data_ptr(), keepalive=arena)
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
            hip_model = binding.HipModel(hp, arena_ptr=arena.data_ptr(), already_filled=True, keepalive=arena)

    B = args.windows
    C = max(1, args.clips_per_batch)
    if B * C > 32:
        raise SystemExit("windows x clips-per-batch must not exceed 32 (rows of the decode gemv)")
    n_frames = WINDOW_SAMPLES // 160

    def make_slot(n_clips):
        """A context for n_clips clip passes in lock step + its inputs; clip j of every slot is the same seeded clip."""
        pcm = torch.from_numpy(np.concatenate([synth_pcm(B, seed=100 + rank + 1000 * j) for j in range(n_clips)])).cuda()
        mel = torch.empty((B * n_clips, hp.n_mels, n_frames), dtype=torch.float32, device="cuda")
        return (binding.HipContext(hip_model, B * n_clips), pcm, mel)

    slots = [make_slot(C) for _ in range(max(1, args.inflight))]
    n_full, rem = divmod(args.steps, C)
    rem_slot = make_slot(rem) if rem else None
    sequence = [slots[i % len(slots)] for i in range(n_full)] + ([rem_slot] if rem_slot else [])
    groups = [slots[0]]
    torch.cuda.synchronize()
    audio_seconds = CLIP_SECONDS * B / 7.0

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if rank == 0:
        log("warmup ...")
    for _ in range(args.warmup):
        run_passes(slots + ([rem_slot] if rem_slot else []), prompt, N_GREEDY, len(slots))
    barrier()
    if rank == 0:
        log("timed region: %d steps ..." % args.steps)
    t0 = time.perf_counter()
    toks = run_passes(sequence, prompt, N_GREEDY, len(slots))
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- per-kernel pass (rank 0): identical work with hipEvent pairs around every launch ----
    roofline, kernels = None, {}
    if rank == 0:
        log("timed region done: %.3f s" % elapsed)
    if rank == 0 and not args.no_roofline:
        # one group at a time (events on concurrent streams would time each other's kernels), summed over the groups
        kernels = {}
        for grp in groups:
            grp[0].profile(True)
            transcribe_clip([grp], prompt, N_GREEDY)
            for k, v in grp[0].profile_read().items():
                acc = kernels.setdefault(k, dict(calls=0, ms=0.0, flops=0.0, bytes=0.0))
                for f in acc:
                    acc[f] += v[f]
            grp[0].profile(False)
        total_ms = sum(k["ms"] for k in kernels.values())
        # The event pairs need eager launches, which cost ~0.6 us more per dispatch than the captured graph the timed region
        # replays. One lone pass from the graph is timed as a whole (wall clock between syncs) and the per-kernel times are
        # rescaled so that they sum to it: that is the per-launch duration inside the graph, the one rocprofv3 reports.
        graph_ms = 1e9
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            transcribe_clip(groups, prompt, N_GREEDY)
            graph_ms = min(graph_ms, 1e3 * (time.perf_counter() - t0))
        scale = min(1.0, graph_ms / total_ms)
        name, dom = max(kernels.items(), key=lambda kv: kv[1]["ms"])
        dom_ms = dom["ms"] * scale
        avg_us = 1e3 * dom_ms / dom["calls"]
        if name in ("gemmTiled", "attentionEnc"):
            ach = dom["flops"] / (dom_ms * 1e-3) / 1e12
            roofline = {"kernel": name, "bound": "mfma", "achieved": round(ach, 2), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(ach / MFMA_PEAK_TFLOPS, 4), "traffic": None}
        else:
            ach = dom["bytes"] / (dom_ms * 1e-3) / 1e9
            roofline = {"kernel": name, "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None}
        # HBM bytes per launch from the PMC pass committed under profiles/ (rocprofv3 cannot run inside this process):
        # FETCH_SIZE x 2 (gfx950 correction) + WRITE_SIZE of the eager single-token decode steps, tools/pmc_probe.py
        try:
            with open(os.path.join(ROOT, "profiles", "r01_pmc_gemv.json")) as f:
                pmc = json.load(f)
            if pmc.get("kernel") == name:
                roofline["traffic"] = pmc["hbm_read_bytes_per_launch"] + pmc["hbm_write_bytes_per_launch"]
                roofline["traffic_source"] = pmc["source"] + " (" + pmc["note"] + ")"
        except (OSError, ValueError, KeyError):
            pass
        roofline.update({"avg_launch_us": round(avg_us, 2), "launches_per_batch_pass": dom["calls"],
                         "share_of_gpu_time": round(dom["ms"] / total_ms, 3),
                         "algorithmic_per_launch": round((dom["flops"] if roofline["bound"] == "mfma" else dom["bytes"]) / dom["calls"], 1),
                         "timing": "hipEvent pairs around every launch of one lone pass (eager, sum %.1f ms), rescaled by %.3f to the "
                                   "%.1f ms the same pass takes from the captured graph" % (total_ms, scale, graph_ms)})

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        if model is None:
            model = gf.synth_model(args.model, seed=1)
        log("cpu baseline (reference CPU path, bounded) ...")
        cpu = cpu_baseline(model, args.model, slots[0][1][0].cpu().numpy(), prompt)
        log("cpu baseline done: %s" % cpu.get("value"))

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = world * audio_seconds * args.steps / elapsed
        line = {
            "metric": "audio-seconds/sec (real-time factor), ggml-medium & large, 30s chunks @1/2/4/8 GPU",
            "value": round(value, 2), "unit": "audio-seconds/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": (round(value / PUBLISHED_AUDIO_S_PER_S[args.model], 2) if args.model in PUBLISHED_AUDIO_S_PER_S and B == 7 else None),
            "dtype": "f16", "data": "synthetic",
            "config": {"workload": "ggml-%s shape (random weights), %.3f s clip = %d x 30 s independent windows per GPU, "
                                   "%d clip pass(es) per lock-step batch, %d batches in flight on separate HIP streams; GPU mel + encoder + %d-token prompt + %d greedy steps per window, "
                                   "device-side sampling (captured hipGraph per token)" % (args.model, audio_seconds, B, C, len(slots), N_PROMPT, N_GREEDY),
                       "model": "ggml-" + args.model, "baseline": "BASELINE.md section 1: reference D3D11 backend, GTX 1080Ti, same clip length (whole job, 1 GPU)",
                       "windows_per_clip": B, "clips_per_batch": C, "batches_in_flight": len(slots), "decode_steps_per_window": N_GREEDY + 1,
                       "parallelism": "dp%d (independent windows, RCCL weight broadcast %.3f s outside the timed region)" % (world, t_bcast)},
            "rtf": round(elapsed / (args.steps * audio_seconds), 6),
            "roofline": roofline,
            "cpu_baseline": cpu,
            "kernels": {k: {"calls": v["calls"], "ms": round(v["ms"], 3),
                            "tflops": round(v["flops"] / max(v["ms"], 1e-9) / 1e9, 2), "gbs": round(v["bytes"] / max(v["ms"], 1e-9) / 1e6, 1)}
                        for k, v in kernels.items()},
            "model_build_s": round(t_load, 1),
            "tokens_checksum": int(np.asarray(toks, np.int64)[:B].sum() % 1000003),
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
