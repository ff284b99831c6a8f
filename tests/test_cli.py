"""whisper-main, the command-line tool over libWhisper.so (whisper_amd/host/cli): the counterpart of the reference's
Examples/main (main.cpp:174-330, params.cpp, textWriter.cpp).

CPU tests: option handling and exit codes, and the .txt / .srt / .vtt writers byte for byte (UTF-8 BOM, CRLF, hh:mm:ss.mmm
with hours running past 24, leading blanks of a segment dropped -- textWriter.cpp:52-63, 96-190).
GPU test: a scripted model + a WAV file through the tool must print and write the segments the reference's whisper_full
produced for that model (tests/golden/ref_hostloop.json)."""
import json
import os
import subprocess
import time as time_mod
import wave

import numpy as np
import pytest

from whisper_amd import build, ggml_format as gf

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden", "ref_hostloop.json")


@pytest.fixture(scope="module")
def exe():
    if not os.path.exists(build.CLI_BIN):
        build.build_all()
    return build.CLI_BIN


def run(exe, *args, env=None):
    return subprocess.run([exe] + list(args), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120,
                          env=dict(os.environ, **env) if env else None)


def stamp(t10ms, comma=False):
    ms = t10ms * 10
    return "%02d:%02d:%02d%s%03d" % (ms // 3600000, ms // 60000 % 60, ms // 1000 % 60, "," if comma else ".", ms % 1000)


def test_writers_byte_for_byte(exe, tmp_path):
    prefix = str(tmp_path / "sample")
    assert run(exe, "--format-sample", prefix).returncode == 0
    bom = b"\xef\xbb\xbf"
    t = [("00:00:00.000", "00:00:03.600", "And so my fellow Americans,"),
         ("00:00:03.600", "01:02:03.456", "ask not what your country can do for you"),
         ("25:01:01.001", "25:01:01.999", "ask what you can do for your country.")]
    txt = bom + b"".join(("[%s --> %s]  %s\r\n" % x).encode() for x in t)
    assert open(prefix + ".txt", "rb").read() == txt
    assert open(prefix + ".nostamps.txt", "rb").read() == bom + b"".join((x[2] + "\r\n").encode() for x in t)
    srt = bom + b"".join(("%d\r\n%s --> %s\r\n%s\r\n\r\n" % (i + 1, a.replace(".", ","), b.replace(".", ","), c)).encode()
                         for i, (a, b, c) in enumerate(t))
    assert open(prefix + ".srt", "rb").read() == srt
    vtt = bom + b"WEBVTT\r\n\r\n" + b"".join(("%s --> %s\r\n%s\r\n\r\n" % x).encode() for x in t)
    assert open(prefix + ".vtt", "rb").read() == vtt


def test_writers_against_the_references_own(exe, tmp_path):
    """The .txt / .srt / .vtt writers against the reference's OWN code: Examples/main/textWriter.cpp compiled unmodified (oracle/Makefile ->
    _ref/libtextwriter_ref.so; ATL / PathCch / wide LPCTSTR through shims) and handed the same segments -- leading blanks and tabs, UTF-8 text, times beyond
    24 h (the reference prints days * 24 + hours), sub-millisecond ticks (truncated, not rounded), an empty text, a path whose directory has a dot."""
    so = os.path.join(ROOT, "oracle", "_ref", "libtextwriter_ref.so")
    if not os.path.exists(so):
        pytest.skip("oracle/_ref/libtextwriter_ref.so not present")
    import ctypes as C
    L = C.CDLL(so)
    L.tw_write.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    segs = [(0, 36000000, " And so my fellow Americans,"),
            (36000000, 3723456 * 10000, "\t ask not what your country can do for you"),
            (90061001 * 10000, 90061999 * 10000, "ask what you can do for your country."),
            (12349999, 12350001, "  gr\u00fc\u00dfe, \u4e16\u754c \U0001f600"),
            (10 ** 7 * 3600 * 24 * 3 + 9999, 10 ** 7 * 3600 * 24 * 3 + 10 ** 7 * 59 + 9990000, ""),
            (5, 6, "trailing blank ")]
    d = tmp_path / "dir.with.dot"
    d.mkdir()
    texts = (C.c_char_p * len(segs))(*[s[2].encode() for s in segs])
    b = (C.c_uint64 * len(segs))(*[s[0] for s in segs])
    e = (C.c_uint64 * len(segs))(*[s[1] for s in segs])
    ref_prefix = str(d / "reference")
    for kind, audio in ((0, ref_prefix + ".wav"), (2, ref_prefix + ".wav"), (3, ref_prefix + ".wav"), (1, str(d / "reference_nostamps.flac"))):      # (a path without an extension fails in the reference: its buffer has room for a REPLACED extension only)
        assert L.tw_write(audio.encode(), kind, len(segs), texts, b, e) == 0
    seg_file = str(tmp_path / "segments.txt")
    with open(seg_file, "wb") as f:
        for s in segs:
            f.write(("%d %d " % (s[0], s[1])).encode() + s[2].encode() + b"\n")
    ours = str(d / "ours")
    assert run(exe, "--format-file", seg_file, ours).returncode == 0
    for ext, ref_name in ((".txt", ref_prefix + ".txt"), (".srt", ref_prefix + ".srt"), (".vtt", ref_prefix + ".vtt"), (".nostamps.txt", str(d / "reference_nostamps.txt"))):
        want = open(ref_name, "rb").read()
        got = open(ours + ext, "rb").read()
        assert want.startswith(b"\xef\xbb\xbf") and got == want, (ext, got[:200], want[:200])


def test_options_and_exit_codes(exe, tmp_path):
    r = run(exe, "--help")
    assert r.returncode == 1 and b"--output-srt" in r.stderr and b"--max-context" in r.stderr
    assert run(exe).returncode == 2                                   # no input files
    assert run(exe, "-l", "xx", "a.wav").returncode == 3              # unknown language
    assert run(exe, "--bogus").returncode == 1
    assert run(exe, "-t").returncode == 1                             # missing value
    r = run(exe, "-m", str(tmp_path / "missing.bin"), "a.wav")
    assert r.returncode == 4 and b"failed to load the model" in r.stderr


@pytest.mark.gpu
def test_transcribes_like_the_reference_host_loop(exe, tmp_path):
    case = [c for c in json.load(open(GOLDEN))["cases"] if c["name"] == "first_window_no_prompt"][0]
    model = str(tmp_path / "m.bin")
    gf.write_model(model, gf.scripted_model(case["script"], case["prompt_len"]))
    rng = np.random.default_rng(case["pcm_seed"])
    pcm = (0.05 * rng.standard_normal(case["n_samples"])).astype(np.float32)
    wav = str(tmp_path / "clip.wav")
    with wave.open(wav, "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(16000)
        w.writeframes(np.clip(np.round(pcm * 32768.0), -32768, 32767).astype("<i2").tobytes())
    r = run(exe, "-m", model, "-f", wav, "-l", case["lang"], "-nc", "-otxt", "-osrt", "-ovtt")
    print(r.stdout.decode(), r.stderr.decode()[-2000:])
    assert r.returncode == 0
    segs = case["segments"]
    want_console = "\n" + "".join("[%s --> %s]  %s\n" % (stamp(s["t0"]), stamp(s["t1"]), s["text"]) for s in segs)
    assert r.stdout.decode() == want_console
    bom = b"\xef\xbb\xbf"
    base = str(tmp_path / "clip")
    assert open(base + ".txt", "rb").read() == bom + b"".join(
        ("[%s --> %s]  %s\r\n" % (stamp(s["t0"]), stamp(s["t1"]), s["text"].lstrip(" \t"))).encode() for s in segs)
    assert open(base + ".srt", "rb").read() == bom + b"".join(
        ("%d\r\n%s --> %s\r\n%s\r\n\r\n" % (i + 1, stamp(s["t0"], True), stamp(s["t1"], True), s["text"].lstrip(" \t"))).encode()
        for i, s in enumerate(segs))
    assert open(base + ".vtt", "rb").read() == bom + b"WEBVTT\r\n\r\n" + b"".join(
        ("%s --> %s\r\n%s\r\n\r\n" % (stamp(s["t0"]), stamp(s["t1"]), s["text"].lstrip(" \t"))).encode() for s in segs)
    # timingsPrint: sections and block names of the reference's profiler output (SampleClips/*.txt)
    err = r.stderr.decode()
    for needle in ("    CPU Tasks", "RunComplete\t", "Spectrogram\t", "Encode\t", "Decode\t", "DecodeStep\t", "    Memory Usage", "Model\t", "Context\t", "Total\t"):
        assert needle in err, needle
    assert "Compute Shaders" not in err
    # WHISPER_PROFILE=1 adds the per-kernel table (eager launches with event pairs); the transcript does not change
    r2 = run(exe, "-m", model, "-f", wav, "-l", case["lang"], "-nc", env={"WHISPER_PROFILE": "1"})
    assert r2.returncode == 0 and r2.stdout == r.stdout
    err2 = r2.stderr.decode()
    table = err2[err2.index("    Compute Shaders"):err2.index("    Memory Usage")].splitlines()[1:]
    assert any(t.startswith("gemvFused\t") for t in table) and any(t.startswith("attentionEnc\t") for t in table)
    assert all(" calls, " in t or t.endswith("seconds") for t in table)


# ----------------------------------------------------------------------------------------------------------------------
# whisper-mgpu: one process per GPU over the C ABI (wh_comm_* + loadModelShared), no torch
# ----------------------------------------------------------------------------------------------------------------------
def test_mgpu_harness_arguments():
    """CPU: the harness exists, refuses a call without model / audio, and libwhisper_hip.so exports the communicator entry
    points it binds (the broadcast itself needs GPUs: see the gpu test below and INTEGRATION.md section E)."""
    if not os.path.exists(build.MGPU_BIN):
        build.build_all()
    r = subprocess.run([build.MGPU_BIN], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=60)
    assert r.returncode == 1 and b"-m and -f are required" in r.stderr
    r = subprocess.run([build.MGPU_BIN, "--bogus"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=60)
    assert r.returncode == 1 and b"usage: whisper-mgpu" in r.stderr
    # the harness deals windows to ranks exactly as the Python host does
    from whisper_amd import distributed as wd
    for n in (0, 1, 7, 8, 23, 256):
        for world in (1, 2, 3, 8):
            for rank in range(world):
                out = subprocess.run([build.MGPU_BIN, "--shard-range", str(n), str(rank), str(world)], stdout=subprocess.PIPE, timeout=30)
                assert out.returncode == 0 and tuple(int(x) for x in out.stdout.split()) == tuple(wd.shard_range(n, rank, world))
    syms = subprocess.run(["nm", "-D", "--defined-only", build.HIP_LIB], stdout=subprocess.PIPE, text=True).stdout
    for name in ("wh_comm_unique_id", "wh_comm_create", "wh_comm_destroy", "wh_comm_info", "wh_comm_barrier", "wh_model_broadcast"):
        assert (" T " + name) in syms, name
    syms = subprocess.run(["nm", "-D", "--defined-only", "-C", build.HOST_LIB], stdout=subprocess.PIPE, text=True).stdout
    assert "Whisper::loadModelShared" in syms


def test_mgpu_harness_ends_the_job_when_a_rank_dies_or_hangs(tmp_path):
    """CPU: no rank may hang the node. WHISPER_MGPU_TEST_FAULT makes a rank exit with a code, or sleep for ever, before it touches a
    device: the parent reaps whichever child ends first, ends the others (SIGTERM, SIGKILL) and returns non-zero well inside the
    deadline; with every rank stuck the job's own deadline ends it. (The collectives' deadlines -- wh_comm_create_timeout,
    wh_comm_set_timeout -- are exported and used by the harness; they need GPUs to run.)"""
    import time
    if not os.path.exists(build.MGPU_BIN):
        build.build_all()
    args = [build.MGPU_BIN, "-m", str(tmp_path / "none.bin"), "-f", str(tmp_path / "none.wav"), "-o", str(tmp_path / "t.txt")]
    # rank 1 dies at once, rank 0 and 2 hang: the parent must not wait for them
    t0 = time.time()
    r = subprocess.run(args + ["-n", "3", "-timeout", "60"], env=dict(os.environ, WHISPER_MGPU_TEST_FAULT="1:exit", HIP_VISIBLE_DEVICES=""),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=60)
    assert r.returncode == 1 and time.time() - t0 < 20, r.stderr.decode()[-500:]
    assert b"ending the other ranks" in r.stderr
    # every rank stuck: the JOB's deadline (-job-timeout; -timeout is the deadline of a collective, not of the job) ends them
    t0 = time.time()
    r = subprocess.run(args + ["-n", "2", "-timeout", "0.5", "-job-timeout", "2"], env=dict(os.environ, WHISPER_MGPU_TEST_FAULT="all:hang"),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=60)
    assert r.returncode == 1 and 1.5 < time.time() - t0 < 20, r.stderr.decode()[-500:]
    assert b"did not finish within" in r.stderr
    assert not any(f.startswith("t.txt") for f in os.listdir(tmp_path))
    # under an external launcher every rank is its own process: the id file must be common (-id) or derivable (MASTER_PORT)
    r = subprocess.run(args, env=dict(os.environ, RANK="1", WORLD_SIZE="2", LOCAL_RANK="1"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=60)
    assert r.returncode == 1 and b"-id" in r.stderr
    # ... and the launch must be nameable: torchrun's default run id is the literal "none", the same for every launch on a port, so a rank could take
    # the id file a crashed earlier launch left behind; without a per-launch id from the launcher the ranks insist on -job
    ext = dict(os.environ, RANK="1", WORLD_SIZE="2", LOCAL_RANK="1", MASTER_PORT="29999", TORCHELASTIC_RUN_ID="none")
    for k in ("SLURM_JOB_ID", "OMPI_MCA_ess_base_jobid", "PMIX_NAMESPACE"):
        ext.pop(k, None)
    r = subprocess.run(args, env=ext, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=60)
    assert r.returncode == 1 and b"-job" in r.stderr and b"per-launch id" in r.stderr
    syms = subprocess.run(["nm", "-D", "--defined-only", build.HIP_LIB], stdout=subprocess.PIPE, text=True).stdout
    for name in ("wh_comm_create_timeout", "wh_comm_set_timeout", "wh_comm_broadcast_i32"):
        assert (" T " + name) in syms, name


def test_mgpu_id_file_carries_the_job_not_a_start_time(tmp_path):
    """The rendezvous file of whisper-mgpu under an external launcher: a rank that starts long after rank 0 published the id must still take
    it -- freshness against the rank's OWN start time rejected the valid file on every poll -- while the file of another job (same path, other token)
    and a file older than the rendezvous deadline allows are refused."""
    if not os.path.exists(build.MGPU_BIN):
        build.build_all()
    path = str(tmp_path / "job.id")

    def run(*a):
        return subprocess.run([build.MGPU_BIN] + [str(x) for x in a], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=30).returncode

    assert run("--id-write", path, "run-17/29500/") == 0
    assert os.path.getsize(path) == 64 + 128
    assert run("--id-read", path, "run-17/29500/", 0, 300) == 0
    assert run("--id-read", path, "run-17/29500/", 120, 300) == 0            # this rank starts two minutes after the file was published
    assert run("--id-read", path, "run-18/29500/", 0, 300) == 1              # an earlier job's file under the same name
    assert run("--id-read", path, "run-17/29500/", 400, 300) == 1            # older than the rendezvous deadline: rank 0 has given up by then
    old = time_mod.time() - 1000
    os.utime(path, (old, old))
    assert run("--id-read", path, "run-17/29500/", 0, 300) == 1
    assert run("--id-read", str(tmp_path / "missing.id"), "x", 0, 300) == 1


@pytest.mark.gpu
def test_mgpu_harness_one_rank_matches_the_cli(exe, tmp_path):
    """One rank end to end: RCCL communicator of size 1, loadModelShared (file -> arena -> ncclBroadcast in place), the
    window range of the rank through runFull. The transcript must be the one whisper-main prints for the same input."""
    case = [c for c in json.load(open(GOLDEN))["cases"] if c["name"] == "first_window_no_prompt"][0]
    model = str(tmp_path / "m.bin")
    gf.write_model(model, gf.scripted_model(case["script"], case["prompt_len"]))
    rng = np.random.default_rng(case["pcm_seed"])
    pcm = (0.05 * rng.standard_normal(case["n_samples"])).astype(np.float32)
    wav = str(tmp_path / "clip.wav")
    with wave.open(wav, "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(16000)
        w.writeframes(np.clip(np.round(pcm * 32768.0), -32768, 32767).astype("<i2").tobytes())
    out = str(tmp_path / "t.txt")
    r = subprocess.run([build.MGPU_BIN, "-n", "1", "-m", model, "-f", wav, "-l", case["lang"], "-o", out],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    print(r.stdout.decode(), r.stderr.decode()[-3000:])
    assert r.returncode == 0
    line = json.loads(r.stdout.decode().strip().splitlines()[-1])
    assert line["ranks"] == 1 and line["windows"] >= 1
    got = open(out).read().splitlines()
    segs = case["segments"]
    assert len(got) == len(segs)
    for g, s in zip(got, segs):
        assert g.endswith("] " + s["text"]), (g, s)
        t0, t1 = float(g[1:10]), float(g[15:24])
        assert abs(t0 - s["t0"] / 100.0) < 0.006 and abs(t1 - s["t1"] / 100.0) < 0.006


@pytest.mark.gpu
def test_mgpu_per_recording_keeps_the_sliding_window(exe, tmp_path):
    """whisper-mgpu -per-recording: whole recordings dealt to the ranks, each ONE stream with the reference's host loop -- windows advance by the
    timestamp the decoder ended on (seek_delta, ContextImpl.cpp:618-625, 785), text carries over as the next window's prompt -- instead of independent 30 s chunks.
    Two recordings on one rank (the batch runner decodes them in lock step): each transcript must be what iContext::runFull returns for that recording alone,
    segment times included; on the multi-window recording that differs from the chunk mode's transcript, which is the trade-off the option makes a choice."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden_runfull", os.path.join(ROOT, "tests", "golden", "make_golden_runfull.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    from whisper_amd import api
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_runfull_conditioned.json")))
    cases = [next(c for c in fx["cases"] if c["name"].startswith("long")), next(c for c in fx["cases"] if c["name"].startswith("jfk"))]
    assert cases[0]["seed"] == cases[1]["seed"]
    model = str(tmp_path / "cond.bin")
    gf.write_model(model, mg.model_for(cases[0]["seed"]))
    wavs, pcms = [], []
    for i, c in enumerate(cases):
        pcm = mg.pcm_for(c["pcm"])
        q = np.clip(np.round(pcm * 32768.0), -32768, 32767).astype("<i2")
        path = str(tmp_path / ("rec%d.wav" % i))
        with wave.open(path, "wb") as w:
            w.setnchannels(1)
            w.setsampwidth(2)
            w.setframerate(16000)
            w.writeframes(q.tobytes())
        wavs.append(path)
        pcms.append(q.astype(np.float32) / 32768.0)
    out = str(tmp_path / "t.txt")
    r = subprocess.run([build.MGPU_BIN, "-n", "1", "-m", model, "-per-recording", "-f", wavs[0], "-f", wavs[1], "-o", out, "-timeout", "60"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    print(r.stdout.decode(), r.stderr.decode()[-2000:])
    assert r.returncode == 0
    text = open(out).read()
    parts = text.split("== ")[1:]
    assert len(parts) == 2 and parts[0].startswith(wavs[0]) and parts[1].startswith(wavs[1])
    m = api.Model(model)
    for part, pcm in zip(parts, pcms):
        ctx = m.create_context()
        assert ctx.run_full(pcm) == 0
        want = ctx.results()
        ctx.close()
        got = part.splitlines()[1:]
        assert len(got) == len(want) and len(want) >= 1
        for g, s in zip(got, want):
            assert g.endswith("] " + s["text"].decode()), (g, s["text"])
            t0, t1 = float(g[1:10]), float(g[15:24])
            assert abs(t0 - s["t0"] / 1e7) < 0.006 and abs(t1 - s["t1"] / 1e7) < 0.006
    m.close()
    # the chunk mode on the long recording: independent 30 s chunks, another transcript
    r2 = subprocess.run([build.MGPU_BIN, "-n", "1", "-m", model, "-f", wavs[0], "-o", out + ".chunks", "-timeout", "60"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r2.returncode == 0
    assert [ln.split("] ", 1)[-1] for ln in open(out + ".chunks").read().splitlines()] != [ln.split("] ", 1)[-1] for ln in parts[0].splitlines()[1:]]
    # several recordings without the option: refused
    r3 = subprocess.run([build.MGPU_BIN, "-n", "1", "-m", model, "-f", wavs[0], "-f", wavs[1]], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=60)
    assert r3.returncode == 1 and b"-per-recording" in r3.stderr


@pytest.mark.gpu
def test_mgpu_two_ranks_over_rccl_or_a_loud_failure(exe, tmp_path):
    """whisper-mgpu -n 2 through the C++ path (wh_comm_* over RCCL, loadModelShared, one batch runner per rank). With two devices visible: two ranks on two
    GPUs, the arena broadcast over the fabric, and the concatenated transcript is the one rank's (chunks are independent recordings). With ONE device
    (the test box): both ranks bind device 0, which RCCL cannot serve -- the job must END with an error inside the rendezvous deadline (RCCL's own
    refusal or wh_comm_create_timeout's), never hang, and leave no partial output."""
    import time
    from whisper_amd import binding
    case = [c for c in json.load(open(GOLDEN))["cases"] if c["name"] == "first_window_no_prompt"][0]
    model = str(tmp_path / "m.bin")
    gf.write_model(model, gf.scripted_model(case["script"], case["prompt_len"]))
    rng = np.random.default_rng(case["pcm_seed"])
    pcm = np.tile((0.05 * rng.standard_normal(case["n_samples"])).astype(np.float32), 3)[:16000 * 75]      # three chunks: ranks get 2 + 1
    wav = str(tmp_path / "clip.wav")
    with wave.open(wav, "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(16000)
        w.writeframes(np.clip(np.round(pcm * 32768.0), -32768, 32767).astype("<i2").tobytes())
    outs = {}
    n_dev = binding.device_count()
    for ranks in (1, 2):
        out = str(tmp_path / ("t%d.txt" % ranks))
        t0 = time.time()
        r = subprocess.run([build.MGPU_BIN, "-n", str(ranks), "-m", model, "-f", wav, "-l", case["lang"], "-o", out, "-timeout", "25", "-job-timeout", "120"],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=200)
        dt = time.time() - t0
        err = r.stderr.decode()
        print("ranks %d on %d device(s): rc %d in %.1f s\n%s" % (ranks, n_dev, r.returncode, dt, err[-1500:]))
        if ranks == 1 or n_dev >= 2:
            assert r.returncode == 0, err[-800:]
            outs[ranks] = open(out).read()
            assert json.loads(r.stdout.decode().strip().splitlines()[-1])["ranks"] == ranks
        else:
            assert r.returncode != 0 and dt < 90, "two ranks on one device must fail loudly and soon"
            assert ("ending the other ranks" in err) and ("gave up" in err or "RCCL" in err or "nccl" in err.lower() or "communicator" in err), err[-800:]
            assert not os.path.exists(out) and not any(f.startswith("t2.txt") for f in os.listdir(tmp_path))
    if 2 in outs:
        assert outs[1] == outs[2] and len(outs[1].splitlines()) >= 3


REF_CLI = os.path.join(ROOT, "oracle", "_ref", "libcliparams_ref.so")


def _ref_parse(args, argv0="main", threads=4, capture=False):
    """The reference's own whisper_params::parse (Examples/main/params.cpp compiled unmodified, oracle/_ref/libcliparams_ref.so)."""
    import ctypes as C
    import tempfile
    L = C.CDLL(REF_CLI)
    argv = (C.c_char_p * (len(args) + 1))(argv0.encode(), *[a.encode() for a in args])
    out = C.create_string_buffer(16384)
    text = None
    if capture:          # what it prints goes to the C library's stderr
        with tempfile.TemporaryFile() as tf:
            saved = os.dup(2)
            os.dup2(tf.fileno(), 2)
            try:
                go = L.cp_parse(len(args) + 1, argv, threads, out, 16384)
                C.CDLL(None).fflush(None)
            finally:
                os.dup2(saved, 2)
                os.close(saved)
            tf.seek(0)
            text = tf.read()
    else:
        go = L.cp_parse(len(args) + 1, argv, threads, out, 16384)
    return go, json.loads(out.value.decode()), text


def test_command_line_against_the_reference_parser(exe):
    """Row f2, the command line: whisper-main's parser (--dump-options) against the reference's whisper_params::parse, and its usage text
    against whisper_print_usage byte for byte (Examples/main/params.cpp compiled unmodified into oracle/_ref/libcliparams_ref.so)."""
    if not os.path.exists(REF_CLI):
        pytest.skip("oracle/_ref/libcliparams_ref.so not built (needs /root/reference)")
    hw = min(4, max(1, os.cpu_count() or 1))
    cases = [
        [],
        ["a.wav"],
        ["-t", "8", "-p", "2", "-ot", "1500", "-on", "3", "-d", "20000", "-mc", "64", "-ml", "40", "-wt", "0.25", "a.wav", "b.wav"],
        ["--threads", "2", "--processors", "1", "--offset-t", "0", "--offset-n", "0", "--duration", "0", "--max-context", "0", "--max-len", "1", "--word-thold", "1", "-f", "x.wav"],
        ["-su", "-tr", "-di", "-otxt", "-ovtt", "-osrt", "-owts", "-ps", "-nc", "-nt", "z.wav"],
        ["--speed-up", "--translate", "--diarize", "--output-txt", "--output-vtt", "--output-srt", "--output-words", "--print-special", "--no-colors", "--no-timestamps", "--file", "z.wav"],
        ["-l", "de", "-m", "models/ggml-medium.bin", "-gpu", "AMD Instinct MI355X", "--prompt", "Hello, \"world\" \u00e9t\u00e9", "clip one.wav"],
        ["--language", "ja", "--model", "m.bin", "--use-gpu", "x", "one.wav", "-f", "two.wav", "three.wav"],
        ["-mc", "4294967295", "-t", "1", "a.wav"],
        ["a.wav", "-otxt", "b.wav", "-osrt"],
    ]
    for args in cases:
        go, want, _ = _ref_parse(args, threads=hw)
        r = run(exe, "--dump-options", *args)
        assert go == 1 and r.returncode == 0, (args, r.stderr)
        got = json.loads(r.stdout.decode())
        assert set(got) == set(want)
        for k in want:
            if k == "word_thold":
                assert abs(got[k] - want[k]) < 1e-6, (args, k)
            else:
                assert got[k] == want[k], (args, k, got[k], want[k])
    # what stops the reference stops whisper-main: help, an unknown argument (both print the usage text)
    for args in (["-h"], ["--help"], ["--bogus", "a.wav"], ["a.wav", "-xyz"]):
        go, _, text = _ref_parse(args, argv0=exe, threads=hw, capture=True)
        r = run(exe, *args)
        assert go == 0 and r.returncode == 1, args
        assert r.stderr == text, (args, r.stderr.decode()[-400:], text.decode()[-400:])
