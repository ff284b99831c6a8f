"""CPU tests of the lock-step batch scheduler: whisper_amd/host/batchScheduler.cpp (Whisper::createBatchRunner / iBatchRunner::run -- what a
caller of libWhisper.so gets the headline throughput through) compiled UNCHANGED into a test library whose compute layer is a test double
(tests/hostloop_cpu/fake_device.cpp: the dozen entry points of include/whisper_hip.h the scheduler calls, one reference CPU model per slot).
The claim under test is the scheduler's contract: the transcript of every stream is the transcript of that stream run ALONE through the same
host loop (tests/hostloop_cpu/driver.cpp; itself pinned on the reference's two host loops, tests/test_hostloop_cpu.py) -- whatever the number
of slots and groups, the order streams finish in, the chunk size, the look-ahead; pieces of a recording are recordings of their own with
times shifted by their start; a stream that cannot run fails alone. The GPU twin: tests/test_batch_api.py (through libWhisper.so).
(Offline, the same comparison over a dozen random arrangements -- 3 to 9 streams with random pieces, 1 / 2 / 3 / 5 / 64 slots, 1 to 3 groups,
chunks of 1 to 64 steps, with and without look-ahead, with and without prompt carry-over, both rule sets -- found no difference.)"""
import ctypes as C
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

import test_hostloop_cpu as H
from whisper_amd import ggml_format as gf

ROOT = H.ROOT
LIB = os.path.join(H.BUILD, "libbatch_cpu.so")
SOURCES = [os.path.join(ROOT, "tests", "hostloop_cpu", f) for f in ("batch_driver.cpp", "fake_device.cpp")] + \
          [os.path.join(ROOT, "whisper_amd", "host", f) for f in ("batchScheduler.cpp", "support.cpp", "tokenTimestamps.cpp")]
HEADERS = H.HEADERS


class StreamDesc(C.Structure):
    _fields_ = [("buffer", C.c_int32), ("firstSample", C.c_int64), ("countSamples", C.c_int64)]


@pytest.fixture(scope="module")
def batch_lib():
    if not os.path.exists(os.path.join(H.REF_DIR, "libwhisper_ref.so")):
        pytest.skip("oracle/_ref/libwhisper_ref.so not built (needs /root/reference)")
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    os.makedirs(H.BUILD, exist_ok=True)
    deps = SOURCES + HEADERS + [os.path.join(H.REF_DIR, "libwhisper_ref.so")]
    if not os.path.exists(LIB) or any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in deps):
        cmd = ["g++", "-std=c++17", "-O1", "-fPIC", "-shared", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "whisper_amd", "host")] + SOURCES + \
              ["-o", LIB, "-L" + H.REF_DIR, "-lwhisper_ref", "-Wl,-rpath," + H.REF_DIR, "-Wl,-Bsymbolic", "-Wl,--no-undefined", "-lpthread"]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert r.returncode == 0, r.stdout
    L = C.CDLL(LIB)
    L.bt_run.argtypes = [C.c_char_p, C.c_int, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(C.c_int32), C.c_int, C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.c_int32),
                         C.c_int, C.POINTER(StreamDesc), C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int]
    L.bt_result.restype = C.c_char_p
    L.fake_device_counters.argtypes = [np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")]
    return L


def run_batch(L, path, rules, buffers, streams, slots, groups, chunk, lookahead, flags=H.FLAG_NO_CONTEXT, prompt=(1000,), n_max_text_ctx=0, lang="en"):
    bufs = [np.ascontiguousarray(b, np.float32) for b in buffers]
    ptrs = (C.POINTER(C.c_float) * len(bufs))(*[b.ctypes.data_as(C.POINTER(C.c_float)) for b in bufs])
    lens = (C.c_int32 * len(bufs))(*[len(b) for b in bufs])
    descs = (StreamDesc * len(streams))(*[StreamDesc(b, f, n) for (b, f, n) in streams])
    pt = (C.c_int32 * max(1, len(prompt)))(*(list(prompt) or [0]))
    hr = L.bt_run(path.encode(), rules, flags, H.language_key(lang), n_max_text_ctx, C.cast(pt, C.POINTER(C.c_int32)) if prompt else None, len(prompt),
                  ptrs, lens, len(bufs), descs, len(streams), slots, groups, chunk, lookahead, 4)
    return hr, json.loads(L.bt_result().decode())


def alone(driver, tmp_path, model, pcm, rules, name, flags=H.FLAG_NO_CONTEXT, prompt=(1000,), n_max_text_ctx=0):
    c = dict(name=name, lang="en", flags=dict(no_context=bool(flags & H.FLAG_NO_CONTEXT)), prompt=list(prompt) or None, n_max_text_ctx=n_max_text_ctx)
    hr, got = H.run_case(driver, tmp_path, c, pcm, rules=rules, model=model)
    return hr, ([(s["t0"], s["t1"], s["text"], [t["id"] for t in s["tokens"]]) for s in got["segments"]] if got else [])


def recordings():
    jfk = np.load(os.path.join(ROOT, "tests", "golden", "ref_test_d128.npz"))["pcm16"].astype(np.float32) / 32768.0
    rng = np.random.default_rng(3)
    noisy = (jfk * 0.5 + 0.01 * rng.standard_normal(len(jfk))).astype(np.float32)
    return [jfk, np.concatenate([jfk[::-1], noisy]).astype(np.float32), (0.3 * jfk[::2]).astype(np.float32), jfk[:8000].copy(),
            np.concatenate([noisy, jfk, jfk[::2]]).astype(np.float32)]


@pytest.mark.parametrize("rules", [0, 1])
def test_every_stream_equals_the_stream_alone(batch_lib, tmp_path, rules):
    """Audio-conditioned model (tokens and timestamps depend on the audio, so streams finish their windows at different steps and their
    seeks part): seven streams -- five recordings of 0.5 .. 27.5 s, two pieces of one of them -- under four (slots, groups, chunk, look-ahead)
    arrangements, among them fewer slots than streams (slots are refilled as streams retire) and more slots than streams (idle slots)."""
    model = gf.conditioned_model(gf.conditioned_layout(gf.hparams_for("test-d128-ml")), 4, kind="test-d128-ml", seed=10)
    path = str(tmp_path / "m.bin")
    gf.write_model(path, model)
    bufs = recordings()
    streams = [(0, 0, 0), (1, 0, 0), (2, 0, 0), (3, 0, 0), (4, 0, 0), (4, 16000 * 5, 16000 * 9), (1, 16000 * 12, 0)]
    drv = _driver()
    want = []
    for i, (b, first, count) in enumerate(streams):
        pcm = bufs[b][first:first + count] if count else bufs[b][first:]
        hr, segs = alone(drv, tmp_path, model, pcm, rules, "alone%d" % i)
        shift = first * 10000000 // 16000
        want.append((hr, [(t0 * 100000 + shift, t1 * 100000 + shift, text, ids) for (t0, t1, text, ids) in segs]))
    assert sum(len(w[1]) for w in want) >= 8 and want[3][0] == 1          # the 0.5 s recording: S_FALSE, nothing transcribed
    for slots, groups, chunk, lookahead in ((2, 2, 4, 0), (3, 1, 7, 1), (64, 2, 4, 0), (1, 1, 3, 0)):
        before = np.zeros(6, np.int64)
        batch_lib.fake_device_counters(before)
        hr, got = run_batch(batch_lib, path, rules, bufs, streams, slots, groups, chunk, lookahead)
        assert hr == 0, hr
        # the runner released every device context it created
        after = np.zeros(6, np.int64)
        batch_lib.fake_device_counters(after)
        assert after[0] == before[0]
        for i, (st, w) in enumerate(zip(got["streams"], want)):
            assert st["hr"] == w[0], (slots, groups, i, st["hr"])
            assert [(s["t0"], s["t1"], s["text"], s["tokens"]) for s in st["segments"]] == w[1], (slots, groups, chunk, lookahead, i)
        assert got["new_segments"] == sum(len(w[1]) for w in want)
        # every new_segment callback asked for the results the reference's way (no NewObject, Release on scope exit), twice: the object is the
        # context's own and survives the Release (a heap object handed out without AddRef was freed by the first such callback)
        assert got["callback_faults"] == 0


def _driver():
    """the sequential driver of tests/test_hostloop_cpu.py (same build rule as its fixture)"""
    if not os.path.exists(os.path.join(H.HIP_DIR, "libwhisper_hip.so")):
        pytest.skip("libwhisper_hip.so not built: the sequential driver links against it")
    deps = H.SOURCES + H.HEADERS + [os.path.join(H.REF_DIR, "libwhisper_ref.so")]
    if not os.path.exists(H.LIB) or any(os.path.getmtime(d) > os.path.getmtime(H.LIB) for d in deps):
        cmd = ["g++", "-std=c++17", "-O1", "-fPIC", "-shared", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "whisper_amd", "host")] + H.SOURCES + \
              ["-o", H.LIB, "-L" + H.HIP_DIR, "-lwhisper_hip", "-L" + H.REF_DIR, "-lwhisper_ref", "-Wl,-rpath," + H.HIP_DIR, "-Wl,-rpath," + H.REF_DIR, "-lpthread"]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert r.returncode == 0, r.stdout
    L = C.CDLL(H.LIB)
    L.hl_run.argtypes = [C.c_char_p, C.c_int, C.POINTER(H.HlParams), np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS"), C.c_int, C.c_int]
    L.hl_result.restype = C.c_char_p
    return L


def test_carry_over_and_a_stream_that_cannot_run(batch_lib, tmp_path):
    """Prompt carry-over (no NoContext flag, n_max_text_ctx = 1): every stream conditions its windows on ITS OWN past text, so the prompts of
    a lock-step batch differ in length (ragged) from the second round on. And a stream that names samples outside its buffer fails alone
    (E_INVALIDARG in perStream, no transcript) while the run returns that first failure and every other stream is complete."""
    hp = gf.hparams_for("test-d128-ml")
    model = gf.conditioned_model(gf.conditioned_layout(hp), 4, kind="test-d128-ml", seed=11)
    path = str(tmp_path / "m.bin")
    gf.write_model(path, model)
    bufs = recordings()
    streams = [(4, 0, 0), (0, 0, 0), (1, 0, 0)]
    drv = _driver()
    want = [alone(drv, tmp_path, model, bufs[b], 0, "co%d" % i, flags=0, prompt=(1000,), n_max_text_ctx=1) for i, (b, _, _) in enumerate(streams)]
    hr, got = run_batch(batch_lib, path, 0, bufs, streams, 2, 2, 4, 0, flags=0, prompt=(1000,), n_max_text_ctx=1)
    assert hr == 0
    for st, w in zip(got["streams"], want):
        assert st["hr"] == w[0] and [(s["t0"] // 100000, s["t1"] // 100000, s["text"], s["tokens"]) for s in st["segments"]] == w[1]
    bad = streams + [(0, 16000 * 20, 16000)]           # jfk.wav has 11 s
    hr, got = run_batch(batch_lib, path, 0, bufs, bad, 2, 2, 4, 0, flags=0, prompt=(1000,), n_max_text_ctx=1)
    assert hr & 0xFFFFFFFF == 0x80070057 and got["streams"][3]["hr"] & 0xFFFFFFFF == 0x80070057 and got["streams"][3]["segments"] == []
    for st, w in zip(got["streams"][:3], want):
        assert st["hr"] == w[0] and [(s["t0"] // 100000, s["t1"] // 100000, s["text"], s["tokens"]) for s in st["segments"]] == w[1]
