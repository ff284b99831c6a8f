"""Generates tests/golden/ref_melstreamer.npz from the REFERENCE's streaming spectrogram -- Whisper/Whisper/MelStreamer.cpp +
melSpectrogram.cpp + MF/AudioBuffer.cpp, and Spectrogram.cpp (`whole`: Spectrogram::pcmToMel of the GPU model's runFull) compiled unmodified into oracle/_ref/libmelstreamer_ref.so (oracle/Makefile; needs
/root/reference, so it runs in the build container only; the fixture travels).

    python tests/golden/make_golden_melstreamer.py

The clip is the first 4 s of the committed test recording (ref_test_d128.npz pcm16) with a quiet tail (the window-local and the
whole-stream maxima differ) minus 83 samples: 63917 samples = 399 whole 160-sample chunks + one partial chunk of 77 samples.
Requests, in the order iContext::runStreamed makes them (offsets only grow, MelInputTensor.cpp:37-39 clamps a request to the
stream's length): a fresh maximum, the re-used maximum when a request ends where the last one ended (MelStreamer.cpp:158-172),
a window-local maximum, and -- MelStreamerSimple only -- a request PAST the length: the frame of the partial chunk is computed,
the frames after it are zero before normalisation. Both streamers (FFTs on demand / background thread, 4 workers) must agree bit
for bit on every request inside the length; past it the threaded one returns zeros for the partial chunk's frame too
(MelStreamer.cpp:287-291 stops at getLength()), which runStreamed can never observe. The in-memory source reader delivers one
160-sample chunk per read: with larger deliveries the reference's PcmReader hands out stale memory after the last partial chunk
(oracle/melstreamer_harness.cpp); the frames that touches are kept as `stale_end_block4096` for the record, not for comparison.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import ref  # noqa: E402
from whisper_amd import ggml_format as gf  # noqa: E402

N_SAMPLES = 64000 - 83
REQUESTS = [(0, 399), (200, 199), (250, 100), (300, 99), (320, 79)]
PAST_END = (350, 60)


def clip():
    g = np.load(os.path.join(HERE, "ref_test_d128.npz"))
    pcm = g["pcm16"].astype(np.float32) / 32768.0
    pcm = pcm[:N_SAMPLES].copy()
    pcm[32000:] *= 0.02
    return pcm


def main():
    pcm = clip()
    filters = np.ascontiguousarray(gf.synth_model("test-d128", seed=1).filters, np.float32)
    out = {"n_samples": np.int64(len(pcm)), "requests": np.asarray(REQUESTS, np.int64), "past_end": np.asarray(PAST_END, np.int64)}
    simple = ref.RefMelStreamer(pcm, filters, threads=1)
    thread = ref.RefMelStreamer(pcm, filters, threads=4)
    assert simple.length == thread.length == len(pcm) // 160
    for i, (off, ln) in enumerate(REQUESTS):
        a, b = simple.make_buffer(off, ln), thread.make_buffer(off, ln)
        assert np.array_equal(a, b), "the reference's two streamers differ inside the stream's length"
        out["window%d" % i] = a
    a, b = simple.make_buffer(*PAST_END), thread.make_buffer(*PAST_END)
    out["past_end_simple"], out["past_end_thread"] = a, b
    # A different delivery size of the source reader changes nothing -- except where the reference reads stale memory: the frames
    # whose 400 samples reach past the last whole chunk (oracle/melstreamer_harness.cpp, "the end of a stream")
    other = ref.RefMelStreamer(pcm, filters, threads=1, block=4096).make_buffer(0, 399)
    assert np.array_equal(other[:, :397], out["window0"][:, :397])
    out["stale_end_block4096"] = other[:, 397:]
    # row a1 on the file SURVEY.md section 8 cites for the GPU model: Spectrogram::pcmToMel (Spectrogram.cpp:64-122), the whole buffer
    # normalised on its global maximum; 1 thread and 4 must agree bit for bit (frames are independent)
    whole = ref.spectrogram_pcm_to_mel(pcm, filters, threads=1)
    assert np.array_equal(whole, ref.spectrogram_pcm_to_mel(pcm, filters, threads=4))
    # a stream of one window: the streamer's buffer IS the whole-buffer spectrogram, bit for bit -- window0 serves as both
    assert np.array_equal(whole, out["window0"])
    path = os.path.join(HERE, "ref_melstreamer.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
