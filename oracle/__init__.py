"""oracle/ -- TEST INFRASTRUCTURE ONLY: the checker of the MI355X path, never the thing measured or shipped.

  _ref/                     the reference compiled UNMODIFIED from where it lies under /root/reference by oracle/Makefile; git-ignored, travels to the GPU box
    libwhisper_ref.so       its CPU path (Whisper/source/ggml.c + whisper.cpp; ref_harness.cpp, shim/Utils/Logger.h, shim/trace_shim.h)
    libmelstreamer_ref.so   its streaming and whole-buffer spectrograms (Whisper/Whisper/MelStreamer.cpp, Spectrogram.cpp, melSpectrogram.cpp,
                            MF/AudioBuffer.cpp; melstreamer_harness.cpp, shim/melstreamer/: Win32 / ATL / DirectXMath names + three overlay headers)
    libcontextimpl_ref.so   its GPU-model iContext (Whisper/Whisper/ContextImpl.cpp, ContextImpl.misc.cpp, Languages.cpp + the spectrogram sources):
                            host loop, sampler, results, token-level timestamps -- the D3D compute context replaced by libwhisper_ref.so
                            (contextimpl_harness.cpp)
    libcliparams_ref.so     the command line of its CLI (Examples/main/params.cpp; cliparams_harness.cpp, shim/cli/)
  ref.py                    ctypes wrappers: RefWhisper, RefMelStreamer / spectrogram_pcm_to_mel, RefContextImpl
  whisper_np.py             numpy restatement of the same arithmetic, pinned against _ref and the fixtures in tests/golden/

Only tests/, __graft_entry__.smoke() and the cpu_baseline leg of bench.py may import this package. Nothing under whisper_amd/
does: the product has no CPU fallback and fails loudly without its HIP library (tests/test_abi.py checks the import graph).
Parity is PINNED: every restated function is checked against outputs of the reference itself.
"""
