"""oracle/ -- TEST INFRASTRUCTURE ONLY: the checker of the MI355X path, never the thing measured or shipped.

  _ref/          the reference's own CPU path (Whisper/source/ggml.c + whisper.cpp) compiled UNMODIFIED from where it lies under
                 /root/reference by oracle/Makefile (+ ref_harness.cpp, two shims under shim/); git-ignored, travels to the GPU box
  ref.py         ctypes wrapper over _ref/libwhisper_ref.so
  whisper_np.py  numpy restatement of the same arithmetic, pinned against _ref and the fixtures in tests/golden/

Only tests/, __graft_entry__.smoke() and the cpu_baseline leg of bench.py may import this package. Nothing under whisper_amd/
does: the product has no CPU fallback and fails loudly without its HIP library (tests/test_abi.py checks the import graph).
Parity is PINNED: every restated function is checked against outputs of the reference itself.
"""
