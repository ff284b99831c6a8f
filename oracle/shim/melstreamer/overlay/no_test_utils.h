// oracle/shim/melstreamer/overlay/no_test_utils.h -- TEST INFRASTRUCTURE ONLY.
// Linked into the build tree of oracle/Makefile under the name Whisper/ML/testUtils.h: Spectrogram.cpp includes the D3D build's
// debug helpers (tensor dumps to disk) and uses none of them outside a commented-out line (Spectrogram.cpp:115).
#pragma once
