// oracle/shim/melstreamer/overlay/filters_only.h -- TEST INFRASTRUCTURE ONLY.
// Linked into the build tree of oracle/Makefile under the name Whisper/Whisper/WhisperModel.h. The real header describes the whole
// model in VRAM (D3D11 tensors, vocabulary, loader); the spectrogram sources use one thing of it: the mel filterbank as loaded from
// the ggml file (WhisperModel.h:12-17), of which melSpectrogram.cpp:374 reads `data` (80 rows of 201 coefficients).
#pragma once
#include <stdint.h>
#include <vector>
namespace Whisper
{
	struct Filters
	{
		std::vector<float> data;	   // [ n_mel ][ n_fft ]
		uint32_t n_mel = 0, n_fft = 0;
	};
}
