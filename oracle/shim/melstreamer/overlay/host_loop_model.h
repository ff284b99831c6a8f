// oracle/shim/melstreamer/overlay/host_loop_model.h -- TEST INFRASTRUCTURE ONLY.
// Linked into the build tree of libcontextimpl_ref.so under the name Whisper/Whisper/WhisperModel.h. The real header is the model
// in VRAM; ContextImpl's host code reads the hyper-parameters, the vocabulary and the mel filterbank from it (model.parameters.*,
// model.shared->vocab, model.shared->filters) and asks for its memory use. `cpuModel` is what replaces the tensors: the
// reference's own CPU model (whisper_context of Whisper/source/whisper.cpp) that oracle/contextimpl_harness.cpp computes with.
#pragma once
#include <stdint.h>
#include <memory>
#include <vector>
#include <emmintrin.h>
#include "Vocabulary.h"
#include "sModelParams.h"
namespace Whisper
{
	struct Filters
	{
		std::vector<float> data;	   // [ n_mel ][ n_fft ]
		uint32_t n_mel = 0, n_fft = 0;
	};
	struct ModelShared
	{
		Vocabulary vocab;
		Filters filters;
	};
	struct WhisperModel
	{
		sModelParams parameters;
		std::shared_ptr<ModelShared> shared;
		void* cpuModel = nullptr;
		int cpuThreads = 4;
		__m128i getMemoryUse() const { return _mm_setzero_si128(); }
	};
}
