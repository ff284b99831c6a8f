// oracle/shim/melstreamer/overlay/cpu_compute_context.h -- TEST INFRASTRUCTURE ONLY.
// Linked into the build tree of libcontextimpl_ref.so under the name Whisper/Whisper/WhisperContext.h. The real class is the D3D11
// compute context (encoder / decoder as compute shader dispatches). ContextImpl -- the host loop this oracle exists for -- uses
// five things of it (ContextImpl.cpp:36, :61, :523, :528, :596; ContextImpl.misc.cpp:121): here encode() and decode() run the
// reference's own CPU model instead (defined in oracle/contextimpl_harness.cpp over oracle/_ref/libwhisper_ref.so).
#pragma once
#include <vector>
#include <emmintrin.h>
#include "sEncodeParams.h"
#include "iSpectrogram.h"
#include "WhisperModel.h"
#include "../Utils/ProfileCollection.h"
namespace DirectCompute
{
	struct EncoderOutput {};	// the real encode() returns the output tensor; ContextImpl only hands it to the tracer
	class WhisperContext
	{
		const Whisper::WhisperModel& model;
	public:
		WhisperContext( const Whisper::WhisperModel& m, Whisper::ProfileCollection& ) : model( m ) {}
		EncoderOutput encode( Whisper::iSpectrogram& mel, const sEncodeParams& ep );	// throws HRESULT
		void decode( const int* tokens, int length, const sDecodeParams& dp, std::vector<float>& probs, int threads );
		struct Nothing {};
		Nothing completeProfiler() { return Nothing{}; }
		Nothing decodeProfiler() { return Nothing{}; }
		HRESULT clearState() { return S_OK; }	// the CPU model's caches are position-addressed: nothing to clear
		__m128i getMemoryUse() const { return _mm_setzero_si128(); }
	};
}
