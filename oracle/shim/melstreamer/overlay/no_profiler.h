// oracle/shim/melstreamer/overlay/no_profiler.h -- TEST INFRASTRUCTURE ONLY.
// Linked into the build trees of oracle/Makefile under the name Whisper/Utils/ProfileCollection.h (the GPU + CPU profiler of the D3D
// build): the streamer and the host loop bracket their stages with profiler.cpuBlock( eCpuBlock::... ) (MelStreamer.cpp:221, :305;
// ContextImpl.cpp:19, :538 ...), iContext::timingsPrint / timingsReset forward to print() / reset(); here nothing is measured.
#pragma once
#include <stdint.h>
namespace Whisper
{
	struct WhisperModel;
	enum struct eCpuBlock : uint8_t { LoadModel, RunComplete, Run, Callbacks, Spectrogram, Sample, VAD, Encode, Decode, DecodeStep, DecodeLayer };
	class ProfileCollection
	{
	public:
		ProfileCollection() = default;
		ProfileCollection( const WhisperModel& ) {}
		struct Nothing {};
		Nothing cpuBlock( eCpuBlock ) { return Nothing{}; }
		void print() {}
		void reset() {}
	};
}
