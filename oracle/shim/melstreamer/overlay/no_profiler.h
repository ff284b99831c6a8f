// oracle/shim/melstreamer/overlay/no_profiler.h -- TEST INFRASTRUCTURE ONLY.
// Linked into the build tree of oracle/Makefile under the name Whisper/Utils/ProfileCollection.h (the GPU + CPU profiler of the D3D
// build): the streamer brackets its FFTs with profiler.cpuBlock( eCpuBlock::Spectrogram ) (MelStreamer.cpp:221, :305); here the
// bracket measures nothing.
#pragma once
#include <stdint.h>
namespace Whisper
{
	enum struct eCpuBlock : uint8_t { Spectrogram };
	class ProfileCollection
	{
	public:
		struct Nothing {};
		Nothing cpuBlock( eCpuBlock ) { return Nothing{}; }
	};
}
