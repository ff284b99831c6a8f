// oracle/shim/melstreamer/overlay/spectrogram_interface.h -- TEST INFRASTRUCTURE ONLY.
// Linked into the build tree of oracle/Makefile under the name Whisper/Whisper/iSpectrogram.h. The reference declares iSpectrogram
// with MSVC's `__interface` keyword (methods implicitly public and pure virtual); g++ has no such keyword, and MelStreamer.h marks its
// implementations `override final`, so the three methods (iSpectrogram.h:12-23) need a virtual declaration g++ understands.
// MelBufferRaii of the real header belongs to the D3D encoder's upload and is not needed.
#pragma once
#include "audioConstants.h"
#include <vector>
namespace Whisper
{
	struct alignas( 8 ) StereoSample { float left, right; };

	struct iSpectrogram
	{
		virtual ~iSpectrogram() = default;
		// length * N_MEL floats from frame `offset` on, row j at buffer + j * stride
		virtual HRESULT makeBuffer( size_t offset, size_t length, const float** buffer, size_t& stride ) = 0;
		// frames of 160 samples
		virtual size_t getLength() const = 0;
		virtual HRESULT copyStereoPcm( size_t offset, size_t length, std::vector<StereoSample>& buffer ) const = 0;
	};
}
