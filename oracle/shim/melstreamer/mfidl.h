// oracle/shim/melstreamer/mfidl.h -- TEST INFRASTRUCTURE ONLY.
// Media Foundation does not exist here. The reference's PcmReader (MF/PcmReader.h) holds a CComPtr<IMFSourceReader>; for the
// streaming-spectrogram oracle that "source reader" is 16 kHz mono PCM in memory, handed out `block` samples per ReadSample.
// The default is one 160-sample chunk per delivery, for a reason: see "the end of a stream" in oracle/melstreamer_harness.cpp.
#pragma once
#include "stdafx.h"
struct IMFSourceReader
{
	const float* pcm = nullptr;
	size_t count = 0, cursor = 0, block = 160;
	long refs = 1;
	void AddRef() { refs++; }
	void Release() { if( 0 == --refs ) delete this; }
};
