// oracle/shim/melstreamer/overlay/no_device.h -- TEST INFRASTRUCTURE ONLY.
// Linked into the build tree of libcontextimpl_ref.so under the name Whisper/ML/Device.h (the D3D11 device and its per-thread
// binding, ContextImpl.cpp:454): there is no device here.
#pragma once
namespace DirectCompute
{
	struct Device
	{
		struct Nothing {};
		Nothing setForCurrentThread() const { return Nothing{}; }
	};
}
