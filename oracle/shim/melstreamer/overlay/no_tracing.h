// oracle/shim/melstreamer/overlay/no_tracing.h -- TEST INFRASTRUCTURE ONLY.
// Linked into the build tree of libcontextimpl_ref.so under the name Whisper/Utils/Trace/tracing.h: the debug tracer of the D3D
// build (ContextImpl.cpp:37 traces the encoder's output; compiled to nothing in the reference's release builds, SAVE_DEBUG_TRACE 0).
#pragma once
namespace Tracing
{
	template<class... A> inline void tensor( A&&... ) {}
	template<class... A> inline void vector( A&&... ) {}
}
