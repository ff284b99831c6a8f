// oracle/shim/trace_shim.h -- TEST INFRASTRUCTURE ONLY.
// Force-included ahead of the reference's Whisper/source/whisper.cpp. The reference calls
// Tracing::delayTensor / tensor / vector / writeDelayedTensors at the probe points listed in
// SURVEY.md section 4 (whisper.cpp:1121-1869; real signatures Whisper/Utils/Trace/tracing.h:57-71).
// Instead of no-ops we CAPTURE the named tensors into a process-wide map so the parity tests can
// diff intermediates (enc.conv1, enc-Qcur, enc-KQV, encode-out, dec-KQV, probs ...).
#pragma once
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <map>
#include <string>
#include <vector>
struct ggml_tensor;
namespace Tracing
{
	struct ItemName
	{
		char text[ 96 ];
		ItemName( const char* s ) { snprintf( text, sizeof( text ), "%s", s ); }
		ItemName( const char* fmt, int a ) { snprintf( text, sizeof( text ), fmt, a ); }
	};
	struct Captured
	{
		int ne[ 4 ];
		std::vector<float> data;
	};
	extern bool g_enabled;
	extern std::map<std::string, Captured> g_captured;
	void captureTensor( const char* name, const ggml_tensor* t );
	void delayTensor( const ItemName& name, const ggml_tensor* t );
	void writeDelayedTensors();
	inline void tensor( const ItemName& name, const ggml_tensor* t ) { if( g_enabled ) captureTensor( name.text, t ); }
	inline void vector( const ItemName& name, const std::vector<float>& v )
	{
		if( !g_enabled ) return;
		Captured& c = g_captured[ name.text ];
		c.ne[ 0 ] = (int)v.size(); c.ne[ 1 ] = c.ne[ 2 ] = c.ne[ 3 ] = 1;
		c.data = v;
	}
}
