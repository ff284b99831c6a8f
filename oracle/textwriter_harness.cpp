// oracle/textwriter_harness.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product).
// The reference CLI's output writers -- Examples/main/textWriter.cpp: writeText / writeSubRip / writeWebVTT -- compiled UNMODIFIED by oracle/Makefile into
// oracle/_ref/libtextwriter_ref.so. tw_write() hands them an iContext whose getResults returns the given
// segments; tests/test_cli.py holds whisper-main's writers (whisper_amd/host/cli/subtitles.cpp) against the files they produce, byte for byte.
#include <string>
#include <vector>
#include <string.h>
#include "Examples/main/textWriter.h"

namespace
{
	using namespace Whisper;
	struct Result : iTranscribeResult
	{
		std::vector<sSegment> segs;
		HRESULT COMLIGHTCALL QueryInterface( REFIID, void** ) override { return E_NOINTERFACE; }
		uint32_t COMLIGHTCALL AddRef() override { return 2; }
		uint32_t COMLIGHTCALL Release() override { return 1; }
		HRESULT COMLIGHTCALL getSize( sTranscribeLength& rdi ) const override { rdi.countSegments = (uint32_t)segs.size(); rdi.countTokens = 0; return S_OK; }
		const sSegment* COMLIGHTCALL getSegments() const override { return segs.data(); }
		const sToken* COMLIGHTCALL getTokens() const override { return nullptr; }
	};
	struct Context : iContext
	{
		mutable Result res;
		HRESULT COMLIGHTCALL QueryInterface( REFIID, void** ) override { return E_NOINTERFACE; }
		uint32_t COMLIGHTCALL AddRef() override { return 2; }
		uint32_t COMLIGHTCALL Release() override { return 1; }
		HRESULT COMLIGHTCALL runFull( const sFullParams&, const iAudioBuffer* ) override { return E_NOTIMPL; }
		HRESULT COMLIGHTCALL runStreamed( const sFullParams&, const sProgressSink&, const iAudioReader* ) override { return E_NOTIMPL; }
		HRESULT COMLIGHTCALL runCapture( const sFullParams&, const sCaptureCallbacks&, const iAudioCapture* ) override { return E_NOTIMPL; }
		HRESULT COMLIGHTCALL getResults( eResultFlags, iTranscribeResult** pp ) const override { *pp = &res; return S_OK; }
		HRESULT COMLIGHTCALL detectSpeaker( const sTimeInterval&, eSpeakerChannel& ) const override { return E_NOTIMPL; }
		HRESULT COMLIGHTCALL getModel( iModel** ) override { return E_NOTIMPL; }
		HRESULT COMLIGHTCALL fullDefaultParams( eSamplingStrategy, sFullParams* ) override { return E_NOTIMPL; }
		HRESULT COMLIGHTCALL timingsPrint() override { return S_OK; }
		HRESULT COMLIGHTCALL timingsReset() override { return S_OK; }
	};
}

// kind: 0 = writeText with timestamps, 1 = writeText without, 2 = writeSubRip, 3 = writeWebVTT. audioPath (UTF-8): the writers replace its extension.
extern "C" __attribute__( ( visibility( "default" ) ) ) int tw_write( const char* audioPath, int kind, int n, const char* const* texts, const uint64_t* begin, const uint64_t* end )
{
	Context ctx;
	for( int i = 0; i < n; i++ )
	{
		sSegment s;
		memset( &s, 0, sizeof( s ) );
		s.text = texts[ i ];
		s.time.begin.ticks = begin[ i ];
		s.time.end.ticks = end[ i ];
		ctx.res.segs.push_back( s );
	}
	std::wstring w;
	for( const char* p = audioPath; *p; p++ ) w.push_back( (wchar_t)(unsigned char)*p );	   // the tests use ASCII paths
	switch( kind )
	{
	case 0: return (int)writeText( &ctx, w.c_str(), true );
	case 1: return (int)writeText( &ctx, w.c_str(), false );
	case 2: return (int)writeSubRip( &ctx, w.c_str() );
	default: return (int)writeWebVTT( &ctx, w.c_str() );
	}
}
