// A caller of libWhisper.so built TWICE from this one source (tests/test_abi_reference_headers.py):
//   -DUSE_REFERENCE_HEADERS  against the reference's OWN public headers (Whisper/API/whisperComLight.h + ComLightLib/comLightClient.h,
//                            found through -I/root/reference, never copied), the way Examples/main/main.cpp:174-330 is built;
//   (default)                against include/whisperApi.h.
// Both binaries link the same libWhisper.so. Mode `layout` prints sizeof / offsetof of every POD structure that crosses the
// boundary and the interface ids; mode `run` drives loadModel -> createContext -> fullDefaultParams -> loadAudioFile -> runFull
// -> getResults and prints the transcript. The test asserts that the two binaries print the same bytes: that is the
// "compiles unchanged apart from the include line" claim of INTEGRATION.md, executed.
#include <cstring>
#include <cstdint>
#include <cstddef>
#include <cstdio>
#include <string>
#include <vector>
#include <climits>
#include <cstdlib>
#ifdef USE_REFERENCE_HEADERS
#include "Whisper/API/whisperComLight.h"
#include "ComLightLib/comLightClient.h"
#else
#include "whisperApi.h"
#endif

using namespace Whisper;

static std::wstring widen( const char* s )
{
	std::wstring w;
	for( ; *s; s++ ) w.push_back( (wchar_t)(unsigned char)*s );
	return w;
}

#define FIELD( S, f ) printf( "  \"%s.%s\": [%zu, %zu],\n", #S, #f, offsetof( S, f ), sizeof( ( (S*)nullptr )->f ) )
#define SIZE( S ) printf( "  \"sizeof %s\": %zu,\n", #S, sizeof( S ) )

template<class I>
static void printIid( const char* name )
{
#ifdef USE_REFERENCE_HEADERS
	const GUID& g = I::iid();
#else
	const ComLight::GUID& g = I::iid();
#endif
	const uint8_t* b = (const uint8_t*)&g;
	printf( "  \"iid %s\": \"", name );
	for( int i = 0; i < 16; i++ ) printf( "%02x", b[ i ] );
	printf( "\",\n" );
}

static int layout()
{
	printf( "{\n" );
	SIZE( sFullParams );
	FIELD( sFullParams, strategy ); FIELD( sFullParams, cpuThreads ); FIELD( sFullParams, n_max_text_ctx ); FIELD( sFullParams, offset_ms );
	FIELD( sFullParams, duration_ms ); FIELD( sFullParams, flags ); FIELD( sFullParams, language ); FIELD( sFullParams, thold_pt );
	FIELD( sFullParams, thold_ptsum ); FIELD( sFullParams, max_len ); FIELD( sFullParams, max_tokens ); FIELD( sFullParams, greedy );
	FIELD( sFullParams, beam_search ); FIELD( sFullParams, audio_ctx ); FIELD( sFullParams, prompt_tokens ); FIELD( sFullParams, prompt_n_tokens );
	FIELD( sFullParams, new_segment_callback ); FIELD( sFullParams, new_segment_callback_user_data );
	FIELD( sFullParams, encoder_begin_callback ); FIELD( sFullParams, encoder_begin_callback_user_data );
	SIZE( sSegment ); FIELD( sSegment, text ); FIELD( sSegment, time ); FIELD( sSegment, firstToken ); FIELD( sSegment, countTokens );
	SIZE( sToken ); FIELD( sToken, text ); FIELD( sToken, time ); FIELD( sToken, probability ); FIELD( sToken, probabilityTimestamp );
	FIELD( sToken, ptsum ); FIELD( sToken, vlen ); FIELD( sToken, id ); FIELD( sToken, flags );
	SIZE( sTimeInterval ); SIZE( sTimeSpan ); SIZE( sTranscribeLength );
	SIZE( sModelSetup ); FIELD( sModelSetup, impl ); FIELD( sModelSetup, flags ); FIELD( sModelSetup, adapter );
	SIZE( SpecialTokens ); SIZE( sLoadModelCallbacks ); SIZE( sProgressSink ); SIZE( sLoggerSetup ); SIZE( sLanguageList ); SIZE( sLanguageEntry );
	printIid<iModel>( "iModel" ); printIid<iContext>( "iContext" ); printIid<iTranscribeResult>( "iTranscribeResult" );
	printIid<iAudioBuffer>( "iAudioBuffer" ); printIid<iAudioReader>( "iAudioReader" ); printIid<iMediaFoundation>( "iMediaFoundation" );
	printf( "  \"S_FALSE\": %d, \"E_NOTIMPL\": %u, \"E_INVALIDARG\": %u\n}\n", (int)S_FALSE, (unsigned)E_NOTIMPL, (unsigned)E_INVALIDARG );
	return 0;
}

static int segmentsSeen = 0;
static HRESULT onSegment( iContext* ctx, uint32_t nNew, void* user ) noexcept
{
	segmentsSeen += (int)nNew;
	return S_OK;
}

#define CHECK_HR( expr )                                                        \
	{                                                                           \
		const HRESULT hr__ = ( expr );                                          \
		if( FAILED( hr__ ) ) { fprintf( stderr, "%s failed: 0x%08x\n", #expr, (unsigned)hr__ ); return 10; } \
	}

static int run( const char* modelPath, const char* wavPath, const char* lang, const char* promptCsv, int nMaxTextCtx )
{
	sLoggerSetup ls;
	memset( &ls, 0, sizeof( ls ) );
	ls.flags = eLoggerFlags::UseStandardError;
	ls.level = eLogLevel::Warning;
	setupLogger( ls );
	if( findLanguageKeyA( lang ) == UINT_MAX ) { fprintf( stderr, "unknown language\n" ); return 3; }

	ComLight::CComPtr<iModel> model;
	sModelSetup setup;
	setup.impl = eModelImplementation::GPU;
	CHECK_HR( loadModel( widen( modelPath ).c_str(), setup, nullptr, &model ) );
	SpecialTokens st;
	CHECK_HR( model->getSpecialTokens( st ) );
	ComLight::CComPtr<iContext> context;
	CHECK_HR( model->createContext( &context ) );
	ComLight::CComPtr<iMediaFoundation> mf;
	CHECK_HR( initMediaFoundation( &mf ) );

	sFullParams wparams;
	CHECK_HR( context->fullDefaultParams( eSamplingStrategy::Greedy, &wparams ) );
	wparams.resetFlag( eFullParamsFlags::PrintRealtime | eFullParamsFlags::PrintProgress );
	wparams.setFlag( eFullParamsFlags::NoContext );
	wparams.language = makeLanguageKey( lang );
	wparams.new_segment_callback = &onSegment;
	std::vector<int> prompt;
	for( const char* p = promptCsv; p && *p; )
	{
		prompt.push_back( atoi( p ) );
		p = strchr( p, ',' );
		if( p ) p++;
	}
	if( !prompt.empty() )
	{
		wparams.prompt_tokens = prompt.data();
		wparams.prompt_n_tokens = (int)prompt.size();
	}
	if( nMaxTextCtx >= 0 ) wparams.n_max_text_ctx = nMaxTextCtx;

	ComLight::CComPtr<iAudioBuffer> buffer;
#ifdef USE_REFERENCE_HEADERS
	CHECK_HR( mf->loadAudioFile( wavPath, false, &buffer ) );	  // LPCTSTR = const char* off Windows (ComLightLib/comLightCommon.h:8)
#else
	CHECK_HR( mf->loadAudioFile( wavPath, false, &buffer ) );
#endif
	CHECK_HR( context->runFull( wparams, buffer ) );

	ComLight::CComPtr<iTranscribeResult> result;
	CHECK_HR( context->getResults( eResultFlags::Timestamps | eResultFlags::Tokens, &result ) );
	sTranscribeLength len;
	CHECK_HR( result->getSize( len ) );
	const sSegment* const segs = result->getSegments();
	const sToken* const toks = result->getTokens();
	printf( "{\"eot\": %d, \"multilingual\": %d, \"callback_segments\": %d, \"segments\": [", st.TranscriptionEnd, model->isMultilingual() == S_OK ? 1 : 0, segmentsSeen );
	for( uint32_t i = 0; i < len.countSegments; i++ )
	{
		const sSegment& s = segs[ i ];
		printf( "%s{\"begin\": %llu, \"end\": %llu, \"ids\": [", i ? ", " : "", (unsigned long long)s.time.begin.ticks, (unsigned long long)s.time.end.ticks );
		for( uint32_t t = 0; t < s.countTokens; t++ ) printf( "%s%d", t ? ", " : "", toks[ s.firstToken + t ].id );
		printf( "]}" );
	}
	printf( "], \"countTokens\": %u}\n", len.countTokens );
	return 0;
}

int main( int argc, char** argv )
{
	if( argc >= 2 && 0 == strcmp( argv[ 1 ], "layout" ) ) return layout();
	if( argc >= 4 && 0 == strcmp( argv[ 1 ], "run" ) )
		return run( argv[ 2 ], argv[ 3 ], argc >= 5 ? argv[ 4 ] : "en", argc >= 6 ? argv[ 5 ] : "", argc >= 7 ? atoi( argv[ 6 ] ) : -1 );
	fprintf( stderr, "usage: %s layout | run <model.bin> <clip.wav> [lang [prompt,ids [n_max_text_ctx]]]\n", argv[ 0 ] );
	return 2;
}
