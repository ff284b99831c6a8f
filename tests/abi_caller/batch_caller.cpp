// The C++ face of the lock-step batch runner, with callbacks (tests/test_batch_api.py::test_batch_caller_cpp). Compiled against
// include/whisperApi.h only. For the recordings given on the command line:
//   1. K sequential iContext::runFull calls, each with new_segment / encoder_begin callbacks (the reference's own calling sequence,
//      Examples/main/main.cpp:174-330); stream 1's encoder_begin callback stops it before its second window (S_FALSE);
//   2. ONE Whisper::runFullBatch over the same buffers with the same per-stream parameters.
// Exit code 0 (and BATCH_CALLER_OK) only when every stream's transcript, its callback counts and what its callbacks saw through
// iContext::getResults agree between the two.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include "whisperApi.h"

using namespace Whisper;

#define CHECK_HR( expr )                                                        \
	{                                                                           \
		const HRESULT hr__ = ( expr );                                          \
		if( FAILED( hr__ ) ) { fprintf( stderr, "%s failed: 0x%08x\n", #expr, (unsigned)hr__ ); return 10; } \
	}

struct Seen
{
	int index = 0;
	int segments = 0, encoderBegins = 0, stopAtWindow = -1;
	std::vector<uint32_t> visible;	   // segments visible through getResults at each new_segment callback
};

static HRESULT onSegment( iContext* ctx, uint32_t nNew, void* user ) noexcept
{
	Seen& s = *(Seen*)user;
	s.segments += (int)nNew;
	iTranscribeResult* r = nullptr;
	if( FAILED( ctx->getResults( eResultFlags::Timestamps | eResultFlags::Tokens | eResultFlags::NewObject, &r ) ) ) return E_FAIL;
	sTranscribeLength len;
	r->getSize( len );
	r->Release();
	// and the reference's own idiom (Examples/main/main.cpp:57-58): no NewObject, released on scope exit -- the context's embedded object
	iTranscribeResult* emb = nullptr;
	if( FAILED( ctx->getResults( eResultFlags::Timestamps | eResultFlags::Tokens, &emb ) ) || !emb ) return E_FAIL;
	sTranscribeLength len2;
	emb->getSize( len2 );
	emb->Release();
	if( len2.countSegments != len.countSegments ) return E_FAIL;
	s.visible.push_back( len.countSegments );
	iModel* m = nullptr;
	if( FAILED( ctx->getModel( &m ) ) || !m ) return E_FAIL;
	m->Release();
	return S_OK;
}
static HRESULT onEncoderBegin( iContext*, void* user ) noexcept
{
	Seen& s = *(Seen*)user;
	const int w = s.encoderBegins++;
	return w == s.stopAtWindow ? S_FALSE : S_OK;
}

static std::string transcript( iTranscribeResult* r )
{
	std::string out;
	sTranscribeLength len;
	r->getSize( len );
	const sSegment* segs = r->getSegments();
	const sToken* toks = r->getTokens();
	char buf[ 128 ];
	for( uint32_t i = 0; i < len.countSegments; i++ )
	{
		snprintf( buf, sizeof( buf ), "[%llu %llu]", (unsigned long long)segs[ i ].time.begin.ticks, (unsigned long long)segs[ i ].time.end.ticks );
		out += buf;
		for( uint32_t t = 0; t < segs[ i ].countTokens; t++ )
		{
			snprintf( buf, sizeof( buf ), " %d", toks[ segs[ i ].firstToken + t ].id );
			out += buf;
		}
		out += " |";
		out += segs[ i ].text ? segs[ i ].text : "";
		out += "\n";
	}
	return out;
}

int main( int argc, char** argv )
{
	if( argc < 4 ) { fprintf( stderr, "usage: %s <model.bin> <a.wav> <b.wav> [...]\n", argv[ 0 ] ); return 2; }
	sLoggerSetup ls;
	memset( &ls, 0, sizeof( ls ) );
	ls.flags = eLoggerFlags::UseStandardError;
	ls.level = eLogLevel::Warning;
	setupLogger( ls );
	std::wstring wpath;
	for( const char* p = argv[ 1 ]; *p; p++ ) wpath.push_back( (wchar_t)(unsigned char)*p );
	ComLight::CComPtr<iModel> model;
	sModelSetup setup;
	setup.impl = eModelImplementation::GPU;
	CHECK_HR( loadModel( wpath.c_str(), setup, nullptr, &model ) );
	ComLight::CComPtr<iMediaFoundation> mf;
	CHECK_HR( initMediaFoundation( &mf ) );
	const int K = argc - 2;
	std::vector<ComLight::CComPtr<iAudioBuffer>> buffers( K );
	for( int i = 0; i < K; i++ ) CHECK_HR( mf->loadAudioFile( argv[ 2 + i ], false, &buffers[ i ] ) );

	ComLight::CComPtr<iContext> context;
	CHECK_HR( model->createContext( &context ) );
	sFullParams base;
	CHECK_HR( context->fullDefaultParams( eSamplingStrategy::Greedy, &base ) );
	base.resetFlag( eFullParamsFlags::PrintRealtime | eFullParamsFlags::PrintProgress );
	base.setFlag( eFullParamsFlags::NoContext );
	const int prompt[ 1 ] = { 1000 };
	base.prompt_tokens = prompt;
	base.prompt_n_tokens = 1;
	base.n_max_text_ctx = 0;
	base.new_segment_callback = &onSegment;
	base.encoder_begin_callback = &onEncoderBegin;

	std::vector<Seen> seq( K ), bat( K );
	std::vector<sFullParams> params( K, base );
	std::vector<std::string> want( K );
	for( int i = 0; i < K; i++ )
	{
		seq[ i ].index = bat[ i ].index = i;
		seq[ i ].stopAtWindow = bat[ i ].stopAtWindow = ( i == 1 ) ? 1 : -1;
		params[ i ].new_segment_callback_user_data = params[ i ].encoder_begin_callback_user_data = &seq[ i ];
		CHECK_HR( context->runFull( params[ i ], buffers[ i ] ) );
		ComLight::CComPtr<iTranscribeResult> r;
		CHECK_HR( context->getResults( eResultFlags::Timestamps | eResultFlags::Tokens | eResultFlags::NewObject, &r ) );
		want[ i ] = transcript( r );
	}

	std::vector<sBatchStream> streams( K );
	for( int i = 0; i < K; i++ )
	{
		params[ i ].new_segment_callback_user_data = params[ i ].encoder_begin_callback_user_data = &bat[ i ];
		streams[ i ] = sBatchStream{ buffers[ i ], 0, 0, &params[ i ] };
	}
	std::vector<iTranscribeResult*> results( K, nullptr );
	std::vector<HRESULT> per( K, E_FAIL );
	sBatchSetup bs{ 2, 1, 3, 0 };	  // two slots for K streams: the third enters when the first one finishes
	CHECK_HR( runFullBatch( model, base, streams.data(), (uint32_t)K, &bs, results.data(), per.data() ) );
	int bad = 0;
	for( int i = 0; i < K; i++ )
	{
		const std::string got = results[ i ] ? transcript( results[ i ] ) : std::string( "<null>" );
		if( results[ i ] ) results[ i ]->Release();
		const bool same = got == want[ i ] && seq[ i ].segments == bat[ i ].segments && seq[ i ].encoderBegins == bat[ i ].encoderBegins &&
			seq[ i ].visible == bat[ i ].visible && per[ i ] == S_OK;
		printf( "stream %d: %s; segments via callback %d / %d, encoder_begin calls %d / %d, hr 0x%08x\n%s", i, same ? "same" : "DIFFERENT", seq[ i ].segments,
			bat[ i ].segments, seq[ i ].encoderBegins, bat[ i ].encoderBegins, (unsigned)per[ i ], got.c_str() );
		if( !same )
		{
			printf( "--- sequential runFull gave:\n%s", want[ i ].c_str() );
			bad++;
		}
	}
	if( seq[ 1 ].encoderBegins != 2 ) { printf( "stream 1 was to be stopped at its second window\n" ); bad++; }
	if( bad ) return 1;
	printf( "BATCH_CALLER_OK\n" );
	return 0;
}
