// tests/hostloop_cpu/batch_driver.cpp -- CPU test harness for the lock-step batch scheduler (not part of the product, never shipped).
// whisper_amd/host/batchScheduler.cpp (Whisper::createBatchRunner / iBatchRunner::run: K streams in lock step, slots refilled as streams
// finish, several groups served by one host thread) is compiled here UNCHANGED together with hostLoop.h / results.h / tokenTimestamps.cpp /
// support.cpp; the compute layer behind it is the test double of fake_device.cpp (the reference's CPU model, one per slot). What is tested
// is scheduling and per-stream semantics: the transcript of every stream must be the transcript of that stream run ALONE through the same
// host loop (driver.cpp hl_run) -- for any number of slots and groups, stream lengths, chunk sizes and look-ahead.
#include "hostLoop.h"
#include "results.h"
#include <cstdio>
#include <cstring>
#include <sstream>

extern "C" {
wh_model* fake_model_create( const char* ggmlPath, int threads );
void ref_hparams( void* ctx, int32_t* out11 );
void* ref_init( const char* path );
void ref_free( void* ctx );
}

namespace Whisper
{
	eHostLoopRules g_hostLoopRules = eHostLoopRules::ReferenceCpu;
	HRESULT createBatchRunner( iModel* model, const sBatchSetup* setup, iBatchRunner** pp );	   // batchScheduler.cpp
}
using namespace Whisper;

namespace
{
	// what createBatchRunner needs of an iModel of this library: the loaded model behind it (hostCommon.h iModelInternals)
	class TestModel : public iModel, public iModelInternals
	{
		std::shared_ptr<LoadedModel> lm;
	public:
		TestModel( const std::shared_ptr<LoadedModel>& m ) : lm( m ) {}
		virtual ~TestModel() = default;
		HRESULT QueryInterface( const ComLight::GUID& riid, void** ppv ) override
		{
			if( riid == iModelInternals::iid() ) { *ppv = static_cast<iModelInternals*>( this ); return S_OK; }
			if( riid == iModel::iid() ) { *ppv = static_cast<iModel*>( this ); return S_OK; }
			*ppv = nullptr;
			return E_NOINTERFACE;
		}
		uint32_t AddRef() override { return 2; }
		uint32_t Release() override { return 1; }
		HRESULT createContext( iContext** ) override { return E_NOTIMPL; }
		HRESULT tokenize( const char*, pfnDecodedTokens, void* ) override { return E_NOTIMPL; }
		HRESULT isMultilingual() override { return lm->vocab.isMultilingual() ? S_OK : S_FALSE; }
		HRESULT getSpecialTokens( SpecialTokens& ) override { return E_NOTIMPL; }
		const char* stringFromToken( whisper_token t ) override { return lm->vocab.string( t ); }
		HRESULT clone( iModel** ) override { return E_NOTIMPL; }
		const std::shared_ptr<LoadedModel>& loaded() const override { return lm; }
	};
	struct MemoryBuffer : iAudioBuffer
	{
		std::vector<float> pcm;
		int64_t time = 0;
		virtual ~MemoryBuffer() = default;
		HRESULT QueryInterface( const ComLight::GUID&, void** ) override { return E_NOINTERFACE; }
		uint32_t AddRef() override { return 2; }
		uint32_t Release() override { return 1; }
		uint32_t countSamples() const override { return (uint32_t)pcm.size(); }
		const float* getPcmMono() const override { return pcm.empty() ? nullptr : pcm.data(); }
		const float* getPcmStereo() const override { return nullptr; }
		HRESULT getTime( int64_t& rdi ) const override { rdi = time; return S_OK; }
	};
	int g_newSegments = 0;
	int g_callbackFaults = 0;
	// The reference's own callback idiom (Examples/main/main.cpp:55-58: `CComPtr<iTranscribeResult> r; ctx->getResults( flags, &r );`): results WITHOUT
	// eResultFlags::NewObject, released on scope exit. The object belongs to the context -- Release must not delete it, and the next getResults (and the
	// context's destructor) must find it alive. Called twice per callback, the second through the first's storage.
	HRESULT newSegment( iContext* ctx, uint32_t nNew, void* ) noexcept
	{
		g_newSegments += (int)nNew;
		uint32_t seen[ 2 ] = { 0, 0 };
		const iTranscribeResult* firstObject = nullptr;
		for( int k = 0; k < 2; k++ )
		{
			iTranscribeResult* r = nullptr;
			if( FAILED( ctx->getResults( eResultFlags::Timestamps | eResultFlags::Tokens, &r ) ) || !r ) { g_callbackFaults++; return S_OK; }
			sTranscribeLength len;
			r->getSize( len );
			seen[ k ] = len.countSegments;
			if( k == 0 ) firstObject = r; else if( r != firstObject ) g_callbackFaults++;
			r->Release();
		}
		if( seen[ 0 ] != seen[ 1 ] || seen[ 0 ] < nNew ) g_callbackFaults++;
		return S_OK;
	}
	std::string g_out;
	void jsonString( std::ostringstream& o, const char* s )
	{
		o << '"';
		for( const unsigned char* p = (const unsigned char*)( s ? s : "" ); *p; p++ )
		{
			if( *p == '"' || *p == '\\' ) o << '\\' << *p;
			else if( *p < 0x20 ) { char b[ 8 ]; snprintf( b, sizeof( b ), "\\u%04x", *p ); o << b; }
			else o << *p;
		}
		o << '"';
	}
}

struct BatchStreamDesc
{
	int32_t buffer;					  // index into the buffers of the call
	int64_t firstSample, countSamples;
};

// buffers: nBuffers recordings (pcm[b], nSamples[b]); streams: nStreams descriptors. flags / language / prompt etc. are common to all streams.
// bt_result() = {"hr":..,"streams":[{"hr":..,"segments":[{"t0","t1","text","tokens":[ids]}]}],"new_segments":N}
extern "C" __attribute__( ( visibility( "default" ) ) ) int bt_run( const char* modelPath, int rules, uint32_t flags, uint32_t language, int nMaxTextCtx,
	const int32_t* promptTokens, int nPrompt, const float* const* pcm, const int32_t* nSamples, int nBuffers, const BatchStreamDesc* streams, int nStreams,
	uint32_t maxSlots, uint32_t groups, uint32_t chunk, uint32_t lookahead, int threads )
{
	g_out.clear();
	g_newSegments = 0;
	g_callbackFaults = 0;
	g_hostLoopRules = (eHostLoopRules)rules;
	std::shared_ptr<LoadedModel> lm = std::make_shared<LoadedModel>();
	HRESULT hr = loadVocabulary( modelPath, lm->vocab );
	if( FAILED( hr ) ) return hr;
	{
		void* w = ref_init( modelPath );
		if( !w ) return E_FAIL;
		int32_t h[ 11 ];
		ref_hparams( w, h );
		ref_free( w );
		lm->hp = wh_hparams{ h[ 0 ], h[ 1 ], h[ 2 ], h[ 3 ], h[ 4 ], h[ 5 ], h[ 6 ], h[ 7 ], h[ 8 ], h[ 9 ], h[ 10 ] };
	}
	lm->gpu = fake_model_create( modelPath, threads );
	TestModel model( lm );

	std::vector<MemoryBuffer> buffers( (size_t)nBuffers );
	for( int b = 0; b < nBuffers; b++ ) buffers[ b ].pcm.assign( pcm[ b ], pcm[ b ] + nSamples[ b ] );
	std::vector<sBatchStream> descs( (size_t)nStreams );
	for( int i = 0; i < nStreams; i++ )
		descs[ i ] = sBatchStream{ streams[ i ].buffer >= 0 ? &buffers[ streams[ i ].buffer ] : nullptr, streams[ i ].firstSample, streams[ i ].countSamples, nullptr };

	sFullParams p{};
	p.strategy = eSamplingStrategy::Greedy;
	p.cpuThreads = threads;
	p.n_max_text_ctx = nMaxTextCtx >= 0 ? nMaxTextCtx : 16384;
	p.flags = (eFullParamsFlags)flags;
	p.language = language;
	p.thold_pt = p.thold_ptsum = 0.01f;
	p.prompt_tokens = promptTokens; p.prompt_n_tokens = nPrompt;
	p.new_segment_callback = &newSegment;

	const sBatchSetup setup{ maxSlots, groups, chunk, lookahead };
	iBatchRunner* runner = nullptr;
	hr = createBatchRunner( &model, &setup, &runner );
	if( FAILED( hr ) ) return hr;
	std::vector<iTranscribeResult*> results( (size_t)nStreams, nullptr );
	std::vector<HRESULT> per( (size_t)nStreams, S_OK );
	hr = runner->run( p, descs.data(), (uint32_t)nStreams, results.data(), per.data() );

	std::ostringstream o;
	o << "{\"hr\":" << hr << ",\"streams\":[";
	for( int i = 0; i < nStreams; i++ )
	{
		o << ( i ? "," : "" ) << "{\"hr\":" << per[ i ] << ",\"segments\":[";
		if( results[ i ] )
		{
			sTranscribeLength len{};
			results[ i ]->getSize( len );
			const sSegment* segs = results[ i ]->getSegments();
			const sToken* toks = results[ i ]->getTokens();
			for( uint32_t s = 0; s < len.countSegments; s++ )
			{
				o << ( s ? "," : "" ) << "{\"t0\":" << segs[ s ].time.begin.ticks << ",\"t1\":" << segs[ s ].time.end.ticks << ",\"text\":";
				jsonString( o, segs[ s ].text );
				o << ",\"tokens\":[";
				for( uint32_t j = 0; j < segs[ s ].countTokens; j++ ) o << ( j ? "," : "" ) << toks[ segs[ s ].firstToken + j ].id;
				o << "]}";
			}
			results[ i ]->Release();
		}
		o << "]}";
	}
	o << "],\"new_segments\":" << g_newSegments << ",\"callback_faults\":" << g_callbackFaults << "}";
	g_out = o.str();
	runner->Release();
	return hr;
}
extern "C" __attribute__( ( visibility( "default" ) ) ) const char* bt_result() { return g_out.c_str(); }
