// oracle/contextimpl_harness.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product).
// The reference's GPU-model iContext -- Whisper/Whisper/ContextImpl.cpp (sampleBest :71-157, token-level timestamps :219-419, the
// host loop runFullImpl :452-793) and ContextImpl.misc.cpp (runFull, runStreamed, makeResults / getResults, wrapSegment,
// fullDefaultParams), with Languages.cpp, Spectrogram.cpp, MelStreamer.cpp, melSpectrogram.cpp -- compiled UNMODIFIED by oracle/Makefile
// into oracle/_ref/libcontextimpl_ref.so, with ONE substitution: the D3D11 compute context (DirectCompute::WhisperContext) is the
// reference's own CPU model, Whisper/source/whisper.cpp, through the flat entry points of oracle/_ref/libwhisper_ref.so
// (oracle/ref_harness.cpp). What comes out is the transcript the reference's GPU model's HOST CODE produces from the reference's CPU
// arithmetic: the oracle for SURVEY.md section 8 rows a10 / a11 / f1 / f4 under the GPU model's rules -- its host loop differs from
// whisper_full's in two rules (no dropped prompt near the end, no retry of a failed window; whisper_amd/host/hostLoop.h).
//
// Defined here: WhisperContext::encode / decode (the substitution), ContextImpl's two methods that live in files this build leaves
// out (runCapture: ContextImpl.capture.cpp, audio devices; detectSpeaker: ContextImpl.diarize.cpp), the logger, and the flat C API.
#include "stdafx.h"
#include "Whisper/ContextImpl.h"
#include "Whisper/Languages.h"
#include "Whisper/MelStreamer.h"
#include "API/iMediaFoundation.cl.h"
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <string>

using namespace Whisper;

// ---- oracle/ref_harness.cpp (libwhisper_ref.so) ----
extern "C" {
void* ref_init( const char* path );
void ref_free( void* ctx );
void ref_set_log_level( int lvl );
void ref_hparams( void* ctx, int32_t* out11 );
int ref_set_mel( void* ctx, const float* mel, int nLen, int nMel );
int ref_encode( void* ctx, int melOffset, int nThreads );
int ref_decode( void* ctx, const int32_t* tokens, int nTokens, int nPast, int nThreads );
size_t ref_logits_size( void* ctx );
void ref_get_probs( void* ctx, float* dst );
int ref_token_special( void* ctx, int which );
const char* ref_token_to_str( void* ctx, int token );
}

// ---- the substitution: MelInputTensor::create (MelInputTensor.cpp:8-63: frames [ mel_offset, mel_offset + 2 n_ctx ) clamped to the
// spectrogram's length, the rest zero) + whisper_encode; whisper_decode + its probabilities of the last token ----
DirectCompute::EncoderOutput DirectCompute::WhisperContext::encode( iSpectrogram& mel, const sEncodeParams& ep )
{
	const size_t ne0 = (size_t)ep.n_ctx * 2;
	std::vector<float> window( ne0 * ep.n_mels, 0.0f );
	const size_t nLen = mel.getLength();
	const size_t i0 = std::min( (size_t)ep.mel_offset, nLen );
	const size_t i1 = std::min( (size_t)ep.mel_offset + ne0, nLen );
	const float* src = nullptr;
	size_t stride = 0;
	check( mel.makeBuffer( i0, i1 - i0, &src, stride ) );
	for( size_t j = 0; j < ep.n_mels; j++ ) memcpy( window.data() + j * ne0, src + j * stride, ( i1 - i0 ) * 4 );
	if( 0 != ref_set_mel( model.cpuModel, window.data(), (int)ne0, (int)ep.n_mels ) ) throw E_FAIL;
	if( 0 != ref_encode( model.cpuModel, 0, model.cpuThreads ) ) throw E_FAIL;
	return EncoderOutput{};
}
void DirectCompute::WhisperContext::decode( const int* tokens, int length, const sDecodeParams& dp, std::vector<float>& probs, int threads )
{
	if( 0 != ref_decode( model.cpuModel, tokens, length, (int)dp.n_past, std::max( 1, threads ) ) ) throw E_FAIL;
	// ContextImpl reads the LAST n_vocab entries (ContextImpl.cpp:161-169)
	std::vector<float> all( ref_logits_size( model.cpuModel ) );
	ref_get_probs( model.cpuModel, all.data() );
	probs.assign( all.end() - dp.n_vocab, all.end() );
	if( getenv( "CI_DEBUG" ) )
	{
		int best = 0;
		for( int i = 1; i < (int)dp.n_vocab; i++ ) if( probs[ i ] > probs[ best ] ) best = i;
		fprintf( stderr, "decode n=%d n_past=%d first=%d last=%d -> argmax %d p=%g\n", length, (int)dp.n_past, tokens[ 0 ], tokens[ length - 1 ], best, probs[ best ] );
	}
}

HRESULT COMLIGHTCALL ContextImpl::runCapture( const sFullParams&, const sCaptureCallbacks&, const iAudioCapture* ) { return E_NOTIMPL; }
HRESULT COMLIGHTCALL ContextImpl::detectSpeaker( const sTimeInterval&, eSpeakerChannel& result ) const noexcept { result = (eSpeakerChannel)0; return S_FALSE; }

// ---- PcmReader over memory for runStreamed, the thread pool, setCurrentThreadName: shared with the streamer's harness ----
#include "memory_reader.inl"

static int g_logLevel = 0;		// 0: errors are kept for ci_last_error only; 1: errors and warnings to stderr
static std::string g_lastError;
static void vlog( const char* level, const char8_t* fmt, va_list ap, bool keep )
{
	char buf[ 1024 ];
	vsnprintf( buf, sizeof( buf ), (const char*)fmt, ap );
	if( keep ) g_lastError = buf;
	if( g_logLevel > 0 ) fprintf( stderr, "[contextimpl_ref %s] %s\n", level, buf );
}
extern "C" {
void logError( const char8_t* fmt, ... ) { va_list ap; va_start( ap, fmt ); vlog( "error", fmt, ap, true ); va_end( ap ); }
void logErrorHr( long, const char8_t* fmt, ... ) { va_list ap; va_start( ap, fmt ); vlog( "error", fmt, ap, true ); va_end( ap ); }
void logWarning( const char8_t* fmt, ... ) { va_list ap; va_start( ap, fmt ); vlog( "warning", fmt, ap, false ); va_end( ap ); }
void logWarningHr( long, const char8_t* fmt, ... ) { va_list ap; va_start( ap, fmt ); vlog( "warning", fmt, ap, false ); va_end( ap ); }
void logInfo( const char8_t*, ... ) {}
void logDebug( const char8_t*, ... ) {}
}

namespace
{
	struct MemoryBuffer : iAudioBuffer
	{
		std::vector<float> pcm;
		int64_t time = 0;
		virtual ~MemoryBuffer() {}
		HRESULT COMLIGHTCALL QueryInterface( REFIID, void** ) override { return E_NOINTERFACE; }
		uint32_t COMLIGHTCALL AddRef() override { return 1; }
		uint32_t COMLIGHTCALL Release() override { return 1; }
		uint32_t COMLIGHTCALL countSamples() const override { return (uint32_t)pcm.size(); }
		const float* COMLIGHTCALL getPcmMono() const override { return pcm.data(); }
		const float* COMLIGHTCALL getPcmStereo() const override { return nullptr; }
		HRESULT COMLIGHTCALL getTime( int64_t& rdi ) const override { rdi = time; return S_OK; }
	};
	struct MemoryReader : iAudioReader
	{
		IMFSourceReader* const source;
		MemoryReader( const float* pcm, size_t n ) : source( new IMFSourceReader() ) { source->pcm = pcm; source->count = n; }
		virtual ~MemoryReader() { source->Release(); }
		HRESULT COMLIGHTCALL QueryInterface( REFIID, void** ) override { return E_NOINTERFACE; }
		uint32_t COMLIGHTCALL AddRef() override { return 1; }
		uint32_t COMLIGHTCALL Release() override { return 1; }
		HRESULT COMLIGHTCALL getDuration( int64_t& rdi ) const override { rdi = (int64_t)source->count * 10000000 / SAMPLE_RATE; return S_OK; }
		HRESULT COMLIGHTCALL getReader( IMFSourceReader** pp ) const override { source->AddRef(); *pp = source; return S_OK; }
		HRESULT COMLIGHTCALL requestedStereo() const override { return S_FALSE; }
	};

	struct Instance
	{
		DirectCompute::Device device;
		WhisperModel model;
		ComLight::CComPtr<ComLight::Object<ContextImpl>> context;
		ComLight::CComPtr<iTranscribeResult> result;
		std::vector<double> progress;
		int newSegmentCalls = 0, newSegments = 0;
		~Instance()
		{
			result = nullptr;
			context = nullptr;
			if( model.cpuModel ) ref_free( model.cpuModel );
		}
	};

	HRESULT __stdcall progressSink( double val, iContext*, void* pv ) noexcept
	{
		( (Instance*)pv )->progress.push_back( val );
		return S_OK;
	}
	HRESULT __cdecl newSegment( iContext*, uint32_t nNew, void* pv ) noexcept
	{
		Instance* i = (Instance*)pv;
		i->newSegmentCalls++;
		i->newSegments += (int)nNew;
		return S_OK;
	}
}

#define CI_API extern "C" __attribute__( ( visibility( "default" ) ) )

// What a test sets of sFullParams; everything else is fullDefaultParams' (ContextImpl.misc.cpp:60-93)
struct CiParams
{
	uint32_t flags;			 // eFullParamsFlags
	uint32_t language;		 // makeLanguageKey
	int32_t cpuThreads, n_max_text_ctx, offset_ms, duration_ms, max_tokens, max_len;
	float thold_pt, thold_ptsum;
	const int32_t* prompt_tokens;
	int32_t prompt_n_tokens;
	int64_t mediaTime;		 // iAudioBuffer::getTime of runFull's buffer
};

CI_API void ci_set_log_level( int lvl ) { g_logLevel = lvl; ref_set_log_level( lvl ); }
CI_API const char* ci_last_error() { return g_lastError.c_str(); }

CI_API void* ci_create( const char* modelPath, const float* filters, int nMel, int nFft, int encoderThreads )
{
	try
	{
		std::unique_ptr<Instance> inst( new Instance() );
		WhisperModel& m = inst->model;
		m.cpuModel = ref_init( modelPath );
		if( !m.cpuModel ) return nullptr;
		m.cpuThreads = std::max( 1, encoderThreads );
		int32_t hp[ 11 ];
		ref_hparams( m.cpuModel, hp );	   // n_vocab, n_audio_ctx, n_audio_state, n_audio_head, n_audio_layer, n_text_ctx, n_text_state, n_text_head, n_text_layer, n_mels, f16
		sModelParams& p = m.parameters;
		p.n_vocab = hp[ 0 ]; p.n_audio_ctx = hp[ 1 ]; p.n_audio_state = hp[ 2 ]; p.n_audio_head = hp[ 3 ]; p.n_audio_layer = hp[ 4 ];
		p.n_text_ctx = hp[ 5 ]; p.n_text_state = hp[ 6 ]; p.n_text_head = hp[ 7 ]; p.n_text_layer = hp[ 8 ]; p.n_mels = hp[ 9 ]; p.f16 = hp[ 10 ];
		m.shared = std::make_shared<ModelShared>();
		Vocabulary& v = m.shared->vocab;
		v.n_vocab = p.n_vocab;
		v.token_eot = ref_token_special( m.cpuModel, 0 ); v.token_sot = ref_token_special( m.cpuModel, 1 ); v.token_prev = ref_token_special( m.cpuModel, 2 );
		v.token_solm = ref_token_special( m.cpuModel, 3 ); v.token_not = ref_token_special( m.cpuModel, 4 ); v.token_beg = ref_token_special( m.cpuModel, 5 );
		v.table.resize( (size_t)p.n_vocab );
		for( int i = 0; i < p.n_vocab; i++ )
		{
			const char* s = ref_token_to_str( m.cpuModel, i );
			v.table[ i ] = s ? s : "";
		}
		m.shared->filters.n_mel = (uint32_t)nMel;
		m.shared->filters.n_fft = (uint32_t)nFft;
		m.shared->filters.data.assign( filters, filters + (size_t)nMel * nFft );
		check( ComLight::Object<ContextImpl>::create( inst->context, inst->device, inst->model, nullptr ) );
		return inst.release();
	}
	catch( ... )
	{
		return nullptr;
	}
}
CI_API void ci_destroy( void* h ) { delete (Instance*)h; }

static HRESULT makeParams( Instance& inst, const CiParams& c, sFullParams& p )
{
	iContext* ctx = inst.context;
	CHECK( ctx->fullDefaultParams( eSamplingStrategy::Greedy, &p ) );
	p.flags = (eFullParamsFlags)c.flags;
	p.language = c.language;
	if( c.cpuThreads > 0 ) p.cpuThreads = c.cpuThreads;
	if( c.n_max_text_ctx >= 0 ) p.n_max_text_ctx = c.n_max_text_ctx;
	p.offset_ms = c.offset_ms; p.duration_ms = c.duration_ms; p.max_tokens = c.max_tokens; p.max_len = c.max_len;
	if( c.thold_pt >= 0 ) p.thold_pt = c.thold_pt;
	if( c.thold_ptsum >= 0 ) p.thold_ptsum = c.thold_ptsum;
	p.prompt_tokens = c.prompt_tokens; p.prompt_n_tokens = c.prompt_n_tokens;
	p.new_segment_callback = &newSegment; p.new_segment_callback_user_data = &inst;
	inst.progress.clear(); inst.newSegmentCalls = inst.newSegments = 0;
	inst.result = nullptr;
	return S_OK;
}
static HRESULT fetch( Instance& inst )
{
	iContext* ctx = inst.context;
	// without eResultFlags::NewObject: the context's own result object, strings left in place. (With it makeResults MOVES the segments' strings
	// into the new object, ContextImpl.misc.cpp:244-248: a second getResults( NewObject ) finds empty texts.)
	return ctx->getResults( (eResultFlags)( (uint32_t)eResultFlags::Timestamps | (uint32_t)eResultFlags::Tokens ), &inst.result );
}
CI_API int ci_run_full( void* h, const CiParams* c, const float* pcm, int nSamples )
{
	Instance& inst = *(Instance*)h;
	sFullParams p;
	HRESULT hr = makeParams( inst, *c, p );
	if( FAILED( hr ) ) return hr;
	MemoryBuffer buffer;
	buffer.pcm.assign( pcm, pcm + nSamples );
	buffer.time = c->mediaTime;
	iContext* ctx = inst.context;
	hr = ctx->runFull( p, &buffer );
	if( FAILED( hr ) ) return hr;
	const HRESULT hr2 = fetch( inst );
	return FAILED( hr2 ) ? hr2 : hr;
}
CI_API int ci_run_streamed( void* h, const CiParams* c, const float* pcm, int nSamples )
{
	Instance& inst = *(Instance*)h;
	sFullParams p;
	HRESULT hr = makeParams( inst, *c, p );
	if( FAILED( hr ) ) return hr;
	std::vector<float> copy( pcm, pcm + nSamples );
	MemoryReader reader( copy.data(), copy.size() );
	const sProgressSink sink{ &progressSink, &inst };
	iContext* ctx = inst.context;
	hr = ctx->runStreamed( p, sink, &reader );
	if( FAILED( hr ) ) return hr;
	const HRESULT hr2 = fetch( inst );
	return FAILED( hr2 ) ? hr2 : hr;
}
// getResults again with other eResultFlags (Tokens = 1, Timestamps = 2): the context's own result object, refilled
CI_API int ci_get_results( void* h, uint32_t flags )
{
	Instance& inst = *(Instance*)h;
	inst.result = nullptr;
	iContext* ctx = inst.context;
	return ctx->getResults( (eResultFlags)flags, &inst.result );
}
CI_API int ci_counts( void* h, int32_t* out4 )
{
	Instance& inst = *(Instance*)h;
	sTranscribeLength len{ 0, 0 };
	if( inst.result ) inst.result->getSize( len );
	out4[ 0 ] = (int32_t)len.countSegments; out4[ 1 ] = (int32_t)len.countTokens; out4[ 2 ] = inst.newSegmentCalls; out4[ 3 ] = inst.newSegments;
	return (int)inst.progress.size();
}
CI_API void ci_progress( void* h, double* dst ) { Instance& inst = *(Instance*)h; std::copy( inst.progress.begin(), inst.progress.end(), dst ); }
CI_API const char* ci_segment( void* h, int i, int64_t* times2, uint32_t* tokens2 )
{
	const sSegment& s = ( (Instance*)h )->result->getSegments()[ i ];
	times2[ 0 ] = (int64_t)s.time.begin.ticks; times2[ 1 ] = (int64_t)s.time.end.ticks;
	tokens2[ 0 ] = s.firstToken; tokens2[ 1 ] = s.countTokens;
	return s.text;
}
CI_API const char* ci_token( void* h, int i, int64_t* times2, float* probs4, int32_t* idFlags2 )
{
	const sToken& t = ( (Instance*)h )->result->getTokens()[ i ];
	times2[ 0 ] = (int64_t)t.time.begin.ticks; times2[ 1 ] = (int64_t)t.time.end.ticks;
	probs4[ 0 ] = t.probability; probs4[ 1 ] = t.probabilityTimestamp; probs4[ 2 ] = t.ptsum; probs4[ 3 ] = t.vlen;
	idFlags2[ 0 ] = t.id; idFlags2[ 1 ] = (int32_t)t.flags;
	return t.text;
}
CI_API int ci_language_id( const char* code ) { return lookupLanguageId( code ); }
