// tests/hostloop_cpu/driver.cpp -- a CPU test harness for the product's HOST LOGIC (not part of the product, never shipped).
// libWhisper.so's host loop -- whisper_amd/host/hostLoop.h: StreamRun (seek range, prompt carry-over, failure handling, segment cutting,
// callbacks) and WindowScan (timestamp tracking, stop rules) -- with whisper_amd/host/tokenTimestamps.cpp and support.cpp (vocabulary,
// languages) is plain C++; what it needs from a device is one thing: the tokens of a window. Here those come from the reference's own
// CPU model (oracle/_ref/libwhisper_ref.so: whisper_encode / whisper_decode / whisper_sample_best of Whisper/source/whisper.cpp) instead
// of libwhisper_hip.so, so the SAME source files that run behind iContext::runFull are exercised without a GPU, under both rule sets:
//   rules 0 = whisper_full's (the oracle: tests/golden/ref_hostloop.json), rules 1 = ContextImpl::runFullImpl's (the reference's GPU
//   model: tests/golden/ref_hostloop_contextimpl.json, from the reference's ContextImpl.cpp compiled unmodified).
// Built by tests/test_hostloop_cpu.py into tests/_build/ (g++, a few seconds); the window loop below is the shape of
// ContextImpl::runFullImpl / whisperImpl.cpp's runFullImpl: nextWindow -> encode -> decode + sample until WindowScan says stop -> finishWindow.
#include "hostLoop.h"
#include "results.h"
#include <cstdio>
#include <cstring>
#include <sstream>

extern "C" {
int ref_token_special( void* ctx, int which );
int ref_tokenize( void* ctx, const char* text, int32_t* tokens, int cap );
const char* ref_token_to_str( void* ctx, int token );
void* ref_init( const char* path );
void ref_free( void* ctx );
void ref_set_log_level( int lvl );
void ref_hparams( void* ctx, int32_t* out11 );
int ref_pcm_to_mel( void* ctx, const float* pcm, int n, int nThreads );
int ref_mel_len( void* ctx );
int ref_set_mel( void* ctx, const float* mel, int nLen, int nMel );
int ref_encode( void* ctx, int melOffset, int nThreads );
int ref_decode( void* ctx, const int32_t* tokens, int nTokens, int nPast, int nThreads );
void ref_sample_best( void* ctx, int32_t* id, int32_t* tid, float* p, float* pt, float* ptsum );
void ref_sample_timestamp( void* ctx, int isInitial, int32_t* id, int32_t* tid, float* p, float* pt, float* ptsum );
}

namespace Whisper
{
	eHostLoopRules g_hostLoopRules = eHostLoopRules::ReferenceCpu;
}
using namespace Whisper;

namespace
{
	struct Sinks
	{
		std::vector<double> progress;
		int newSegmentCalls = 0, newSegments = 0;
	};
	HRESULT progressSink( double val, iContext*, void* pv ) noexcept
	{
		( (Sinks*)pv )->progress.push_back( val );
		return S_OK;
	}
	HRESULT newSegment( iContext*, uint32_t nNew, void* pv ) noexcept
	{
		Sinks* s = (Sinks*)pv;
		s->newSegmentCalls++;
		s->newSegments += (int)nNew;
		return S_OK;
	}
	std::string g_out;
	void jsonString( std::ostringstream& o, const std::string& s )
	{
		o << '"';
		for( unsigned char c : s )
		{
			if( c == '"' || c == '\\' ) o << '\\' << c;
			else if( c < 0x20 ) { char b[ 8 ]; snprintf( b, sizeof( b ), "\\u%04x", c ); o << b; }
			else o << c;
		}
		o << '"';
	}
}

struct HlParams
{
	uint32_t flags, language;
	int32_t n_max_text_ctx, offset_ms, duration_ms, max_tokens, max_len;
	float thold_pt, thold_ptsum;
	const int32_t* prompt_tokens;
	int32_t prompt_n_tokens;
	int32_t withProgress;
	uint32_t resultFlags;	 // eResultFlags of the "pods" section (iContext::getResults through results.h fillResultData)
	int64_t mediaTime;		 // iAudioBuffer::getTime
	const float* mel;		 // optional: the whole-buffer spectrogram [80][melLen] to run on (else whisper.cpp's own of the PCM)
	int32_t melLen;
};

// Returns the HRESULT of the run; hl_result() = {"pods":[...] (see below),"segments":[{"t0","t1","text","tokens":[{"id","tid","p","pt","ptsum","t0","t1","vlen"}]}],"progress":[..],"new_segment":[calls,sum]}
extern "C" __attribute__( ( visibility( "default" ) ) ) int hl_run( const char* modelPath, int rules, const HlParams* hp, const float* pcm, int nSamples, int threads )
{
	g_out.clear();
	g_hostLoopRules = (eHostLoopRules)rules;
	Vocabulary vocab;
	HRESULT hr = loadVocabulary( modelPath, vocab );
	if( FAILED( hr ) ) return hr;
	ref_set_log_level( 0 );
	void* cpu = ref_init( modelPath );
	if( !cpu ) return E_FAIL;
	int32_t h[ 11 ];
	ref_hparams( cpu, h );
	wh_hparams hparams{ h[ 0 ], h[ 1 ], h[ 2 ], h[ 3 ], h[ 4 ], h[ 5 ], h[ 6 ], h[ 7 ], h[ 8 ], h[ 9 ], h[ 10 ] };

	Sinks sinks;
	sFullParams p{};
	p.strategy = eSamplingStrategy::Greedy;
	p.cpuThreads = threads;
	p.n_max_text_ctx = hp->n_max_text_ctx >= 0 ? hp->n_max_text_ctx : 16384;
	p.offset_ms = hp->offset_ms; p.duration_ms = hp->duration_ms;
	p.flags = (eFullParamsFlags)hp->flags;
	p.language = hp->language;
	p.thold_pt = hp->thold_pt >= 0 ? hp->thold_pt : 0.01f;
	p.thold_ptsum = hp->thold_ptsum >= 0 ? hp->thold_ptsum : 0.01f;
	p.max_len = hp->max_len; p.max_tokens = hp->max_tokens;
	p.prompt_tokens = hp->prompt_tokens; p.prompt_n_tokens = hp->prompt_n_tokens;
	p.new_segment_callback = &newSegment; p.new_segment_callback_user_data = &sinks;
	const sProgressSink sink{ hp->withProgress ? &progressSink : nullptr, &sinks };

	std::vector<Segment> resultAll;
	std::vector<int> promptPast;
	TokenTimestamper stamper;
	if( p.flag( eFullParamsFlags::TokenTimestamps ) ) stamper.begin( pcm, (size_t)nSamples );
	StreamRun run( p, vocab, hparams, nullptr, sink, resultAll, promptPast, &stamper );

	// the whole-buffer spectrogram, as iContext::runFull makes it before the loop
	if( hp->mel )
	{
		if( 0 != ref_set_mel( cpu, hp->mel, hp->melLen, hparams.n_mels ) ) { ref_free( cpu ); return E_FAIL; }
	}
	else if( nSamples > 0 && 0 != ref_pcm_to_mel( cpu, pcm, nSamples, threads ) ) { ref_free( cpu ); return E_FAIL; }
	const int64_t melLen = hp->mel ? hp->melLen : ( nSamples > 0 ? ref_mel_len( cpu ) : 0 );
	hr = run.begin( melLen );
	if( hr == S_OK )
	{
		std::vector<int> prompt;
		while( true )
		{
			hr = run.nextWindow( prompt );
			if( hr != S_OK ) break;
			if( 0 != ref_encode( cpu, run.seek, threads ) ) { hr = E_FAIL; break; }
			WindowScan scan( run.fullParams(), vocab, run.seek, run.seekEnd(), run.maxTokens() );
			std::vector<int32_t> feed( prompt.begin(), prompt.end() );
			int nPast = 0;
			for( int i = 0; !scan.over; i++ )
			{
				if( 0 != ref_decode( cpu, feed.data(), (int)feed.size(), nPast, threads ) ) { hr = E_FAIL; break; }
				nPast += (int)feed.size();
				TokenData t;
				int32_t id = 0, tid = 0;
				if( i == 0 ) ref_sample_timestamp( cpu, 1, &id, &tid, &t.p, &t.pt, &t.ptsum );
				else ref_sample_best( cpu, &id, &tid, &t.p, &t.pt, &t.ptsum );
				t.id = id; t.tid = tid;
				feed.assign( 1, id );
				scan.feed( t );
			}
			if( FAILED( hr ) ) break;
			hr = run.finishWindow( scan );
			if( FAILED( hr ) ) break;
		}
		if( hr == S_FALSE ) hr = run.end();
	}
	ref_free( cpu );
	if( FAILED( hr ) ) return hr;

	std::ostringstream o;
	o.precision( 9 );
	o << "{\"segments\":[";
	for( size_t i = 0; i < resultAll.size(); i++ )
	{
		const Segment& s = resultAll[ i ];
		o << ( i ? "," : "" ) << "{\"t0\":" << s.t0 << ",\"t1\":" << s.t1 << ",\"text\":";
		jsonString( o, s.text );
		o << ",\"tokens\":[";
		for( size_t j = 0; j < s.tokens.size(); j++ )
		{
			const TokenData& t = s.tokens[ j ];
			o << ( j ? "," : "" ) << "{\"id\":" << t.id << ",\"tid\":" << t.tid << ",\"p\":" << t.p << ",\"pt\":" << t.pt << ",\"ptsum\":" << t.ptsum
			  << ",\"t0\":" << t.t0 << ",\"t1\":" << t.t1 << ",\"vlen\":" << t.vlen << "}";
		}
		o << "]}";
	}
	// the same transcript as iContext::getResults hands it out: the product's own conversion into the reference's POD layout
	ResultData pods;
	fillResultData( resultAll, vocab, hp->mediaTime, (eResultFlags)hp->resultFlags, pods );
	o << "],\"pods\":[";
	for( size_t i = 0; i < pods.segments.size(); i++ )
	{
		const sSegment& s = pods.segments[ i ];
		o << ( i ? "," : "" ) << "{\"t0\":" << s.time.begin.ticks << ",\"t1\":" << s.time.end.ticks << ",\"first_token\":" << s.firstToken << ",\"count_tokens\":" << s.countTokens << ",\"text\":";
		jsonString( o, s.text );
		o << ",\"tokens\":[";
		for( uint32_t j = 0; j < s.countTokens && s.firstToken + j < pods.tokens.size(); j++ )
		{
			const sToken& t = pods.tokens[ s.firstToken + j ];
			o << ( j ? "," : "" ) << "{\"id\":" << t.id << ",\"flags\":" << (uint32_t)t.flags << ",\"t0\":" << t.time.begin.ticks << ",\"t1\":" << t.time.end.ticks << ",\"text\":";
			jsonString( o, t.text ? t.text : "" );
			o << "}";
		}
		o << "]}";
	}
	o << "],\"progress\":[";
	for( size_t i = 0; i < sinks.progress.size(); i++ ) o << ( i ? "," : "" ) << sinks.progress[ i ];
	o << "],\"new_segment\":[" << sinks.newSegmentCalls << "," << sinks.newSegments << "]}";
	g_out = o.str();
	return hr;
}
extern "C" __attribute__( ( visibility( "default" ) ) ) const char* hl_result() { return g_out.c_str(); }

// support.cpp's language table (Whisper/Whisper/Languages.cpp, languageCodez.inl): the id behind a makeLanguageKey() key, -1 when unknown
extern "C" __attribute__( ( visibility( "default" ) ) ) int hl_language_id( uint32_t key ) { return lookupLanguageId( key ); }

// support.cpp's vocabulary loader (loadVocabulary + Vocabulary::finalize: special ids, the tokens the file does not store) against the
// reference CPU model's own vocabulary of the same file: returns the number of differences (token strings of all ids, the six special ids)
extern "C" __attribute__( ( visibility( "default" ) ) ) int hl_vocabulary_differences( const char* modelPath, int* nVocabOut )
{
	Vocabulary vocab;
	if( FAILED( loadVocabulary( modelPath, vocab ) ) ) return -1;
	ref_set_log_level( 0 );
	void* cpu = ref_init( modelPath );
	if( !cpu ) return -1;
	int32_t h[ 11 ];
	ref_hparams( cpu, h );
	int diffs = 0;
	const int special[ 6 ] = { vocab.token_eot, vocab.token_sot, vocab.token_prev, vocab.token_solm, vocab.token_not, vocab.token_beg };
	for( int i = 0; i < 6; i++ ) diffs += special[ i ] != ref_token_special( cpu, i );
	diffs += vocab.token_translate != ref_token_special( cpu, 6 ) || vocab.token_transcribe != ref_token_special( cpu, 7 );
	diffs += vocab.n_vocab != h[ 0 ];
	for( int id = 0; id < h[ 0 ]; id++ )
	{
		const char* a = vocab.string( id );
		const char* b = ref_token_to_str( cpu, id );
		if( !a || !b || 0 != strcmp( a, b ) )
		{
			if( diffs < 8 ) fprintf( stderr, "vocabulary: token %d '%s' vs the reference's '%s'\n", id, a ? a : "(null)", b ? b : "(null)" );
			diffs++;
		}
	}
	if( nVocabOut ) *nVocabOut = h[ 0 ];
	ref_free( cpu );
	return diffs;
}

// Vocabulary::tokenize (iModel::tokenize, the --prompt of the CLI) against the reference's whisper_tokenize on the same file: texts are
// '\n'-separated; returns the number of texts whose token lists differ
extern "C" __attribute__( ( visibility( "default" ) ) ) int hl_tokenize_differences( const char* modelPath, const char* texts, int* nTextsOut, int* nTokensOut )
{
	Vocabulary vocab;
	if( FAILED( loadVocabulary( modelPath, vocab ) ) ) return -1;
	ref_set_log_level( 0 );
	void* cpu = ref_init( modelPath );
	if( !cpu ) return -1;
	int diffs = 0, nTexts = 0, nTokens = 0;
	std::istringstream in( texts );
	std::string line;
	while( std::getline( in, line ) )
	{
		std::vector<int> ours;
		if( FAILED( vocab.tokenize( line.c_str(), ours ) ) ) { diffs++; continue; }
		std::vector<int32_t> theirs( 4096 );
		const int n = ref_tokenize( cpu, line.c_str(), theirs.data(), (int)theirs.size() );
		theirs.resize( (size_t)std::max( n, 0 ) );
		if( n < 0 || std::vector<int32_t>( ours.begin(), ours.end() ) != theirs )
		{
			if( diffs < 5 ) fprintf( stderr, "tokenize: '%s' -> %zu tokens vs the reference's %d\n", line.c_str(), ours.size(), n );
			diffs++;
		}
		nTexts++;
		nTokens += (int)ours.size();
	}
	if( nTextsOut ) *nTextsOut = nTexts;
	if( nTokensOut ) *nTokensOut = nTokens;
	ref_free( cpu );
	return diffs;
}
