// The per-stream rules of the reference's host loop -- ContextImpl::runFullImpl (Whisper/Whisper/ContextImpl.cpp:452-793), itself a
// port of whisper_full (Whisper/source/whisper.cpp:2765-3120) -- as two small state machines, so that ONE stream through
// iContext::runFull / runStreamed and the K streams of Whisper::runFullBatch (lock-step batches, batchScheduler.cpp) apply the very same
// code to their tokens:
//   StreamRun   one recording: seek range, prompt carry-over, the window loop's entry / exit rules, failure handling, segment
//               cutting, callbacks. The caller supplies the device work between nextWindow() and finishWindow().
//   WindowScan  the tokens of one window, fed in order: timestamp tracking and the stop rules (ContextImpl.cpp:597-673).
#pragma once
#include "hostCommon.h"

namespace Whisper
{
	// The reference ships TWO host loops that differ in two rules: its CPU model (Whisper/source/whisper.cpp:2765-3120, the oracle
	// every parity test is pinned to) drops the past prompt when < 5 s of audio remain and retries a failed window once without it;
	// its GPU model's port (Whisper/Whisper/ContextImpl.cpp:452-793) does neither -- and its token-level timestamps start from 0
	// instead of "unknown" (see finishWindow). The default follows the CPU path, because that is what north_star asks token ids to
	// match; whisperc_set_host_loop_rules( 1 ) selects the GPU model's behaviour. Both are pinned on the reference's own code:
	// whisper_full (tests/golden/ref_hostloop.json) and ContextImpl.cpp compiled unmodified (ref_hostloop_contextimpl.json), on the
	// CPU (tests/test_hostloop_cpu.py: these objects over the reference's CPU model) and through libWhisper.so (tests/test_host_api.py).
	enum struct eHostLoopRules : int { ReferenceCpu = 0, ContextImpl = 1 };
	extern eHostLoopRules g_hostLoopRules;

	constexpr int CHUNK_FRAMES = 3000;	 // 30 s of 10 ms frames (WHISPER_CHUNK_SIZE * 100)

	class WindowScan
	{
		const sFullParams& params;
		const Vocabulary& vocab;
		const int seek, seekEnd, nMax;
		int i = 0;
		bool hasTs = false;
	public:
		int seekDelta = CHUNK_FRAMES;
		int resultLen = 0;
		bool failed = false, over = false;
		std::vector<TokenData> tokens;
		WindowScan( const sFullParams& p, const Vocabulary& v, int seek_, int seekEnd_, int nMax_ ) : params( p ), vocab( v ), seek( seek_ ), seekEnd( seekEnd_ ), nMax( nMax_ ) {}
		int consumed() const { return i; }
		// the constants feed() tests, for the device-side restatement of these rules (beam search without the host in the loop: wh_beam_rules)
		void constants( int& seekOut, int& seekEndOut, int& nMaxOut, int& maxTokensOut, bool& singleSegmentOut ) const
		{
			seekOut = seek; seekEndOut = seekEnd; nMaxOut = nMax; maxTokensOut = params.max_tokens; singleSegmentOut = params.flag( eFullParamsFlags::SingleSegment );
		}
		// beam search: the window's result becomes that of the hypothesis that won (same window: same seek, bounds and parameters)
		void adopt( const WindowScan& o )
		{
			i = o.i; hasTs = o.hasTs; seekDelta = o.seekDelta; resultLen = o.resultLen; failed = o.failed; over = o.over; tokens = o.tokens;
		}
		// Token i of the window. Returns true when the window is over: `failed`, or resultLen tokens stand and the next window
		// starts seekDelta frames later.
		bool feed( const TokenData& token )
		{
			if( over ) return true;
			if( token.id > vocab.token_beg )
			{
				// a timestamp token moves the sliding window; going back in time ends the window
				const int seekDeltaNew = 2 * ( token.id - vocab.token_beg );
				if( hasTs && seekDelta > seekDeltaNew && resultLen < i ) return over = true;
				seekDelta = seekDeltaNew;
				resultLen = i + 1;
				hasTs = true;
			}
			tokens.push_back( token );
			const bool endOfAudio = hasTs && seek + seekDelta + 100 >= seekEnd;
			if( token.id == vocab.token_eot || ( params.max_tokens > 0 && i >= params.max_tokens ) || endOfAudio )
			{
				if( resultLen == 0 )
				{
					if( seek + seekDelta + 100 >= seekEnd )
						resultLen = i + 1;
					else
					{
						failed = true;
						return over = true;
					}
				}
				if( params.flag( eFullParamsFlags::SingleSegment ) )
				{
					resultLen = i + 1;
					seekDelta = CHUNK_FRAMES;
				}
				return over = true;
			}
			// stuck in a repetition loop: give up on this window (ContextImpl.cpp:665-672)
			if( i == nMax - 1 && ( resultLen == 0 || seekDelta < CHUNK_FRAMES / 2 ) )
			{
				failed = true;
				return over = true;
			}
			i++;
			if( i >= nMax ) over = true;	// the loop's own bound: nMax tokens examined
			return over;
		}
	};

	class StreamRun
	{
		const sFullParams params;	  // a copy: the batch scheduler outlives the caller's per-stream structures
		const Vocabulary& vocab;
		const wh_hparams& hp;
		iContext* const self;		  // what the callbacks receive
		const sProgressSink progress;
		std::vector<Segment>& resultAll;
		std::vector<int>& promptPast;
		TokenTimestamper* const stamper;   // TokenTimestamps flag, or nullptr
		std::vector<int> promptInit;
		int seekStart = 0, seekEndV = 0;
		bool stoppedPrematurely = false;
	public:
		int seek = 0;
		StreamRun( const sFullParams& p, const Vocabulary& v, const wh_hparams& h, iContext* ctx, const sProgressSink& sink, std::vector<Segment>& results,
			std::vector<int>& past, TokenTimestamper* ts ) : params( p ), vocab( v ), hp( h ), self( ctx ), progress( sink ), resultAll( results ), promptPast( past ), stamper( ts ) {}
		const sFullParams& fullParams() const { return params; }
		int seekEnd() const { return seekEndV; }
		int maxTokens() const { return hp.n_text_ctx / 2 - 4; }

		// S_OK: run the window loop; S_FALSE: less than a second of audio, nothing to do (ContextImpl.cpp:469-473); failure otherwise
		HRESULT begin( int64_t melLen )
		{
			resultAll.clear();
			if( params.flag( eFullParamsFlags::SpeedupAudio ) )
			{
				logError( "GPU model doesn't implement the SpeedupAudio flag" );
				return E_NOTIMPL;
			}
			// sFullParams::audio_ctx (ContextImpl.cpp:488-489: exp_n_audio_ctx = params.audio_ctx): the callers hand it to the device context
			// (wh_context_set_audio_ctx) before the first window; here only the range is checked
			if( params.audio_ctx < 0 || params.audio_ctx > hp.n_audio_ctx )
			{
				logError( "audio_ctx %d is outside [ 0, %d ]", params.audio_ctx, hp.n_audio_ctx );
				return E_INVALIDARG;
			}
			seekStart = params.offset_ms / 10;
			seekEndV = seekStart + ( params.duration_ms == 0 ? (int)melLen : params.duration_ms / 10 );
			if( seekEndV < 100 + seekStart ) return S_FALSE;
			if( params.flag( eFullParamsFlags::NoContext ) ) promptPast.clear();
			if( params.prompt_tokens && params.prompt_n_tokens > 0 )
				promptPast.insert( promptPast.begin(), params.prompt_tokens, params.prompt_tokens + params.prompt_n_tokens );
			// the tokens that select the task
			promptInit = { vocab.token_sot };
			if( vocab.isMultilingual() )
			{
				const int langId = lookupLanguageId( params.language );
				if( langId < 0 )
				{
					char lang[ 5 ] = { 0 };
					memcpy( lang, &params.language, 4 );
					logError( "runFull: unknown language '%s'", lang );
					return E_INVALIDARG;
				}
				promptInit.push_back( vocab.token_sot + 1 + langId );
				promptInit.push_back( params.flag( eFullParamsFlags::Translate ) ? vocab.token_translate : vocab.token_transcribe );
			}
			seek = seekStart;
			stoppedPrematurely = false;
			return S_OK;
		}

		// Top of the loop (ContextImpl.cpp:531-576). S_OK: `prompt` is the prompt of the window at `seek`, encode + decode it;
		// S_FALSE: the stream is finished (call end()); failure: a callback failed.
		HRESULT nextWindow( std::vector<int>& prompt )
		{
			if( progress.pfn )
			{
				const double percentage = (double)( seek - seekStart ) / (double)( seekEndV - seekStart );
				CHECK( progress.pfn( percentage, self, progress.pv ) );
			}
			if( seek + 100 >= seekEndV ) return S_FALSE;
			// whisper.cpp only: with less than 5 s left the past prompt is dropped, "since it tends to confuse the decoder"
			// (Whisper/source/whisper.cpp:2874-2878; absent from ContextImpl.cpp)
			if( g_hostLoopRules == eHostLoopRules::ReferenceCpu && seek > seekStart && seek + 500 >= seekEndV ) promptPast.clear();
			if( params.encoder_begin_callback )
			{
				const HRESULT hr = params.encoder_begin_callback( self, params.encoder_begin_callback_user_data );
				if( FAILED( hr ) ) return hr;
				if( hr != S_OK )
				{
					stoppedPrematurely = true;
					return S_FALSE;
				}
			}
			// previous text conditions this window: [prev] + the last n_take tokens + the task tokens (ContextImpl.cpp:565-576)
			prompt.clear();
			if( !promptPast.empty() )
			{
				const int nTake = std::min( std::min( params.n_max_text_ctx, hp.n_text_ctx / 2 ), (int)promptPast.size() );
				prompt.push_back( vocab.token_prev );
				prompt.insert( prompt.end(), promptPast.end() - nTake, promptPast.end() );
				promptPast.assign( prompt.begin() + 1, prompt.end() );
			}
			prompt.insert( prompt.end(), promptInit.begin(), promptInit.end() );
			return S_OK;
		}

		// Bottom of the loop (ContextImpl.cpp:675-785): the scanned window becomes segments and the stream moves on -- or, when
		// the window failed, is retried without the past prompt / skipped by a second.
		HRESULT finishWindow( WindowScan& scan )
		{
			if( scan.failed )
			{
				// whisper.cpp retries the same window once without the past prompt before skipping a second
				// (whisper.cpp:3006-3016); ContextImpl.cpp:675-680 skips right away
				if( g_hostLoopRules == eHostLoopRules::ReferenceCpu && !promptPast.empty() )
				{
					promptPast.clear();
					return S_OK;
				}
				logError( "runFull: failed to generate timestamp token - skipping one second" );
				seek += 100;
				return S_OK;
			}
			std::vector<TokenData>& tokensCur = scan.tokens;
			tokensCur.resize( std::min( (size_t)scan.resultLen, tokensCur.size() ) );
			for( const TokenData& t : tokensCur ) promptPast.push_back( t.id );

			// cut the window's tokens into segments at the timestamp tokens (ContextImpl.cpp:689-784)
			if( !tokensCur.empty() )
			{
				const bool special = params.flag( eFullParamsFlags::PrintSpecial );
				const bool single = params.flag( eFullParamsFlags::SingleSegment );
				int i0 = 0;
				int t0 = seek + 2 * ( tokensCur.front().tid - vocab.token_beg );
				std::string text;
				auto emit = [ & ]( int t1, int last ) -> HRESULT
				{
					Segment s;
					s.t0 = t0; s.t1 = t1; s.text = text;
					s.tokens.assign( tokensCur.begin() + i0, tokensCur.begin() + last + 1 );
					if( params.flag( eFullParamsFlags::PrintRealtime ) ) logDebug( "[%d --> %d]  %s", t0, t1, text.c_str() );
					resultAll.push_back( std::move( s ) );
					uint32_t nNew = 1;
					if( params.flag( eFullParamsFlags::TokenTimestamps ) && stamper && stamper->ready() )
					{
						// whisper.cpp:3063-3069 / ContextImpl.cpp:741-749. The GPU model's sampler starts a token's times at 0
						// (`sTokenData result = { 0 }`, ContextImpl.cpp:77) where whisper.cpp starts them at -1 = unknown
						// (whisper.cpp:1880-1882), and its port of the algorithm still tests `t1 < 0` (ContextImpl.cpp:306): there the
						// proportional split of the unknown intervals never runs. Under its rules the same happens here.
						if( g_hostLoopRules == eHostLoopRules::ContextImpl )
							for( TokenData& t : resultAll.back().tokens ) t.t0 = t.t1 = 0;
						stamper->compute( resultAll.back(), vocab, params.thold_pt, params.thold_ptsum );
						if( params.max_len > 0 ) nNew = (uint32_t)TokenTimestamper::wrapLast( resultAll, vocab, params.max_len );
					}
					if( params.new_segment_callback )
					{
						const HRESULT hr = params.new_segment_callback( self, nNew, params.new_segment_callback_user_data );
						if( FAILED( hr ) ) return hr;
					}
					return S_OK;
				};
				for( int i = 0; i < (int)tokensCur.size(); i++ )
				{
					const int id = tokensCur[ i ].id;
					if( special || id < vocab.token_eot ) text += vocab.string( id );
					if( id > vocab.token_beg && !single )
					{
						const int t1 = seek + 2 * ( tokensCur[ i ].tid - vocab.token_beg );
						if( !text.empty() ) CHECK( emit( t1, i ) );
						text.clear();
						while( i < (int)tokensCur.size() && tokensCur[ i ].id > vocab.token_beg ) i++;
						i--;
						t0 = t1;
						i0 = i + 1;
					}
				}
				if( !text.empty() ) CHECK( emit( seek + scan.seekDelta, (int)tokensCur.size() - 1 ) );
			}
			seek += scan.seekDelta;
			return S_OK;
		}

		// ContextImpl.cpp:788-792
		HRESULT end()
		{
			if( progress.pfn && !stoppedPrematurely ) CHECK( progress.pfn( 1.0, self, progress.pv ) );
			return S_OK;
		}
	};
}
