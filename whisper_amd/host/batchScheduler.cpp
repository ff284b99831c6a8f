// Whisper::runFullBatch -- K streams in lock step behind the drop-in boundary (extension next to loadModelShared).
//
// The reference transcribes one recording per iContext, window after window (ContextImpl::runFullImpl, Whisper/Whisper/ContextImpl.cpp:
// 452-793); its only multi-instance facilities are whisper_full_parallel (Whisper/source/whisper.cpp:3127: n processors, each a slice of
// the recording) and iModel::clone + a context per thread (Whisper/Whisper/ModelImpl.cpp:40-60). On MI355X a decode step of ONE sequence
// is a chain of ~200 launches at their latency floors that leaves the chip idle; the same chain carries 64 or 112 sequences for about
// the same time. So the streams' NEXT windows are made one encoder batch and one decode chain:
//
//   group   = one wh_context of `slots` windows (own HIP stream, own captured decode graph, own KV caches)
//   slot    = the stream currently assigned to a row of the group's batch; a finished stream's slot takes the next pending stream
//   round   = every slot's next window: StreamRun::nextWindow (progress / encoder_begin callbacks, prompt with the stream's own past
//             text) -> wh_encode_windows (each slot its own spectrogram and seek) -> wh_decode_window_start_ragged (prompts of
//             different lengths, a position per sequence) -> greedy chunks, scanned chunk by chunk with WindowScan until every slot's
//             window is over -> StreamRun::finishWindow per slot (segments, callbacks, seek += the stream's own delta)
//
// Rounds of different groups overlap on the GPU (the MFMA-bound encoder of one under the latency-bound decode chain of the other);
// ONE host thread -- the caller's, so callbacks arrive on the calling thread like runFull's -- serves all groups through
// wh_decode_window_ready polls. Per stream the rules are hostLoop.h's, the same objects iContext::runFull uses: a stream's transcript
// is the transcript of runFull on the same samples (tests/test_batch_api.py).
#include "hostCommon.h"
#include "hostLoop.h"
#include "results.h"
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <thread>

namespace Whisper
{
	namespace
	{
		// What a stream's callbacks receive: results so far and the model; running anything through it is not possible.
		class StreamContext : public ComObject<iContext>
		{
			iModel* const owner;
			const Vocabulary& vocab;
			// what getResults hands out without NewObject: embedded, Release never deletes -- the ownership ContextImpl has
			// (a heap object held in a raw pointer here was freed by the first callback that released it the reference's way)
			mutable TranscribeResultStatic live;
		public:
			std::vector<Segment> resultAll;
			int64_t mediaTimeOffset = 0;
			StreamContext( iModel* m, const Vocabulary& v ) : owner( m ), vocab( v ) {}
			HRESULT runFull( const sFullParams&, const iAudioBuffer* ) override { return E_NOTIMPL; }
			HRESULT runStreamed( const sFullParams&, const sProgressSink&, const iAudioReader* ) override { return E_NOTIMPL; }
			HRESULT runCapture( const sFullParams&, const sCaptureCallbacks&, const iAudioCapture* ) override { return E_NOTIMPL; }
			HRESULT getResults( eResultFlags flags, iTranscribeResult** pp ) const override
			{
				if( !pp ) return E_POINTER;
				if( flags & eResultFlags::NewObject )
				{
					TranscribeResult* r = new TranscribeResult();
					const HRESULT hr = fillResultData( resultAll, vocab, mediaTimeOffset, flags, *r );
					if( FAILED( hr ) ) { r->Release(); return hr; }
					*pp = r;
					return S_OK;
				}
				// without NewObject the reference hands out an object that lives as long as the context (TranscribeResult.h:34-43)
				CHECK( fillResultData( resultAll, vocab, mediaTimeOffset, flags, live ) );
				*pp = &live;
				return S_OK;
			}
			HRESULT detectSpeaker( const sTimeInterval&, eSpeakerChannel& result ) const override { result = eSpeakerChannel::NoStereoData; return S_FALSE; }
			HRESULT getModel( iModel** pp ) override
			{
				if( !pp ) return E_POINTER;
				owner->AddRef();
				*pp = owner;
				return S_OK;
			}
			HRESULT fullDefaultParams( eSamplingStrategy, sFullParams* ) override { return E_NOTIMPL; }
			HRESULT timingsPrint() override { return S_OK; }
			HRESULT timingsReset() override { return S_OK; }
		};

		struct Stream
		{
			uint32_t index = 0;
			const float* pcm = nullptr;
			int64_t nSamples = 0;
			sFullParams params{};
			StreamContext* ctx = nullptr;
			std::vector<int> promptPast;
			TokenTimestamper stamper;
			std::unique_ptr<StreamRun> run;
			void *pcmDev = nullptr, *melDev = nullptr;
			int64_t melLen = 0;
			// the window of the current round
			std::vector<int> prompt;
			std::unique_ptr<WindowScan> scan;
			HRESULT status = S_OK;
		};

		// Device buffers of the streams (PCM, spectrograms), recycled: hipFree waits for the whole device, i.e. for the OTHER group's
		// queued chunk or encoder, so nothing is freed while a run is under way -- a retired stream's buffers serve the next stream.
		class BufferPool
		{
			struct Buf { void* dev; int64_t bytes; };
			std::vector<Buf> idle;
			std::map<void*, int64_t> sizes;
		public:
			HRESULT acquire( int64_t bytes, void*& dev )
			{
				int best = -1;
				for( int i = 0; i < (int)idle.size(); i++ )
					if( idle[ i ].bytes >= bytes && ( best < 0 || idle[ i ].bytes < idle[ best ].bytes ) ) best = i;
				if( best >= 0 && idle[ best ].bytes <= 2 * bytes + 65536 )
				{
					dev = idle[ best ].dev;
					idle.erase( idle.begin() + best );
					return S_OK;
				}
				CHECK_WH( wh_buffer_alloc( bytes, &dev ) );
				sizes[ dev ] = bytes;
				return S_OK;
			}
			void release( void* dev )
			{
				if( dev ) idle.push_back( Buf{ dev, sizes[ dev ] } );
			}
			// only with the device idle (the runner's destructor)
			void freeAll()
			{
				for( auto& kv : sizes ) wh_buffer_free( kv.first );
				sizes.clear();
				idle.clear();
			}
		};
		// What the groups of a runner share: the pool and a one-window context whose stream carries the uploads and spectrograms of
		// newly admitted streams, so that admission waits for its own copies only -- not for a group's queued decode work.
		struct Shared
		{
			wh_context* loader = nullptr;
			BufferPool pool;
		};

		struct Group
		{
			wh_context* gpu = nullptr;
			int slots = 0;
			std::vector<Stream*> slot;
			enum ePhase { Idle, Decoding, Done } phase = Idle;
			int enqueued = 0, fetched = 0, promptMax = 0;
			std::vector<wh_token_data> buf;
			std::vector<int32_t> tokens, lens;
			std::vector<wh_mel_window> windows;
		};

		// One call of iBatchRunner::run over the runner's groups (their wh_contexts outlive the call: KV caches, captured graphs)
		class Scheduler
		{
			const std::shared_ptr<LoadedModel> model;
			iModel* const owner;
			const sFullParams& common;
			const sBatchStream* const descs;
			const uint32_t count;
			iTranscribeResult** const results;
			HRESULT* const perStream;
			std::vector<Group>& groups;
			Shared& shared;
			const int chunk, lookahead;
			bool loaderBusy = false;
			std::deque<uint32_t> pending;
			HRESULT firstFailure = S_OK;

			void fail( Stream& s, HRESULT hr )
			{
				s.status = hr;
				if( SUCCEEDED( firstFailure ) ) firstFailure = hr;
			}
			// The stream's device buffers go, its transcript becomes results[index]
			void retire( Group& g, int b )
			{
				Stream* s = g.slot[ b ];
				g.slot[ b ] = nullptr;
				shared.pool.release( s->melDev );
				shared.pool.release( s->pcmDev );
				if( perStream ) perStream[ s->index ] = s->status;
				if( SUCCEEDED( s->status ) )
				{
					TranscribeResult* r = new TranscribeResult();
					fillResultData( s->ctx->resultAll, model->vocab, s->ctx->mediaTimeOffset, eResultFlags::Tokens | eResultFlags::Timestamps, *r );
					results[ s->index ] = r;
				}
				s->run.reset();
				s->ctx->Release();
				delete s;
			}
			// Next pending stream -> slot b: PCM to the device, its spectrogram (normalised on the stream's own maximum, like runFull's on
			// its buffer), StreamRun::begin. Returns false when nothing is pending. A stream that ends here (too short, bad parameters)
			// is retired on the spot and the next one is tried.
			bool admit( Group& g, int b )
			{
				while( !pending.empty() )
				{
					const uint32_t i = pending.front();
					pending.pop_front();
					const sBatchStream& d = descs[ i ];
					Stream* s = new Stream();
					s->index = i;
					s->params = d.params ? *d.params : common;
					s->ctx = new StreamContext( owner, model->vocab );
					g.slot[ b ] = s;
					const int64_t total = d.buffer ? (int64_t)d.buffer->countSamples() : 0;
					const float* const mono = d.buffer ? d.buffer->getPcmMono() : nullptr;
					if( !d.buffer || d.firstSample < 0 || d.firstSample > total || d.countSamples < 0 || d.firstSample + d.countSamples > total || ( total > 0 && !mono ) )
					{
						logError( "runFullBatch: stream %u names samples outside its buffer", i );
						fail( *s, E_INVALIDARG );
						retire( g, b );
						continue;
					}
					s->nSamples = d.countSamples ? d.countSamples : total - d.firstSample;
					s->pcm = mono ? mono + d.firstSample : nullptr;
					int64_t bufferTime = 0;
					d.buffer->getTime( bufferTime );
					s->ctx->mediaTimeOffset = bufferTime + d.firstSample * 10000000ll / 16000;
					s->melLen = s->nSamples / 160;
					HRESULT hr = S_OK;
					if( s->melLen > 0 )
					{
						// enqueued on the loader's stream; startRound waits for it once, after the last admission of the round
						int64_t got = 0;
						wh_context_bind( shared.loader );
						hr = shared.pool.acquire( s->nSamples * 4, s->pcmDev );
						if( SUCCEEDED( hr ) ) hr = shared.pool.acquire( s->melLen * model->hp.n_mels * 4, s->melDev );
						if( SUCCEEDED( hr ) && ( 0 != wh_buffer_upload_async( shared.loader, s->pcmDev, s->pcm, s->nSamples * 4 ) ||
							0 != wh_mel_spectrogram( shared.loader, (const float*)s->pcmDev, s->nSamples, (float*)s->melDev, &got ) ) )
							hr = hrFromStatus( -1, "runFullBatch: stream admission" );
						loaderBusy = true;
					}
					if( SUCCEEDED( hr ) )
					{
						if( s->params.flag( eFullParamsFlags::TokenTimestamps ) ) s->stamper.begin( s->pcm, (size_t)s->nSamples );
						const sProgressSink none{ nullptr, nullptr };
						s->run.reset( new StreamRun( s->params, model->vocab, model->hp, s->ctx, none, s->ctx->resultAll, s->promptPast, &s->stamper ) );
						hr = s->run->begin( s->melLen );
					}
					if( hr != S_OK )
					{
						if( FAILED( hr ) ) fail( *s, hr );
						else s->status = hr;	  // S_FALSE: shorter than a second, an empty transcript (ContextImpl.cpp:469-473)
						retire( g, b );
						continue;
					}
					return true;
				}
				return false;
			}

			// Every slot's next window; streams that end here leave and their slots are refilled. Then the round's device work is enqueued.
			HRESULT startRound( Group& g )
			{
				bool any = false;
				for( int b = 0; b < g.slots; b++ )
				{
					while( true )
					{
						if( !g.slot[ b ] && !admit( g, b ) ) break;
						Stream& s = *g.slot[ b ];
						const HRESULT hr = s.run->nextWindow( s.prompt );
						if( hr == S_OK )
						{
							s.scan.reset( new WindowScan( s.run->fullParams(), model->vocab, s.run->seek, s.run->seekEnd(), s.run->maxTokens() ) );
							any = true;
							break;
						}
						if( FAILED( hr ) ) fail( s, hr );
						else
						{
							const HRESULT hrEnd = s.run->end();
							if( FAILED( hrEnd ) ) fail( s, hrEnd );
						}
						retire( g, b );
					}
				}
				if( loaderBusy )
				{
					CHECK_WH( wh_context_synchronize( shared.loader ) );
					loaderBusy = false;
				}
				if( !any )
				{
					g.phase = Group::Done;
					return S_OK;
				}
				// one encoder batch: each slot its own spectrogram at its own seek; an idle slot is a window of zeros with a one-token prompt
				const int sot = model->vocab.token_sot;
				g.promptMax = 1;
				for( int b = 0; b < g.slots; b++ )
					if( g.slot[ b ] ) g.promptMax = std::max( g.promptMax, (int)g.slot[ b ]->prompt.size() );
				g.tokens.assign( (size_t)g.slots * g.promptMax, 0 );
				g.lens.assign( (size_t)g.slots, 1 );
				g.windows.assign( (size_t)g.slots, wh_mel_window{ nullptr, 0, 0, 0 } );
				for( int b = 0; b < g.slots; b++ )
				{
					int32_t* const row = g.tokens.data() + (size_t)b * g.promptMax;
					const Stream* s = g.slot[ b ];
					if( !s ) { row[ 0 ] = sot; continue; }
					std::copy( s->prompt.begin(), s->prompt.end(), row );
					g.lens[ b ] = (int32_t)s->prompt.size();
					g.windows[ b ] = wh_mel_window{ (const float*)s->melDev, s->melLen, (int32_t)s->run->seek, 0 };
				}
				CHECK_WH( wh_encode_windows( g.gpu, g.windows.data(), g.slots ) );
				const int room = model->hp.n_text_ctx - g.promptMax;
				const int n0 = std::max( 0, std::min( chunk, room ) );
				CHECK_WH( wh_decode_window_start_ragged( g.gpu, g.slots, g.tokens.data(), g.lens.data(), g.promptMax, n0, 1, 1 ) );
				g.enqueued = 1 + n0;
				g.fetched = 0;
				// lookahead: chunks kept queued BEHIND the one the host waits for. 0 = a chunk is enqueued only once the previous one has
				// been scanned: the stream idles for the host's reaction (a poll + a few graph launches, under the other group's work)
				// and at most one chunk is decoded past the end of a round; 1 = the device never waits, up to two chunks are
				for( int k = 0; k < lookahead; k++ ) CHECK( enqueue( g ) );
				g.phase = Group::Decoding;
				return S_OK;
			}
			HRESULT enqueue( Group& g )
			{
				const int room = model->hp.n_text_ctx - ( g.promptMax + g.enqueued - 1 );
				const int n = std::min( chunk, room );
				if( n <= 0 ) return S_FALSE;
				CHECK_WH( wh_decode_window_continue( g.gpu, n ) );
				g.enqueued += n;
				return S_OK;
			}
			// samples the next fetch of this group takes: the first sample alone, then chunk by chunk
			int nextCount( const Group& g ) const { return g.fetched == 0 ? std::min( g.enqueued, 1 + chunk ) : std::min( chunk, g.enqueued - g.fetched ); }

			HRESULT consume( Group& g )
			{
				const int n = nextCount( g );
				if( n <= 0 )
				{
					// n_text_ctx reached with windows still open: WindowScan's own bound (n_text_ctx / 2 - 4 tokens) fires first for every
					// prompt the host loop can build, so this is unreachable; close the round rather than spin
					return finishRound( g );
				}
				g.buf.resize( (size_t)n * g.slots );
				CHECK_WH( wh_decode_window_fetch( g.gpu, g.fetched, n, g.buf.data() ) );
				g.fetched += n;
				bool open = false;
				for( int b = 0; b < g.slots; b++ )
				{
					Stream* s = g.slot[ b ];
					if( !s || s->scan->over ) continue;
					for( int k = 0; k < n && !s->scan->over; k++ )
					{
						const wh_token_data& t = g.buf[ (size_t)k * g.slots + b ];
						TokenData td;
						td.id = t.id; td.tid = t.tid; td.p = t.p; td.pt = t.pt; td.ptsum = t.ptsum;
						s->scan->feed( td );
					}
					open = open || !s->scan->over;
				}
				if( !open ) return finishRound( g );
				while( g.enqueued - g.fetched < ( 1 + lookahead ) * chunk )
					if( enqueue( g ) != S_OK ) break;
				return S_OK;
			}
			HRESULT finishRound( Group& g )
			{
				for( int b = 0; b < g.slots; b++ )
				{
					Stream* s = g.slot[ b ];
					if( !s ) continue;
					if( !s->scan->over ) s->scan->failed = s->scan->over = true;	// see consume(): not reachable through the host loop's prompts
					const HRESULT hr = s->run->finishWindow( *s->scan );
					s->scan.reset();
					if( FAILED( hr ) )
					{
						fail( *s, hr );
						retire( g, b );
					}
				}
				g.phase = Group::Idle;
				return S_OK;
			}

		public:
			Scheduler( const std::shared_ptr<LoadedModel>& m, iModel* o, const sFullParams& p, const sBatchStream* d, uint32_t n, iTranscribeResult** r, HRESULT* per,
				std::vector<Group>& grp, Shared& sh, int chunk_, int lookahead_ )
				: model( m ), owner( o ), common( p ), descs( d ), count( n ), results( r ), perStream( per ), groups( grp ), shared( sh ), chunk( chunk_ ), lookahead( lookahead_ ) {}
			~Scheduler()
			{
				// a failure left streams in their slots: their buffers go, the device work they queued is awaited
				for( Group& g : groups )
				{
					for( int b = 0; b < (int)g.slot.size(); b++ )
						if( g.slot[ b ] )
						{
							g.slot[ b ]->status = E_FAIL;
							retire( g, b );
						}
					if( g.gpu ) wh_context_synchronize( g.gpu );
					g.phase = Group::Idle;
				}
				if( shared.loader ) wh_context_synchronize( shared.loader );
			}
			HRESULT run( int slots, int nGroups )
			{
				for( uint32_t i = 0; i < count; i++ )
				{
					results[ i ] = nullptr;
					if( perStream ) perStream[ i ] = S_OK;
					pending.push_back( i );
				}
				for( int gi = 0; gi < (int)groups.size(); gi++ )
				{
					Group& g = groups[ gi ];
					g.slots = slots;
					g.slot.assign( (size_t)slots, nullptr );
					g.phase = gi < nGroups ? Group::Idle : Group::Done;
				}
				while( true )
				{
					bool progressed = false, alive = false;
					for( Group& g : groups )
					{
						if( g.phase == Group::Idle )
						{
							CHECK( startRound( g ) );
							progressed = true;
						}
						else if( g.phase == Group::Decoding )
						{
							const int n = nextCount( g );
							const int ready = n > 0 ? wh_decode_window_ready( g.gpu, g.fetched, n ) : 1;
							if( ready < 0 ) return hrFromStatus( ready, "wh_decode_window_ready" );
							if( ready )
							{
								CHECK( consume( g ) );
								progressed = true;
							}
						}
						alive = alive || g.phase != Group::Done;
					}
					if( !alive ) break;
					if( !progressed ) std::this_thread::sleep_for( std::chrono::microseconds( 50 ) );
				}
				return firstFailure;
			}
		};

		// iBatchRunner: the groups' device contexts (KV caches for maxSlots windows each, captured decode graphs) live as long as the
		// runner, so a service that transcribes batch after batch pays for them once.
		class BatchRunner : public ComObject<iBatchRunner>
		{
			const std::shared_ptr<LoadedModel> model;
			iModel* const owner;
			uint32_t maxSlots = 64, nGroups = 2;
			int chunk = 4, lookahead = 0;
			std::vector<Group> groups;
			Shared shared;
		public:
			BatchRunner( const std::shared_ptr<LoadedModel>& m, iModel* o, const sBatchSetup* setup ) : model( m ), owner( o )
			{
				owner->AddRef();
				if( setup && setup->maxSlots ) maxSlots = setup->maxSlots;
				if( setup && setup->groups ) nGroups = setup->groups;
				if( setup && setup->greedyChunk ) chunk = (int)setup->greedyChunk;
				if( setup && ( setup->flags & 1u ) ) lookahead = 1;
				if( const char* e = getenv( "WHISPER_BATCH_SLOTS" ) ) maxSlots = (uint32_t)std::max( 1, atoi( e ) );
				if( const char* e = getenv( "WHISPER_BATCH_GROUPS" ) ) nGroups = (uint32_t)std::max( 1, atoi( e ) );
				if( const char* e = getenv( "WHISPER_BATCH_CHUNK" ) ) chunk = atoi( e );
				if( const char* e = getenv( "WHISPER_BATCH_LOOKAHEAD" ) ) lookahead = atoi( e ) ? 1 : 0;
				chunk = std::max( 1, std::min( chunk, 64 ) );
				maxSlots = std::max<uint32_t>( 1, std::min<uint32_t>( maxSlots, 512 ) );
				nGroups = std::max<uint32_t>( 1, std::min<uint32_t>( nGroups, 8 ) );
			}
			~BatchRunner() override
			{
				for( Group& g : groups )
					if( g.gpu ) wh_context_synchronize( g.gpu );
				if( shared.loader )
				{
					wh_context_synchronize( shared.loader );
					shared.pool.freeAll();
					wh_context_destroy( shared.loader );
				}
				for( Group& g : groups )
					if( g.gpu ) wh_context_destroy( g.gpu );
				owner->Release();
			}
			HRESULT run( const sFullParams& params, const sBatchStream* streams, uint32_t count, iTranscribeResult** results, HRESULT* perStream ) override
			{
				if( !results || ( count && !streams ) ) return E_POINTER;
				if( count == 0 ) return S_OK;
				// the lock-step batch decodes greedily (one sequence per stream); beam search is iContext::runFull's (hypotheses of ONE window in lock step)
				bool beam = params.strategy == eSamplingStrategy::BeamSearch;
				for( uint32_t i = 0; i < count; i++ ) beam = beam || ( streams[ i ].params && streams[ i ].params->strategy == eSamplingStrategy::BeamSearch );
				if( beam )
				{
					logError( "runFullBatch: eSamplingStrategy::BeamSearch is not available in a lock-step batch; use iContext::runFull" );
					return E_NOTIMPL;
				}
				// streams dealt evenly: no more groups than hold two streams each, every group the same number of slots
				const uint32_t useGroups = std::max<uint32_t>( 1, std::min<uint32_t>( nGroups, count / 2 ) );
				const uint32_t slots = std::max<uint32_t>( 1, std::min<uint32_t>( maxSlots, ( count + useGroups - 1 ) / useGroups ) );
				while( groups.size() < useGroups )
				{
					Group g;
					CHECK_WH( wh_context_create( model->gpu, (int)maxSlots, nullptr, &g.gpu ) );
					groups.push_back( std::move( g ) );
				}
				if( !shared.loader ) CHECK_WH( wh_context_create( model->gpu, 1, nullptr, &shared.loader ) );
				// one sFullParams for every stream of the batch: its audio_ctx is the groups' (ContextImpl.cpp:488-489)
				if( params.audio_ctx < 0 || params.audio_ctx > model->hp.n_audio_ctx ) return E_INVALIDARG;
				for( Group& g : groups ) CHECK_WH( wh_context_set_audio_ctx( g.gpu, params.audio_ctx ) );
				Scheduler s( model, owner, params, streams, count, results, perStream, groups, shared, chunk, lookahead );
				const HRESULT hr = s.run( (int)slots, (int)useGroups );
				if( FAILED( hr ) ) logError( "runFullBatch: failed, HRESULT 0x%08x", (unsigned)hr );
				return hr;
			}
		};
	}	// namespace

	HRESULT createBatchRunner( iModel* model, const sBatchSetup* setup, iBatchRunner** pp )
	{
		if( !model || !pp ) return E_POINTER;
		iModelInternals* mi = nullptr;
		if( FAILED( model->QueryInterface( iModelInternals::iid(), (void**)&mi ) ) || !mi )
		{
			logError( "createBatchRunner: this iModel was not created by this library" );
			return E_INVALIDARG;
		}
		const std::shared_ptr<LoadedModel> lm = mi->loaded();
		mi->Release();
		*pp = new BatchRunner( lm, model, setup );
		return S_OK;
	}

	HRESULT runFullBatch( iModel* model, const sFullParams& params, const sBatchStream* streams, uint32_t count, const sBatchSetup* setup,
		iTranscribeResult** results, HRESULT* perStream )
	{
		if( !model || !results || ( count && !streams ) ) return E_POINTER;
		if( count == 0 ) return S_OK;
		iBatchRunner* runner = nullptr;
		sBatchSetup st = setup ? *setup : sBatchSetup{ 0, 0, 0, 0 };
		if( st.maxSlots == 0 ) st.maxSlots = std::min<uint32_t>( 64, count );	 // a one-off call sizes its contexts for what it was given
		CHECK( createBatchRunner( model, &st, &runner ) );
		const HRESULT hr = runner->run( params, streams, count, results, perStream );
		runner->Release();
		return hr;
	}
}
