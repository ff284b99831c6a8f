// iModel / iContext / iTranscribeResult / iAudioBuffer / iMediaFoundation implementations and the library exports.
//
// Host logic only (plain C++): windows, prompts, stop rules and segment assembly follow the reference's
// ContextImpl::runFullImpl (Whisper/Whisper/ContextImpl.cpp:452-793, itself a port of whisper_full), the arithmetic is
// behind include/whisper_hip.h. Structure here is ours: one WindowDecoder per 30 s window feeds a token stream (filled
// from the device-side greedy loop in chunks) through the reference's stop rules, then SegmentBuilder cuts it at the
// timestamp tokens.
#include <cstdlib>
#include "hostCommon.h"
#include "hostLoop.h"
#include "results.h"
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>

namespace Whisper
{
	eHostLoopRules g_hostLoopRules = eHostLoopRules::ReferenceCpu;

	namespace
	{
		// greedy steps enqueued per chunk; one chunk always runs behind the one being scanned, so up to two chunks are decoded in
		// vain when a window ends: one sequential 199 s clip through runFull, medium shape, ran at 265 / 308 / 330 / 351 audio-s/s
		// with chunks of 8 / 4 / 2 / 1 in round 3 (1.13 ms per step; a fetch polls the sampler's pinned mailbox, it enqueues
		// nothing). WHISPER_GREEDY_CHUNK overrides (1 .. 64).
		// eSamplingStrategy::BeamSearch: candidates ranked on the host after every step (round 4's decoder, the checker) instead of on the device
		bool g_beamRankingOnHost = []() { const char* e = getenv( "WHISPER_BEAM_HOST" ); return e && atoi( e ) != 0; }();
		static const int GREEDY_CHUNK = []() {
			const char* e = getenv( "WHISPER_GREEDY_CHUNK" );
			const int v = e ? atoi( e ) : 0;
			return v >= 1 && v <= 64 ? v : 1;
		}();


		// ---- iAudioBuffer -----------------------------------------------------------------------------------------
		class AudioBuffer : public ComObject<iAudioBuffer>
		{
		public:
			std::vector<float> mono, stereo;
			uint32_t countSamples() const override { return (uint32_t)mono.size(); }
			const float* getPcmMono() const override { return mono.empty() ? nullptr : mono.data(); }
			const float* getPcmStereo() const override { return stereo.empty() ? nullptr : stereo.data(); }
			HRESULT getTime( int64_t& rdi ) const override { rdi = 0; return S_OK; }
		};

		// ---- iAudioReader -----------------------------------------------------------------------------------------
		// The reference's readers wrap an IMFSourceReader that iContext::runStreamed pulls 10 ms chunks from
		// (Whisper/MF/PcmReader.cpp:393-430). Media Foundation does not exist here; the reader holds the decoded 16 kHz PCM
		// of a WAV file and hands it to runStreamed through a private interface (getReader keeps its slot: E_NOTIMPL).
		// {5d6b0c1e-7f0a-4f53-9f6e-3c1d2a7b9e41}
		struct iPcmSource : public ComLight::IUnknown
		{
			static constexpr ComLight::GUID iid() { return { 0x5d6b0c1e, 0x7f0a, 0x4f53, { 0x9f, 0x6e, 0x3c, 0x1d, 0x2a, 0x7b, 0x9e, 0x41 } }; }
			virtual const std::vector<float>& pcmMono() const = 0;
		};
		class WavReader : public iAudioReader, public iPcmSource
		{
			std::atomic<uint32_t> rc{ 1 };
			std::vector<float> mono;
			bool stereo;
		public:
			WavReader( std::vector<float>&& m, bool wantStereo ) : mono( std::move( m ) ), stereo( wantStereo ) {}
			virtual ~WavReader() = default;
			HRESULT QueryInterface( const ComLight::GUID& riid, void** ppv ) override
			{
				if( !ppv ) return E_POINTER;
				if( riid == iAudioReader::iid() || riid == ComLight::IID_IUnknown ) { *ppv = static_cast<iAudioReader*>( this ); AddRef(); return S_OK; }
				if( riid == iPcmSource::iid() ) { *ppv = static_cast<iPcmSource*>( this ); AddRef(); return S_OK; }
				*ppv = nullptr;
				return E_NOINTERFACE;
			}
			uint32_t AddRef() override { return ++rc; }
			uint32_t Release() override
			{
				const uint32_t r = --rc;
				if( r == 0 ) delete this;
				return r;
			}
			// 100 ns ticks, like MF_PD_DURATION (PcmReader.cpp getDuration)
			HRESULT getDuration( int64_t& rdi ) const override { rdi = (int64_t)mono.size() * 10000000ll / 16000; return S_OK; }
			HRESULT getReader( IMFSourceReader** pp ) const override { if( pp ) *pp = nullptr; return E_NOTIMPL; }
			HRESULT requestedStereo() const override { return stereo ? S_OK : S_FALSE; }
			const std::vector<float>& pcmMono() const override { return mono; }
		};

		// ---- iContext ---------------------------------------------------------------------------------------------
		class ContextImpl : public ComObject<iContext>
		{
			std::shared_ptr<LoadedModel> model;
			iModel* owner;	  // strong reference, like ContextImpl::modelPtr (ContextImpl.h:16)
			wh_context* gpu = nullptr;
			wh_context* gpuBeam = nullptr;	   // eSamplingStrategy::BeamSearch: one window x `beamSlots` hypotheses sharing its cross-attention K/V
			int beamSlots = 0;
			void* melDev = nullptr;
			int64_t melCapacity = 0;
			void* pcmDev = nullptr;
			int64_t pcmCapacity = 0;
			std::vector<Segment> resultAll;
			std::vector<int> promptPast;
			int64_t mediaTimeOffset = 0;
			mutable TranscribeResultStatic results;
			// timings, the blocks of ProfileCollection (Whisper/Utils/ProfileCollection.h)
			double msSpectrogram = 0, msEncode = 0, msDecode = 0, msRun = 0;
			int nEncode = 0, nDecodeSteps = 0, nRuns = 0, nSpectrogram = 0, nDecodeWindows = 0;
			TokenTimestamper stamper;	  // TokenTimestamps flag: token-level times + max_len wrapping (host-only post-processing)
			bool gpuProfile = false;	  // WHISPER_PROFILE=1: per-kernel hipEvent timing, printed as the "Compute Shaders" table

			using Clock = std::chrono::steady_clock;
			static double msSince( Clock::time_point t ) { return std::chrono::duration<double, std::milli>( Clock::now() - t ).count(); }

			// iSpectrogram of the reference: runFull hands the loop a whole-buffer spectrogram (global maximum), runStreamed a
			// MelStreamer that makes each window on demand with the window's own maximum (MelStreamer.cpp:125-245)
			struct MelSource
			{
				bool streamed = false;
				int64_t length = 0;			  // iSpectrogram::getLength(): 10 ms frames
				int64_t nSamples = 0, nChunks = 0;
				int64_t lastBufferEnd = -1;	  // MelStreamer::lastBufferEnd
			};
			MelSource mel;

			HRESULT ensureBuffer( void*& dev, int64_t& cap, int64_t bytes )
			{
				CHECK_WH( wh_context_bind( gpu ) );
				if( bytes <= cap ) return S_OK;
				if( dev ) { wh_buffer_free( dev ); dev = nullptr; cap = 0; }
				CHECK_WH( wh_buffer_alloc( bytes, &dev ) );
				cap = bytes;
				return S_OK;
			}
			HRESULT fillResults( eResultFlags flags, ResultData& res ) const;
			HRESULT runFullImpl( const sFullParams& params, const sProgressSink& progress );
			HRESULT encodeWindow( wh_context* ctx, int seek );
			int audioCtx = 0;	   // sFullParams::audio_ctx of the run in progress (0 = the model's)
			HRESULT beamContext( int width );
			HRESULT decodeWindowBeam( const std::vector<int>& prompt, int width, WindowScan& scan, int& steps );
			HRESULT decodeWindowBeamDevice( const std::vector<int>& prompt, int width, WindowScan& scan, int& steps );

		public:
			ContextImpl( const std::shared_ptr<LoadedModel>& m, iModel* o ) : model( m ), owner( o )
			{
				if( owner ) owner->AddRef();
			}
			~ContextImpl() override
			{
				if( gpu ) wh_context_bind( gpu );
				if( melDev ) wh_buffer_free( melDev );
				if( pcmDev ) wh_buffer_free( pcmDev );
				if( gpuBeam ) wh_context_destroy( gpuBeam );
				if( gpu ) wh_context_destroy( gpu );
				if( owner ) owner->Release();
			}
			HRESULT init()
			{
				CHECK_WH( wh_context_create( model->gpu, 1, nullptr, &gpu ) );
				// The reference profiles every dispatch all the time (GpuProfiler). Here the event pairs force eager launches
				// (no hipGraph replay), so the per-kernel table is opt-in.
				const char* const env = getenv( "WHISPER_PROFILE" );
				gpuProfile = env && env[ 0 ] && env[ 0 ] != '0';
				if( gpuProfile ) CHECK_WH( wh_profile_enable( gpu, 1 ) );
				return S_OK;
			}

			HRESULT runFull( const sFullParams& params, const iAudioBuffer* buffer ) override;
			HRESULT runStreamed( const sFullParams& params, const sProgressSink& progress, const iAudioReader* reader ) override;
			HRESULT runCapture( const sFullParams&, const sCaptureCallbacks&, const iAudioCapture* ) override { return E_NOTIMPL; }
			HRESULT getResults( eResultFlags flags, iTranscribeResult** pp ) const override;
			HRESULT detectSpeaker( const sTimeInterval&, eSpeakerChannel& result ) const override
			{
				result = eSpeakerChannel::NoStereoData;
				return S_FALSE;
			}
			HRESULT getModel( iModel** pp ) override
			{
				if( !pp ) return E_POINTER;
				if( !owner ) return E_UNEXPECTED;
				owner->AddRef();
				*pp = owner;
				return S_OK;
			}
			HRESULT fullDefaultParams( eSamplingStrategy strategy, sFullParams* rdi ) override;
			HRESULT timingsPrint() override;
			HRESULT timingsReset() override
			{
				msSpectrogram = msEncode = msDecode = msRun = 0;
				nEncode = nDecodeSteps = nRuns = nSpectrogram = nDecodeWindows = 0;
				if( gpuProfile ) wh_profile_enable( gpu, 1 );
				return S_OK;
			}
		};

		HRESULT ContextImpl::fullDefaultParams( eSamplingStrategy strategy, sFullParams* rdi )
		{
			// whisper_full_default_params as the reference restates it (ContextImpl.misc.cpp:61-93)
			if( !rdi ) return E_POINTER;
			memset( rdi, 0, sizeof( sFullParams ) );
			rdi->strategy = strategy;
			rdi->cpuThreads = 4;
			rdi->n_max_text_ctx = 16384;
			rdi->flags = eFullParamsFlags::PrintProgress | eFullParamsFlags::PrintTimestamps;
			rdi->thold_pt = rdi->thold_ptsum = 0.01f;
			rdi->language = makeLanguageKey( "en" );
			switch( strategy )
			{
			case eSamplingStrategy::Greedy:
				rdi->beam_search.n_past = rdi->beam_search.beam_width = rdi->beam_search.n_best = -1;
				return S_OK;
			case eSamplingStrategy::BeamSearch:
				rdi->greedy.n_past = -1;
				rdi->beam_search.beam_width = 10;
				rdi->beam_search.n_best = 5;
				return S_OK;
			}
			logError( "Unknown sampling strategy %i", (int)strategy );
			return E_INVALIDARG;
		}

		HRESULT ContextImpl::timingsPrint()
		{
			// Sections, block names and line format of the reference's profiler output (ProfileCollection::print,
			// ContextImpl.misc.cpp:170-182, e.g. SampleClips/columbia-medium-1080ti.txt) so that logs stay comparable.
			constexpr double ticksPerMs = 1.0e4;
			logInfo( "    CPU Tasks" );
			if( nRuns ) logInfo( "%s", formatMeasure( "RunComplete", msRun * ticksPerMs, nRuns ).c_str() );
			if( nSpectrogram ) logInfo( "%s", formatMeasure( "Spectrogram", msSpectrogram * ticksPerMs, nSpectrogram ).c_str() );
			if( nEncode ) logInfo( "%s", formatMeasure( "Encode", msEncode * ticksPerMs, nEncode ).c_str() );
			if( nDecodeWindows ) logInfo( "%s", formatMeasure( "Decode", msDecode * ticksPerMs, nDecodeWindows ).c_str() );
			if( nDecodeSteps ) logInfo( "%s", formatMeasure( "DecodeStep", msDecode * ticksPerMs, nDecodeSteps ).c_str() );
			if( gpuProfile )
			{
				// one row per kernel class, longest first -- the "Compute Shaders" table of the reference
				wh_profile_entry rows[ 32 ];
				int n = 0;
				if( 0 == wh_profile_read( gpu, rows, 32, &n ) && n > 0 )
				{
					std::sort( rows, rows + n, []( const wh_profile_entry& a, const wh_profile_entry& b ) { return a.ms > b.ms; } );
					logInfo( "    Compute Shaders" );
					for( int i = 0; i < n; i++ )
						if( rows[ i ].calls > 0 ) logInfo( "%s", formatMeasure( rows[ i ].name, rows[ i ].ms * ticksPerMs, (uint64_t)rows[ i ].calls ).c_str() );
				}
			}
			int64_t ctxVram = 0, modelVram = 0;
			void* arena = nullptr;
			wh_context_memory( gpu, &ctxVram );
			wh_model_arena( model->gpu, &arena, &modelVram );
			int64_t modelRam = 0;
			for( const std::string& t : model->vocab.idToToken ) modelRam += (int64_t)t.size() * 2 + 2 * (int64_t)sizeof( std::string );
			const int64_t ctxRam = (int64_t)( resultAll.capacity() * sizeof( Segment ) + promptPast.capacity() * sizeof( int ) );
			logInfo( "    Memory Usage" );
			logInfo( "Model\t%s RAM, %s VRAM", formatBytes( (double)modelRam ).c_str(), formatBytes( (double)modelVram ).c_str() );
			logInfo( "Context\t%s RAM, %s VRAM", formatBytes( (double)ctxRam ).c_str(), formatBytes( (double)ctxVram ).c_str() );
			logInfo( "Total\t%s RAM, %s VRAM", formatBytes( (double)( modelRam + ctxRam ) ).c_str(), formatBytes( (double)( modelVram + ctxVram ) ).c_str() );
			return S_OK;
		}

		HRESULT ContextImpl::runFull( const sFullParams& params, const iAudioBuffer* buffer )
		{
			if( !buffer ) return E_POINTER;
			const auto tRun = Clock::now();
			CHECK( buffer->getTime( mediaTimeOffset ) );
			const uint32_t n = buffer->countSamples();
			const float* pcm = buffer->getPcmMono();
			const int64_t melLen = n / 160;
			mel = MelSource{};
			mel.length = melLen;
			if( melLen > 0 )
			{
				const auto t = Clock::now();
				CHECK( ensureBuffer( pcmDev, pcmCapacity, (int64_t)n * 4 ) );
				CHECK( ensureBuffer( melDev, melCapacity, melLen * model->hp.n_mels * 4 ) );
				CHECK_WH( wh_buffer_upload( gpu, pcmDev, pcm, (int64_t)n * 4 ) );
				int64_t got = 0;
				CHECK_WH( wh_mel_spectrogram( gpu, (const float*)pcmDev, n, (float*)melDev, &got ) );
				CHECK_WH( wh_context_synchronize( gpu ) );
				msSpectrogram += msSince( t );
				nSpectrogram++;
			}
			if( params.flag( eFullParamsFlags::TokenTimestamps ) ) stamper.begin( pcm, n );
			const sProgressSink noProgress{ nullptr, nullptr };
			const HRESULT hr = runFullImpl( params, noProgress );
			msRun += msSince( tRun );
			nRuns++;
			return hr;
		}

		// ContextImpl::runStreamed (Whisper/Whisper/ContextImpl.misc.cpp:391-419): the same loop over a spectrogram that is
		// made window by window. What differs from runFull in the RESULTS is the normalisation: every window is clamped
		// against its own maximum (floor 1e-20, FP32 arithmetic) instead of the whole recording's, and a request that ends
		// where the previous one ended re-uses the previous maximum (MelStreamer.cpp:148-166). The PCM of the reader is
		// resident in HBM (a 3 h recording is 690 MB of 288 GB); only the window's frames are ever transformed.
		HRESULT ContextImpl::runStreamed( const sFullParams& params, const sProgressSink& progress, const iAudioReader* reader )
		{
			if( !reader ) return E_POINTER;
			if( params.flag( eFullParamsFlags::TokenTimestamps ) )
			{
				logError( "eFullParamsFlags.TokenTimestamps flag is not supported in streaming mode" );
				return E_NOTIMPL;
			}
			iPcmSource* src = nullptr;
			if( FAILED( const_cast<iAudioReader*>( reader )->QueryInterface( iPcmSource::iid(), (void**)&src ) ) || !src )
			{
				logError( "runStreamed: this reader was not created by iMediaFoundation::openAudioFile / loadAudioFileData of this library" );
				return E_INVALIDARG;
			}
			const auto tRun = Clock::now();
			mediaTimeOffset = 0;
			const std::vector<float>& pcm = src->pcmMono();
			mel = MelSource{};
			mel.streamed = true;
			mel.nSamples = (int64_t)pcm.size();
			mel.length = mel.nSamples / 160;			   // PcmReader::getLength(): whole 10 ms chunks (PcmReader.h:50-55)
			mel.nChunks = ( mel.nSamples + 159 ) / 160;	   // readChunk pads the last incomplete chunk with zeros (PcmReader.cpp:416-424)
			HRESULT hr = S_OK;
			if( mel.nSamples > 0 )
			{
				hr = ensureBuffer( pcmDev, pcmCapacity, mel.nSamples * 4 );
				if( SUCCEEDED( hr ) ) hr = ensureBuffer( melDev, melCapacity, (int64_t)std::max( CHUNK_FRAMES, 2 * model->hp.n_audio_ctx ) * model->hp.n_mels * 4 );	// encodeWindow asks for up to 2 * n_audio_ctx frames
				if( SUCCEEDED( hr ) && 0 != wh_buffer_upload( gpu, pcmDev, pcm.data(), mel.nSamples * 4 ) ) hr = E_FAIL;
			}
			if( SUCCEEDED( hr ) ) hr = runFullImpl( params, progress );
			src->Release();
			msRun += msSince( tRun );
			nRuns++;
			return hr;
		}

		// ContextImpl::encode( iSpectrogram&, seek ) + MelInputTensor::create (MelInputTensor.cpp:8-63)
		HRESULT ContextImpl::encodeWindow( wh_context* gpu, int seek )
		{
			const wh_hparams& hp = model->hp;
			if( !mel.streamed )
			{
				const int32_t off = seek;
				CHECK_WH( wh_encode( gpu, (const float*)melDev, 1, mel.length, mel.length * hp.n_mels, &off ) );
				return S_OK;
			}
			const int64_t i0 = std::min( (int64_t)seek, mel.length );
			const int64_t i1 = std::min( (int64_t)seek + 2 * ( audioCtx > 0 ? audioCtx : hp.n_audio_ctx ), mel.length );
			if( i1 <= i0 ) return E_BOUNDS;
			const auto t = Clock::now();
			const bool reuse = mel.lastBufferEnd == i1;
			CHECK_WH( wh_mel_spectrogram_window( gpu, (const float*)pcmDev, mel.nSamples, i0, i1 - i0, mel.nChunks, reuse ? 1 : 0, (float*)melDev ) );
			if( !reuse ) mel.lastBufferEnd = i1;
			msSpectrogram += msSince( t );	   // enqueue time only: the transform overlaps nothing else and is < 1 % of a window
			nSpectrogram++;
			const int32_t zero = 0;
			CHECK_WH( wh_encode( gpu, (const float*)melDev, 1, i1 - i0, ( i1 - i0 ) * hp.n_mels, &zero ) );
			return S_OK;
		}

		// The token stream of one window. The prompt step, the first sample (sampleTimestamp(true) of the reference) and the
		// first chunk of greedy steps are enqueued at once; the host then reads chunk k while chunk k+1 runs and, if no stop
		// rule fired on chunk k, enqueues chunk k+2 before it waits for k+1 -- the device never waits for the host, and one
		// chunk (GREEDY_CHUNK steps) is decoded in vain when a window ends.
		class WindowDecoder
		{
			wh_context* gpu;
			int nTextCtx, nPrompt = 0;
			int enqueued = 0;	  // samples enqueued so far (1 + greedy steps)
			int fetched = 0;	  // samples copied to `buf`
			std::vector<wh_token_data> buf;
			size_t cursor = 0;
			int room() const { return nTextCtx - ( nPrompt + enqueued - 1 ); }	// positions left for further greedy steps
			HRESULT enqueueChunk()
			{
				const int n = std::min( GREEDY_CHUNK, room() );
				if( n <= 0 ) return S_FALSE;
				CHECK_WH( wh_decode_window_continue( gpu, n ) );
				enqueued += n;
				return S_OK;
			}
		public:
			int steps = 0;
			double msFetch = 0, msEnqueue = 0;	   // host time waiting for samples / enqueueing further steps (WHISPER_HOSTPROF)
			WindowDecoder( wh_context* c, int nTextCtx_ ) : gpu( c ), nTextCtx( nTextCtx_ ) {}
			HRESULT start( const std::vector<int>& prompt, TokenData& first )
			{
				nPrompt = (int)prompt.size();
				const int n0 = std::max( 0, std::min( GREEDY_CHUNK, nTextCtx - nPrompt ) );
				CHECK_WH( wh_decode_window_start( gpu, 1, prompt.data(), nPrompt, n0, 1, 1 ) );
				enqueued = 1 + n0;
				buf.resize( 1 );
				CHECK_WH( wh_decode_window_fetch( gpu, 0, 1, buf.data() ) );
				fetched = 1;
				cursor = 1;
				const wh_token_data& t = buf[ 0 ];
				first.id = t.id; first.tid = t.tid; first.p = t.p; first.pt = t.pt; first.ptsum = t.ptsum;
				steps = 1;
				return S_OK;
			}
			HRESULT next( TokenData& out )
			{
				if( cursor >= buf.size() )
				{
					// The caller has looked at everything handed out so far and wants more: NOW the chunk after the one in flight is
					// queued (the one in flight has been running since the previous fetch returned, so the device does not wait
					// for this), then the chunk in flight is awaited. A window that ends on the token just examined therefore
					// leaves ONE chunk decoded in vain, not two.
					const auto t0 = std::chrono::steady_clock::now();
					CHECK( enqueueChunk() );
					const auto t1 = std::chrono::steady_clock::now();
					if( fetched >= enqueued ) return E_BOUNDS;
					// the chunk that follows what has been read: everything up to the next chunk boundary
					const int n = std::min( GREEDY_CHUNK, enqueued - fetched );
					buf.resize( (size_t)n );
					CHECK_WH( wh_decode_window_fetch( gpu, fetched, n, buf.data() ) );
					fetched += n;
					cursor = 0;
					msEnqueue += std::chrono::duration<double, std::milli>( t1 - t0 ).count();
					msFetch += std::chrono::duration<double, std::milli>( std::chrono::steady_clock::now() - t1 ).count();
				}
				const wh_token_data& t = buf[ cursor++ ];
				out.id = t.id; out.tid = t.tid; out.p = t.p; out.pt = t.pt; out.ptsum = t.ptsum;
				steps++;
				return S_OK;
			}
		};

		// A context of one window x `slots` hypotheses (1, 2, 3, 4, 5 or 8 per window: wh_context_create_hyp)
		HRESULT ContextImpl::beamContext( int width )
		{
			const int slots = width <= 5 ? width : 8;
			if( gpuBeam && beamSlots == slots ) return S_OK;
			if( gpuBeam ) { wh_context_destroy( gpuBeam ); gpuBeam = nullptr; }
			CHECK_WH( wh_context_create_hyp( model->gpu, 1, slots, nullptr, &gpuBeam ) );
			beamSlots = slots;
			return S_OK;
		}

		// Beam search over one window (extension; the reference only declares the strategy). `width` hypotheses decode in lock step and
		// share the window's cross-attention K/V (wh_context_create_hyp). A step: every live hypothesis proposes its `width` best
		// continuations under sampleBest's own rules (wh_beam_candidates: candidate 0 is the greedy token), the pool is ranked by
		// cumulative log-probability, the best `width` survive -- each either stays live (its parent's self-attention cache rows are
		// copied into its slot, wh_reorder_self_cache) or, when the reference's stop rules end its window (WindowScan: EOT, a timestamp
		// that goes back, end of audio, max_tokens ...), joins the finished list. The window's transcript is the finished hypothesis with
		// the best log-probability per token; hypotheses the stop rules call failed count only if nothing else finished.
		// Width 1 is the greedy decoder token for token (tests/test_host_api.py::test_beam_search).
		HRESULT ContextImpl::decodeWindowBeam( const std::vector<int>& prompt, int width, WindowScan& result, int& steps )
		{
			const int B = beamSlots, n = (int)prompt.size();
			struct Hyp
			{
				std::unique_ptr<WindowScan> scan;
				double sum = 0.0;
				int slot = 0;
			};
			std::vector<Hyp> live, finished;
			std::vector<int32_t> tokens( (size_t)B * std::max( n, 1 ) ), parents( (size_t)B, 0 ), next( (size_t)B, 0 );
			for( int b = 0; b < B; b++ ) std::copy( prompt.begin(), prompt.end(), tokens.begin() + (size_t)b * n );
			CHECK_WH( wh_decode( gpuBeam, tokens.data(), B, n, 0, nullptr, nullptr ) );
			std::vector<wh_token_data> cand( (size_t)B * width );
			CHECK_WH( wh_beam_candidates( gpuBeam, B, width, 1, 1, cand.data() ) );
			steps = 1;
			// what a parent proposes: (parent index in `live`, candidate) ranked by parent score + log p; ties keep the parent's order
			struct Prop { int parent, k; double score; };
			auto expand = [ & ]( const std::vector<Hyp>& parentsHyp, const std::vector<Prop>& pool, std::vector<Hyp>& newLive )
			{
				int accepted = 0;
				for( const Prop& pr : pool )
				{
					if( accepted >= width ) break;
					const Hyp& par = parentsHyp[ (size_t)pr.parent ];
					const wh_token_data& t = cand[ (size_t)par.slot * width + pr.k ];
					Hyp h;
					h.scan.reset( new WindowScan( *par.scan ) );
					h.sum = pr.score;
					h.slot = par.slot;	 // its cache rows live in the parent's slot until the reorder
					TokenData td;
					td.id = t.id; td.tid = t.tid; td.p = t.p; td.pt = t.pt; td.ptsum = t.ptsum;
					const bool over = h.scan->feed( td );
					accepted++;
					if( over ) finished.push_back( std::move( h ) );
					else newLive.push_back( std::move( h ) );
				}
			};
			{
				// the first sample: every slot holds the same prompt, slot 0 speaks for all
				std::vector<Hyp> root( 1 );
				root[ 0 ].scan.reset( new WindowScan( result ) );
				root[ 0 ].slot = 0;
				std::vector<Prop> pool;
				for( int k = 0; k < width; k++ ) pool.push_back( Prop{ 0, k, log( std::max( (double)cand[ (size_t)k ].p, 1e-30 ) ) } );
				std::stable_sort( pool.begin(), pool.end(), []( const Prop& a, const Prop& b ) { return a.score > b.score; } );
				expand( root, pool, live );
			}
			// The search goes on while a live hypothesis could still win: a finished list that is full ends it only once every live
			// hypothesis has fallen below the best finished one (a cumulative log-probability only ever decreases). Low-probability
			// continuations that end their window at once (an EOT proposed with p ~ 1e-5) fill the list early and must not stop the
			// hypothesis that is still being transcribed.
			auto keepSearching = [ & ]() -> bool
			{
				if( live.empty() ) return false;
				if( (int)finished.size() < width ) return true;
				double bestFinished = -1e300, bestLive = -1e300;
				for( const Hyp& h : finished )
					if( !h.scan->failed ) bestFinished = std::max( bestFinished, h.sum );
				for( const Hyp& h : live ) bestLive = std::max( bestLive, h.sum );
				return bestLive >= bestFinished;
			};
			auto perToken = []( const Hyp& h ) { return h.sum / (double)std::max<size_t>( 1, h.scan->tokens.size() ); };
			for( int s = 0; keepSearching(); s++ )
			{
				// the finished list keeps its best `width` entries (successful windows first, then log-probability per token)
				if( (int)finished.size() > 2 * width )
				{
					std::stable_sort( finished.begin(), finished.end(), [ & ]( const Hyp& a, const Hyp& b )
						{ return a.scan->failed != b.scan->failed ? !a.scan->failed : perToken( a ) > perToken( b ); } );
					finished.resize( (size_t)width );
				}
				if( n + s >= model->hp.n_text_ctx ) break;	   // WindowScan's own bound (n_text_ctx / 2 - 4 tokens) fires first for every prompt the loop builds
				// live hypothesis i moves to slot i; the idle slots repeat slot 0 and are ignored
				for( int b = 0; b < B; b++ )
				{
					const Hyp& h = live[ (size_t)( b < (int)live.size() ? b : 0 ) ];
					parents[ (size_t)b ] = h.slot;
					next[ (size_t)b ] = h.scan->tokens.back().id;
				}
				// a hypothesis that stopped on a token WindowScan did not keep (a timestamp going back) never reaches this point: it is finished
				CHECK_WH( wh_reorder_self_cache( gpuBeam, B, parents.data(), n + s ) );
				for( size_t i = 0; i < live.size(); i++ ) live[ i ].slot = (int)i;
				CHECK_WH( wh_decode( gpuBeam, next.data(), B, 1, n + s, nullptr, nullptr ) );
				CHECK_WH( wh_beam_candidates( gpuBeam, B, width, 0, 0, cand.data() ) );
				steps++;
				std::vector<Prop> pool;
				for( int i = 0; i < (int)live.size(); i++ )
					for( int k = 0; k < width; k++ )
						pool.push_back( Prop{ i, k, live[ (size_t)i ].sum + log( std::max( (double)cand[ (size_t)live[ (size_t)i ].slot * width + k ].p, 1e-30 ) ) } );
				std::stable_sort( pool.begin(), pool.end(), []( const Prop& a, const Prop& b ) { return a.score > b.score; } );
				std::vector<Hyp> newLive;
				expand( live, pool, newLive );
				live = std::move( newLive );
			}
			// hypotheses still live when the list filled up (or the context ran out) are candidates too: they end where they stand, as failed
			// windows do when the loop's bound is reached
			for( Hyp& h : live )
			{
				if( !h.scan->over ) h.scan->failed = h.scan->over = true;
				finished.push_back( std::move( h ) );
			}
			const Hyp* best = nullptr;
			for( const Hyp& h : finished )
				if( !best || ( best->scan->failed && !h.scan->failed ) || ( best->scan->failed == h.scan->failed && perToken( h ) > perToken( *best ) ) ) best = &h;
			if( !best ) return E_UNEXPECTED;
			result.adopt( *best->scan );
			return S_OK;
		}

		// The same search with the ranking ON THE DEVICE (round 5): wh_beam_window_* keeps pool, ranking, stop rules, finished / live lists and the
		// "can a live hypothesis still win" test in a kernel behind every decode step, the whole step is one captured graph, and this loop only polls
		// `done` every 16 steps -- a window of ~50 tokens costs 4 host round trips instead of 100. What comes back: the search state (who finished with
		// which score) and one record per accepted proposal; the winner is chosen by decodeWindowBeam's own rule and its tokens are REPLAYED through the
		// window's WindowScan, which both builds the result and checks the device's restatement of the stop rules against the host's.
		HRESULT ContextImpl::decodeWindowBeamDevice( const std::vector<int>& prompt, int width, WindowScan& result, int& steps )
		{
			const int n = (int)prompt.size(), nCtx = model->hp.n_text_ctx;
			wh_beam_rules rules{};
			bool single = false;
			{
				int seek = 0, seekEnd = 0, nMax = 0, maxTokens = 0;
				result.constants( seek, seekEnd, nMax, maxTokens, single );
				rules.seek = seek; rules.seekEnd = seekEnd; rules.nMax = nMax; rules.maxTokens = maxTokens;
			}
			rules.singleSegment = single ? 1 : 0;
			rules.tokenBeg = model->vocab.token_beg;
			rules.tokenEot = model->vocab.token_eot;
			rules.forced = 0;
			constexpr int CHUNK = 16;
			int enqueued = std::max( 0, std::min( CHUNK - 1, nCtx - n ) );
			std::vector<int32_t> tokens( prompt.begin(), prompt.end() );
			CHECK_WH( wh_beam_window_start( gpuBeam, 1, tokens.data(), n, width, &rules, enqueued ) );
			wh_beam_window st{};
			while( true )
			{
				CHECK_WH( wh_beam_window_status( gpuBeam, &st ) );
				if( st.done ) break;
				const int more = std::min( CHUNK, nCtx - n - enqueued );
				if( more <= 0 ) break;
				CHECK_WH( wh_beam_window_continue( gpuBeam, more ) );
				enqueued += more;
			}
			steps = st.step;
			if( st.step <= 0 ) return E_UNEXPECTED;
			std::vector<wh_beam_record> rec( (size_t)st.step * width );
			CHECK_WH( wh_beam_window_records( gpuBeam, 0, st.step, rec.data() ) );
			// the candidates in decodeWindowBeam's order: the finished list, then whoever was still live (those end where they stand, as failed windows)
			struct Cand { wh_beam_hyp h; bool wasLive; };
			std::vector<Cand> cands;
			for( int i = 0; i < st.nFinished; i++ ) cands.push_back( Cand{ st.finished[ i ], false } );
			for( int i = 0; i < st.nLive; i++ )
			{
				Cand c{ st.live[ i ], true };
				if( !c.h.over ) c.h.failed = c.h.over = 1;
				cands.push_back( c );
			}
			auto perToken = []( const wh_beam_hyp& h ) { return h.sum / (double)std::max( 1, h.nTok ); };
			const Cand* best = nullptr;
			for( const Cand& c : cands )
				if( !best || ( best->h.failed && !c.h.failed ) || ( ( best->h.failed != 0 ) == ( c.h.failed != 0 ) && perToken( c.h ) > perToken( best->h ) ) ) best = &c;
			if( !best ) return E_UNEXPECTED;
			// its tokens, oldest first
			std::vector<const wh_beam_record*> chain;
			for( int r = best->h.rec; r >= 0; )
			{
				if( r >= st.step * width ) return E_UNEXPECTED;
				const wh_beam_record& x = rec[ (size_t)r ];
				chain.push_back( &x );
				r = x.parent;
			}
			WindowScan replay( result );
			for( size_t k = chain.size(); k-- > 0; )
			{
				const wh_beam_record& x = *chain[ k ];
				TokenData td;
				td.id = x.t.id; td.tid = x.t.tid; td.p = x.t.p; td.pt = x.t.pt; td.ptsum = x.t.ptsum;
				replay.feed( td );
			}
			if( best->wasLive && !replay.over ) replay.failed = replay.over = true;
			if( replay.failed != ( best->h.failed != 0 ) || (int)replay.tokens.size() != best->h.nTok || replay.resultLen != best->h.resultLen )
			{
				logError( "beam search: the device's stop rules and the host's disagree on the winning hypothesis (failed %d / %d, tokens %d / %d, resultLen %d / %d)",
					(int)replay.failed, best->h.failed, (int)replay.tokens.size(), best->h.nTok, replay.resultLen, best->h.resultLen );
				return E_UNEXPECTED;
			}
			result.adopt( replay );
			return S_OK;
		}

		HRESULT ContextImpl::runFullImpl( const sFullParams& params, const sProgressSink& progress )
		{
			// the stream's rules (seek range, prompt carry-over, stop rules, segments, callbacks) live in hostLoop.h, shared with the
			// lock-step scheduler of runFullBatch; here: the device work of ONE stream, window after window
			const Vocabulary& vocab = model->vocab;
			const wh_hparams& hp = model->hp;
			StreamRun run( params, vocab, hp, this, progress, resultAll, promptPast, &stamper );
			const HRESULT hrBegin = run.begin( mel.length );
			if( hrBegin != S_OK ) return hrBegin;
			// eSamplingStrategy::BeamSearch (declared by the reference, implemented only here: sFullParams.h:10-13): beam_width hypotheses per
			// window through decodeWindowBeam (width 1 included: there it must reproduce the greedy loop token for token). The Greedy strategy
			// takes the device-side greedy loop.
			int beamWidth = 0;
			if( params.strategy == eSamplingStrategy::BeamSearch && params.beam_search.beam_width >= 1 )
			{
				beamWidth = params.beam_search.beam_width;
				if( beamWidth > 8 )
				{
					logWarning( "runFull: beam_width %d is more than this build decodes in lock step; using 8", beamWidth );
					beamWidth = 8;
				}
				CHECK( beamContext( beamWidth ) );
			}
			// the ONE device context this run works on: the window x hypotheses context of a beam search, else the stream's own
			wh_context* const active = beamWidth >= 1 ? gpuBeam : gpu;
			// overwrite audio_ctx (ContextImpl.cpp:488-489): encoder positions and cross-attention keys of every window of this run
			audioCtx = params.audio_ctx;
			CHECK_WH( wh_context_set_audio_ctx( active, audioCtx ) );
			std::vector<int> prompt;
			while( true )
			{
				const HRESULT hrNext = run.nextWindow( prompt );
				if( FAILED( hrNext ) ) return hrNext;
				if( hrNext != S_OK ) break;
				{
					// enqueued without a host sync: the decoder's launches line up behind the encoder's on the context's stream.
					// With WHISPER_PROFILE=1 the two are separated so that the "Encode" block means what it means in the reference.
					const auto t = Clock::now();
					CHECK( encodeWindow( active, run.seek ) );
					if( gpuProfile ) CHECK_WH( wh_context_synchronize( active ) );	   // the context the encoder was queued on ("Encode" measures the encoder, not its enqueue)
					msEncode += msSince( t );
					nEncode++;
				}
				const auto tDec = Clock::now();
				WindowDecoder dec( active, hp.n_text_ctx );
				WindowScan scan( run.fullParams(), vocab, run.seek, run.seekEnd(), run.maxTokens() );
				if( beamWidth >= 1 )
					CHECK( g_beamRankingOnHost ? decodeWindowBeam( prompt, beamWidth, scan, dec.steps ) : decodeWindowBeamDevice( prompt, beamWidth, scan, dec.steps ) );
				else
				for( bool first = true; !scan.over; first = false )
				{
					TokenData token;
					if( first )
						CHECK( dec.start( prompt, token ) );
					else
						CHECK( dec.next( token ) );
					scan.feed( token );
				}
				msDecode += msSince( tDec );
				if( getenv( "WHISPER_HOSTPROF" ) )
					fprintf( stderr, "[hostprof] window at %d: decode %.2f ms for %d tokens, of which waiting for samples %.2f ms, enqueueing steps %.2f ms\n", run.seek,
						msSince( tDec ), dec.steps, dec.msFetch, dec.msEnqueue );
				nDecodeSteps += dec.steps;	   // tokens the loop consumed (the reference counts one DecodeStep per token)
				nDecodeWindows++;
				CHECK( run.finishWindow( scan ) );
			}
			return run.end();
		}

		HRESULT ContextImpl::fillResults( eResultFlags flags, ResultData& res ) const
		{
			return fillResultData( resultAll, model->vocab, mediaTimeOffset, flags, res );
		}

		HRESULT ContextImpl::getResults( eResultFlags flags, iTranscribeResult** pp ) const
		{
			if( !pp ) return E_POINTER;
			if( flags & eResultFlags::NewObject )
			{
				TranscribeResult* r = new TranscribeResult();
				const HRESULT hr = fillResults( flags, *r );
				if( FAILED( hr ) ) { r->Release(); return hr; }
				*pp = r;
				return S_OK;
			}
			CHECK( fillResults( flags, results ) );
			*pp = &results;
			return S_OK;
		}

		// ---- iModel -----------------------------------------------------------------------------------------------
		class ModelImpl : public ComObject<iModel>, public iModelInternals
		{
			std::shared_ptr<LoadedModel> model;
		public:
			explicit ModelImpl( const std::shared_ptr<LoadedModel>& m ) : model( m ) {}
			HRESULT QueryInterface( const ComLight::GUID& riid, void** ppv ) override
			{
				if( ppv && riid == iModelInternals::iid() )
				{
					*ppv = static_cast<iModelInternals*>( this );
					ComObject<iModel>::AddRef();
					return S_OK;
				}
				return ComObject<iModel>::QueryInterface( riid, ppv );
			}
			uint32_t AddRef() override { return ComObject<iModel>::AddRef(); }
			uint32_t Release() override { return ComObject<iModel>::Release(); }
			const std::shared_ptr<LoadedModel>& loaded() const override { return model; }
			HRESULT createContext( iContext** pp ) override { return createContextImpl( model, this, pp ); }
			HRESULT tokenize( const char* text, pfnDecodedTokens pfn, void* pv ) override
			{
				if( !pfn ) return E_POINTER;
				std::vector<int> toks;
				CHECK( model->vocab.tokenize( text, toks ) );
				pfn( toks.empty() ? nullptr : toks.data(), (int)toks.size(), pv );
				return S_OK;
			}
			HRESULT isMultilingual() override { return model->vocab.isMultilingual() ? S_OK : S_FALSE; }
			HRESULT getSpecialTokens( SpecialTokens& r ) override
			{
				const Vocabulary& v = model->vocab;
				r.TranscriptionEnd = v.token_eot; r.TranscriptionStart = v.token_sot; r.PreviousWord = v.token_prev;
				r.SentenceStart = v.token_solm; r.Not = v.token_not; r.TranscriptionBegin = v.token_beg;
				r.TaskTranslate = v.token_translate; r.TaskTranscribe = v.token_transcribe;
				return S_OK;
			}
			const char* stringFromToken( whisper_token token ) override { return model->vocab.string( token ); }
			// Weights are immutable and shared (WhisperModel.h:28-30): a clone is another handle on the same arena, so that
			// a second context can run from another host thread (ModelImpl.cpp:40-60 needs OpenSharedResource for this).
			HRESULT clone( iModel** rdi ) override { return createModelImpl( model, rdi ); }
		};

		// ---- iMediaFoundation stand-in: WAV (PCM16 / float32, 16 kHz) ------------------------------------------------
		class WavLoader : public ComObject<iMediaFoundation>
		{
		public:
			HRESULT loadAudioFile( LPCTSTR path, bool stereo, iAudioBuffer** pp ) const override;
			HRESULT openAudioFile( LPCTSTR path, bool stereo, iAudioReader** pp ) override;
			HRESULT loadAudioFileData( const void* data, uint64_t size, bool stereo, iAudioReader** pp ) override;
			HRESULT listCaptureDevices( pfnFoundCaptureDevices, void* ) override { return E_NOTIMPL; }
			HRESULT openCaptureDevice( LPCTSTR, const sCaptureParams&, iAudioCapture** ) override { return E_NOTIMPL; }
		};

		// RIFF/WAVE, 16 kHz, mono or stereo, PCM16 or float32 -> mono (and interleaved stereo when asked for)
		static HRESULT decodeWav( const char* bytes, size_t size, const std::string& what, bool wantStereo, std::vector<float>& mono, std::vector<float>& st )
		{
			if( size < 44 || memcmp( bytes, "RIFF", 4 ) || memcmp( bytes + 8, "WAVE", 4 ) )
			{
				logError( "'%s' is not a RIFF/WAVE file (only WAV is supported on this platform)", what.c_str() );
				return E_INVALIDARG;
			}
			uint16_t fmt = 0, channels = 0, bits = 0;
			uint32_t rate = 0;
			const char* pcm = nullptr;
			size_t pcmBytes = 0;
			for( size_t o = 12; o + 8 <= size; )
			{
				uint32_t len;
				memcpy( &len, bytes + o + 4, 4 );
				const char* body = bytes + o + 8;
				if( !memcmp( bytes + o, "fmt ", 4 ) && len >= 16 )
				{
					memcpy( &fmt, body, 2 ); memcpy( &channels, body + 2, 2 ); memcpy( &rate, body + 4, 4 ); memcpy( &bits, body + 14, 2 );
				}
				else if( !memcmp( bytes + o, "data", 4 ) )
				{
					pcm = body;
					pcmBytes = std::min( (size_t)len, size - ( o + 8 ) );
				}
				o += 8 + (size_t)len + ( len & 1 );
			}
			const bool isFloat = fmt == 3 && bits == 32, isPcm16 = fmt == 1 && bits == 16;
			if( !pcm || !( isFloat || isPcm16 ) || channels < 1 || channels > 2 || rate != 16000 )
			{
				logError( "'%s': need 16 kHz mono/stereo PCM16 or float32 WAV (got format %u, %u bit, %u ch, %u Hz)", what.c_str(), fmt, bits, channels, rate );
				return E_INVALIDARG;
			}
			const size_t frames = pcmBytes / ( ( bits / 8 ) * channels );
			mono.resize( frames );
			st.clear();
			if( wantStereo ) st.resize( frames * 2 );
			auto sample = [ & ]( size_t i ) -> float
			{
				if( isFloat ) { float v; memcpy( &v, pcm + i * 4, 4 ); return v; }
				int16_t v; memcpy( &v, pcm + i * 2, 2 ); return (float)v / 32768.0f;
			};
			for( size_t i = 0; i < frames; i++ )
			{
				const float l = sample( i * channels ), r = channels == 2 ? sample( i * 2 + 1 ) : l;
				mono[ i ] = channels == 2 ? 0.5f * ( l + r ) : l;
				if( wantStereo ) { st[ 2 * i ] = l; st[ 2 * i + 1 ] = r; }
			}
			return S_OK;
		}
		static HRESULT readWavFile( LPCTSTR path, bool stereo, std::vector<float>& mono, std::vector<float>& st )
		{
			const std::string p = path;	  // LPCTSTR is UTF-8 off Windows (ComLightLib/comLightCommon.h:8)
			std::ifstream f( p, std::ios::binary );
			if( !f ) { logError( "failed to open audio file '%s'", p.c_str() ); return (HRESULT)0x80070002; }
			std::vector<char> data( ( std::istreambuf_iterator<char>( f ) ), std::istreambuf_iterator<char>() );
			return decodeWav( data.data(), data.size(), p, stereo, mono, st );
		}

		HRESULT WavLoader::loadAudioFile( LPCTSTR path, bool stereo, iAudioBuffer** pp ) const
		{
			if( !pp || !path ) return E_POINTER;
			std::vector<float> mono, st;
			CHECK( readWavFile( path, stereo, mono, st ) );
			return createAudioBuffer( std::move( mono ), std::move( st ), pp );
		}
		HRESULT WavLoader::openAudioFile( LPCTSTR path, bool stereo, iAudioReader** pp )
		{
			if( !pp || !path ) return E_POINTER;
			std::vector<float> mono, st;
			CHECK( readWavFile( path, false, mono, st ) );
			*pp = new WavReader( std::move( mono ), stereo );
			return S_OK;
		}
		HRESULT WavLoader::loadAudioFileData( const void* data, uint64_t size, bool stereo, iAudioReader** pp )
		{
			if( !pp || !data ) return E_POINTER;
			std::vector<float> mono, st;
			CHECK( decodeWav( (const char*)data, (size_t)size, "<memory>", false, mono, st ) );
			*pp = new WavReader( std::move( mono ), stereo );
			return S_OK;
		}
	}	// namespace

	HRESULT createContextImpl( const std::shared_ptr<LoadedModel>& model, iModel* owner, iContext** pp )
	{
		if( !pp ) return E_POINTER;
		ContextImpl* c = new ContextImpl( model, owner );
		const HRESULT hr = c->init();
		if( FAILED( hr ) ) { c->Release(); return hr; }
		*pp = c;
		return S_OK;
	}
	HRESULT createModelImpl( const std::shared_ptr<LoadedModel>& model, iModel** pp )
	{
		if( !pp ) return E_POINTER;
		*pp = new ModelImpl( model );
		return S_OK;
	}
	HRESULT createAudioBuffer( std::vector<float>&& mono, std::vector<float>&& stereo, iAudioBuffer** pp )
	{
		if( !pp ) return E_POINTER;
		AudioBuffer* b = new AudioBuffer();
		b->mono = std::move( mono );
		b->stereo = std::move( stereo );
		*pp = b;
		return S_OK;
	}

	// ---- exports ----------------------------------------------------------------------------------------------------
	static HRESULT loadModelImpl( const wchar_t* path, const sModelSetup& setup, const sLoadModelCallbacks* callbacks, wh_comm* comm, int root, iModel** pp );
	HRESULT loadModel( const wchar_t* path, const sModelSetup& setup, const sLoadModelCallbacks* callbacks, iModel** pp )
	{
		return loadModelImpl( path, setup, callbacks, nullptr, 0, pp );
	}
	// One process per GPU (extension; the reference has one GPU per model, ModelImpl.cpp:40-60): every rank calls this with its
	// own sModelSetup.adapter and the communicator of include/whisper_hip.h; rank `root` reads the file, the others receive the
	// weights over xGMI (RCCL broadcast) and read only the file's header.
	HRESULT loadModelShared( const wchar_t* path, const sModelSetup& setup, const sLoadModelCallbacks* callbacks, void* whComm, int root, iModel** pp )
	{
		if( !whComm ) return E_POINTER;
		return loadModelImpl( path, setup, callbacks, (wh_comm*)whComm, root, pp );
	}
	static HRESULT loadModelImpl( const wchar_t* path, const sModelSetup& setup, const sLoadModelCallbacks* callbacks, wh_comm* comm, int root, iModel** pp )
	{
		if( !path || !pp ) return E_POINTER;
		if( setup.impl != eModelImplementation::GPU )
		{
			// Hybrid is compiled out in the reference as well (stdafx.h:34); Reference is the vendored CPU model, which this
			// repository keeps as its test oracle (oracle/) and never links into the product.
			logError( "loadModel: only eModelImplementation::GPU is available in this build" );
			return E_NOTIMPL;
		}
		int device = 0;
		if( setup.adapter && *setup.adapter )
		{
			// What listGPUs hands out is "<index>: <name>"; the reference matches the adapter by its listed name
			// (Whisper/D3D/createDevice.cpp). Accepted: the full listed string, the name alone, or the bare index ("2" / "2:").
			const std::string a = utf8( setup.adapter );
			const int nDev = wh_device_count();
			device = -1;
			size_t digits = 0;
			while( digits < a.size() && isdigit( (unsigned char)a[ digits ] ) ) digits++;
			if( digits > 0 && ( digits == a.size() || ( digits + 1 == a.size() && a[ digits ] == ':' ) ) )
				device = atoi( a.c_str() );
			else
				for( int i = 0; i < nDev && device < 0; i++ )
				{
					char name[ 256 ];
					if( 0 != wh_device_info( i, name, sizeof( name ), nullptr, nullptr ) ) continue;
					if( a == std::to_string( i ) + ": " + name || a == name ) device = i;
				}
			if( device < 0 || device >= nDev )
			{
				logError( "loadModel: adapter '%s' not found (see listGPUs)", a.c_str() );
				return E_INVALIDARG;
			}
		}
		std::shared_ptr<LoadedModel> lm;
		CHECK( loadGgmlFile( utf8( path ), device, callbacks, lm, comm, root ) );
		return createModelImpl( lm, pp );
	}

	HRESULT initMediaFoundation( iMediaFoundation** pp )
	{
		if( !pp ) return E_POINTER;
		*pp = new WavLoader();
		return S_OK;
	}
}

// ---- flat C mirror for FFI callers (ctypes / cgo / JNI): see include/whisper_c.h -----------------------------------
extern "C" {
using namespace Whisper;

WHISPER_EXPORT int32_t whisperc_load_model( const char* pathUtf8, int device, void** modelOut )
{
	if( !pathUtf8 || !modelOut ) return E_POINTER;
	std::wstring w;
	for( const unsigned char* p = (const unsigned char*)pathUtf8; *p; p++ ) w += (wchar_t)*p;	// paths used by the tests are ASCII
	std::wstring adapter = std::to_wstring( device ) + L":";
	sModelSetup setup;
	setup.adapter = adapter.c_str();
	iModel* m = nullptr;
	const HRESULT hr = loadModel( w.c_str(), setup, nullptr, &m );
	*modelOut = m;
	return hr;
}
WHISPER_EXPORT void whisperc_release( void* unknown )
{
	if( unknown ) ( (ComLight::IUnknown*)unknown )->Release();
}
WHISPER_EXPORT int32_t whisperc_create_context( void* model, void** ctxOut )
{
	if( !model || !ctxOut ) return E_POINTER;
	iContext* c = nullptr;
	const HRESULT hr = ( (iModel*)model )->createContext( &c );
	*ctxOut = c;
	return hr;
}
WHISPER_EXPORT int32_t whisperc_special_tokens( void* model, int32_t* out8 )
{
	SpecialTokens st;
	const HRESULT hr = ( (iModel*)model )->getSpecialTokens( st );
	memcpy( out8, &st, sizeof( st ) );
	return hr;
}
WHISPER_EXPORT const char* whisperc_token_string( void* model, int token ) { return ( (iModel*)model )->stringFromToken( token ); }
WHISPER_EXPORT int32_t whisperc_is_multilingual( void* model ) { return ( (iModel*)model )->isMultilingual(); }
// flags: eFullParamsFlags bits; language: ASCII code; returns the HRESULT of runFull
WHISPER_EXPORT int32_t whisperc_run_full( void* ctx, const float* pcm, uint32_t nSamples, const char* language, uint32_t flags, int maxTokens,
	const int32_t* promptTokens, int nPrompt, int nMaxTextCtx )
{
	if( !ctx || ( !pcm && nSamples ) ) return E_POINTER;
	iContext* c = (iContext*)ctx;
	sFullParams p;
	CHECK( c->fullDefaultParams( eSamplingStrategy::Greedy, &p ) );
	p.flags = (eFullParamsFlags)flags;
	p.language = makeLanguageKey( language ? language : "en" );
	p.max_tokens = maxTokens;
	p.prompt_tokens = promptTokens;
	p.prompt_n_tokens = nPrompt;
	if( nMaxTextCtx >= 0 ) p.n_max_text_ctx = nMaxTextCtx;
	iAudioBuffer* buf = nullptr;
	CHECK( createAudioBuffer( std::vector<float>( pcm, pcm + nSamples ), {}, &buf ) );
	const HRESULT hr = c->runFull( p, buf );
	buf->Release();
	return hr;
}
WHISPER_EXPORT int32_t whisperc_run_full_audio_ctx( void* ctx, const float* pcm, uint32_t nSamples, const char* language, uint32_t flags, int maxTokens,
	const int32_t* promptTokens, int nPrompt, int nMaxTextCtx, int audioCtx )
{
	if( !ctx || ( !pcm && nSamples ) ) return E_POINTER;
	iContext* c = (iContext*)ctx;
	sFullParams p;
	CHECK( c->fullDefaultParams( eSamplingStrategy::Greedy, &p ) );
	p.flags = (eFullParamsFlags)flags;
	p.language = makeLanguageKey( language ? language : "en" );
	p.max_tokens = maxTokens;
	p.prompt_tokens = promptTokens;
	p.prompt_n_tokens = nPrompt;
	p.audio_ctx = audioCtx;
	if( nMaxTextCtx >= 0 ) p.n_max_text_ctx = nMaxTextCtx;
	iAudioBuffer* buf = nullptr;
	CHECK( createAudioBuffer( std::vector<float>( pcm, pcm + nSamples ), {}, &buf ) );
	const HRESULT hr = c->runFull( p, buf );
	buf->Release();
	return hr;
}
// initMediaFoundation -> loadAudioFileData( WAV bytes ) -> iContext::runStreamed; progress values are appended to
// progressOut (up to progressCap), their count is returned through progressCount
WHISPER_EXPORT int32_t whisperc_run_streamed( void* ctx, const void* wavBytes, uint64_t wavSize, const char* language, uint32_t flags, int maxTokens,
	const int32_t* promptTokens, int nPrompt, int nMaxTextCtx, double* progressOut, int progressCap, int* progressCount )
{
	if( !ctx || !wavBytes ) return E_POINTER;
	iContext* c = (iContext*)ctx;
	sFullParams p;
	CHECK( c->fullDefaultParams( eSamplingStrategy::Greedy, &p ) );
	p.flags = (eFullParamsFlags)flags;
	p.language = makeLanguageKey( language ? language : "en" );
	p.max_tokens = maxTokens;
	p.prompt_tokens = promptTokens;
	p.prompt_n_tokens = nPrompt;
	if( nMaxTextCtx >= 0 ) p.n_max_text_ctx = nMaxTextCtx;
	iMediaFoundation* mf = nullptr;
	CHECK( initMediaFoundation( &mf ) );
	iAudioReader* reader = nullptr;
	HRESULT hr = mf->loadAudioFileData( wavBytes, wavSize, false, &reader );
	mf->Release();
	if( FAILED( hr ) ) return hr;
	struct Sink { double* out; int cap, n; } sink{ progressOut, progressCap, 0 };
	sProgressSink ps;
	ps.pfn = []( double v, iContext*, void* pv ) noexcept -> HRESULT
	{
		Sink* s = (Sink*)pv;
		if( s->out && s->n < s->cap ) s->out[ s->n ] = v;
		s->n++;
		return S_OK;
	};
	ps.pv = &sink;
	hr = c->runStreamed( p, ps, reader );
	reader->Release();
	if( progressCount ) *progressCount = sink.n;
	return hr;
}
// iContext::fullDefaultParams( BeamSearch ) + beam_search.beam_width = beamWidth (1 .. 8), then iContext::runFull
WHISPER_EXPORT int32_t whisperc_run_full_beam( void* ctx, const float* pcm, uint32_t nSamples, const char* language, uint32_t flags, int maxTokens,
	const int32_t* promptTokens, int nPrompt, int nMaxTextCtx, int beamWidth )
{
	if( !ctx || ( !pcm && nSamples ) ) return E_POINTER;
	iContext* c = (iContext*)ctx;
	sFullParams p;
	CHECK( c->fullDefaultParams( eSamplingStrategy::BeamSearch, &p ) );
	p.beam_search.beam_width = beamWidth;
	p.flags = (eFullParamsFlags)flags;
	p.language = makeLanguageKey( language ? language : "en" );
	p.max_tokens = maxTokens;
	p.prompt_tokens = promptTokens;
	p.prompt_n_tokens = nPrompt;
	if( nMaxTextCtx >= 0 ) p.n_max_text_ctx = nMaxTextCtx;
	iAudioBuffer* buf = nullptr;
	CHECK( createAudioBuffer( std::vector<float>( pcm, pcm + nSamples ), {}, &buf ) );
	const HRESULT hr = c->runFull( p, buf );
	buf->Release();
	return hr;
}
// The same with the token-timestamp parameters of sFullParams (thold_pt, thold_ptsum, max_len; sFullParams.h:79-86)
WHISPER_EXPORT int32_t whisperc_run_full_tt( void* ctx, const float* pcm, uint32_t nSamples, const char* language, uint32_t flags, int maxTokens,
	const int32_t* promptTokens, int nPrompt, int nMaxTextCtx, float tholdPt, float tholdPtsum, int maxLen )
{
	if( !ctx || ( !pcm && nSamples ) ) return E_POINTER;
	iContext* c = (iContext*)ctx;
	sFullParams p;
	CHECK( c->fullDefaultParams( eSamplingStrategy::Greedy, &p ) );
	p.flags = (eFullParamsFlags)flags;
	p.language = makeLanguageKey( language ? language : "en" );
	p.max_tokens = maxTokens;
	p.prompt_tokens = promptTokens;
	p.prompt_n_tokens = nPrompt;
	if( nMaxTextCtx >= 0 ) p.n_max_text_ctx = nMaxTextCtx;
	p.thold_pt = tholdPt;
	p.thold_ptsum = tholdPtsum;
	p.max_len = maxLen;
	iAudioBuffer* buf = nullptr;
	CHECK( createAudioBuffer( std::vector<float>( pcm, pcm + nSamples ), {}, &buf ) );
	const HRESULT hr = c->runFull( p, buf );
	buf->Release();
	return hr;
}
// Copies the results out: segment times in 10 ms units are returned as 100 ns ticks like the COM API.
WHISPER_EXPORT int32_t whisperc_result_counts( void* ctx, uint32_t* segments, uint32_t* tokens )
{
	iTranscribeResult* r = nullptr;
	CHECK( ( (iContext*)ctx )->getResults( eResultFlags::Tokens | eResultFlags::Timestamps, &r ) );
	sTranscribeLength len;
	r->getSize( len );
	*segments = len.countSegments;
	*tokens = len.countTokens;
	return S_OK;
}
WHISPER_EXPORT int32_t whisperc_result_segment( void* ctx, uint32_t index, uint64_t* t0, uint64_t* t1, uint32_t* firstToken, uint32_t* countTokens,
	char* text, uint32_t textCap )
{
	iTranscribeResult* r = nullptr;
	CHECK( ( (iContext*)ctx )->getResults( eResultFlags::Tokens | eResultFlags::Timestamps, &r ) );
	sTranscribeLength len;
	r->getSize( len );
	if( index >= len.countSegments ) return E_BOUNDS;
	const sSegment& s = r->getSegments()[ index ];
	*t0 = s.time.begin.ticks; *t1 = s.time.end.ticks; *firstToken = s.firstToken; *countTokens = s.countTokens;
	if( text && textCap ) snprintf( text, textCap, "%s", s.text ? s.text : "" );
	return S_OK;
}
WHISPER_EXPORT int32_t whisperc_result_token( void* ctx, uint32_t index, int32_t* id, float* p, float* pt, float* ptsum )
{
	iTranscribeResult* r = nullptr;
	CHECK( ( (iContext*)ctx )->getResults( eResultFlags::Tokens | eResultFlags::Timestamps, &r ) );
	sTranscribeLength len;
	r->getSize( len );
	if( index >= len.countTokens ) return E_BOUNDS;
	const sToken& t = r->getTokens()[ index ];
	*id = t.id; *p = t.probability; *pt = t.probabilityTimestamp; *ptsum = t.ptsum;
	return S_OK;
}
WHISPER_EXPORT int32_t whisperc_result_token_times( void* ctx, uint32_t index, uint64_t* t0, uint64_t* t1, float* vlen )
{
	iTranscribeResult* r = nullptr;
	CHECK( ( (iContext*)ctx )->getResults( eResultFlags::Tokens | eResultFlags::Timestamps, &r ) );
	sTranscribeLength len;
	r->getSize( len );
	if( index >= len.countTokens ) return E_BOUNDS;
	const sToken& t = r->getTokens()[ index ];
	*t0 = t.time.begin.ticks; *t1 = t.time.end.ticks; *vlen = t.vlen;
	return S_OK;
}
WHISPER_EXPORT int32_t whisperc_tokenize( void* model, const char* text, int32_t* out, int cap )
{
	struct Sink { int32_t* out; int cap; int n; } sink{ out, cap, 0 };
	const HRESULT hr = ( (iModel*)model )->tokenize( text, []( const int* toks, int n, void* pv ) {
		Sink* s = (Sink*)pv;
		s->n = n;
		for( int i = 0; i < n && i < s->cap; i++ ) s->out[ i ] = toks[ i ];
	}, &sink );
	return FAILED( hr ) ? hr : sink.n;
}
WHISPER_EXPORT int32_t whisperc_timings_print( void* ctx ) { return ( (iContext*)ctx )->timingsPrint(); }

// ---- the lock-step batch runner (Whisper::createBatchRunner / iBatchRunner::run) for FFI callers ----
namespace
{
	// a caller's PCM as an iAudioBuffer without a copy; the memory stays the caller's for the duration of the call
	class PcmView : public Whisper::ComObject<Whisper::iAudioBuffer>
	{
		const float* const pcm;
		const uint32_t n;
	public:
		PcmView( const float* p, uint32_t count ) : pcm( p ), n( count ) {}
		uint32_t countSamples() const override { return n; }
		const float* getPcmMono() const override { return n ? pcm : nullptr; }
		const float* getPcmStereo() const override { return nullptr; }
		HRESULT getTime( int64_t& rdi ) const override { rdi = 0; return S_OK; }
	};
}
WHISPER_EXPORT int32_t whisperc_batch_create( void* model, uint32_t maxSlots, uint32_t groups, uint32_t greedyChunk, uint32_t flags, void** runnerOut )
{
	if( !model || !runnerOut ) return E_POINTER;
	const sBatchSetup setup{ maxSlots, groups, greedyChunk, flags };
	iBatchRunner* r = nullptr;
	const HRESULT hr = createBatchRunner( (iModel*)model, &setup, &r );
	*runnerOut = r;
	return hr;
}
WHISPER_EXPORT int32_t whisperc_batch_run( void* runner, uint32_t count, const float* const* pcm, const uint32_t* nSamples, const int64_t* firstSample,
	const int64_t* countSamples, const char* language, uint32_t flags, int maxTokens, const int32_t* promptTokens, int nPrompt, int nMaxTextCtx,
	void** resultsOut, int32_t* perStream )
{
	if( !runner || !resultsOut || ( count && ( !pcm || !nSamples ) ) ) return E_POINTER;
	sFullParams p;
	memset( &p, 0, sizeof( p ) );
	p.strategy = eSamplingStrategy::Greedy;
	p.cpuThreads = 4;
	p.n_max_text_ctx = 16384;
	p.thold_pt = p.thold_ptsum = 0.01f;
	p.beam_search.n_past = p.beam_search.beam_width = p.beam_search.n_best = -1;
	p.flags = (eFullParamsFlags)flags;
	p.language = makeLanguageKey( language ? language : "en" );
	p.max_tokens = maxTokens;
	p.prompt_tokens = promptTokens;
	p.prompt_n_tokens = nPrompt;
	if( nMaxTextCtx >= 0 ) p.n_max_text_ctx = nMaxTextCtx;
	// one view per distinct buffer: the chunks of a recording share it
	std::map<std::pair<const float*, uint32_t>, PcmView*> views;
	std::vector<sBatchStream> streams( count );
	for( uint32_t i = 0; i < count; i++ )
	{
		PcmView*& v = views[ { pcm[ i ], nSamples[ i ] } ];
		if( !v ) v = new PcmView( pcm[ i ], nSamples[ i ] );
		streams[ i ] = sBatchStream{ v, firstSample ? firstSample[ i ] : 0, countSamples ? countSamples[ i ] : 0, nullptr };
	}
	std::vector<iTranscribeResult*> res( count, nullptr );
	static_assert( sizeof( HRESULT ) == sizeof( int32_t ), "HRESULT" );
	const HRESULT hr = ( (iBatchRunner*)runner )->run( p, streams.data(), count, res.data(), perStream );
	for( uint32_t i = 0; i < count; i++ ) resultsOut[ i ] = res[ i ];
	for( auto& kv : views ) kv.second->Release();
	return hr;
}
WHISPER_EXPORT int32_t whisperc_tr_counts( void* result, uint32_t* segments, uint32_t* tokens )
{
	if( !result || !segments || !tokens ) return E_POINTER;
	sTranscribeLength len;
	( (iTranscribeResult*)result )->getSize( len );
	*segments = len.countSegments;
	*tokens = len.countTokens;
	return S_OK;
}
WHISPER_EXPORT int32_t whisperc_tr_segment( void* result, uint32_t index, uint64_t* t0, uint64_t* t1, uint32_t* firstToken, uint32_t* countTokens, char* text, uint32_t textCap )
{
	if( !result ) return E_POINTER;
	iTranscribeResult* r = (iTranscribeResult*)result;
	sTranscribeLength len;
	r->getSize( len );
	if( index >= len.countSegments ) return E_BOUNDS;
	const sSegment& s = r->getSegments()[ index ];
	*t0 = s.time.begin.ticks; *t1 = s.time.end.ticks; *firstToken = s.firstToken; *countTokens = s.countTokens;
	if( text && textCap ) snprintf( text, textCap, "%s", s.text ? s.text : "" );
	return S_OK;
}
WHISPER_EXPORT int32_t whisperc_tr_token( void* result, uint32_t index, int32_t* id, float* p, float* pt, float* ptsum, uint64_t* t0, uint64_t* t1, float* vlen )
{
	if( !result ) return E_POINTER;
	iTranscribeResult* r = (iTranscribeResult*)result;
	sTranscribeLength len;
	r->getSize( len );
	if( index >= len.countTokens ) return E_BOUNDS;
	const sToken& t = r->getTokens()[ index ];
	*id = t.id; *p = t.probability; *pt = t.probabilityTimestamp; *ptsum = t.ptsum;
	*t0 = t.time.begin.ticks; *t1 = t.time.end.ticks; *vlen = t.vlen;
	return S_OK;
}
// Host-only post-processing of token data into token times, on its own (no device): what runFull does per segment under the
// TokenTimestamps flag. Segments are processed in order (the anchors carry state from one to the next).
WHISPER_EXPORT int32_t whisperc_debug_token_timestamps( const char* modelPath, const float* pcm, uint64_t nSamples, int32_t nSegments,
	const int64_t* segTimes, const int32_t* segTokenCounts, const int32_t* ids, const int32_t* tids, const float* p, const float* pt,
	const float* ptsum, float tholdPt, float tholdPtsum, int32_t maxLen, int32_t segCap, int32_t tokCap, int32_t* outSegCount,
	int64_t* outSegTimes, int32_t* outSegTokenCounts, char* outTexts, uint32_t textCap, int64_t* outTokTimes, float* outVlen )
{
	using namespace Whisper;
	if( !modelPath || !pcm || !segTimes || !segTokenCounts || !ids || !tids || !p || !pt || !ptsum || !outSegCount ) return E_POINTER;
	Vocabulary vocab;
	const HRESULT hr = loadVocabulary( modelPath, vocab );
	if( FAILED( hr ) ) return hr;
	TokenTimestamper stamper;
	stamper.begin( pcm, (size_t)nSamples );
	std::vector<Segment> all;
	size_t at = 0;
	for( int i = 0; i < nSegments; i++ )
	{
		Segment s;
		s.t0 = segTimes[ 2 * i ]; s.t1 = segTimes[ 2 * i + 1 ];
		for( int j = 0; j < segTokenCounts[ i ]; j++, at++ )
		{
			TokenData t;
			t.id = ids[ at ]; t.tid = tids[ at ]; t.p = p[ at ]; t.pt = pt[ at ]; t.ptsum = ptsum[ at ];
			s.tokens.push_back( t );
			if( t.id < vocab.token_eot && vocab.string( t.id ) ) s.text += vocab.string( t.id );
		}
		all.push_back( std::move( s ) );
		stamper.compute( all.back(), vocab, tholdPt, tholdPtsum );
		if( maxLen > 0 ) TokenTimestamper::wrapLast( all, vocab, maxLen );
	}
	size_t nTok = 0, textAt = 0;
	for( const Segment& s : all ) nTok += s.tokens.size();
	if( (int)all.size() > segCap || (int)nTok > tokCap ) return E_BOUNDS;
	*outSegCount = (int32_t)all.size();
	size_t k = 0;
	for( size_t i = 0; i < all.size(); i++ )
	{
		const Segment& s = all[ i ];
		outSegTimes[ 2 * i ] = s.t0; outSegTimes[ 2 * i + 1 ] = s.t1;
		outSegTokenCounts[ i ] = (int32_t)s.tokens.size();
		if( textAt + s.text.size() + 1 > textCap ) return E_BOUNDS;
		memcpy( outTexts + textAt, s.text.c_str(), s.text.size() + 1 );	   // NUL-separated
		textAt += s.text.size() + 1;
		for( const TokenData& t : s.tokens )
		{
			outTokTimes[ 2 * k ] = t.t0; outTokTimes[ 2 * k + 1 ] = t.t1;
			outVlen[ k ] = t.vlen;
			k++;
		}
	}
	return S_OK;
}
// Vocabulary + tokenizer of a model file on their own (host only): what iModel::tokenize / stringFromToken / getSpecialTokens answer
WHISPER_EXPORT int32_t whisperc_debug_tokenize( const char* modelPath, const char* text, int32_t* out, int cap )
{
	if( !modelPath || !text || ( !out && cap > 0 ) ) return E_POINTER;
	Whisper::Vocabulary vocab;
	const HRESULT hr = Whisper::loadVocabulary( modelPath, vocab );
	if( FAILED( hr ) ) return hr;
	std::vector<int> toks;
	const HRESULT hr2 = vocab.tokenize( text, toks );
	if( FAILED( hr2 ) ) return hr2;
	if( (int)toks.size() > cap ) return E_BOUNDS;
	for( size_t i = 0; i < toks.size(); i++ ) out[ i ] = toks[ i ];
	return (int32_t)toks.size();
}
WHISPER_EXPORT int32_t whisperc_debug_token_string( const char* modelPath, int32_t token, char* out, uint32_t outCap, int32_t* specials8 )
{
	if( !modelPath || !out || outCap == 0 ) return E_POINTER;
	Whisper::Vocabulary vocab;
	const HRESULT hr = Whisper::loadVocabulary( modelPath, vocab );
	if( FAILED( hr ) ) return hr;
	const char* const str = vocab.string( token );
	snprintf( out, outCap, "%s", str ? str : "" );
	if( specials8 )
	{
		const int v[ 8 ] = { vocab.token_eot, vocab.token_sot, vocab.token_prev, vocab.token_solm, vocab.token_not, vocab.token_beg,
			vocab.token_translate, vocab.token_transcribe };
		for( int i = 0; i < 8; i++ ) specials8[ i ] = v[ i ];
	}
	return str ? S_OK : S_FALSE;
}
WHISPER_EXPORT int32_t whisperc_format_measure( const char* name, double ticks, uint64_t count, char* out, uint32_t outCap )
{
	if( !name || !out || outCap == 0 ) return -1;
	const std::string s = Whisper::formatMeasure( name, ticks, count );
	const size_t n = std::min( s.size(), (size_t)outCap - 1 );
	memcpy( out, s.data(), n );
	out[ n ] = 0;
	return (int32_t)n;
}
WHISPER_EXPORT int32_t whisperc_set_beam_ranking( int onHost )
{
	if( onHost != 0 && onHost != 1 ) return E_INVALIDARG;
	g_beamRankingOnHost = onHost != 0;
	return S_OK;
}
WHISPER_EXPORT int32_t whisperc_set_host_loop_rules( int mode )
{
	if( mode != 0 && mode != 1 ) return E_INVALIDARG;
	g_hostLoopRules = (eHostLoopRules)mode;
	return S_OK;
}
}
