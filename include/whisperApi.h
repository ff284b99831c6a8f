// whisperApi.h -- public C++ API of libWhisper.so: the COM-style surface of Const-me/Whisper's Whisper.dll on Linux.
//
// A caller written against the reference (Examples/main/main.cpp:174-330, or the C# interop in WhisperNet/Internal)
// binds to these names, vtable layouts and POD structures unchanged:
//   exports      Whisper/whisper.def:1-8 ; declarations Whisper/API/iContext.cl.h:62-70, iMediaFoundation.cl.h:47
//   interfaces   iModel / iContext (Whisper/API/iContext.cl.h:23-60), iTranscribeResult (iTranscribeResult.cl.h:7-15),
//                iAudioBuffer / iAudioReader / iAudioCapture / iMediaFoundation (iMediaFoundation.cl.h:8-45)
//   structures   sFullParams (sFullParams.h:45-108), sModelSetup (sModelSetup.h), sSegment / sToken (TranscribeStructs.h)
// Binary contract = ComLight's (ComLightLib/comLightCommon.h, unknwn.h): every interface starts with
// QueryInterface / AddRef / Release, then its methods in declaration order; objects are intrusively ref-counted and
// factories return them with one reference through a T** out-parameter; HRESULT everywhere, S_FALSE = benign "no".
// This header is written from that contract, not copied: types that the Linux build cannot honour (Media Foundation
// readers, capture devices) are kept so vtable slots line up, and return E_NOTIMPL.
#pragma once
#include <stdint.h>
#include <string.h>

#ifndef WHISPER_EXPORT
#define WHISPER_EXPORT __attribute__( ( visibility( "default" ) ) )
#endif

typedef int32_t HRESULT;
#ifndef S_OK
#define S_OK ( (HRESULT)0 )
#define S_FALSE ( (HRESULT)1 )
#define E_NOTIMPL ( (HRESULT)0x80004001 )
#define E_NOINTERFACE ( (HRESULT)0x80004002 )
#define E_POINTER ( (HRESULT)0x80004003 )
#define E_FAIL ( (HRESULT)0x80004005 )
#define E_UNEXPECTED ( (HRESULT)0x8000FFFF )
#define E_OUTOFMEMORY ( (HRESULT)0x8007000E )
#define E_INVALIDARG ( (HRESULT)0x80070057 )
#define E_BOUNDS ( (HRESULT)0x8000000B )
#define SUCCEEDED( hr ) ( ( (HRESULT)( hr ) ) >= 0 )
#define FAILED( hr ) ( ( (HRESULT)( hr ) ) < 0 )
#endif

struct IMFSourceReader;	   // Windows-only type, never defined here
// The reference's path type for media files: TCHAR* = UTF-16 on Windows, `using LPCTSTR = const char*` (UTF-8) everywhere
// else (ComLightLib/comLightCommon.h:5-9). A caller compiled from the reference's own headers on Linux passes const char*.
#ifndef _MSC_VER
using LPCTSTR = const char*;
#endif

namespace ComLight
{
	struct GUID
	{
		uint32_t Data1;
		uint16_t Data2, Data3;
		uint8_t Data4[ 8 ];
		bool operator==( const GUID& o ) const { return 0 == memcmp( this, &o, sizeof( GUID ) ); }
	};

	// {00000000-0000-0000-C000-000000000046}
	struct IUnknown
	{
		virtual HRESULT QueryInterface( const GUID& riid, void** ppvObject ) = 0;
		virtual uint32_t AddRef() = 0;
		virtual uint32_t Release() = 0;
	};
	constexpr GUID IID_IUnknown = { 0, 0, 0, { 0xC0, 0, 0, 0, 0, 0, 0, 0x46 } };

	// minimal smart pointer for callers and tests
	template<class I>
	class CComPtr
	{
		I* p = nullptr;
	public:
		CComPtr() = default;
		CComPtr( const CComPtr& o ) : p( o.p ) { if( p ) p->AddRef(); }
		~CComPtr() { release(); }
		CComPtr& operator=( const CComPtr& o ) { if( o.p ) o.p->AddRef(); release(); p = o.p; return *this; }
		void release() { if( p ) { p->Release(); p = nullptr; } }
		I** operator&() { release(); return &p; }
		I* operator->() const { return p; }
		operator I*() const { return p; }
		I* detach() { I* r = p; p = nullptr; return r; }
	};
}

namespace Whisper
{
	using whisper_token = int;
	struct iContext;
	struct iModel;

	// ---- plain-old-data ----------------------------------------------------------------------------------------
	enum struct eModelImplementation : uint32_t
	{
		GPU = 1,		// here: the MI355X HIP path
		Hybrid = 2,		// not built (disabled in the reference too, Whisper/stdafx.h:34)
		Reference = 3,	// the vendored CPU model; not part of the product library (it is the test oracle)
	};
	enum struct eGpuModelFlags : uint32_t
	{
		Wave32 = 1, Wave64 = 2, NoReshapedMatMul = 4, UseReshapedMatMul = 8, Cloneable = 0x10,
	};
	struct sModelSetup
	{
		eModelImplementation impl = eModelImplementation::GPU;
		uint32_t flags = 0;
		const wchar_t* adapter = nullptr;	 // a name reported by listGPUs, or nullptr for device 0
	};
	using pfnListAdapters = void ( * )( const wchar_t* name, void* pv );
	using pfnDecodedTokens = void ( * )( const int* tokens, int tokensLength, void* pv );

	using pfnLoadProgress = HRESULT ( * )( double val, void* pv ) noexcept;
	using pfnCancel = HRESULT ( * )( void* pv ) noexcept;
	struct sLoadModelCallbacks
	{
		pfnLoadProgress progress;
		pfnCancel cancel;
		void* pv;
	};

	enum struct eLogLevel : uint8_t { Error = 0, Warning = 1, Info = 2, Debug = 3 };
	enum struct eLoggerFlags : uint8_t { UseStandardError = 1, SkipFormatMessage = 2 };
	using pfnLoggerSink = void ( * )( void* context, eLogLevel lvl, const char* message );
	struct sLoggerSetup
	{
		pfnLoggerSink sink = nullptr;
		void* context = nullptr;
		eLogLevel level = eLogLevel::Warning;
		eLoggerFlags flags = (eLoggerFlags)0;
	};

	struct sLanguageEntry
	{
		uint32_t key;	 // up to 4 ASCII characters packed little-endian, see makeLanguageKey
		int id;
		const char* name;
	};
	struct sLanguageList
	{
		uint32_t length;
		const sLanguageEntry* pointer;
	};

	struct SpecialTokens
	{
		int TranscriptionEnd, TranscriptionStart, PreviousWord, SentenceStart, Not, TranscriptionBegin, TaskTranslate, TaskTranscribe;
	};

	// times are in 100-nanosecond ticks
	struct sTimeSpan { uint64_t ticks; };
	struct sTimeInterval { sTimeSpan begin, end; };
	struct sSegment
	{
		const char* text;
		sTimeInterval time;
		uint32_t firstToken, countTokens;
	};
	enum eTokenFlags : uint32_t { None = 0, Special = 1 };
	struct sToken
	{
		const char* text;
		sTimeInterval time;
		float probability, probabilityTimestamp, ptsum, vlen;
		int id;
		eTokenFlags flags;
	};
	struct sTranscribeLength { uint32_t countSegments, countTokens; };
	enum struct eResultFlags : uint32_t { None = 0, Tokens = 1, Timestamps = 2, NewObject = 0x100 };
	inline eResultFlags operator|( eResultFlags a, eResultFlags b ) { return (eResultFlags)( (uint32_t)a | (uint32_t)b ); }
	inline bool operator&( eResultFlags a, eResultFlags b ) { return 0 != ( (uint32_t)a & (uint32_t)b ); }
	enum struct eSpeakerChannel : uint8_t { Unsure = 0, Left = 1, Right = 2, NoStereoData = 0xFF };

	enum struct eSamplingStrategy : int { Greedy, BeamSearch };
	using pfnNewSegment = HRESULT ( * )( iContext* ctx, uint32_t n_new, void* user_data ) noexcept;
	using pfnEncoderBegin = HRESULT ( * )( iContext* ctx, void* user_data ) noexcept;	 // S_FALSE stops the run
	enum struct eFullParamsFlags : uint32_t
	{
		Translate = 1, NoContext = 2, SingleSegment = 4, PrintSpecial = 8, PrintProgress = 0x10, PrintRealtime = 0x20,
		PrintTimestamps = 0x40, TokenTimestamps = 0x100, SpeedupAudio = 0x200,
	};
	inline eFullParamsFlags operator|( eFullParamsFlags a, eFullParamsFlags b ) { return (eFullParamsFlags)( (uint32_t)a | (uint32_t)b ); }

	struct sFullParams
	{
		eSamplingStrategy strategy;
		int cpuThreads;
		int n_max_text_ctx;
		int offset_ms;		// start offset in ms
		int duration_ms;	// audio duration to process in ms, 0 = all
		eFullParamsFlags flags;
		uint32_t language;
		float thold_pt, thold_ptsum;
		int max_len;
		int max_tokens;		// max tokens per segment, 0 = no limit
		struct { int n_past; } greedy;
		struct { int n_past, beam_width, n_best; } beam_search;
		int audio_ctx;		// overwrite the audio context size, 0 = default: the encoder takes 2 * audio_ctx frames per window, the decoder attends to audio_ctx keys
		const whisper_token* prompt_tokens;
		int prompt_n_tokens;
		pfnNewSegment new_segment_callback;
		void* new_segment_callback_user_data;
		pfnEncoderBegin encoder_begin_callback;
		void* encoder_begin_callback_user_data;

		bool flag( eFullParamsFlags f ) const { return 0 != ( (uint32_t)flags & (uint32_t)f ); }
		void resetFlag( eFullParamsFlags bit ) { flags = (eFullParamsFlags)( (uint32_t)flags & ~(uint32_t)bit ); }
		void setFlag( eFullParamsFlags bit, bool set = true )
		{
			uint32_t f = (uint32_t)flags;
			f = set ? ( f | (uint32_t)bit ) : ( f & ~(uint32_t)bit );
			flags = (eFullParamsFlags)f;
		}
	};

	inline uint32_t makeLanguageKey( const char* code )
	{
		uint32_t res = 0;
		for( uint32_t i = 0; i < 4 && code[ i ]; i++ ) res |= (uint32_t)(uint8_t)code[ i ] << ( 8 * i );
		return res;
	}

	using pfnReportProgress = HRESULT ( * )( double val, iContext* ctx, void* pv ) noexcept;
	struct sProgressSink
	{
		pfnReportProgress pfn;
		void* pv;
	};
	struct sCaptureCallbacks;
	struct sCaptureParams;
	struct sCaptureDevice;
	using pfnFoundCaptureDevices = HRESULT ( * )( int len, const sCaptureDevice* buffer, void* pv ) noexcept;

	// ---- interfaces ----------------------------------------------------------------------------------------------
	// {2871a73f-5ce3-48f8-8779-6582ee11935e}
	struct iTranscribeResult : public ComLight::IUnknown
	{
		static constexpr ComLight::GUID iid() { return { 0x2871a73f, 0x5ce3, 0x48f8, { 0x87, 0x79, 0x65, 0x82, 0xee, 0x11, 0x93, 0x5e } }; }
		virtual HRESULT getSize( sTranscribeLength& rdi ) const = 0;
		virtual const sSegment* getSegments() const = 0;
		virtual const sToken* getTokens() const = 0;
	};

	// {013583aa-c9eb-42bc-83db-633c2c317051}
	struct iAudioBuffer : public ComLight::IUnknown
	{
		static constexpr ComLight::GUID iid() { return { 0x013583aa, 0xc9eb, 0x42bc, { 0x83, 0xdb, 0x63, 0x3c, 0x2c, 0x31, 0x70, 0x51 } }; }
		virtual uint32_t countSamples() const = 0;
		virtual const float* getPcmMono() const = 0;
		virtual const float* getPcmStereo() const = 0;
		virtual HRESULT getTime( int64_t& rdi ) const = 0;
	};
	// {35b988da-04a6-476a-a193-d8891d5dc390}
	struct iAudioReader : public ComLight::IUnknown
	{
		static constexpr ComLight::GUID iid() { return { 0x35b988da, 0x04a6, 0x476a, { 0xa1, 0x93, 0xd8, 0x89, 0x1d, 0x5d, 0xc3, 0x90 } }; }
		virtual HRESULT getDuration( int64_t& rdi ) const = 0;
		virtual HRESULT getReader( IMFSourceReader** pp ) const = 0;
		virtual HRESULT requestedStereo() const = 0;
	};
	// {747752c2-d9fd-40df-8847-583c781bf013}
	struct iAudioCapture : public ComLight::IUnknown
	{
		static constexpr ComLight::GUID iid() { return { 0x747752c2, 0xd9fd, 0x40df, { 0x88, 0x47, 0x58, 0x3c, 0x78, 0x1b, 0xf0, 0x13 } }; }
		virtual HRESULT getReader( IMFSourceReader** pp ) const = 0;
		virtual const sCaptureParams& getParams() const = 0;
	};
	// {fb9763a5-d77d-4b6e-aff8-f494813cebd8}  -- on Linux: a WAV / raw-PCM loader stands in for Media Foundation
	struct iMediaFoundation : public ComLight::IUnknown
	{
		static constexpr ComLight::GUID iid() { return { 0xfb9763a5, 0xd77d, 0x4b6e, { 0xaf, 0xf8, 0xf4, 0x94, 0x81, 0x3c, 0xeb, 0xd8 } }; }
		virtual HRESULT loadAudioFile( LPCTSTR path, bool stereo, iAudioBuffer** pp ) const = 0;
		virtual HRESULT openAudioFile( LPCTSTR path, bool stereo, iAudioReader** pp ) = 0;
		virtual HRESULT loadAudioFileData( const void* data, uint64_t size, bool stereo, iAudioReader** pp ) = 0;
		virtual HRESULT listCaptureDevices( pfnFoundCaptureDevices pfn, void* pv ) = 0;
		virtual HRESULT openCaptureDevice( LPCTSTR endpoint, const sCaptureParams& captureParams, iAudioCapture** pp ) = 0;
	};

	// {b9956374-3b18-4943-90f2-2ab18a404537}
	struct iContext : public ComLight::IUnknown
	{
		static constexpr ComLight::GUID iid() { return { 0xb9956374, 0x3b18, 0x4943, { 0x90, 0xf2, 0x2a, 0xb1, 0x8a, 0x40, 0x45, 0x37 } }; }
		// PCM -> log-mel -> encoder -> greedy decoder -> segments, the complete model
		virtual HRESULT runFull( const sFullParams& params, const iAudioBuffer* buffer ) = 0;
		virtual HRESULT runStreamed( const sFullParams& params, const sProgressSink& progress, const iAudioReader* reader ) = 0;
		virtual HRESULT runCapture( const sFullParams& params, const sCaptureCallbacks& callbacks, const iAudioCapture* reader ) = 0;
		virtual HRESULT getResults( eResultFlags flags, iTranscribeResult** pp ) const = 0;
		virtual HRESULT detectSpeaker( const sTimeInterval& time, eSpeakerChannel& result ) const = 0;
		virtual HRESULT getModel( iModel** pp ) = 0;
		virtual HRESULT fullDefaultParams( eSamplingStrategy strategy, sFullParams* rdi ) = 0;
		virtual HRESULT timingsPrint() = 0;
		virtual HRESULT timingsReset() = 0;
	};

	// {abefb4c9-e8d8-46a3-8747-5afbadef1adb}
	struct iModel : public ComLight::IUnknown
	{
		static constexpr ComLight::GUID iid() { return { 0xabefb4c9, 0xe8d8, 0x46a3, { 0x87, 0x47, 0x5a, 0xfb, 0xad, 0xef, 0x1a, 0xdb } }; }
		virtual HRESULT createContext( iContext** pp ) = 0;
		virtual HRESULT tokenize( const char* text, pfnDecodedTokens pfn, void* pv ) = 0;
		virtual HRESULT isMultilingual() = 0;
		virtual HRESULT getSpecialTokens( SpecialTokens& rdi ) = 0;
		virtual const char* stringFromToken( whisper_token token ) = 0;
		virtual HRESULT clone( iModel** rdi ) = 0;
	};

	// ---- the seven exports of Whisper.dll (Whisper/whisper.def) ---------------------------------------------------
	WHISPER_EXPORT HRESULT setupLogger( const sLoggerSetup& setup );
	WHISPER_EXPORT HRESULT loadModel( const wchar_t* path, const sModelSetup& setup, const sLoadModelCallbacks* callbacks, iModel** pp );
	// Extension (no counterpart in whisper.def): one process per GPU. whComm is a wh_comm* of include/whisper_hip.h; rank `root`
	// reads the tensors, the other ranks receive the weight arena over RCCL and read only the header of the file.
	WHISPER_EXPORT HRESULT loadModelShared( const wchar_t* path, const sModelSetup& setup, const sLoadModelCallbacks* callbacks, void* whComm, int root, iModel** pp );
	// Extension (no counterpart in whisper.def): K recordings -- or K pieces of recordings -- transcribed in LOCK STEP on one GPU.
	// Every stream keeps the semantics of iContext::runFull on its own (ContextImpl.cpp:452-793: seek by the last timestamp, prompt
	// carry-over unless NoContext, stop rules, failure handling, callbacks on the calling thread), and its transcript is the transcript
	// runFull gives for the same samples; what changes is the device work: the next windows of up to `maxSlots` streams are ONE encoder
	// batch and ONE decode chain (a decode step costs about the same for 1 and for 64 sequences), `groups` such batches are in flight
	// on separate HIP streams (the encoder of one runs under the decode chain of the other), and a slot whose stream has finished is
	// refilled with the next stream. The reference's nearest facilities: whisper_full_parallel (Whisper/source/whisper.cpp:3127) and
	// iModel::clone + one iContext per thread (Whisper/Whisper/ModelImpl.cpp:40-60).
	struct sBatchStream
	{
		// Samples [firstSample, firstSample + countSamples) of the buffer's mono PCM (16 kHz), transcribed as a recording of its own:
		// its spectrogram is normalised on its own maximum and windows never read past its end -- how independent 30 s chunks of one
		// recording are declared. countSamples 0 = to the end of the buffer. Segment and token times are relative to the BUFFER
		// (the stream's first sample adds firstSample / 16000 s), plus the buffer's media time (iAudioBuffer::getTime).
		const iAudioBuffer* buffer;
		int64_t firstSample, countSamples;
		const sFullParams* params;	  // nullptr = the call's common parameters (callbacks receive a per-stream iContext: getResults, getModel)
	};
	struct sBatchSetup
	{
		uint32_t maxSlots;		// streams per lock-step batch (the device contexts are sized for it); 0 = 64, at most 512
		uint32_t groups;		// lock-step batches in flight; 0 = 2
		uint32_t greedyChunk;	// greedy steps enqueued at a time; 0 = 4. The host applies its stop rules chunk by chunk, so up to one chunk is decoded past the end of a round
		uint32_t flags;			// 1 = keep one more chunk queued behind the one in flight (the device never waits for the host; up to two chunks are decoded in vain)
	};
	// {6f0c9a1e-3b52-4d7c-8e21-9a4f5c7d2b10}
	struct iBatchRunner : public ComLight::IUnknown
	{
		static constexpr ComLight::GUID iid() { return { 0x6f0c9a1e, 0x3b52, 0x4d7c, { 0x8e, 0x21, 0x9a, 0x4f, 0x5c, 0x7d, 0x2b, 0x10 } }; }
		// results[i] receives the transcript of streams[i] (a new object, the caller releases it; nullptr when the stream failed before it
		// produced one). Returns the first failure of any stream, S_OK otherwise; perStream, when non-null, receives every stream's own
		// HRESULT (S_FALSE: shorter than a second, an empty transcript -- runFull's S_FALSE). Not re-entrant; callbacks of sFullParams
		// arrive on the calling thread and receive a per-stream iContext (getResults, getModel).
		virtual HRESULT run( const sFullParams& params, const sBatchStream* streams, uint32_t count, iTranscribeResult** results, HRESULT* perStream ) = 0;
	};
	// The runner owns the device contexts of its groups (KV caches for maxSlots windows each, captured decode graphs): a service creates
	// it once. runFullBatch = createBatchRunner + run + Release for a one-off call.
	WHISPER_EXPORT HRESULT createBatchRunner( iModel* model, const sBatchSetup* setup, iBatchRunner** pp );
	WHISPER_EXPORT HRESULT runFullBatch( iModel* model, const sFullParams& params, const sBatchStream* streams, uint32_t count, const sBatchSetup* setup,
		iTranscribeResult** results, HRESULT* perStream );
	WHISPER_EXPORT HRESULT initMediaFoundation( iMediaFoundation** pp );
	WHISPER_EXPORT uint32_t findLanguageKeyW( const wchar_t* lang );
	WHISPER_EXPORT uint32_t findLanguageKeyA( const char* lang );
	WHISPER_EXPORT HRESULT getSupportedLanguages( sLanguageList& rdi );
	WHISPER_EXPORT HRESULT listGPUs( pfnListAdapters pfn, void* pv );
}

// Flat C mirror of the above for FFI callers that cannot consume C++ vtables (ctypes, cgo ...): whisper_c.h
