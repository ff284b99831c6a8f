// Internal helpers of libWhisper.so (plain C++17, no HIP headers: the GPU is reached only through include/whisper_hip.h).
#pragma once
#include "whisperApi.h"
#include "whisper_hip.h"
#include <atomic>
#include <map>
#include <memory>
#include <string>
#include <vector>

namespace Whisper
{
	// ---- logging (Whisper/Utils/Logger.cpp: a sink callback with a level filter, or stderr) ----
	void logMessage( eLogLevel lvl, const char* fmt, ... ) __attribute__( ( format( printf, 2, 3 ) ) );
#define logError( ... ) ::Whisper::logMessage( ::Whisper::eLogLevel::Error, __VA_ARGS__ )
#define logWarning( ... ) ::Whisper::logMessage( ::Whisper::eLogLevel::Warning, __VA_ARGS__ )
#define logInfo( ... ) ::Whisper::logMessage( ::Whisper::eLogLevel::Info, __VA_ARGS__ )
#define logDebug( ... ) ::Whisper::logMessage( ::Whisper::eLogLevel::Debug, __VA_ARGS__ )

	// wh_status -> HRESULT, logging the library's message
	HRESULT hrFromStatus( int status, const char* what );
#define CHECK_WH( expr )                                         \
	do                                                           \
	{                                                            \
		const int st__ = ( expr );                               \
		if( st__ != 0 ) return ::Whisper::hrFromStatus( st__, #expr ); \
	} while( 0 )
#define CHECK( expr )                     \
	do                                    \
	{                                     \
		const HRESULT hr__ = ( expr );    \
		if( FAILED( hr__ ) ) return hr__; \
	} while( 0 )

	std::string utf8( const wchar_t* w );

	// ================================================================================================================
	// profiler output -- one line per measure, the format of ProfileCollection::Measure::print
	// (Whisper/Utils/ProfileCollection.cpp:113-170): time in 100 ns ticks scaled to seconds / milliseconds / microseconds,
	// "%g"; a measure taken once prints only its total, otherwise "total, N calls, avg average"
	std::string formatMeasure( const char* name, double ticks, uint64_t count );
	// "877.966 KB" / "1.42785 GB": bytes scaled by 1024 steps, "%g"
	std::string formatBytes( double bytes );

	// ---- intrusive ref-counting for the COM-style objects (the role of ComLight::ObjectRoot / Object<T>) ----
	template<class I>
	class ComObject : public I
	{
		std::atomic<uint32_t> refs{ 1 };
	public:
		virtual ~ComObject() = default;
		HRESULT QueryInterface( const ComLight::GUID& riid, void** ppv ) override
		{
			if( !ppv ) return E_POINTER;
			if( riid == I::iid() || riid == ComLight::IID_IUnknown )
			{
				*ppv = static_cast<I*>( this );
				AddRef();
				return S_OK;
			}
			*ppv = nullptr;
			return E_NOINTERFACE;
		}
		uint32_t AddRef() override { return ++refs; }
		uint32_t Release() override
		{
			const uint32_t r = --refs;
			if( r == 0 ) delete this;
			return r;
		}
	};

	// ---- vocabulary (Whisper/Whisper/Vocabulary.{h,cpp}; whisper.cpp:540-606) ----
	struct Vocabulary
	{
		int n_vocab = 0;
		int token_eot = 50256, token_sot = 50257, token_prev = 50360, token_solm = 50361, token_not = 50362, token_beg = 50363;
		int token_translate = 50358, token_transcribe = 50359;
		std::vector<std::string> idToToken;
		std::map<std::string, int> tokenToId;
		bool isMultilingual() const { return n_vocab >= 51865; }
		const char* string( int id ) const { return ( id >= 0 && id < (int)idToToken.size() ) ? idToToken[ id ].c_str() : nullptr; }
		// fills the special ids and synthesises the tokens the file does not store
		void finalize( int nVocabModel );
		// GPT-2 style greedy longest-match over the regex-split words (whisper.cpp:2186-2248 "tokenize")
		HRESULT tokenize( const char* text, std::vector<int>& out ) const;
	};

	// ---- transcript under construction (whisper_segment / whisper_token_data, whisper.cpp:398-407, whisper.h:71-85) ----
	struct TokenData
	{
		int id = 0, tid = 0;
		float p = 0, pt = 0, ptsum = 0, vlen = 0;
		int64_t t0 = -1, t1 = -1;	// 10 ms units, -1 = unknown
	};
	struct Segment
	{
		int64_t t0 = 0, t1 = 0;	   // 10 ms units
		std::string text;
		std::vector<TokenData> tokens;
	};

	// ---- token-level timestamps + max_len wrapping (TokenTimestamps flag; whisper.cpp:3320-3575, 2711-2760) ----
	class TokenTimestamper
	{
		std::vector<float> energy;
		int64_t tBeg = 0, tLast = 0;
		int tidLast = 0;
	public:
		// start of a run: smoothed |signal| (65-sample box) and a clean timestamp state (whisper.cpp:2803-2808, 3352-3369)
		void begin( const float* pcm, size_t samples );
		bool ready() const { return !energy.empty(); }
		// Fills t0 / t1 / vlen of the segment's tokens: timestamp tokens the model is confident about anchor the text tokens
		// before them, the gaps are split in proportion to a "voice length" of the token text, then every token is grown or
		// shrunk to the nearest change of voice activity.
		void compute( Segment& segment, const Vocabulary& vocab, float tholdPt, float tholdPtsum );
		// Splits the LAST segment into pieces of at most maxLen characters at token boundaries; returns the number of
		// segments it became (>= 1).
		static int wrapLast( std::vector<Segment>& all, const Vocabulary& vocab, int maxLen );
	};

	// ---- languages (Whisper/Whisper/Languages.cpp, languageCodez.inl; whisper.cpp:31-133) ----
	int lookupLanguageId( uint32_t key );
	const sLanguageList& languageList();

	// ---- ggml model file (Whisper/Whisper/WhisperModel.cpp:434-492, 257-340) ----
	struct LoadedModel
	{
		wh_hparams hp{};
		wh_model* gpu = nullptr;
		Vocabulary vocab;
		~LoadedModel();
	};
	// comm != nullptr: one process per GPU -- only rank `root` reads the tensors, every other rank reads the header (hparams, filterbank,
	// vocabulary) and receives the weight arena over RCCL (wh_model_broadcast)
	HRESULT loadGgmlFile( const std::string& path, int device, const sLoadModelCallbacks* callbacks, std::shared_ptr<LoadedModel>& out,
		wh_comm* comm = nullptr, int root = 0 );
	// only the vocabulary of a model file (no device needed)
	HRESULT loadVocabulary( const std::string& path, Vocabulary& vocab );

	// what runFullBatch needs from an iModel of this library: the loaded weights behind it
	// {8c1f0d3a-52b7-4e0e-9a64-0f6d2b7c41e5}
	struct iModelInternals : public ComLight::IUnknown
	{
		static constexpr ComLight::GUID iid() { return { 0x8c1f0d3a, 0x52b7, 0x4e0e, { 0x9a, 0x64, 0x0f, 0x6d, 0x2b, 0x7c, 0x41, 0xe5 } }; }
		virtual const std::shared_ptr<LoadedModel>& loaded() const = 0;
	};

	HRESULT createContextImpl( const std::shared_ptr<LoadedModel>& model, iModel* owner, iContext** pp );
	HRESULT createModelImpl( const std::shared_ptr<LoadedModel>& model, iModel** pp );
	HRESULT createAudioBuffer( std::vector<float>&& mono, std::vector<float>&& stereo, iAudioBuffer** pp );
}
