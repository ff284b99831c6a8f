// Logger, language table, vocabulary and the ggml file loader of libWhisper.so.
#include "hostCommon.h"
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <mutex>
#include <regex>

namespace Whisper
{
	// ================================================================================================================
	// profiler output
	namespace
	{
		struct ScaledTime
		{
			double value;
			const char* unit;
			explicit ScaledTime( double ticks )
			{
				if( ticks >= 1.0e7 ) { value = ticks / 1.0e7; unit = "seconds"; }
				else if( ticks >= 1.0e4 ) { value = ticks / 1.0e4; unit = "milliseconds"; }
				else { value = ticks / 10.0; unit = "microseconds"; }
			}
		};
	}

	std::string formatMeasure( const char* name, double ticks, uint64_t count )
	{
		char buf[ 256 ];
		const ScaledTime total( ticks );
		if( count == 1 )
			snprintf( buf, sizeof( buf ), "%s\t%g %s", name, total.value, total.unit );
		else
		{
			const ScaledTime avg( count ? ticks / (double)count : 0.0 );
			snprintf( buf, sizeof( buf ), "%s\t%g %s, %zu calls, %g %s average", name, total.value, total.unit, (size_t)count, avg.value, avg.unit );
		}
		return buf;
	}

	std::string formatBytes( double bytes )
	{
		const char* unit = "bytes";
		if( bytes >= 1024.0 * 1024.0 * 1024.0 ) { bytes /= 1024.0 * 1024.0 * 1024.0; unit = "GB"; }
		else if( bytes >= 1024.0 * 1024.0 ) { bytes /= 1024.0 * 1024.0; unit = "MB"; }
		else if( bytes >= 1024.0 ) { bytes /= 1024.0; unit = "KB"; }
		char buf[ 64 ];
		snprintf( buf, sizeof( buf ), "%g %s", bytes, unit );
		return buf;
	}

	// ================================================================================================================
	// logger -- Whisper/Utils/Logger.cpp: messages go to the sink registered with setupLogger, filtered by level
	// ================================================================================================================
	namespace
	{
		std::mutex g_logMutex;
		sLoggerSetup g_logger;
		bool g_loggerSet = false;
	}

	HRESULT setupLogger( const sLoggerSetup& setup )
	{
		std::lock_guard<std::mutex> lk( g_logMutex );
		g_logger = setup;
		g_loggerSet = true;
		return S_OK;
	}

	void logMessage( eLogLevel lvl, const char* fmt, ... )
	{
		sLoggerSetup ls;
		bool set;
		{
			std::lock_guard<std::mutex> lk( g_logMutex );
			ls = g_logger;
			set = g_loggerSet;
		}
		// without a registered sink only errors and warnings are shown, on stderr
		if( set ? (uint8_t)lvl > (uint8_t)ls.level : (uint8_t)lvl > (uint8_t)eLogLevel::Warning ) return;
		char buf[ 2048 ];
		va_list a;
		va_start( a, fmt );
		vsnprintf( buf, sizeof( buf ), fmt, a );
		va_end( a );
		if( set && ls.sink ) ls.sink( ls.context, lvl, buf );
		if( !set || !ls.sink || ( (uint8_t)ls.flags & (uint8_t)eLoggerFlags::UseStandardError ) ) fprintf( stderr, "%s\n", buf );
	}

	HRESULT hrFromStatus( int status, const char* what )
	{
		logError( "%s failed: %s", what, wh_last_error() );
		switch( status )
		{
		case WH_E_INVALIDARG: return E_INVALIDARG;
		case WH_E_OUTOFMEMORY: return E_OUTOFMEMORY;
		case WH_E_BOUNDS: return E_BOUNDS;
		case WH_E_NOT_READY: return E_UNEXPECTED;
		default: return E_FAIL;
		}
	}

	std::string utf8( const wchar_t* w )
	{
		std::string r;
		if( !w ) return r;
		for( ; *w; w++ )
		{
			uint32_t c = (uint32_t)*w;
			if( c < 0x80 ) r += (char)c;
			else if( c < 0x800 ) { r += (char)( 0xC0 | ( c >> 6 ) ); r += (char)( 0x80 | ( c & 0x3F ) ); }
			else if( c < 0x10000 ) { r += (char)( 0xE0 | ( c >> 12 ) ); r += (char)( 0x80 | ( ( c >> 6 ) & 0x3F ) ); r += (char)( 0x80 | ( c & 0x3F ) ); }
			else { r += (char)( 0xF0 | ( c >> 18 ) ); r += (char)( 0x80 | ( ( c >> 12 ) & 0x3F ) ); r += (char)( 0x80 | ( ( c >> 6 ) & 0x3F ) ); r += (char)( 0x80 | ( c & 0x3F ) ); }
		}
		return r;
	}

	// ================================================================================================================
	// languages -- the 99 ids of the multilingual models, language token = sot + 1 + id (ContextImpl.cpp:505)
	// ================================================================================================================
	namespace
	{
		struct Lang { const char* code; const char* name; };
		const Lang g_langs[] = {
			{ "en", "english" }, { "zh", "chinese" }, { "de", "german" }, { "es", "spanish" }, { "ru", "russian" }, { "ko", "korean" },
			{ "fr", "french" }, { "ja", "japanese" }, { "pt", "portuguese" }, { "tr", "turkish" }, { "pl", "polish" }, { "ca", "catalan" },
			{ "nl", "dutch" }, { "ar", "arabic" }, { "sv", "swedish" }, { "it", "italian" }, { "id", "indonesian" }, { "hi", "hindi" },
			{ "fi", "finnish" }, { "vi", "vietnamese" }, { "iw", "hebrew" }, { "uk", "ukrainian" }, { "el", "greek" }, { "ms", "malay" },
			{ "cs", "czech" }, { "ro", "romanian" }, { "da", "danish" }, { "hu", "hungarian" }, { "ta", "tamil" }, { "no", "norwegian" },
			{ "th", "thai" }, { "ur", "urdu" }, { "hr", "croatian" }, { "bg", "bulgarian" }, { "lt", "lithuanian" }, { "la", "latin" },
			{ "mi", "maori" }, { "ml", "malayalam" }, { "cy", "welsh" }, { "sk", "slovak" }, { "te", "telugu" }, { "fa", "persian" },
			{ "lv", "latvian" }, { "bn", "bengali" }, { "sr", "serbian" }, { "az", "azerbaijani" }, { "sl", "slovenian" }, { "kn", "kannada" },
			{ "et", "estonian" }, { "mk", "macedonian" }, { "br", "breton" }, { "eu", "basque" }, { "is", "icelandic" }, { "hy", "armenian" },
			{ "ne", "nepali" }, { "mn", "mongolian" }, { "bs", "bosnian" }, { "kk", "kazakh" }, { "sq", "albanian" }, { "sw", "swahili" },
			{ "gl", "galician" }, { "mr", "marathi" }, { "pa", "punjabi" }, { "si", "sinhala" }, { "km", "khmer" }, { "sn", "shona" },
			{ "yo", "yoruba" }, { "so", "somali" }, { "af", "afrikaans" }, { "oc", "occitan" }, { "ka", "georgian" }, { "be", "belarusian" },
			{ "tg", "tajik" }, { "sd", "sindhi" }, { "gu", "gujarati" }, { "am", "amharic" }, { "yi", "yiddish" }, { "lo", "lao" },
			{ "uz", "uzbek" }, { "fo", "faroese" }, { "ht", "haitian creole" }, { "ps", "pashto" }, { "tk", "turkmen" }, { "nn", "nynorsk" },
			{ "mt", "maltese" }, { "sa", "sanskrit" }, { "lb", "luxembourgish" }, { "my", "myanmar" }, { "bo", "tibetan" }, { "tl", "tagalog" },
			{ "mg", "malagasy" }, { "as", "assamese" }, { "tt", "tatar" }, { "haw", "hawaiian" }, { "ln", "lingala" }, { "ha", "hausa" },
			{ "ba", "bashkir" }, { "jw", "javanese" }, { "su", "sundanese" },
		};
		constexpr int N_LANGS = (int)( sizeof( g_langs ) / sizeof( g_langs[ 0 ] ) );
		static_assert( N_LANGS == 99, "the multilingual models know 99 languages" );

		struct LangTable
		{
			sLanguageEntry entries[ N_LANGS ];
			sLanguageList list;
			LangTable()
			{
				for( int i = 0; i < N_LANGS; i++ ) entries[ i ] = { makeLanguageKey( g_langs[ i ].code ), i, g_langs[ i ].name };
				list.length = N_LANGS;
				list.pointer = entries;
			}
		};
		const LangTable& langTable()
		{
			static const LangTable t;
			return t;
		}
	}

	const sLanguageList& languageList() { return langTable().list; }

	int lookupLanguageId( uint32_t key )
	{
		const LangTable& t = langTable();
		for( int i = 0; i < N_LANGS; i++ )
			if( t.entries[ i ].key == key ) return i;
		return -1;
	}

	uint32_t findLanguageKeyA( const char* lang )
	{
		if( !lang ) return UINT32_MAX;
		std::string s( lang );
		for( char& c : s ) c = (char)tolower( (unsigned char)c );
		const LangTable& t = langTable();
		for( int i = 0; i < N_LANGS; i++ )
			if( s == g_langs[ i ].code || s == g_langs[ i ].name ) return t.entries[ i ].key;
		return UINT32_MAX;
	}
	uint32_t findLanguageKeyW( const wchar_t* lang ) { return findLanguageKeyA( utf8( lang ).c_str() ); }

	HRESULT getSupportedLanguages( sLanguageList& rdi )
	{
		rdi = languageList();
		return S_OK;
	}

	// ================================================================================================================
	// vocabulary
	// ================================================================================================================
	void Vocabulary::finalize( int nVocabModel )
	{
		const int nWords = (int)idToToken.size();
		n_vocab = nVocabModel;
		if( isMultilingual() )
		{
			token_eot++; token_sot++; token_prev++; token_solm++; token_not++; token_beg++;
			// vocabularies beyond the reference's 51865 (the large-v3 shape has one more language token): everything behind
			// the language block moves up with it, the task tokens included
			const int more = n_vocab - 51865;
			token_prev += more; token_solm += more; token_not += more; token_beg += more;
			token_translate += more; token_transcribe += more;
		}
		// the file stores the byte-pair vocabulary only; the special tokens get printable stand-ins
		idToToken.resize( std::max( nWords, nVocabModel ) );
		for( int i = nWords; i < nVocabModel; i++ )
		{
			std::string w;
			if( i > token_beg ) w = "[_TT_" + std::to_string( i - token_beg ) + "]";
			else if( i == token_eot ) w = "[_EOT_]";
			else if( i == token_sot ) w = "[_SOT_]";
			else if( i == token_prev ) w = "[_PREV_]";
			else if( i == token_not ) w = "[_NOT_]";
			else if( i == token_beg ) w = "[_BEG_]";
			else w = "[_extra_token_" + std::to_string( i ) + "]";
			idToToken[ i ] = w;
		}
		// Every token, the synthesised names included, is findable by the tokenizer: the reference does the same in both of
		// its loaders (Whisper/Whisper/Vocabulary.cpp:36-40 completeBuild() maps ALL `tokens`; Whisper/source/whisper.cpp:601
		// `vocab.token_to_id[word] = i` inside the extra-token loop), so "[_EOT_]" in a prompt tokenizes to the control id there too.
		tokenToId.clear();
		for( int i = 0; i < (int)idToToken.size(); i++ ) tokenToId[ idToToken[ i ] ] = i;
	}

	HRESULT Vocabulary::tokenize( const char* text, std::vector<int>& out ) const
	{
		out.clear();
		if( !text ) return E_POINTER;
		// word split of GPT-2's encoder, then greedy longest match of each word against the vocabulary
		static const std::regex re( R"('s|'t|'re|'ve|'m|'ll|'d| ?[[:alpha:]]+| ?[[:digit:]]+| ?[^\s[:alpha:][:digit:]]+|\s+(?!\S)|\s+)" );
		std::string str = text;
		std::vector<std::string> words;
		std::smatch m;
		while( std::regex_search( str, m, re ) )
		{
			words.push_back( m.str( 0 ) );
			str = m.suffix();
		}
		for( const std::string& word : words )
		{
			const int n = (int)word.size();
			int i = 0;
			while( i < n )
			{
				int j = n;
				for( ; j > i; j-- )
				{
					auto it = tokenToId.find( word.substr( i, j - i ) );
					if( it != tokenToId.end() )
					{
						out.push_back( it->second );
						break;
					}
				}
				if( j > i )
					i = j;
				else
				{
					logWarning( "tokenize: unknown token '%s'", word.substr( i, 1 ).c_str() );
					i++;
				}
			}
		}
		return S_OK;
	}

	// ================================================================================================================
	// ggml file -> device arena
	// ================================================================================================================
	LoadedModel::~LoadedModel()
	{
		if( gpu ) wh_model_destroy( gpu );
	}

	namespace
	{
		template<class T>
		bool rd( std::ifstream& f, T& v )
		{
			f.read( (char*)&v, sizeof( T ) );
			return (bool)f;
		}
	}

	// magic, hparams, mel filters, vocabulary: everything in front of the tensors (Appendix A of SURVEY.md); host only
	static HRESULT readGgmlHeader( std::ifstream& f, const std::string& path, wh_hparams& hp, int32_t& nMel, int32_t& nFft,
		std::vector<float>& filters, Vocabulary& vocab )
	{
		uint32_t magic = 0;
		if( !rd( f, magic ) || magic != 0x67676d6c )
		{
			logError( "invalid model file '%s' (bad magic)", path.c_str() );
			return E_INVALIDARG;
		}
		static_assert( sizeof( wh_hparams ) == 44, "hparams are 11 x int32 in file order" );
		if( !rd( f, hp ) ) return E_INVALIDARG;
		if( !rd( f, nMel ) || !rd( f, nFft ) || nMel <= 0 || nFft <= 0 || (int64_t)nMel * nFft > ( 1 << 20 ) ) return E_INVALIDARG;
		filters.resize( (size_t)nMel * nFft );
		f.read( (char*)filters.data(), filters.size() * 4 );
		int32_t nWords = 0;
		if( !rd( f, nWords ) || nWords < 0 || nWords > ( 1 << 20 ) ) return E_INVALIDARG;
		vocab.idToToken.resize( nWords );
		for( int i = 0; i < nWords; i++ )
		{
			uint32_t len = 0;
			if( !rd( f, len ) || len > ( 1u << 16 ) ) return E_INVALIDARG;
			std::string w( len, '\0' );
			if( len ) f.read( &w[ 0 ], len );
			vocab.idToToken[ i ] = std::move( w );
		}
		if( !f ) return E_INVALIDARG;
		vocab.finalize( hp.n_vocab );
		return S_OK;
	}

	HRESULT loadVocabulary( const std::string& path, Vocabulary& vocab )
	{
		std::ifstream f( path, std::ios::binary );
		if( !f ) return (HRESULT)0x80070002;
		wh_hparams hp{};
		int32_t nMel = 0, nFft = 0;
		std::vector<float> filters;
		return readGgmlHeader( f, path, hp, nMel, nFft, filters, vocab );
	}

	HRESULT loadGgmlFile( const std::string& path, int device, const sLoadModelCallbacks* callbacks, std::shared_ptr<LoadedModel>& out, wh_comm* comm, int root )
	{
		int rank = 0, world = 1;
		if( comm ) CHECK_WH( wh_comm_info( comm, &rank, &world ) );
		const bool readsTensors = !comm || rank == root;
		std::ifstream f;
		int64_t fileSize = 0;
		auto lm = std::make_shared<LoadedModel>();
		// Failure has to be collective: a rank that cannot open the file, parse its header or create its device model -- the root or any other --
		// must not leave the others inside ncclBroadcast. So EVERYTHING up to the arena's broadcast runs under one HRESULT per rank, the ranks
		// exchange those (4 bytes each, every rank in turn the root of a status broadcast), and only when all are S_OK does the arena travel.
		auto prepare = [ & ]() -> HRESULT
		{
			f.open( path, std::ios::binary );
			if( !f )
			{
				logError( "failed to open model file '%s'", path.c_str() );
				return (HRESULT)0x80070002;	   // HRESULT_FROM_WIN32( ERROR_FILE_NOT_FOUND )
			}
			f.seekg( 0, std::ios::end );
			fileSize = (int64_t)f.tellg();
			f.seekg( 0 );
			int32_t nMel = 0, nFft = 0;
			std::vector<float> filters;
			CHECK( readGgmlHeader( f, path, lm->hp, nMel, nFft, filters, lm->vocab ) );
			CHECK_WH( wh_device_set( device ) );
			CHECK_WH( wh_model_create( &lm->hp, nullptr, 0, &lm->gpu ) );
			CHECK_WH( wh_model_set_filters( lm->gpu, nMel, nFft, filters.data() ) );
			return S_OK;
		};
		auto readTensors = [ & ]() -> HRESULT
		{
			std::vector<char> payload;
			while( true )
			{
				int32_t nDims = 0, nameLen = 0, ftype = 0;
				if( !rd( f, nDims ) ) break;	// clean end of file
				if( !rd( f, nameLen ) || !rd( f, ftype ) || nDims < 1 || nDims > 3 || nameLen <= 0 || nameLen > 256 ) return E_INVALIDARG;
				int32_t ne[ 3 ] = { 1, 1, 1 };
				int64_t count = 1;
				for( int i = 0; i < nDims; i++ )
				{
					if( !rd( f, ne[ i ] ) || ne[ i ] <= 0 ) return E_INVALIDARG;
					count *= ne[ i ];
				}
				std::string name( (size_t)nameLen, '\0' );
				f.read( &name[ 0 ], nameLen );
				const int64_t bytes = count * ( ftype == 0 ? 4 : 2 );
				if( !f || bytes > fileSize ) return E_INVALIDARG;
				payload.resize( (size_t)bytes );
				f.read( payload.data(), bytes );
				if( !f )
				{
					logError( "model file '%s' is truncated inside tensor '%s'", path.c_str(), name.c_str() );
					return E_INVALIDARG;
				}
				CHECK_WH( wh_model_set_tensor( lm->gpu, name.c_str(), nDims, ne, ftype != 0, payload.data() ) );
				if( callbacks )
				{
					if( callbacks->cancel && S_OK != callbacks->cancel( callbacks->pv ) ) return (HRESULT)0x800704C7;	 // ERROR_CANCELLED
					if( callbacks->progress ) CHECK( callbacks->progress( (double)f.tellg() / (double)fileSize, callbacks->pv ) );
				}
			}
			CHECK_WH( wh_model_finalize( lm->gpu ) );
			return S_OK;
		};
		HRESULT hrRead = prepare();
		if( SUCCEEDED( hrRead ) && readsTensors ) hrRead = readTensors();
		if( comm )
		{
			HRESULT firstPeer = S_OK;
			int failedRank = -1;
			for( int r = 0; r < world; r++ )
			{
				int32_t status = r == rank ? (int32_t)hrRead : 0;
				CHECK_WH( wh_comm_broadcast_i32( comm, r, &status ) );
				if( FAILED( (HRESULT)status ) && failedRank < 0 ) { failedRank = r; firstPeer = (HRESULT)status; }
			}
			if( FAILED( hrRead ) ) return hrRead;
			if( failedRank >= 0 )
			{
				logError( "loadModelShared: rank %d could not load the model (0x%08x); rank %d gives up with it", failedRank, (unsigned)firstPeer, rank );
				return firstPeer;
			}
			double seconds = 0;
			CHECK_WH( wh_model_broadcast( lm->gpu, comm, root, &seconds ) );
			void* dev = nullptr;
			int64_t bytes = 0;
			(void)wh_model_arena( lm->gpu, &dev, &bytes );
			// every rank reports what it saw: a slow link shows up as one rank's figure
			logInfo( "rank %d of %d: model arena %s, %.1f MB in %.3f s (%.1f GB/s)", rank, world, rank == root ? "sent" : "received", bytes / 1e6, seconds,
				seconds > 0 ? bytes / 1e9 / seconds : 0.0 );
		}
		else if( FAILED( hrRead ) )
			return hrRead;
		out = lm;
		return S_OK;
	}

	HRESULT listGPUs( pfnListAdapters pfn, void* pv )
	{
		if( !pfn ) return E_POINTER;
		const int n = wh_device_count();
		for( int i = 0; i < n; i++ )
		{
			char name[ 256 ];
			uint64_t mem = 0;
			int cus = 0;
			if( 0 != wh_device_info( i, name, sizeof( name ), &mem, &cus ) ) continue;
			char full[ 320 ];
			snprintf( full, sizeof( full ), "%d: %s", i, name );
			std::wstring w;
			for( const char* p = full; *p; p++ ) w += (wchar_t)(unsigned char)*p;
			pfn( w.c_str(), pv );
		}
		return S_OK;
	}
}
