// Command-line transcription tool over the iModel / iContext API of libWhisper.so: the Linux counterpart of the
// reference's Examples/main (main.cpp:174-330, params.cpp:24-56) with the same options, console output, exit codes and
// .txt / .srt / .vtt writers. Audio comes from the WAV stand-in for Media Foundation (16 kHz PCM16 / float32).
//
//   whisper-main -m ggml-medium.bin -f clip.wav -osrt
//
// Like the reference's tool the audio goes through iMediaFoundation::openAudioFile + iContext::runStreamed (per-window
// spectrogram normalisation) unless token timestamps are requested, which need the whole buffer (main.cpp:305-321).
// Differences from the reference, all due to features this build does not have: -di and -su report
// what the library reports for them; -owts turns token timestamps on but
// writes no karaoke script (the reference's tool does not either: params.h declares the option, main.cpp never reads it).
#include <stdio.h>
#include <string.h>
#include <unistd.h>
#include <atomic>
#include <limits>
#include <string>
#include <thread>
#include <vector>
#include "subtitles.h"
using namespace Whisper;

namespace
{
	struct Options
	{
		uint32_t threads = 4, processors = 1, offsetMs = 0, offsetN = 0, durationMs = 0;
		uint32_t maxContext = std::numeric_limits<uint32_t>::max(), maxLen = 0;
		float wordThreshold = 0.01f;
		bool speedUp = false, translate = false, diarize = false;
		bool outputTxt = false, outputVtt = false, outputSrt = false, outputWords = false;
		bool printSpecial = false, colors = true, noTimestamps = false;
		std::string language = "en", model = "models/ggml-base.en.bin", gpu, prompt;
		std::vector<std::string> inputs;
	};

	enum struct eKind { Flag, NegFlag, U32, F32, Str, Input };
	struct OptionSpec
	{
		const char* brief;
		const char* full;
		eKind kind;
		void* target;
		const char* left;	   // the option's two name columns as the reference's usage text prints them (Examples/main/params.cpp:26-55)
		const char* help;
	};

	std::vector<OptionSpec> optionTable( Options& o )
	{
		return {
			{ "-gpu", "--use-gpu", eKind::Str, &o.gpu, "-gpu,     --use-gpu       ", "The graphic adapter to use for inference" },
			{ "-t", "--threads", eKind::U32, &o.threads, "-t N,     --threads N     ", "number of threads to use during computation" },
			{ "-p", "--processors", eKind::U32, &o.processors, "-p N,     --processors N  ", "number of processors to use during computation" },
			{ "-ot", "--offset-t", eKind::U32, &o.offsetMs, "-ot N,    --offset-t N    ", "time offset in milliseconds" },
			{ "-on", "--offset-n", eKind::U32, &o.offsetN, "-on N,    --offset-n N    ", "segment index offset" },
			{ "-d", "--duration", eKind::U32, &o.durationMs, "-d  N,    --duration N    ", "duration of audio to process in milliseconds" },
			{ "-mc", "--max-context", eKind::U32, &o.maxContext, "-mc N,    --max-context N ", "maximum number of text context tokens to store" },
			{ "-ml", "--max-len", eKind::U32, &o.maxLen, "-ml N,    --max-len N     ", "maximum segment length in characters" },
			{ "-wt", "--word-thold", eKind::F32, &o.wordThreshold, "-wt N,    --word-thold N  ", "word timestamp probability threshold" },
			{ "-su", "--speed-up", eKind::Flag, &o.speedUp, "-su,      --speed-up      ", "speed up audio by x2 (reduced accuracy)" },
			{ "-tr", "--translate", eKind::Flag, &o.translate, "-tr,      --translate     ", "translate from source language to english" },
			{ "-di", "--diarize", eKind::Flag, &o.diarize, "-di,      --diarize       ", "stereo audio diarization" },
			{ "-otxt", "--output-txt", eKind::Flag, &o.outputTxt, "-otxt,    --output-txt    ", "output result in a text file" },
			{ "-ovtt", "--output-vtt", eKind::Flag, &o.outputVtt, "-ovtt,    --output-vtt    ", "output result in a vtt file" },
			{ "-osrt", "--output-srt", eKind::Flag, &o.outputSrt, "-osrt,    --output-srt    ", "output result in a srt file" },
			{ "-owts", "--output-words", eKind::Flag, &o.outputWords, "-owts,    --output-words  ", "output script for generating karaoke video" },
			{ "-ps", "--print-special", eKind::Flag, &o.printSpecial, "-ps,      --print-special ", "print special tokens" },
			{ "-nc", "--no-colors", eKind::NegFlag, &o.colors, "-nc,      --no-colors     ", "do not print colors" },
			{ "-nt", "--no-timestamps", eKind::Flag, &o.noTimestamps, "-nt,      --no-timestamps ", "do not print timestamps" },
			{ "-l", "--language", eKind::Str, &o.language, "-l LANG,  --language LANG ", "spoken language" },
			{ "-m", "--model", eKind::Str, &o.model, "-m FNAME, --model FNAME   ", "model path" },
			{ "-f", "--file", eKind::Input, &o.inputs, "-f FNAME, --file FNAME    ", "path of the input audio file" },
			{ nullptr, "--prompt", eKind::Str, &o.prompt, nullptr, "initial prompt for the model" },
		};
	}

	// whisper_print_usage (Examples/main/params.cpp:22-56), line for line: the defaults in a 7-column bracket, numbers through %d (so the
	// unlimited text context prints as -1), the adapter without a default, --prompt on a line of its own shape
	void printUsage( const char* exe, Options& o )
	{
		fprintf( stderr, "\n" );
		fprintf( stderr, "usage: %s [options] file0.wav file1.wav ...\n", exe );
		fprintf( stderr, "\n" );
		fprintf( stderr, "options:\n" );
		fprintf( stderr, "  -h,       --help          [default] show this help message and exit\n" );
		fprintf( stderr, "  -la,      --list-adapters List graphic adapters and exit\n" );
		for( const OptionSpec& s : optionTable( o ) )
		{
			if( !s.left )
			{
				fprintf( stderr, "  %-36s%s\n", s.full, s.help );
				continue;
			}
			char def[ 128 ] = "";
			switch( s.kind )
			{
			case eKind::Flag: snprintf( def, sizeof( def ), "[%-7s] ", *(bool*)s.target ? "true" : "false" ); break;
			case eKind::NegFlag: snprintf( def, sizeof( def ), "[%-7s] ", *(bool*)s.target ? "false" : "true" ); break;
			case eKind::U32: snprintf( def, sizeof( def ), "[%-7d] ", (int)*(uint32_t*)s.target ); break;
			case eKind::F32: snprintf( def, sizeof( def ), "[%-7.2f] ", *(float*)s.target ); break;
			case eKind::Str: if( s.target != &o.gpu ) snprintf( def, sizeof( def ), "[%-7s] ", ( (std::string*)s.target )->c_str() ); break;
			case eKind::Input: snprintf( def, sizeof( def ), "[%-7s] ", "" ); break;
			}
			fprintf( stderr, "  %s%s%s\n", s.left, def, s.help );
		}
		fprintf( stderr, "\n" );
	}

	void jsonEscaped( const std::string& s )
	{
		putchar( '"' );
		for( unsigned char c : s )
		{
			if( c == '"' || c == '\\' ) { putchar( '\\' ); putchar( c ); }
			else if( c < 0x20 ) printf( "\\u%04x", c );
			else putchar( c );
		}
		putchar( '"' );
	}
	// --dump-options (first argument; a test hook): the parsed command line as JSON under the reference's own parameter names
	// (Examples/main/params.h), for the comparison with the reference's parser (tests/test_cli.py)
	void dumpOptions( const Options& o )
	{
		printf( "{\"threads\":%u,\"processors\":%u,\"offset_t_ms\":%u,\"offset_n\":%u,\"duration_ms\":%u,\"max_context\":%u,\"max_len\":%u,\"word_thold\":%g,",
			o.threads, o.processors, o.offsetMs, o.offsetN, o.durationMs, o.maxContext, o.maxLen, o.wordThreshold );
		printf( "\"speed_up\":%d,\"translate\":%d,\"diarize\":%d,\"output_txt\":%d,\"output_vtt\":%d,\"output_srt\":%d,\"output_wts\":%d,\"print_special\":%d,\"print_colors\":%d,\"no_timestamps\":%d,",
			o.speedUp, o.translate, o.diarize, o.outputTxt, o.outputVtt, o.outputSrt, o.outputWords, o.printSpecial, o.colors, o.noTimestamps );
		printf( "\"language\":" ); jsonEscaped( o.language );
		printf( ",\"model\":" ); jsonEscaped( o.model );
		printf( ",\"gpu\":" ); jsonEscaped( o.gpu );
		printf( ",\"prompt\":" ); jsonEscaped( o.prompt );
		printf( ",\"inputs\":[" );
		for( size_t i = 0; i < o.inputs.size(); i++ ) { if( i ) putchar( ',' ); jsonEscaped( o.inputs[ i ] ); }
		printf( "]}\n" );
	}

	void printFailure( const char* what, HRESULT hr ) { fprintf( stderr, "%s: HRESULT 0x%08X\n", what, (unsigned)hr ); }

	void listAdapters()
	{
		printf( "    Available graphic adapters:\n" );
		const HRESULT hr = listGPUs( []( const wchar_t* name, void* ) { printf( "\"%ls\"\n", name ); }, nullptr );
		if( FAILED( hr ) ) printFailure( "Unable to enumerate GPUs", hr );
	}

	// 0 = run, otherwise the process exit code + 1 (so that "stop with code 0" is expressible)
	int parse( int argc, char** argv, Options& o )
	{
		const std::vector<OptionSpec> table = optionTable( o );
		for( int i = 1; i < argc; i++ )
		{
			const std::string arg = argv[ i ];
			if( arg.empty() || arg[ 0 ] != '-' )
			{
				o.inputs.push_back( arg );
				continue;
			}
			if( arg == "-h" || arg == "--help" )
			{
				printUsage( argv[ 0 ], o );
				return 1 + 1;
			}
			if( arg == "-la" || arg == "--list-adapters" )
			{
				listAdapters();
				return 1 + 1;
			}
			const OptionSpec* spec = nullptr;
			for( const OptionSpec& s : table )
				if( ( s.brief && arg == s.brief ) || arg == s.full ) spec = &s;
			if( !spec )
			{
				fprintf( stderr, "error: unknown argument: %s\n", arg.c_str() );
				printUsage( argv[ 0 ], o );
				return 1 + 1;
			}
			const bool takesValue = spec->kind != eKind::Flag && spec->kind != eKind::NegFlag;
			if( takesValue && i + 1 >= argc )
			{
				fprintf( stderr, "error: %s needs a value\n", arg.c_str() );
				return 1 + 1;
			}
			try
			{
				switch( spec->kind )
				{
				case eKind::Flag: *(bool*)spec->target = true; break;
				case eKind::NegFlag: *(bool*)spec->target = false; break;
				case eKind::U32: *(uint32_t*)spec->target = (uint32_t)std::stoul( argv[ ++i ] ); break;
				case eKind::F32: *(float*)spec->target = std::stof( argv[ ++i ] ); break;
				case eKind::Str: *(std::string*)spec->target = argv[ ++i ]; break;
				case eKind::Input: ( (std::vector<std::string>*)spec->target )->push_back( argv[ ++i ] ); break;
				}
			}
			catch( const std::exception& )
			{
				fprintf( stderr, "error: bad value for %s\n", arg.c_str() );
				return 1 + 1;
			}
		}
		return 0;
	}

	std::wstring widen( const std::string& s )
	{
		// UTF-8 -> UTF-32 (wchar_t is 32 bits here)
		std::wstring w;
		for( size_t i = 0; i < s.size(); )
		{
			const unsigned char c = (unsigned char)s[ i ];
			uint32_t cp = c;
			int extra = 0;
			if( c >= 0xF0 ) { cp = c & 7; extra = 3; }
			else if( c >= 0xE0 ) { cp = c & 15; extra = 2; }
			else if( c >= 0xC0 ) { cp = c & 31; extra = 1; }
			i++;
			for( ; extra > 0 && i < s.size(); extra--, i++ ) cp = ( cp << 6 ) | ( (unsigned char)s[ i ] & 63 );
			w.push_back( (wchar_t)cp );
		}
		return w;
	}

	// confidence colours: red (p^3 < 0.1) ... yellow ... green
	const char* const kColors[ 10 ] = {
		"\033[38;5;196m", "\033[38;5;202m", "\033[38;5;208m", "\033[38;5;214m", "\033[38;5;220m",
		"\033[38;5;226m", "\033[38;5;190m", "\033[38;5;154m", "\033[38;5;118m", "\033[38;5;82m" };

	const char* colorOf( const sToken& t )
	{
		const float p = t.probability;
		int idx = (int)( p * p * p * 10.0f );
		idx = idx < 0 ? 0 : ( idx > 9 ? 9 : idx );
		return kColors[ idx ];
	}

	void printTokens( const sSegment& seg, const sToken* tokens, const Options& o )
	{
		for( uint32_t j = 0; j < seg.countTokens; j++ )
		{
			const sToken& t = tokens[ seg.firstToken + j ];
			if( !o.printSpecial && ( t.flags & eTokenFlags::Special ) ) continue;
			printf( "%s%s\033[0m", colorOf( t ), t.text ? t.text : "" );
		}
	}

	const char* speakerLabel( iContext* ctx, const sSegment& seg )
	{
		eSpeakerChannel ch;
		if( FAILED( ctx->detectSpeaker( seg.time, ch ) ) ) return "";
		switch( ch )
		{
		case eSpeakerChannel::Unsure: return "(speaker ?)";
		case eSpeakerChannel::Left: return "(speaker 0)";
		case eSpeakerChannel::Right: return "(speaker 1)";
		default: return "";
		}
	}

	// new_segment_callback: print the last n_new segments as they are produced
	HRESULT onNewSegments( iContext* ctx, uint32_t nNew, void* user ) noexcept
	{
		const Options& o = *(const Options*)user;
		ComLight::CComPtr<iTranscribeResult> res;
		HRESULT hr = ctx->getResults( eResultFlags::Timestamps | eResultFlags::Tokens, &res );
		if( FAILED( hr ) ) return hr;
		sTranscribeLength len;
		hr = res->getSize( len );
		if( FAILED( hr ) ) return hr;
		const uint32_t first = len.countSegments - nNew;
		if( first == 0 ) printf( "\n" );
		const sSegment* const segs = res->getSegments();
		const sToken* const toks = res->getTokens();
		for( uint32_t i = first; i < len.countSegments; i++ )
		{
			const sSegment& s = segs[ i ];
			if( o.noTimestamps )
			{
				if( o.colors ) printTokens( s, toks, o );
				else printf( "%s", s.text ? s.text : "" );
				fflush( stdout );
				continue;
			}
			const char* const who = o.diarize ? speakerLabel( ctx, s ) : "";
			const std::string t0 = cli::formatStamp( s.time.begin.ticks ), t1 = cli::formatStamp( s.time.end.ticks );
			if( o.colors )
			{
				printf( "[%s --> %s] %s ", t0.c_str(), t1.c_str(), who );
				printTokens( s, toks, o );
				printf( "\n" );
			}
			else
				printf( "[%s --> %s]  %s%s\n", t0.c_str(), t1.c_str(), who, s.text ? s.text : "" );
		}
		fflush( stdout );
		return S_OK;
	}

	HRESULT onEncoderBegin( iContext*, void* user ) noexcept
	{
		return ( (std::atomic_bool*)user )->load() ? S_FALSE : S_OK;
	}

	// hidden: --format-sample PREFIX writes PREFIX.txt / .srt / .vtt from a fixed three-segment transcript (no GPU needed);
	// tests/test_cli.py checks the bytes
	int formatSample( const std::string& prefix )
	{
		const sSegment segs[ 3 ] = {
			{ " And so my fellow Americans,", { { 0 }, { 36000000 } }, 0, 0 },
			{ "\t ask not what your country can do for you", { { 36000000 }, { 3723456 * 10000ull } }, 0, 0 },
			{ "ask what you can do for your country.", { { 90061001 * 10000ull }, { 90061999 * 10000ull } }, 0, 0 },
		};
		const struct { const char* ext; cli::eFormat f; } outs[] = { { ".txt", cli::eFormat::Text }, { ".nostamps.txt", cli::eFormat::TextNoStamps },
			{ ".srt", cli::eFormat::SubRip }, { ".vtt", cli::eFormat::WebVTT } };
		for( const auto& o : outs )
		{
			const std::string bytes = cli::renderTranscript( segs, 3, o.f );
			FILE* f = fopen( ( prefix + o.ext ).c_str(), "wb" );
			if( !f ) return 20;
			fwrite( bytes.data(), 1, bytes.size(), f );
			fclose( f );
		}
		return 0;
	}
}

// hidden: --format-file SEGMENTS PREFIX writes PREFIX.txt / .nostamps.txt / .srt / .vtt from the segments of a file (one per line: begin ticks, end ticks,
// then the text up to the end of the line, verbatim); tests/test_cli.py holds the bytes against the reference's own writers (oracle/_ref/libtextwriter_ref.so)
static int formatFile( const char* segFile, const std::string& prefix )
{
	FILE* f = fopen( segFile, "rb" );
	if( !f ) return 20;
	std::vector<std::string> texts;
	std::vector<sSegment> segs;
	char line[ 4096 ];
	while( fgets( line, sizeof( line ), f ) )
	{
		unsigned long long b = 0, e = 0;
		int used = 0;
		if( sscanf( line, "%llu %llu %n", &b, &e, &used ) < 2 ) continue;
		std::string t = line + used;
		while( !t.empty() && ( t.back() == '\n' || t.back() == '\r' ) ) t.pop_back();
		// the separator after the second number is ONE blank: what follows (leading blanks included) is the text
		const char* p = line;
		int fields = 0;
		while( *p && fields < 2 ) { while( *p == ' ' ) p++; while( *p && *p != ' ' ) p++; fields++; }
		if( *p == ' ' ) p++;
		t = p;
		while( !t.empty() && ( t.back() == '\n' || t.back() == '\r' ) ) t.pop_back();
		texts.push_back( t );
		sSegment s{};
		s.time.begin.ticks = b;
		s.time.end.ticks = e;
		segs.push_back( s );
	}
	fclose( f );
	for( size_t i = 0; i < segs.size(); i++ ) segs[ i ].text = texts[ i ].c_str();
	const struct { const char* ext; cli::eFormat fm; } outs[] = { { ".txt", cli::eFormat::Text }, { ".nostamps.txt", cli::eFormat::TextNoStamps },
		{ ".srt", cli::eFormat::SubRip }, { ".vtt", cli::eFormat::WebVTT } };
	for( const auto& o : outs )
	{
		const std::string bytes = cli::renderTranscript( segs.data(), segs.size(), o.fm );
		FILE* w = fopen( ( prefix + o.ext ).c_str(), "wb" );
		if( !w ) return 20;
		fwrite( bytes.data(), 1, bytes.size(), w );
		fclose( w );
	}
	return 0;
}

int main( int argc, char** argv )
{
	if( argc == 3 && !strcmp( argv[ 1 ], "--format-sample" ) ) return formatSample( argv[ 2 ] );
	if( argc == 4 && !strcmp( argv[ 1 ], "--format-file" ) ) return formatFile( argv[ 2 ], argv[ 3 ] );

	sLoggerSetup log{};
	log.flags = eLoggerFlags::UseStandardError;
	log.level = eLogLevel::Debug;
	setupLogger( log );

	Options o;
	o.threads = std::min( 4u, std::max( 1u, std::thread::hardware_concurrency() ) );
	const bool dump = argc > 1 && !strcmp( argv[ 1 ], "--dump-options" );
	if( dump ) { argv[ 1 ] = argv[ 0 ]; argv++; argc--; }
	if( const int stop = parse( argc, argv, o ) ) return stop - 1;
	if( dump ) { dumpOptions( o ); return 0; }
	if( o.colors && !isatty( STDOUT_FILENO ) ) o.colors = false;

	if( o.inputs.empty() )
	{
		fprintf( stderr, "error: no input files specified\n" );
		printUsage( argv[ 0 ], o );
		return 2;
	}
	if( findLanguageKeyA( o.language.c_str() ) == std::numeric_limits<uint32_t>::max() )
	{
		fprintf( stderr, "error: unknown language '%s'\n", o.language.c_str() );
		printUsage( argv[ 0 ], o );
		return 3;
	}

	ComLight::CComPtr<iModel> model;
	HRESULT hr;
	{
		sModelSetup setup{};
		setup.impl = eModelImplementation::GPU;
		const std::wstring gpu = widen( o.gpu );
		if( !gpu.empty() ) setup.adapter = gpu.c_str();
		hr = loadModel( widen( o.model ).c_str(), setup, nullptr, &model );
	}
	if( FAILED( hr ) ) { printFailure( "failed to load the model", hr ); return 4; }

	std::vector<int> prompt;
	if( !o.prompt.empty() )
	{
		hr = model->tokenize( o.prompt.c_str(), []( const int* p, int n, void* pv ) { if( n > 0 ) ( (std::vector<int>*)pv )->assign( p, p + n ); }, &prompt );
		if( FAILED( hr ) ) { printFailure( "failed to tokenize the initial prompt", hr ); return 5; }
	}

	ComLight::CComPtr<iContext> context;
	hr = model->createContext( &context );
	if( FAILED( hr ) ) { printFailure( "failed to initialize whisper context", hr ); return 6; }

	ComLight::CComPtr<iMediaFoundation> media;
	hr = initMediaFoundation( &media );
	if( FAILED( hr ) ) { printFailure( "failed to initialize the audio loader", hr ); return 7; }

	for( const std::string& input : o.inputs )
	{
		if( model->isMultilingual() == S_FALSE && ( o.language != "en" || o.translate ) )
		{
			o.language = "en";
			o.translate = false;
			fprintf( stderr, "main: WARNING: model is not multilingual, ignoring language and translation options\n" );
		}

		sFullParams p;
		context->fullDefaultParams( eSamplingStrategy::Greedy, &p );
		p.setFlag( eFullParamsFlags::PrintRealtime, false );
		p.setFlag( eFullParamsFlags::PrintProgress, false );
		p.setFlag( eFullParamsFlags::PrintTimestamps, !o.noTimestamps );
		p.setFlag( eFullParamsFlags::PrintSpecial, o.printSpecial );
		p.setFlag( eFullParamsFlags::Translate, o.translate );
		p.setFlag( eFullParamsFlags::NoContext );		// several input files are independent clips
		p.language = makeLanguageKey( o.language.c_str() );
		p.cpuThreads = (int)o.threads;
		if( o.maxContext != std::numeric_limits<uint32_t>::max() ) p.n_max_text_ctx = (int)o.maxContext;
		p.offset_ms = (int)o.offsetMs;
		p.duration_ms = (int)o.durationMs;
		p.setFlag( eFullParamsFlags::TokenTimestamps, o.outputWords || o.maxLen > 0 );
		p.thold_pt = o.wordThreshold;
		p.max_len = ( o.outputWords && o.maxLen == 0 ) ? 60 : (int)o.maxLen;
		p.setFlag( eFullParamsFlags::SpeedupAudio, o.speedUp );
		if( !prompt.empty() )
		{
			p.prompt_tokens = prompt.data();
			p.prompt_n_tokens = (int)prompt.size();
		}
		p.new_segment_callback = &onNewSegments;
		p.new_segment_callback_user_data = &o;
		std::atomic_bool aborted{ false };
		p.encoder_begin_callback = &onEncoderBegin;
		p.encoder_begin_callback_user_data = &aborted;

		hr = E_NOTIMPL;
		if( !p.flag( eFullParamsFlags::TokenTimestamps ) )
		{
			ComLight::CComPtr<iAudioReader> reader;
			if( SUCCEEDED( media->openAudioFile( input.c_str(), o.diarize, &reader ) ) )
			{
				sProgressSink sink{ nullptr, nullptr };
				hr = context->runStreamed( p, sink, reader );
			}
		}
		if( hr == E_NOTIMPL )
		{
			ComLight::CComPtr<iAudioBuffer> buffer;
			hr = media->loadAudioFile( input.c_str(), o.diarize, &buffer );
			if( SUCCEEDED( hr ) ) hr = context->runFull( p, buffer );
		}
		if( FAILED( hr ) ) { printFailure( "Unable to process audio", hr ); return 10; }

		if( o.outputTxt && FAILED( hr = cli::writeTranscript( context, input, ".txt", o.noTimestamps ? cli::eFormat::TextNoStamps : cli::eFormat::Text ) ) )
			printFailure( "Unable to produce the text file", hr );
		if( o.outputSrt && FAILED( hr = cli::writeTranscript( context, input, ".srt", cli::eFormat::SubRip ) ) )
			printFailure( "Unable to produce the text file", hr );
		if( o.outputVtt && FAILED( hr = cli::writeTranscript( context, input, ".vtt", cli::eFormat::WebVTT ) ) )
			printFailure( "Unable to produce the text file", hr );
	}

	context->timingsPrint();
	context.release();
	return 0;
}
