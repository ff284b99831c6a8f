// oracle/cliparams_harness.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product).
// The command line of the reference's CLI -- Examples/main/params.cpp (whisper_params::parse, the defaults of params.h) -- compiled
// UNMODIFIED by oracle/Makefile into oracle/_ref/libcliparams_ref.so: the oracle for the option handling of whisper-main
// (whisper_amd/host/cli). cp_parse() parses a UTF-8 argument vector the way wmain's would be and prints the
// resulting whisper_params as JSON; tests/test_cli.py holds whisper-main's own parser (--dump-options) against it.
#include <climits>
#include <cstdio>
#include <cstring>
#include <sstream>
#include <string>
#include <vector>
#include "Examples/main/params.h"
#include "Whisper/API/iContext.cl.h"

std::wstring utf16( const std::string& u8 )
{
	std::wstring w;
	for( size_t i = 0; i < u8.size(); )
	{
		const unsigned char c = (unsigned char)u8[ i ];
		uint32_t cp = c;
		int extra = 0;
		if( c >= 0xF0 ) { cp = c & 7; extra = 3; }
		else if( c >= 0xE0 ) { cp = c & 15; extra = 2; }
		else if( c >= 0xC0 ) { cp = c & 31; extra = 1; }
		i++;
		for( int k = 0; k < extra && i < u8.size(); k++, i++ ) cp = ( cp << 6 ) | ( (unsigned char)u8[ i ] & 63 );
		w.push_back( (wchar_t)cp );
	}
	return w;
}
std::string utf8( const std::wstring& w )
{
	std::string s;
	for( wchar_t ch : w )
	{
		const uint32_t cp = (uint32_t)ch;
		if( cp < 0x80 ) s.push_back( (char)cp );
		else if( cp < 0x800 ) { s.push_back( (char)( 0xC0 | ( cp >> 6 ) ) ); s.push_back( (char)( 0x80 | ( cp & 63 ) ) ); }
		else if( cp < 0x10000 ) { s.push_back( (char)( 0xE0 | ( cp >> 12 ) ) ); s.push_back( (char)( 0x80 | ( ( cp >> 6 ) & 63 ) ) ); s.push_back( (char)( 0x80 | ( cp & 63 ) ) ); }
		else { s.push_back( (char)( 0xF0 | ( cp >> 18 ) ) ); s.push_back( (char)( 0x80 | ( ( cp >> 12 ) & 63 ) ) ); s.push_back( (char)( 0x80 | ( ( cp >> 6 ) & 63 ) ) ); s.push_back( (char)( 0x80 | ( cp & 63 ) ) ); }
	}
	return s;
}
void printError( const char* what, HRESULT hr ) { fprintf( stderr, "%s: error code %i (0x%08X)\n", what, (int)hr, (unsigned)hr ); }
namespace Whisper
{
	HRESULT COMLIGHTCALL listGPUs( pfnListAdapters, void* ) { return S_OK; }	   // -la prints the adapters of the library: none here
}

static void jsonString( std::ostringstream& o, const std::string& s )
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

// returns 1 when parse() says "go on", 0 when it says "stop" (help, list adapters, unknown argument); out = the parameters as JSON
extern "C" __attribute__( ( visibility( "default" ) ) ) int cp_parse( int argc, const char* const* argvUtf8, int threadsDefault, char* out, int cap )
{
	std::vector<std::wstring> wide;
	for( int i = 0; i < argc; i++ ) wide.push_back( utf16( argvUtf8[ i ] ) );
	std::vector<wchar_t*> argv;
	for( auto& w : wide ) argv.push_back( w.data() );
	whisper_params p;
	p.n_threads = (uint32_t)threadsDefault;	  // the constructor's min( 4, hardware threads ) depends on the machine
	const bool go = p.parse( argc, argv.data() );
	std::ostringstream o;
	o << "{\"threads\":" << p.n_threads << ",\"processors\":" << p.n_processors << ",\"offset_t_ms\":" << p.offset_t_ms << ",\"offset_n\":" << p.offset_n
	  << ",\"duration_ms\":" << p.duration_ms << ",\"max_context\":" << p.max_context << ",\"max_len\":" << p.max_len << ",\"word_thold\":" << p.word_thold
	  << ",\"speed_up\":" << p.speed_up << ",\"translate\":" << p.translate << ",\"diarize\":" << p.diarize << ",\"output_txt\":" << p.output_txt
	  << ",\"output_vtt\":" << p.output_vtt << ",\"output_srt\":" << p.output_srt << ",\"output_wts\":" << p.output_wts << ",\"print_special\":" << p.print_special
	  << ",\"print_colors\":" << p.print_colors << ",\"no_timestamps\":" << p.no_timestamps << ",\"language\":";
	jsonString( o, p.language );
	o << ",\"model\":";
	jsonString( o, utf8( p.model ) );
	o << ",\"gpu\":";
	jsonString( o, utf8( p.gpu ) );
	o << ",\"prompt\":";
	jsonString( o, p.prompt );
	o << ",\"inputs\":[";
	for( size_t i = 0; i < p.fname_inp.size(); i++ ) { o << ( i ? "," : "" ); jsonString( o, utf8( p.fname_inp[ i ] ) ); }
	o << "]}";
	snprintf( out, (size_t)cap, "%s", o.str().c_str() );
	return go ? 1 : 0;
}
