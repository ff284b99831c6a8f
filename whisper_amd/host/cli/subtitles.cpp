#include "subtitles.h"
#include <stdio.h>
#include <fstream>

namespace cli
{
	std::string formatStamp( uint64_t ticks, bool comma )
	{
		const uint64_t totalMs = ticks / 10000;
		const unsigned ms = (unsigned)( totalMs % 1000 );
		const uint64_t totalSec = totalMs / 1000;
		const unsigned sec = (unsigned)( totalSec % 60 );
		const unsigned min = (unsigned)( ( totalSec / 60 ) % 60 );
		const unsigned hours = (unsigned)( totalSec / 3600 );
		char buf[ 48 ];
		snprintf( buf, sizeof( buf ), "%02u:%02u:%02u%c%03u", hours, min, sec, comma ? ',' : '.', ms );
		return buf;
	}

	static const char* withoutLeadingBlanks( const char* s )
	{
		if( !s ) return "";
		while( *s == ' ' || *s == '\t' ) s++;
		return s;
	}

	std::string renderTranscript( const Whisper::sSegment* segments, size_t count, eFormat format )
	{
		std::string out = "\xEF\xBB\xBF";
		if( format == eFormat::WebVTT ) out += "WEBVTT\r\n\r\n";
		for( size_t i = 0; i < count; i++ )
		{
			const Whisper::sSegment& s = segments[ i ];
			const char* const text = withoutLeadingBlanks( s.text );
			switch( format )
			{
			case eFormat::Text:
				out += "[" + formatStamp( s.time.begin.ticks ) + " --> " + formatStamp( s.time.end.ticks ) + "]  ";
				out += text;
				out += "\r\n";
				break;
			case eFormat::TextNoStamps:
				out += text;
				out += "\r\n";
				break;
			case eFormat::SubRip:
				out += std::to_string( i + 1 ) + "\r\n";
				out += formatStamp( s.time.begin.ticks, true ) + " --> " + formatStamp( s.time.end.ticks, true ) + "\r\n";
				out += text;
				out += "\r\n\r\n";
				break;
			case eFormat::WebVTT:
				out += formatStamp( s.time.begin.ticks ) + " --> " + formatStamp( s.time.end.ticks ) + "\r\n";
				out += text;
				out += "\r\n\r\n";
				break;
			}
		}
		return out;
	}

	std::string replaceExtension( const std::string& path, const char* ext )
	{
		const size_t slash = path.find_last_of( "/\\" );
		const size_t dot = path.find_last_of( '.' );
		if( dot == std::string::npos || ( slash != std::string::npos && dot < slash ) )
			return path + ext;
		return path.substr( 0, dot ) + ext;
	}

	HRESULT writeTranscript( Whisper::iContext* context, const std::string& audioPath, const char* ext, eFormat format )
	{
		using namespace Whisper;
		ComLight::CComPtr<iTranscribeResult> result;
		const HRESULT hr = context->getResults( eResultFlags::Timestamps | eResultFlags::Tokens, &result );
		if( FAILED( hr ) ) return hr;
		sTranscribeLength len;
		const HRESULT hr2 = result->getSize( len );
		if( FAILED( hr2 ) ) return hr2;
		const std::string bytes = renderTranscript( result->getSegments(), len.countSegments, format );
		std::ofstream f( replaceExtension( audioPath, ext ), std::ios::binary | std::ios::trunc );
		if( !f ) return E_FAIL;
		f.write( bytes.data(), (std::streamsize)bytes.size() );
		return f ? S_OK : E_FAIL;
	}
}
