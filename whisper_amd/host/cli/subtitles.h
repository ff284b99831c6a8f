// Transcript writers of the command-line tool: plain text, SubRip and WebVTT, byte-compatible with what the reference's
// Examples/main/textWriter.cpp produces (UTF-8 BOM, CRLF line ends, hh:mm:ss.mmm stamps, leading blanks of a segment dropped).
#pragma once
#include <stdint.h>
#include <string>
#include "whisperApi.h"

namespace cli
{
	// hh:mm:ss.mmm (or hh:mm:ss,mmm for SubRip) from 100 ns ticks; hours keep counting past 24
	std::string formatStamp( uint64_t ticks, bool comma = false );

	enum struct eFormat { Text, TextNoStamps, SubRip, WebVTT };
	// The whole file as a byte string
	std::string renderTranscript( const Whisper::sSegment* segments, size_t count, eFormat format );
	// audio.wav -> audio.<ext> next to it; returns S_OK or an error HRESULT
	HRESULT writeTranscript( Whisper::iContext* context, const std::string& audioPath, const char* ext, eFormat format );
	std::string replaceExtension( const std::string& path, const char* ext );
}
