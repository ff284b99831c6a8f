// TEST INFRASTRUCTURE: CAtlFile::Create / Write as textWriter.cpp calls them, over stdio; the wide path is converted to UTF-8 (oracle/Makefile, libtextwriter_ref.so).
#pragma once
#include <stdio.h>
#include <string>
#include "atlstr.h"
constexpr unsigned GENERIC_WRITE = 0x40000000u, CREATE_ALWAYS = 2;
class CAtlFile
{
	FILE* f = nullptr;
public:
	~CAtlFile() { if( f ) fclose( f ); }
	HRESULT Create( const wchar_t* path, unsigned, unsigned, unsigned )
	{
		std::string u8;
		for( const wchar_t* p = path; *p; p++ )
		{
			const unsigned cp = (unsigned)*p;
			if( cp < 0x80 ) u8.push_back( (char)cp );
			else if( cp < 0x800 ) { u8.push_back( (char)( 0xC0 | ( cp >> 6 ) ) ); u8.push_back( (char)( 0x80 | ( cp & 63 ) ) ); }
			else { u8.push_back( (char)( 0xE0 | ( cp >> 12 ) ) ); u8.push_back( (char)( 0x80 | ( ( cp >> 6 ) & 63 ) ) ); u8.push_back( (char)( 0x80 | ( cp & 63 ) ) ); }
		}
		f = fopen( u8.c_str(), "wb" );
		return f ? S_OK : E_FAIL;
	}
	HRESULT Write( const void* p, DWORD n ) { return fwrite( p, 1, n, f ) == n ? S_OK : E_FAIL; }
};
