// TEST INFRASTRUCTURE: the handful of ATL string members Examples/main/textWriter.cpp uses, over std::basic_string (oracle/Makefile, libtextwriter_ref.so).
#pragma once
#include <string>
#include <vector>
#include <stdarg.h>
#include <stdio.h>
#include <wchar.h>
typedef unsigned int DWORD;
class CString
{
	std::wstring s;
	std::vector<wchar_t> buf;
public:
	CString& operator=( const wchar_t* p ) { s = p ? p : L""; return *this; }
	int GetLength() const { return (int)s.size(); }
	// a writable buffer of n characters holding the current text (ATL: the string's own storage, grown)
	wchar_t* GetBufferSetLength( int n )
	{
		buf.assign( (size_t)n + 1, 0 );
		wcsncpy( buf.data(), s.c_str(), (size_t)n );
		return buf.data();
	}
	void ReleaseBuffer() { s = buf.data(); }
	operator const wchar_t*() const { return s.c_str(); }
	const std::wstring& str() const { return s; }
};
class CStringA
{
	std::string s;
public:
	CStringA& operator=( const char* p ) { s = p ? p : ""; return *this; }
	CStringA& operator+=( const char* p ) { s += p; return *this; }
	int GetLength() const { return (int)s.size(); }
	operator const char*() const { return s.c_str(); }
	void AppendFormat( const char* fmt, ... )
	{
		char tmp[ 512 ];
		va_list ap;
		va_start( ap, fmt );
		vsnprintf( tmp, sizeof( tmp ), fmt, ap );
		va_end( ap );
		s += tmp;
	}
	void Format( const char* fmt, ... )
	{
		char tmp[ 512 ];
		va_list ap;
		va_start( ap, fmt );
		vsnprintf( tmp, sizeof( tmp ), fmt, ap );
		va_end( ap );
		s = tmp;
	}
};
