// TEST INFRASTRUCTURE: PathCchRenameExtension (the extension after the last dot of the last path component is replaced, or appended when there is none).
#pragma once
#include <wchar.h>
inline HRESULT PathCchRenameExtension( wchar_t* path, size_t cch, const wchar_t* ext )
{
	wchar_t* dot = nullptr;
	for( wchar_t* p = path; *p; p++ )
	{
		if( *p == L'.' ) dot = p;
		else if( *p == L'/' || *p == L'\\' ) dot = nullptr;
	}
	wchar_t* const end = dot ? dot : path + wcslen( path );
	if( (size_t)( end - path ) + wcslen( ext ) + 1 > cch ) return E_INVALIDARG;
	wcscpy( end, ext );
	return S_OK;
}
