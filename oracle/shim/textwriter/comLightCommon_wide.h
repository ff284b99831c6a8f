// TEST INFRASTRUCTURE (oracle/Makefile, libtextwriter_ref.so): overlay of ComLightLib/comLightCommon.h for compiling Examples/main/textWriter.cpp unmodified.
// The reference's portable header says `using LPCTSTR = const char*` outside MSVC, while textWriter.cpp is Windows code that passes L".txt" literals and
// fills wchar_t buffers through LPCTSTR: here LPCTSTR is what it is on Windows, a wide string. Everything else is the reference's own header text order.
#pragma once
#include "hresult.h"
#include "pal/guiddef.h"
using LPCTSTR = const wchar_t*;
#include "unknwn.h"
