// oracle/shim/cli/miscUtils_overlay.h -- TEST INFRASTRUCTURE ONLY.
// Linked into the build tree of oracle/Makefile (libcliparams_ref.so) under the name Examples/main/miscUtils.h. The real header declares
// `using HRESULT = long;` (32 bits on Windows), which collides on Linux with ComLightLib's own `using HRESULT = int32_t` that params.cpp pulls
// in through API/iContext.cl.h; the three functions are the ones params.cpp calls (miscUtils.h:4-8), defined in oracle/cliparams_harness.cpp.
#pragma once
#include <string>
#include "../../ComLightLib/hresult.h"
std::string utf8( const std::wstring& utf16 );
std::wstring utf16( const std::string& u8 );
void printError( const char* what, HRESULT hr );
