// oracle/shim/Utils/Logger.h -- TEST INFRASTRUCTURE ONLY (never linked into the product).
// The reference's vendored CPU path (Whisper/source/ggml.c:100,1055 and Whisper/source/whisper.cpp:18,452...)
// expects MSVC's forced-include of stdafx.h to provide logError/logWarning/logInfo/logDebug. This shim
// declares them with C linkage so the unmodified sources compile with gcc/g++. Definitions: ref_harness.cpp.
#pragma once
#ifdef __cplusplus
extern "C" {
void logError( const char8_t* fmt, ... );
void logWarning( const char8_t* fmt, ... );
void logInfo( const char8_t* fmt, ... );
void logDebug( const char8_t* fmt, ... );
}
#else
void logError( const char* fmt, ... );
void logWarning( const char* fmt, ... );
void logInfo( const char* fmt, ... );
void logDebug( const char* fmt, ... );
#endif
