// oracle/shim/melstreamer/stdafx.h -- TEST INFRASTRUCTURE ONLY (never linked into the product).
// Stands in for the reference's precompiled header (Whisper/stdafx.h) so that the reference's streaming spectrogram --
// Whisper/Whisper/MelStreamer.cpp, melSpectrogram.cpp and MF/AudioBuffer.cpp -- compiles UNMODIFIED with g++ from where it lies
// (oracle/Makefile, target _ref/libmelstreamer_ref.so). The real header pulls <windows.h>, <d3d11.h>, <DirectXMath.h> and ATL; what the
// three sources actually use of them is declared here, over the standard library:
//   * Win32 scalars and the few HRESULT names ComLightLib's own Linux headers (ComLightLib/pal/hresult.h) do not carry;
//   * MSVC keywords (__forceinline, __stdcall, __vectorcall, __interface, __int64);
//   * the thread / critical section / condition variable calls of MelStreamerThread over std::thread / mutex / condition_variable_any;
//   * ATL's CComPtr, CComAutoCriticalSection, CComCritSecLock, CHandle;
//   * DirectX::XMVectorSinCos / XMScalarSinCos: DirectXMath is not in this image, so sine and cosine come from libm, rounded from
//     double. The library's own minimax polynomials differ from that by about 1 ulp of a twiddle factor: the only place where this
//     build can differ from a Windows build of the same sources, far below the 1e-5 the test allows on log-mel values.
#pragma once
#define _USE_MATH_DEFINES
#include <assert.h>
#include <math.h>
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <new>
#include <thread>
#include <vector>
#include <immintrin.h>

#define __forceinline inline __attribute__( ( always_inline ) )
#define __stdcall
#define __cdecl
#define __vectorcall
#define __interface struct
#define __int64 long long

#include "ComLightLib/hresult.h"	// HRESULT, S_OK ..., HRESULT_FROM_WIN32, E_EOF, E_BOUNDS, OLE_E_BLANK (the reference's own Linux definitions)

using BYTE = uint8_t;
using DWORD = uint32_t;
using LONG = int32_t;
using PVOID = void*;
using HANDLE = void*;
using PTP_WORK = void*;
using PTP_CALLBACK_INSTANCE = void*;
constexpr DWORD INFINITE = 0xFFFFFFFFu;
constexpr DWORD WAIT_OBJECT_0 = 0;
constexpr DWORD WAIT_TIMEOUT = 258;
inline int GetLastError() { return 0; }

// ---- critical sections and condition variables (synchapi.h). The objects are never destroyed: the reference leaves its
// background thread asleep on a condition variable when a streamer dies idle (MelStreamer.cpp:478-493), which Windows permits. ----
struct CRITICAL_SECTION
{
	std::mutex* const mx = new std::mutex();
};
inline void EnterCriticalSection( CRITICAL_SECTION* cs ) { cs->mx->lock(); }
inline void LeaveCriticalSection( CRITICAL_SECTION* cs ) { cs->mx->unlock(); }
struct CONDITION_VARIABLE
{
	std::condition_variable_any* cv = nullptr;
};
inline void InitializeConditionVariable( CONDITION_VARIABLE* v ) { v->cv = new std::condition_variable_any(); }
inline void WakeAllConditionVariable( CONDITION_VARIABLE* v ) { v->cv->notify_all(); }
inline int SleepConditionVariableCS( CONDITION_VARIABLE* v, CRITICAL_SECTION* cs, DWORD )
{
	v->cv->wait( *cs->mx );	 // entered with the section held once, like the Win32 call
	return 1;
}

// ---- threads (processthreadsapi.h): a handle is a heap record of a std::thread and its exit code ----
struct ShimThread
{
	std::thread thread;
	std::mutex mx;
	std::condition_variable cv;
	bool done = false, started = false;
	DWORD code = 0;
};
inline HANDLE CreateThread( void*, size_t, DWORD ( *proc )( void* ), void* arg, DWORD, DWORD* )
{
	ShimThread* t = new ShimThread();
	t->thread = std::thread( [ t, proc, arg ]() {
		{
			std::lock_guard<std::mutex> lk( t->mx );
			t->started = true;
			t->cv.notify_all();
		}
		const DWORD c = proc( arg );
		std::lock_guard<std::mutex> lk( t->mx );
		t->code = c;
		t->done = true;
		t->cv.notify_all();
	} );
	// MelStreamerThread has a start-up race: a makeBuffer() that runs before the new thread has set threadStatus = Working sees
	// NotStarted, falls through its wait loop (MelStreamer.cpp:436-452; the assert is compiled out of release builds) and returns zeros
	// for every frame not produced yet. On Windows the caller spends its first hundred microseconds in D3D calls and the thread wins;
	// here nothing stands between the constructor and the first makeBuffer(), so thread creation is made "slow": the oracle must
	// never lose that race.
	{
		std::unique_lock<std::mutex> lk( t->mx );
		t->cv.wait( lk, [ t ] { return t->started; } );		// the thread runs (however loaded the machine is) ...
	}
	std::this_thread::sleep_for( std::chrono::milliseconds( 30 ) );	// ... and has had time for the few instructions up to `Working`
	return t;
}
inline DWORD WaitForSingleObject( HANDLE h, DWORD ms )
{
	ShimThread* t = (ShimThread*)h;
	std::unique_lock<std::mutex> lk( t->mx );
	if( ms == INFINITE ) { t->cv.wait( lk, [ t ] { return t->done; } ); return WAIT_OBJECT_0; }
	return t->cv.wait_for( lk, std::chrono::milliseconds( ms ), [ t ] { return t->done; } ) ? WAIT_OBJECT_0 : WAIT_TIMEOUT;
}
inline int GetExitCodeThread( HANDLE h, DWORD* code )
{
	ShimThread* t = (ShimThread*)h;
	std::lock_guard<std::mutex> lk( t->mx );
	*code = t->done ? t->code : 259u;	// STILL_ACTIVE
	return 1;
}
// The record outlives the handle when the thread is still asleep (see above): a finished thread is joined, any other detached --
// after a patient wait: ~MelStreamerThread gives a working thread 100 ms to see `shuttingDown` (MelStreamer.cpp:478-493) and then
// lets go of it while it may still touch the object; a thread that is merely slow must not outlive the stack frame here.
inline void CloseHandle( HANDLE h )
{
	ShimThread* t = (ShimThread*)h;
	bool done;
	{
		std::unique_lock<std::mutex> lk( t->mx );
		done = t->cv.wait_for( lk, std::chrono::seconds( 3 ), [ t ] { return t->done; } );
	}
	if( done ) { t->thread.join(); delete t; }
	else t->thread.detach();
}

// ---- ATL (atlbase.h) ----
template<class T>
class CComPtr
{
public:
	T* p = nullptr;
	CComPtr() = default;
	CComPtr( const CComPtr& ) = delete;
	~CComPtr() { if( p ) p->Release(); }
	T** operator&() { return &p; }
	T* operator->() const { return p; }
	operator T*() const { return p; }
};
class CComAutoCriticalSection
{
public:
	CRITICAL_SECTION m_sec;
};
template<class CS>
class CComCritSecLock
{
	CS& cs;
public:
	CComCritSecLock( CS& c ) : cs( c ) { EnterCriticalSection( &cs.m_sec ); }
	~CComCritSecLock() { LeaveCriticalSection( &cs.m_sec ); }
};
class CHandle
{
	HANDLE h = nullptr;
public:
	CHandle() = default;
	CHandle( const CHandle& ) = delete;
	~CHandle() { if( h ) CloseHandle( h ); }
	void Attach( HANDLE v ) { h = v; }
	operator HANDLE() const { return h; }
};

// ATL's hash map as Languages.cpp uses it: SetAt, Lookup returning a node with m_value (atlcoll.h)
#include <unordered_map>
template<class K, class V>
class CAtlMap
{
public:
	struct CPair { K m_key; V m_value; };
private:
	std::unordered_map<K, CPair> map;
public:
	CAtlMap( unsigned = 17, float = 0.75f, float = 0.25f, float = 2.25f, unsigned = 10 ) {}
	void SetAt( const K& k, const V& v ) { map[ k ] = CPair{ k, v }; }
	const CPair* Lookup( const K& k ) const
	{
		const auto it = map.find( k );
		return it == map.end() ? nullptr : &it->second;
	}
};

// ---- what ContextImpl.misc.cpp asks the system: processor counts (sysinfoapi.h) and Media Foundation's 64-bit a * b / c ----
#include <climits>
#define __declspec( x )
struct SYSTEM_INFO { DWORD dwNumberOfProcessors; };
inline void GetSystemInfo( SYSTEM_INFO* si ) { si->dwNumberOfProcessors = std::max( 1u, std::thread::hardware_concurrency() ); }
enum LOGICAL_PROCESSOR_RELATIONSHIP { RelationProcessorCore = 0 };
struct SYSTEM_LOGICAL_PROCESSOR_INFORMATION { uint64_t ProcessorMask; LOGICAL_PROCESSOR_RELATIONSHIP Relationship; uint64_t pad[ 2 ]; };
inline int GetLogicalProcessorInformation( SYSTEM_LOGICAL_PROCESSOR_INFORMATION*, DWORD* n ) { *n = 0; return 1; }
inline long long MFllMulDiv( long long a, long long b, long long c, long long d ) { return (long long)( ( (__int128)a * b + d ) / c ); }

// ---- DirectXMath (DirectXMathVector.inl XMVectorSinCos, DirectXMathMisc.inl XMScalarSinCos): see the header comment ----
namespace DirectX
{
	inline void XMScalarSinCos( float* pSin, float* pCos, float value )
	{
		*pSin = (float)sin( (double)value );
		*pCos = (float)cos( (double)value );
	}
	inline void XMVectorSinCos( __m128* pSin, __m128* pCos, __m128 v )
	{
		alignas( 16 ) float a[ 4 ], s[ 4 ], c[ 4 ];
		_mm_store_ps( a, v );
		for( int i = 0; i < 4; i++ ) XMScalarSinCos( &s[ i ], &c[ i ], a[ i ] );
		*pSin = _mm_load_ps( s );
		*pCos = _mm_load_ps( c );
	}
}

#include "Utils/Logger.h"		// oracle/shim/Utils/Logger.h: logError ... with C linkage (defined in the harness)
extern "C" {
void logErrorHr( long hr, const char8_t* fmt, ... );	  // Utils/Logger.h:9, :12 of the reference
void logWarningHr( long hr, const char8_t* fmt, ... );
}
#include "Utils/miscUtils.h"	// the reference's own: CHECK, check(), setCurrentThreadName (through the tree of oracle/Makefile)
