// oracle/melstreamer_harness.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product).
// Flat C entry points over the reference's streaming spectrogram, compiled UNMODIFIED from /root/reference by oracle/Makefile
// (Whisper/Whisper/MelStreamer.cpp + Spectrogram.cpp + melSpectrogram.cpp, MF/AudioBuffer.cpp) into oracle/_ref/libmelstreamer_ref.so:
// the oracle of SURVEY.md section 8 row f1 (iContext::runStreamed's per-window spectrogram, MelStreamer.cpp:125-187) and, for row a1,
// the whole-buffer spectrogram of the GPU model's runFull (Spectrogram::pcmToMel, Spectrogram.cpp:64-122).
//
// What this file and memory_reader.inl (shared with contextimpl_harness.cpp) supply is what those sources link against and Windows / Media
// Foundation would provide:
//   * PcmReader's three methods (declared in the reference's MF/PcmReader.h, defined in MF/PcmReader.cpp over IMFSourceReader):
//     here over PCM in memory (shim/melstreamer/mfidl.h). The rules are PcmReader.cpp's: length = samples / 160 chunks
//     (:277, :300), whole chunks while 160 samples are buffered (:399-405), ONE final partial chunk padded with zeros (:418-425,
//     copyMono :24-30), then E_EOF;
//     THE END OF A STREAM: readNextSample() compacts the buffer and zeroes bufferReadOffset BEFORE it learns that the stream has
//     ended (PcmReader.cpp:307-322, :345), and readChunk() then finishes with the offset and count it read before the call
//     (:398-399, :418-425: `bufferReadOffset = off + availableSamples`). Whenever chunks had been consumed from the last delivery
//     (off > 0), the next readChunk() computes size - offset below zero as an unsigned number, finds "enough data" and hands out
//     chunk after chunk of whatever lies behind the vector instead of E_EOF. The streamer asks for two chunks more than frames
//     (MelStreamer.cpp:32), so on Windows the LAST TWO FRAMES of a stream whose length is not a multiple of 160 samples contain
//     stale heap memory where zeros belong. The methods of memory_reader.inl keep the reference's statements in their order, so the behaviour is
//     reproduced (block = 1000: frame length-1 moves by 1e-2); with one chunk per delivery (block = 160, the default) the last
//     delivery is the remainder itself, off is 0 and the end of the stream is clean -- that is the configuration the fixture is made
//     with, and what the product defines (zeros past the last sample).
//   * ThreadPoolWork (Utils/parallelFor.h; Utils/parallelFor.cpp is the Win32 thread pool): n std::threads per call;
//   * the logger's functions and setCurrentThreadName.
#include "stdafx.h"
#include "Whisper/MelStreamer.h"
#include "Whisper/Spectrogram.h"
#include <chrono>
#include <cstdarg>
#include <cstdio>

using namespace Whisper;

#include "memory_reader.inl"

static void vlog( const char* level, const char8_t* fmt, va_list ap )
{
	fprintf( stderr, "[melstreamer_ref %s] ", level );
	vfprintf( stderr, (const char*)fmt, ap );
	fputc( '\n', stderr );
}
extern "C" {
void logError( const char8_t* fmt, ... ) { va_list ap; va_start( ap, fmt ); vlog( "error", fmt, ap ); va_end( ap ); }
void logWarning( const char8_t* fmt, ... ) { va_list ap; va_start( ap, fmt ); vlog( "warning", fmt, ap ); va_end( ap ); }
void logInfo( const char8_t*, ... ) {}
void logDebug( const char8_t*, ... ) {}
}

namespace
{
	// iAudioReader over PCM the caller owns (the reference's own is MF/loadAudioFile.cpp's MediaFileReader)
	struct MemoryReader : iAudioReader
	{
		IMFSourceReader* const source;
		MemoryReader( const float* pcm, size_t n, size_t block ) : source( new IMFSourceReader() )
		{
			source->pcm = pcm; source->count = n; source->block = block ? block : 160;
		}
		virtual ~MemoryReader() { source->Release(); }
		HRESULT COMLIGHTCALL QueryInterface( REFIID, void** ) override { return E_NOINTERFACE; }
		uint32_t COMLIGHTCALL AddRef() override { return 1; }
		uint32_t COMLIGHTCALL Release() override { return 1; }
		HRESULT COMLIGHTCALL getDuration( int64_t& rdi ) const override { rdi = (int64_t)source->count * 10000000 / SAMPLE_RATE; return S_OK; }
		HRESULT COMLIGHTCALL getReader( IMFSourceReader** pp ) const override { source->AddRef(); *pp = source; return S_OK; }
		HRESULT COMLIGHTCALL requestedStereo() const override { return S_FALSE; }
	};
	// iAudioBuffer over PCM the caller owns (what iContext::runFull receives)
	struct MemoryBuffer : iAudioBuffer
	{
		const float* const pcm;
		const uint32_t count;
		MemoryBuffer( const float* p, uint32_t n ) : pcm( p ), count( n ) {}
		virtual ~MemoryBuffer() {}
		HRESULT COMLIGHTCALL QueryInterface( REFIID, void** ) override { return E_NOINTERFACE; }
		uint32_t COMLIGHTCALL AddRef() override { return 1; }
		uint32_t COMLIGHTCALL Release() override { return 1; }
		uint32_t COMLIGHTCALL countSamples() const override { return count; }
		const float* COMLIGHTCALL getPcmMono() const override { return pcm; }
		const float* COMLIGHTCALL getPcmStereo() const override { return nullptr; }
		HRESULT COMLIGHTCALL getTime( int64_t& rdi ) const override { rdi = 0; return S_OK; }
	};
	struct Streamer
	{
		Filters filters;
		ProfileCollection profiler;
		std::vector<float> pcm;
		std::unique_ptr<MemoryReader> reader;
		std::unique_ptr<iSpectrogram> mel;
	};
}

extern "C" {
// threads <= 1: MelStreamerSimple (FFTs on demand); >= 2: MelStreamerThread( threads ) (ContextImpl.cpp runStreamed picks by cpuThreads)
__attribute__( ( visibility( "default" ) ) ) void* ms_create( const float* filters, int nMel, int nFft, const float* pcm, long long nSamples, int threads, int block )
{
	try
	{
		if( nMel != (int)N_MEL || nFft != 1 + (int)FFT_SIZE / 2 ) return nullptr;
		Streamer* s = new Streamer();
		s->filters.n_mel = (uint32_t)nMel;
		s->filters.n_fft = (uint32_t)nFft;
		s->filters.data.assign( filters, filters + (size_t)nMel * nFft );
		s->pcm.assign( pcm, pcm + nSamples );
		s->reader.reset( new MemoryReader( s->pcm.data(), s->pcm.size(), (size_t)block ) );
		if( threads >= 2 ) s->mel.reset( new MelStreamerThread( s->filters, s->profiler, s->reader.get(), threads ) );	// (start-up race: shim/melstreamer/stdafx.h CreateThread)
		else s->mel.reset( new MelStreamerSimple( s->filters, s->profiler, s->reader.get() ) );
		return s;
	}
	catch( ... )
	{
		return nullptr;
	}
}
__attribute__( ( visibility( "default" ) ) ) long long ms_length( void* h ) { return (long long)( (Streamer*)h )->mel->getLength(); }
// out = N_MEL rows of `length` floats (the layout makeBuffer returns: stride == length)
__attribute__( ( visibility( "default" ) ) ) int ms_make_buffer( void* h, long long offset, long long length, float* out )
{
	const float* buffer = nullptr;
	size_t stride = 0;
	const HRESULT hr = ( (Streamer*)h )->mel->makeBuffer( (size_t)offset, (size_t)length, &buffer, stride );
	if( FAILED( hr ) ) return (int)hr;
	for( size_t j = 0; j < N_MEL; j++ ) memcpy( out + j * (size_t)length, buffer + j * stride, (size_t)length * 4 );
	return (int)hr;
}
// Spectrogram::pcmToMel (Spectrogram.cpp:64-122): the whole buffer, normalised on its global maximum; out = N_MEL rows of nSamples / 160 floats
__attribute__( ( visibility( "default" ) ) ) int ms_pcm_to_mel( const float* filters, int nMel, int nFft, const float* pcm, long long nSamples, int threads, float* out )
{
	if( nMel != (int)N_MEL || nFft != 1 + (int)FFT_SIZE / 2 ) return (int)E_INVALIDARG;
	Filters f;
	f.n_mel = (uint32_t)nMel;
	f.n_fft = (uint32_t)nFft;
	f.data.assign( filters, filters + (size_t)nMel * nFft );
	MemoryBuffer buffer( pcm, (uint32_t)nSamples );
	Spectrogram mel;
	HRESULT hr = mel.pcmToMel( &buffer, f, threads );
	if( FAILED( hr ) ) return (int)hr;
	iSpectrogram& is = mel;
	const float* p = nullptr;
	size_t stride = 0;
	hr = is.makeBuffer( 0, is.getLength(), &p, stride );
	if( FAILED( hr ) ) return (int)hr;
	for( size_t j = 0; j < N_MEL; j++ ) memcpy( out + j * is.getLength(), p + j * stride, is.getLength() * 4 );
	return (int)hr;
}
__attribute__( ( visibility( "default" ) ) ) void ms_destroy( void* h )
{
	Streamer* s = (Streamer*)h;
	s->mel.reset();		// the streamer first: it holds the reader
	delete s;
}
}
